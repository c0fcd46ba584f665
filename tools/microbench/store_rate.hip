// store_rate.hip -- what a pure 16-byte-store kernel reaches on one MI355X at the output sizes
// of the K1 fill (30 MB ... 4 GB): the practical ceiling beside which mpcasm_fill_su is read.
//   flat : grid-stride, consecutive lanes on consecutive 16-byte words, G workgroups of 256
//   chunk: every wavefront writes ONE contiguous chunk (the fill's pattern: a wave owns the
//          S, U of a few systems), one 1 KiB store instruction after the other
// Build: hipcc -O3 --offload-arch=gfx950 store_rate.hip -o store_rate ; run: ./store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void flat(double2* out, long n2) {
  const double2 v = {1.0, 2.0};
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n2; q += (long)gridDim.x * 256) out[q] = v;
}

__global__ __launch_bounds__(64) void chunk(double2* out, long n2, int per_wave2) {
  const double2 v = {1.0, 2.0};
  const long base = (long)blockIdx.x * per_wave2;
  for (int q = threadIdx.x; q < per_wave2; q += 64)
    if (base + q < n2) out[base + q] = v;
}

// the LTV fill's pattern: every wave owns a contiguous region and writes it in BURSTS of
// `burst2` words with a pause (a step of the recurrence) between them
template <int SLEEP>
__global__ __launch_bounds__(64) void bursty(double2* out, long n2, int per_wave2, int burst2) {
  const double2 v = {1.0, 2.0};
  const long base = (long)blockIdx.x * per_wave2;
  for (int b = 0; b < per_wave2; b += burst2) {
    for (int q = threadIdx.x; q < burst2; q += 64)
      if (b + q < per_wave2 && base + b + q < n2) out[base + b + q] = v;
    for (int i = 0; i < burst2 / 150; ++i) __builtin_amdgcn_s_sleep(SLEEP);
  }
}

static float timed(void (*launch)(double2*, long, int), double2* buf, long n2, int arg, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch(buf, n2, arg);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch(buf, n2, arg);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

static void launch_flat(double2* b, long n2, int wgs) { flat<<<wgs, 256>>>(b, n2); }
static int g_burst2 = 150;
template <int SLEEP>
static void launch_bursty(double2* b, long n2, int per_wave2) {
  bursty<SLEEP><<<(unsigned)((n2 + per_wave2 - 1) / per_wave2), 64>>>(b, n2, per_wave2, g_burst2);
}
static void launch_chunk(double2* b, long n2, int per_wave2) {
  chunk<<<(unsigned)((n2 + per_wave2 - 1) / per_wave2), 64>>>(b, n2, per_wave2);
}

int main() {
  const double mbs[] = {30.3, 60.6, 121.1, 484.4, 526.0, 2493.0};
  double2* buf;
  hipMalloc(&buf, (size_t)4300 << 20);
  printf("%-10s %-28s %10s %9s %6s\n", "MB", "kernel", "us/launch", "GB/s", "frac");
  for (double mb : mbs) {
    const long n2 = (long)(mb * 1e6 / 16);
    for (int wgs : {1024, 2048, 4096, 8192}) {
      const float us = timed(launch_flat, buf, n2, wgs, 20);
      char name[64];
      snprintf(name, sizeof name, "flat, %d workgroups", wgs);
      printf("%-10.1f %-28s %10.2f %9.0f %6.3f\n", mb, name, us, mb / us * 1e3, mb / us / 8);
    }
    for (int kb : {7, 29, 116, 988}) {
      const float us = timed(launch_chunk, buf, n2, kb * 64, 20);
      char name[64];
      snprintf(name, sizeof name, "chunk, %d KiB per wave", kb);
      printf("%-10.1f %-28s %10.2f %9.0f %6.3f\n", mb, name, us, mb / us * 1e3, mb / us / 8);
    }
  }
  // 2048 and 16384 waves x 257 KB (the C5 shape), bursts of 1, 4, 16 rows of 2400 B
  for (int waves : {2048, 16384}) {
    const int per_wave2 = 16068;
    const long n2 = (long)waves * per_wave2;
    const double mb = n2 * 16 / 1e6;
    if (mb > 4300.0 * 1.048) continue;
    for (int rows : {1, 4, 16, 100}) {
      g_burst2 = 150 * rows;
      const float us0 = timed(launch_bursty<0>, buf, n2, per_wave2, 10);
      const float us8 = timed(launch_bursty<8>, buf, n2, per_wave2, 10);
      const float us20 = timed(launch_bursty<20>, buf, n2, per_wave2, 10);
      printf("%-10.1f bursty %5d waves, %3d rows per burst: pause 0 / 512 / 1280 cycles per row  "
             "%8.2f %8.2f %8.2f us  frac %.3f %.3f %.3f\n", mb, waves, rows, us0, us8, us20,
             mb / us0 / 8, mb / us8 / 8, mb / us20 / 8);
    }
  }
  return 0;
}
