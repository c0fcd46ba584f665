// store_rate2.hip -- which shape of a pure store kernel gets closest to HBM's write rate on one
// MI355X (torch's fill reaches 6.3-6.7 TB/s on 0.5-8 GB; store_rate.hip's kernels 5.5).
// Build: hipcc -O3 --offload-arch=gfx950 store_rate2.hip -o store_rate2 ; run: ./store_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>

// one wavefront per workgroup, a contiguous chunk per wavefront, 16 bytes per lane and store
template <bool NT>
__global__ __launch_bounds__(64) void chunk64(double2* out, long n2, int per_wave2, double2 v) {
  const long base = (long)blockIdx.x * per_wave2;
  for (int q = threadIdx.x; q < per_wave2; q += 64)
    if (base + q < n2) {
      if (NT) {
        __builtin_nontemporal_store(v.x, &out[base + q].x);
        __builtin_nontemporal_store(v.y, &out[base + q].y);
      } else
        out[base + q] = v;
    }
}
// 256 threads, every thread 32 contiguous bytes (two 16-byte stores), a workgroup 8 KiB: the
// shape of a vectorised elementwise kernel
__global__ __launch_bounds__(256) void vec32(double2* out, long n2, int, double2 v) {
  const long q = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (q + 1 < n2) {
    out[q] = v;
    out[q + 1] = v;
  }
}
// 256 threads, 16 bytes per thread and store, a workgroup `per2` contiguous words
__global__ __launch_bounds__(256) void blk256(double2* out, long n2, int per2, double2 v) {
  const long base = (long)blockIdx.x * per2;
  for (int q = threadIdx.x; q < per2; q += 256)
    if (base + q < n2) out[base + q] = v;
}
// 512 threads (the persistent kernel's workgroup), grid = 512 workgroups, each walks the buffer
// in `per2`-word pieces (an instance's outputs), piece i to workgroup i % grid
__global__ __launch_bounds__(512) void persistent(double2* out, long n2, int per2, double2 v) {
  const long pieces = n2 / per2;
  for (long i = blockIdx.x; i < pieces; i += gridDim.x)
    for (int q = threadIdx.x; q < per2; q += 512) out[i * per2 + q] = v;
}

typedef void (*kern_t)(double2*, long, int, double2);
static float timed(kern_t k, unsigned grid, unsigned block, double2* buf, long n2, int arg, double2 v) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, block, 0, 0, buf, n2, arg, v);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, grid, block, 0, 0, buf, n2, arg, v);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10 * 1e3f;
}

int main() {
  double2* buf;
  hipMalloc(&buf, (size_t)2 << 30);
  const double2 v12 = {1.0, 2.0}, v11 = {1.0, 1.0};
  for (long bytes : {(long)484400000, (long)512 << 20, (long)2 << 30}) {
    const long n2 = bytes / 16;
    auto report = [&](const char* name, float us) {
      printf("%8.1f MB  %-44s %9.2f us %7.0f GB/s  %.3f\n", bytes / 1e6, name, us, bytes / us / 1e3,
             bytes / us / 8e6);
    };
    report("chunk64 8 KiB/wave {1,2}", timed(chunk64<false>, (n2 + 511) / 512, 64, buf, n2, 512, v12));
    report("chunk64 8 KiB/wave {1,1}", timed(chunk64<false>, (n2 + 511) / 512, 64, buf, n2, 512, v11));
    report("chunk64 8 KiB/wave nontemporal", timed(chunk64<true>, (n2 + 511) / 512, 64, buf, n2, 512, v12));
    report("chunk64 32 KiB/wave", timed(chunk64<false>, (n2 + 2047) / 2048, 64, buf, n2, 2048, v12));
    report("vec32 (256 thr x 32 B)", timed(vec32, (n2 / 2 + 255) / 256, 256, buf, n2, 0, v12));
    report("blk256 4 KiB/workgroup", timed(blk256, (n2 + 255) / 256, 256, buf, n2, 256, v12));
    report("blk256 8 KiB/workgroup", timed(blk256, (n2 + 511) / 512, 256, buf, n2, 512, v12));
    report("blk256 32 KiB/workgroup", timed(blk256, (n2 + 2047) / 2048, 256, buf, n2, 2048, v12));
    report("blk256 128 KiB/workgroup", timed(blk256, (n2 + 8191) / 8192, 256, buf, n2, 8192, v12));
    report("persistent 512 wg x 512 thr, 7.4 KB pieces", timed(persistent, 512, 512, buf, n2, 462, v12));
    report("persistent 512 wg x 512 thr, 8 KiB pieces", timed(persistent, 512, 512, buf, n2, 512, v12));
    report("persistent 512 wg x 512 thr, 64 KiB pieces", timed(persistent, 512, 512, buf, n2, 4096, v12));
    report("persistent 1024 wg x 512 thr, 8 KiB pieces", timed(persistent, 1024, 512, buf, n2, 512, v12));
  }
  return 0;
}
