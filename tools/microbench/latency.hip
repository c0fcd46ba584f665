// latency.hip -- ground-truth cycle costs on gfx950 for the pieces of the persistent
// kernel's tile loop: dependent LDS reads, dependent fp64 MFMAs, one "trip"
// (8 ds_read_b64 + 4 MFMA), scalar round trips through v_readfirstlane.
// Build: hipcc -O3 --offload-arch=gfx950 latency.hip -o latency ; run: ./latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void bench(unsigned long long* out, int n, int active_waves) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = (double)((i * 7 + 3) & 1023);
  __syncthreads();
  if (wave >= active_waves) return;
  unsigned long long t0, t1;
  // (a) dependent LDS reads: index chase
  int idx = lane;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) idx = (int)lds[idx & 8191] + lane;
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 8 + 0] = t1 - t0;
  // (b) dependent MFMA chain
  f64x4 acc = {0, 0, 0, 0};
  double a = 1.0 + lane, b = 2.0 + idx;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 8 + 1] = t1 - t0;
  // (c) trip: 8 loads (stride 38 doubles) -> wait -> scale -> 4 dependent MFMAs
  const int li = lane & 15, lk = lane >> 4;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    const double* ap = lds + ((i * 16) & 1023) + li;
    const double* bp = lds + ((i * 16 + 608) & 2047) + li;
    double x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = ap[(4 * u + lk) * 38];
      y[u] = bp[(4 * u + lk) * 38];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1.5 * x[u], y[u], acc, 0, 0, 0);
  }
  asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 8 + 2] = t1 - t0;
  // (d) LDS read -> readfirstlane -> LDS read chains (scalar round trip)
  int s = wave;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) s = __builtin_amdgcn_readfirstlane((int)lds[(s + i) & 8191]);
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 8 + 3] = t1 - t0 + (s & 1);
  // (e) independent VALU fp64 stream
  double z = a;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) z = __builtin_fma(z, 1.0000001, 0.5);
  asm volatile("" ::"v"(z));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 8 + 4] = t1 - t0;
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 8 + 5] = (unsigned long long)(acc[0] + z + idx);
}

int main() {
  unsigned long long* d;
  const int nblk = 512;
  hipMalloc(&d, nblk * 64 * sizeof(unsigned long long));
  unsigned long long* h = (unsigned long long*)malloc(nblk * 64 * sizeof(unsigned long long));
  const int n = 256;
  const char* names[5] = {"dependent ds_read_b64", "dependent mfma f64 16x16x4", "trip 8 ds_read + 4 mfma",
                          "ds_read->readfirstlane chain", "dependent v_fma_f64"};
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int blocks = cfg < 2 ? 1 : nblk, waves = (cfg & 1) ? 8 : 1;
    hipMemset(d, 0, nblk * 64 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench, dim3(blocks), dim3(512), 65536, 0, d, n, waves);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, nblk * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    printf("blocks %d, active waves per block %d (kernel %.3f ms)\n", blocks, waves, ms);
    for (int k = 0; k < 5; ++k) printf("  %-32s %8.1f ticks each (wave 0 of block 0)\n", names[k], (double)h[k] / n);
  }
  return 0;
}
