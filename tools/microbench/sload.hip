// sload.hip -- latency of a dependent chain of scalar loads (s_load_dword through the scalar
// data cache) on gfx950, one wave alone and with every SIMD of the chip busy doing the same;
// and of a dependent chain of uniform vector loads (global_load + readfirstlane) for comparison.
// Build: hipcc -O3 --offload-arch=gfx950 sload.hip -o sload
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef const __attribute__((address_space(4))) int* const_int_ptr;

__global__ __launch_bounds__(512) void chase(const int* tbl, unsigned long long* out, int n, int span) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const const_int_ptr ct = (const_int_ptr)(uintptr_t)tbl;
  int idx = __builtin_amdgcn_readfirstlane(wave * 16) % span;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) idx = ct[idx];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 4 + 0] = t1 - t0 + (idx == -1);
  int j = __builtin_amdgcn_readfirstlane(wave * 16) % span;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) j = __builtin_amdgcn_readfirstlane(tbl[j + (lane & 0)]);
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 4 + 1] = t1 - t0 + (j == -1);
}

int main() {
  const int maxspan = 1 << 20;
  int* h = (int*)malloc(maxspan * 4);
  int* d;
  unsigned long long *o, ho[4];
  hipMalloc(&d, maxspan * 4);
  hipMalloc(&o, 512 * 8 * 4 * 8);
  const int spans[4] = {256, 2048, 16384, 262144};  // ints: 1 KB, 8 KB, 64 KB, 1 MB
  for (int s = 0; s < 4; ++s) {
    const int span = spans[s];
    for (int i = 0; i < span; ++i) h[i] = (i + 16 * 37) % span;  // a stride of 37 cache lines
    hipMemcpy(d, h, span * 4, hipMemcpyHostToDevice);
    for (int blocks = 1; blocks <= 512; blocks *= 512) {
      const int n = 2000;
      chase<<<blocks, 512>>>(d, o, n, span);
      hipDeviceSynchronize();
      hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
      printf("table %7d B, %3d blocks x 8 waves: s_load chain %7.1f ticks/load, global_load chain %7.1f\n",
             span * 4, blocks, (double)ho[0] / n, (double)ho[1] / n);
    }
  }
  return 0;
}
