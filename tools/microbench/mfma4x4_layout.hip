// mfma4x4_layout.hip -- operand and result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found
// by feeding unit operands: for every pair (lane of A, lane of B) which result lanes light up.
// Build: hipcc -O3 --offload-arch=gfx950 mfma4x4_layout.hip -o mfma4x4_layout
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(double* out) {
  const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  out[blockIdx.x * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

int main() {
  double* d;
  if (hipMalloc(&d, 4096 * 64 * 8) != hipSuccess) return 1;
  probe<<<4096, 64>>>(d);
  static double h[4096 * 64];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  // for every A lane: the B lanes it pairs with and the result lane of each pair
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[(la * 64 + lb) * 64 + l] != 0.0) printf("  B%-2d->D%-2d", lb, l);
    printf("\n");
  }
  return 0;
}
