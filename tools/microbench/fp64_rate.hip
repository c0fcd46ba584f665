// fp64_rate.hip -- sustained fp64 rates of one MI355X: v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64
// and v_fma_f64, every SIMD busy with W waves of independent accumulators.
// Build: hipcc -O3 --offload-arch=gfx950 fp64_rate.hip -o fp64_rate ; run: ./fp64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void rate(double* out, int n) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double z0 = a, z1 = b, z2 = a + b, z3 = a - b, z4 = a, z5 = b, z6 = a, z7 = b;
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    } else if (KIND == 1) {
      z0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, z0, 0, 0, 0);
      z1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, z1, 0, 0, 0);
      z2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, z2, 0, 0, 0);
      z3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, z3, 0, 0, 0);
    } else {
      z0 = __builtin_fma(z0, a, b); z1 = __builtin_fma(z1, a, b);
      z2 = __builtin_fma(z2, a, b); z3 = __builtin_fma(z3, a, b);
      z4 = __builtin_fma(z4, a, b); z5 = __builtin_fma(z5, a, b);
      z6 = __builtin_fma(z6, a, b); z7 = __builtin_fma(z7, a, b);
    }
  }
  double s = c0[0] + c1[1] + c2[2] + c3[3] + z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7;
  if (s == 12345.678) out[0] = s;
}

template <int KIND>
static void run(const char* name, double flop_per_iter_per_wave, int waves_per_simd) {
  double* out;
  hipMalloc(&out, 8);
  const int n = 20000, blocks = 256 * waves_per_simd;  // 256 threads = one wave per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate<KIND><<<blocks, 256>>>(out, n);
  hipEventRecord(e0);
  rate<KIND><<<blocks, 256>>>(out, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = blocks * 4.0, flop = waves * n * flop_per_iter_per_wave;
  // cycles per instruction per SIMD at 2.4 GHz
  const double instr_per_simd = waves_per_simd * (double)n * (KIND == 2 ? 8 : 4);
  printf("%-28s %d waves/SIMD  %7.3f ms  %7.1f TFLOP/s  %6.1f cycles/instr/SIMD @2.4GHz\n", name,
         waves_per_simd, ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / instr_per_simd);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("v_mfma_f64_16x16x4_f64", 4 * 2.0 * 16 * 16 * 4, w);
    run<1>("v_mfma_f64_4x4x4_4b_f64", 4 * 2.0 * 4 * 4 * 4 * 4, w);
    run<2>("v_fma_f64", 8 * 2.0 * 64, w);
  }
  return 0;
}
