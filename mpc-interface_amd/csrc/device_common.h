// device_common.h -- device helpers shared by the assembly kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace mpcasm {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16) + C on the fp64 matrix core.
// Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; D register r of lane l
// is element (row = (l >> 4) + 4 r, col = l & 15).
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// four independent 4x4x4 products: lane l feeds A[x][k] / B[k][x] and receives D[k][x] of
// block g, where x = l & 3, g = (l >> 2) & 3, k = l >> 4 (tools/microbench/mfma4x4_layout.hip)
__device__ __forceinline__ double mfma_f64_4x4x4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// the vector-memory counter (s_waitcnt vmcnt(0)), i.e. it waits for every global
// store in flight to be acknowledged by HBM -- a full memory round trip per
// barrier in kernels that stream results out between LDS phases.  LDS operations
// of a wavefront complete in order, so lgkmcnt(0) + s_barrier is sufficient for
// data handed over through LDS; loads whose results feed LDS writes are waited
// for by the compiler through the data dependency.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// the value of lane T of every quad, in all four lanes of the quad (DPP quad_perm)
template <int T>
__device__ __forceinline__ double quad_broadcast(double v) {
  constexpr int ctl = T | (T << 2) | (T << 4) | (T << 6);
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), ctl, 0xF, 0xF, true),
                          __builtin_amdgcn_mov_dpp(__double2loint(v), ctl, 0xF, 0xF, true));
}

// sum over the 64 lanes of a wavefront, result in every lane
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Contribution of the diagonal gterms (GT_FLAG_DIAG, plan_tables.h) to P[c][c] and q[c]:
// rows coef_k e_{c0+k} never enter the workspace; body.py:292-300 reduces to
// P[c][c] += (w coef) coef and q[c] += w (coef (0 - aim)).  `prm`: the instance's params.
__device__ __forceinline__ void diagonal_terms(const PlanDev& p, const double* prm, int c,
                                               double& dP, double& dq) {
  dP = 0.0;
  dq = 0.0;
  if (p.ndiag == 0) return;
  const int32_t* gt = p.itab + p.off_gterm;
  const double* cf = p.dtab + p.doff_diagcoef;
  for (int g = 0; g < p.ngterm; ++g) {
    const int32_t* rec = gt + g * GT_WORDS;
    if (!(rec[GT_FLAGS] & GT_FLAG_DIAG)) continue;
    const int k = c - rec[GT_AOFF];
    if (k < 0 || k >= rec[GT_NROWS]) continue;
    const double w = prm[rec[GT_WPARAM]], aim = prm[rec[GT_AIMPARAM]];
    const double co = cf[rec[GT_BOFF] + k];
    dP += (w * co) * co;
    dq += w * (co * (0.0 - aim));
  }
}

// One element of a composed row: sum_e coef[e] * base_row(entbase[e], entk[e])[c]
// (flattened definition graph, body.py:158-193).  Column c belongs to at most
// one segment of each base variable (colseg); a segment is either an identity
// block (domain variable, dynamics.py:277-281) or a strided slice of a source
// (state of an ExtendedSystem: matrices[dID][k, :, sID], dynamics.py:283-295).
__device__ __forceinline__ double compose_element(const PlanDev& p, const SrcTable& src, long inst,
                                                  int W, int c, int e0, int e1,
                                                  const int32_t* __restrict__ entbase,
                                                  const int32_t* __restrict__ entk,
                                                  const double* __restrict__ coef) {
  const int32_t* colseg = p.itab + p.off_colseg;
  const int32_t* segs = p.itab + p.off_seg;
  double acc = 0.0;
  for (int e = e0; e < e1; ++e) {
    const int sg = colseg[entbase[e] * W + c];
    if (sg >= 0) {
      const int32_t* s = segs + sg * SEG_WORDS;
      const int k = entk[e];
      const int j = c - s[SEG_DST0];
      double val;
      if (s[SEG_KIND] == SEG_KIND_IDENTITY) {
        val = (j == k) ? 1.0 : 0.0;
      } else {
        const int sid = s[SEG_SRC];
        val = src.ptr[sid][inst * src.stride[sid] + s[SEG_OFF0] + (long)k * s[SEG_ROWSTRIDE] +
                           (long)j * s[SEG_ELEMSTRIDE]];
      }
      acc = fma(coef[e], val, acc);
    }
  }
  return acc;
}

}  // namespace mpcasm
