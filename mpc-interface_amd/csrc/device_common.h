// device_common.h -- device helpers shared by the assembly kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_prims.h"
#include "kernels.h"

namespace mpcasm {

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

// P[c][c] and q[c] of the diagonal gterms on column c (a cost on a free variable itself): from
// the plan's per-column table when no column carries more than RS_DIAG_MAX of them
// (H_OFF_RS_DPAR: weight, aim slots; H_DOFF_RS_DCOEF), else by walking the gterm list
__device__ __forceinline__ void diagonal_of_column(const PlanDev& p, const double* pb, int c,
                                                   double& dP, double& dq) {
  if (p.rs_diag_table) {
    const int4 sl = *reinterpret_cast<const int4*>(p.itab + p.off_rs_dpar + 4 * c);
    const double* cf = p.dtab + p.doff_rs_dcoef + 2 * c;
    // (a free slot points at the parameter behind the last: read as 0 weight through its 0 coefficient)
    const double w0 = sl.x < p.nparams ? pb[sl.x] : 0.0, a0 = sl.y < p.nparams ? pb[sl.y] : 0.0;
    const double w1 = sl.z < p.nparams ? pb[sl.z] : 0.0, a1 = sl.w < p.nparams ? pb[sl.w] : 0.0;
    dP = (w0 * cf[0]) * cf[0] + (w1 * cf[1]) * cf[1];
    dq = w0 * (cf[0] * (0.0 - a0)) + w1 * (cf[1] * (0.0 - a1));
  } else {
    diagonal_terms(p, pb, c, dP, dq);
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
