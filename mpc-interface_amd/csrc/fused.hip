// fused.hip -- K2 + K3 + K4 in ONE launch for problems whose per-instance state
// fits on chip (the biped C1/C2 and the 3-D LIPM C3 of SURVEY.md section 8d).
//
// One workgroup of NW wavefronts assembles one QP instance:
//   phase 1  stage everything the instance reads -- its horizon matrices
//            (sources), given vector and parameters -- into LDS with coalesced
//            loads; zero the workspace V[rtot][ldv];
//   phase 2  K2: every structurally non-zero element of V (rows of [Mo | d] of the
//            consumed definitions, body.py:149-193) is one lane's short op list
//            sum coef * arena[src] (* given[g]) over LDS; nothing of the preview
//            matrices ever touches HBM;
//   phase 3  K3: P = sum_gterms (w A)^T B on the fp64 matrix core
//            (v_mfma_f64_16x16x4_f64) straight from LDS, skipping 16x16 tiles that
//            are structurally zero (tile masks of the plan); q by a vector pass;
//            K4: G rows = sum_axes arrow * V row, streamed to HBM with 16-byte
//            stores, h from the d column (body.py:236-329);
//   phase 4  reduce the per-wave q partials and store q.
// HBM traffic per instance is the algorithmic minimum: sources + given + params
// in, P + q + G + h out; the plan tables are shared by all workgroups and stay in
// L2.  Reference semantics are those of assemble.hip (same plan tables).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace mpcasm {

int g_phase_mask = 0xBF;  // diagnostic (timing-only ablation): see mpcasm_set_option

namespace {

constexpr int TPW = 9;       // MFMA output tiles a wavefront may own
constexpr int QSLOT = 2;     // q columns per lane (no <= 64 * QSLOT)
constexpr int AXMAX = 4;     // axes of one constraint

struct FusedLayout {
  // offsets in doubles from the start of dynamic LDS
  int v, arena, g, prm, qpart, rr_arrow, rr_voff /* ints */, total_doubles;
  int nop;  // no rounded up to even
};

__host__ __device__ inline int even_up_i(int x) { return (x + 1) & ~1; }

__host__ __device__ inline FusedLayout fused_layout(const PlanDev& p, int nw) {
  FusedLayout L;
  L.nop = even_up_i(p.no);
  int o = 0;
  L.v = o;        o += even_up_i(p.rtot * p.ldv) + 16;
  L.arena = o;    o += even_up_i(p.arena_total);
  L.g = o;        o += even_up_i(p.ng);
  L.prm = o;      o += even_up_i(p.nparams);
  L.qpart = o;    o += nw * L.nop;
  L.rr_arrow = o; o += p.nc * AXMAX;
  L.rr_voff = o;  o += even_up_i(p.nc * (AXMAX + 1)) / 2 + 1;  // ints: voff[AXMAX] + naxes
  L.total_doubles = o;
  return L;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void fused_assemble_kernel(
    PlanDev p, SrcTable src, const double* __restrict__ params, const double* __restrict__ given,
    double* __restrict__ P, double* __restrict__ q, double* __restrict__ G,
    double* __restrict__ h, int batch, int phases) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long inst = blockIdx.x;
  const FusedLayout L = fused_layout(p, NW);
  const int no = p.no, ng = p.ng, nc = p.nc, ldv = p.ldv;

  double* V = lds + L.v;
  double* arena = lds + L.arena;
  double* gl = lds + L.g;
  double* prm = lds + L.prm;
  double* qpart = lds + L.qpart;
  double* rr_arrow = lds + L.rr_arrow;
  int* rr_voff = reinterpret_cast<int*>(lds + L.rr_voff);  // [nc][AXMAX + 1]

  // ---- phase 1: stage inputs, zero the workspace -----------------------------
  {
    double2* V2 = reinterpret_cast<double2*>(V);
    const int n2 = (even_up_i(p.rtot * ldv) + 16) / 2;
    for (int i = tid; i < n2; i += NT) V2[i] = double2{0.0, 0.0};
    if (tid == 0) arena[0] = 1.0;
    const int32_t* arec = p.itab + p.off_arena;
    for (int s = 0; s < p.nsrc; ++s) {
      const int off = arec[2 * s], size = arec[2 * s + 1];
      const double* sp = src.ptr[s] + inst * src.stride[s];
      for (int i = tid; i < size; i += NT) arena[off + i] = sp[i];
    }
    const double* gb = given + inst * ng;
    for (int i = tid; i < ng; i += NT) gl[i] = gb[i];
    const double* pb = params + inst * p.nparams;
    for (int i = tid; i < p.nparams; i += NT) prm[i] = pb[i];
  }
  lds_barrier();

  // ---- phase 2: compose the workspace (K2) -------------------------------------
  if (phases & 1) {
    const int32_t* fd_idx = p.itab + p.off_fd_idx;
    const int32_t* fd_ptr = p.itab + p.off_fd_ptr;
    const int2* ops = reinterpret_cast<const int2*>(p.itab + p.off_op);
    const double* pool = p.dtab + p.doff_coefpool;
    for (int i = tid; i < p.nfd; i += NT) {
      const int o0 = fd_ptr[i], o1 = fd_ptr[i + 1];
      double acc = 0.0;
      for (int o = o0; o < o1; ++o) {
        const int2 rec = ops[o];
        const unsigned packed = (unsigned)rec.y;
        const int gi = (int)(packed >> 16) - 1;
        double v = pool[packed & 0xFFFFu] * arena[rec.x];
        if (gi >= 0) v *= gl[gi];
        acc += v;
      }
      V[fd_idx[i]] = acc;
    }
  }
  {
    // per output row of G: arrows and workspace row offsets of its axes
    const int32_t* rowlimit = p.itab + p.off_rowlimit;
    const int32_t* limits = p.itab + p.off_limit;
    const int32_t* lax = p.itab + p.off_lax;
    for (int R = tid; R < nc; R += NT) {
      const int32_t* lm = limits + rowlimit[R] * LM_WORDS;
      const int r = R - lm[LM_OUT0];
      const int naxes = lm[LM_NAXES];
      const int32_t* lx = lax + lm[LM_LAX0] * LX_WORDS;
      const int abase = lm[LM_ARROW_P] + (lm[LM_ARROW_ROWS] == 1 ? 0 : r) * naxes;
      for (int ax = 0; ax < naxes; ++ax) {
        const int rr = lx[ax * LX_WORDS + LX_ROWS] == 1 ? 0 : r;
        rr_arrow[R * AXMAX + ax] = prm[abase + ax];
        rr_voff[R * (AXMAX + 1) + ax] = (lx[ax * LX_WORDS + LX_ROWOFF] + rr) * ldv;
      }
      rr_voff[R * (AXMAX + 1) + AXMAX] = naxes;
    }
  }
  lds_barrier();

  const int32_t* gt = p.itab + p.off_gterm;

  if (P != nullptr && (phases & 2)) {
    // ---- phase 3a: Hessian on the matrix core ---------------------------------
    const int nt = (no + 15) >> 4;  // 16-column tiles per dimension
    const int li = lane & 15, lk = lane >> 4;
    f64x4 acc[TPW];
    int tcol[TPW], trow[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      acc[s] = f64x4{0.0, 0.0, 0.0, 0.0};
      const int t = wave + s * NW;
      trow[s] = t < nt * nt ? t / nt : -1;
      tcol[s] = t < nt * nt ? t - (t / nt) * nt : 0;
    }
    for (int g = 0; g < p.ngterm; ++g) {
      const int32_t* rec = gt + g * GT_WORDS;
      if (!(rec[GT_FLAGS] & GT_FLAG_P)) continue;
      const double w = prm[rec[GT_WPARAM]];
      if (w == 0.0) continue;  // contributes exact zeros
      const int aoff = rec[GT_AOFF], boff = rec[GT_BOFF], nrows = rec[GT_NROWS];
      const unsigned ma = (unsigned)rec[GT_MASKA], mb = (unsigned)rec[GT_MASKB];
      bool on[TPW];
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int ti = trow[s], tj = tcol[s];
        on[s] = __builtin_amdgcn_readfirstlane(
            ti >= 0 && ((ma >> min(ti, 30)) & 1u) && ((mb >> min(tj, 30)) & 1u));
      }
      for (int k0 = 0; k0 < nrows; k0 += 4) {
        const int k = k0 + lk;
        const bool valid = k < nrows;
        const double* arow = V + (aoff + k) * ldv + li;
        const double* brow = V + (boff + k) * ldv + li;
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
          if (on[s]) {
            const double a = valid ? w * arow[trow[s] * 16] : 0.0;
            const double b = valid ? brow[tcol[s] * 16] : 0.0;
            acc[s] = mfma_f64_16x16x4(a, b, acc[s]);
          }
        }
      }
    }
    double* Pb = P + (size_t)inst * no * no;
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      if (trow[s] >= 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = trow[s] * 16 + lk + 4 * reg;
          const int col = tcol[s] * 16 + li;
          if (row < no && col < no) {
            double dP = 0.0, dq = 0.0;
            if (row == col) diagonal_terms(p, prm, row, dP, dq);
            Pb[(size_t)row * no + col] = acc[s][reg] + dP;
          }
        }
      }
    }

  }
  if (P != nullptr && (phases & 4)) {
    // ---- phase 3b: gradient, rows split over the wavefronts ---------------------
    double qa[QSLOT];
#pragma unroll
    for (int c = 0; c < QSLOT; ++c) qa[c] = 0.0;
    for (int g = 0; g < p.ngterm; ++g) {
      const int32_t* rec = gt + g * GT_WORDS;
      if (rec[GT_FLAGS] & GT_FLAG_DIAG) continue;  // added analytically below
      const double w = prm[rec[GT_WPARAM]];
      if (w == 0.0) continue;
      const double aim = prm[rec[GT_AIMPARAM]];
      const double scale = (rec[GT_FLAGS] & GT_FLAG_HALF) ? 0.5 : 1.0;
      const int aoff = rec[GT_AOFF], doff = rec[GT_DOFF], nrows = rec[GT_NROWS];
      for (int k = wave; k < nrows; k += NW) {
        const double r = scale * (V[(doff + k) * ldv + no] - aim);
        const double* arow = V + (aoff + k) * ldv;
#pragma unroll
        for (int c = 0; c < QSLOT; ++c) {
          const int col = lane + 64 * c;
          if (col < no) qa[c] = fma(w * arow[col], r, qa[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < QSLOT; ++c) {
      const int col = lane + 64 * c;
      if (col < no) qpart[wave * L.nop + col] = qa[c];
    }
  }

  if (G != nullptr && (phases & 8)) {
    // ---- phase 3c: constraint rows (K4) -----------------------------------------
    double* Gb = G + (size_t)inst * nc * no;
    if ((no & 1) == 0) {
      const int npair = no >> 1;
      const int total = nc * npair;
      const int dR = NT / npair, dcp = NT - dR * npair;
      int e = tid, R = tid / npair, cp = tid - (tid / npair) * npair;
      double2* G2 = reinterpret_cast<double2*>(Gb);
      while (e < total) {
        const int naxes = rr_voff[R * (AXMAX + 1) + AXMAX];
        double2 accv{0.0, 0.0};
        for (int ax = 0; ax < naxes; ++ax) {
          const double a = rr_arrow[R * AXMAX + ax];
          const double2 v =
              *reinterpret_cast<const double2*>(V + rr_voff[R * (AXMAX + 1) + ax] + 2 * cp);
          accv.x = fma(a, v.x, accv.x);
          accv.y = fma(a, v.y, accv.y);
        }
        G2[e] = accv;
        e += NT;
        cp += dcp;
        R += dR;
        if (cp >= npair) {
          cp -= npair;
          ++R;
        }
      }
    } else {
      const int total = nc * no;
      const int dR = NT / no, dc = NT - dR * no;
      int e = tid, R = tid / no, c = tid - (tid / no) * no;
      while (e < total) {
        const int naxes = rr_voff[R * (AXMAX + 1) + AXMAX];
        double accv = 0.0;
        for (int ax = 0; ax < naxes; ++ax)
          accv = fma(rr_arrow[R * AXMAX + ax], V[rr_voff[R * (AXMAX + 1) + ax] + c], accv);
        Gb[e] = accv;
        e += NT;
        c += dc;
        R += dR;
        if (c >= no) {
          c -= no;
          ++R;
        }
      }
    }
    const int32_t* rowlimit = p.itab + p.off_rowlimit;
    const int32_t* limits = p.itab + p.off_limit;
    double* hb = h + (size_t)inst * nc;
    for (int R = tid; R < nc; R += NT) {
      const int32_t* lm = limits + rowlimit[R] * LM_WORDS;
      const int r = R - lm[LM_OUT0];
      const int naxes = lm[LM_NAXES];
      const int cbase = lm[LM_CENTER_P] + (lm[LM_CENTER_ROWS] == 1 ? 0 : r) * naxes;
      const double extreme = prm[lm[LM_EXTREME_P] + (lm[LM_EXTREME_ROWS] == 1 ? 0 : r)];
      double ac = 0.0, ad = 0.0;
      for (int ax = 0; ax < naxes; ++ax) {
        const double a = rr_arrow[R * AXMAX + ax];
        ac += a * prm[cbase + ax];
        ad = fma(a, V[rr_voff[R * (AXMAX + 1) + ax] + no], ad);
      }
      hb[R] = (extreme + ac) - ad;
    }
  }

  // ---- phase 4: reduce q ------------------------------------------------------------
  if (P != nullptr) {
    lds_barrier();
    double* qb = q + (size_t)inst * no;
    for (int c = tid; c < no; c += NT) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += qpart[w * L.nop + c];
      double dP, dq;
      diagonal_terms(p, prm, c, dP, dq);
      qb[c] = s + dq;
    }
  }
}

}  // namespace

// 0 when the fused kernel cannot take this plan, else its dynamic LDS bytes
size_t fused_lds_bytes(const PlanDev& p, int nw) {
  if (!p.fused_ok) return 0;
  const int nt = (p.no + 15) / 16;
  if (nt * nt > nw * TPW || p.no > 64 * QSLOT || p.max_axes > AXMAX) return 0;
  if ((long)p.rtot * p.ldv > (1 << 20)) return 0;
  return (size_t)fused_layout(p, nw).total_doubles * sizeof(double);
}

int launch_assemble_fused(const PlanDev& p, const SrcTable& src, const double* params,
                          const double* given, double* P, double* q, double* G, double* h,
                          int batch, size_t lds_bytes, hipStream_t stream, hipError_t* err) {
  constexpr int NW = 4;
  auto kernel = fused_assemble_kernel<NW>;
  if (lds_bytes > 64 * 1024) {
    *err = allow_whole_lds(reinterpret_cast<const void*>(kernel));
    if (*err != hipSuccess) return MPCASM_ERR_HIP;
  }
  hipLaunchKernelGGL(kernel, dim3(batch), dim3(NW * 64), lds_bytes, stream, p, src, params, given,
                     P, q, G, h, batch, g_phase_mask);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

}  // namespace mpcasm
