// sweep.hip -- K1 + K2 + K3 + K4 for a dynamics with its own (A_k, B_k) at every step and in every
// instance, x+ = A_k x + B_k u (BASELINE config C5: LTV LIPM, N = 100), WITHOUT a horizon matrix.
//
// Reference semantics: tools.extend_matrices (tools.py:14-33) gives S[k] = Phi(k, 0)^T and
// U_j[k][l] = Phi(k, l+1) B_l[:, j] for one (A, B); "time variant" there means re-extending ONE pair per
// tick (dynamics.py:222-231), so this path is pinned to the reference where all steps share one pair
// and is the oracle's own generalisation beyond (oracle/qp_oracle.py extend_matrices_ltv).  The QP
// blocks are those of body.py:236-329 on such matrices.
//
// The fill + assembly route writes 247 KB of S, U per C5 system to HBM and reads it back.  Here, with
// Phi(k, l) = A_k ... A_l (identity for l > k), row k of an output c . x holds c^T Phi(k, l+1) B_l in
// the column of step l <= k, and everything follows from recursions of n x n matrices:
//     x_k = A_k x_{k-1}                      the free response from the given initial state: d, h
//     Psi_l = W_l + A_{l+1}^T Psi_{l+1} A_{l+1}   W_l = sum w c c^T over the cost rows of step l
//     lam_l = rho_l + A_{l+1}^T lam_{l+1}         rho_l = sum w (c . x_l - aim) c
//     q[(j,l)] = B_l[:,j] . lam_l
//     P[(j,l)][(j',l')] = (Psi_l B_l[:,j]) . u,   u = Phi(l, l'+1) B_l'[:,j']      for l' <= l  (forward sweep)
//                       = B_l[:,j] . z,           z = Phi(l', l+1)^T Psi_l' B_l'[:,j']  for l' > l  (backward sweep)
//     G[line of step l][(j',l')] = arrow (c . u)                                         (forward sweep)
// -- O(n) multiply-adds per element of P and G, every element written once, nothing but the results
// and 8 N (n^2 + n m) bytes of (A_k, B_k) crosses HBM.  One workgroup per instance; a thread owns a
// column (an axis, an input, a step) and carries its u, then its z, in registers; all steps'
// (A_k, B_k) sit in LDS (the "per-step reload" is an LDS read).  The plan states what lets a
// formulation run here (plan.py _sweep_tables): one system shared by up to four axes, every unknown
// one of its inputs, every given value an initial state, every row of a cost or a limit a fixed
// combination of the states of one step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_common.h"
#include "kernels.h"

namespace mpcasm {

namespace {

constexpr int SW_BLOCK = 256;
constexpr int SW_WAVES = SW_BLOCK / 64;

__device__ __forceinline__ double ldsd(const double* base, int i) { return base[i]; }

// LDS of one workgroup (doubles): all steps' [A_k | B_k], the free responses, Psi_l B_l[:, j], q, the
// parameters, a scratch for the recursions' exchanges, the combinations c, then the plan's small
// integer tables (terms, limits, the per-step lists, the axes)
struct SweepLds {
  int ab, xbar, gv, qs, par, scratch, cvec, ints, total;
  int i_term, i_lim, i_cptr, i_cent, i_gptr, i_gent, i_axis, nints;
};
constexpr int LIMW = SW_LIM_WORDS + SW_AXMAX * SW_LAX_WORDS;
__host__ __device__ inline SweepLds sweep_lds(const PlanDev& p) {
  const int n = p.sw_n, m = p.sw_m, N = p.sw_horizon, naxes = p.sw_naxes;
  SweepLds x;
  x.ab = 0;
  x.xbar = x.ab + N * (n * n + n * m);
  x.gv = x.xbar + naxes * N * n;
  x.qs = x.gv + naxes * m * N * n;
  x.par = x.qs + p.no;
  x.scratch = x.par + p.nparams + 1;
  x.cvec = x.scratch + 2 * SW_AXMAX * SW_NMAX * SW_NMAX + SW_AXMAX * SW_NMAX;
  x.cvec += x.cvec & 1;
  x.ints = x.cvec + p.sw_ncvec * SW_NMAX;
  x.i_term = 0;
  x.i_lim = x.i_term + p.sw_nterm * SW_TERM_WORDS;
  x.i_cptr = x.i_lim + p.sw_nlim * LIMW;
  x.i_cent = x.i_cptr + N + 1;
  x.i_gptr = x.i_cent + p.sw_ncent;
  x.i_gent = x.i_gptr + N + 1;
  x.i_gent += x.i_gent & 1;
  x.i_axis = x.i_gent + 2 * p.sw_ngent;
  x.nints = x.i_axis + naxes * SW_AXIS_WORDS;
  x.total = x.ints + (x.nints + 1) / 2;
  x.total += x.total & 1;
  return x;
}

// CPT: columns per thread (no <= SW_BLOCK * CPT)
template <int CPT>
__global__ __launch_bounds__(SW_BLOCK) void ltv_sweep_kernel(
    PlanDev p, const double* __restrict__ sysA, long long strideA, const double* __restrict__ sysB,
    long long strideB, const double* __restrict__ params, const double* __restrict__ given,
    double* __restrict__ P, double* __restrict__ q, double* __restrict__ G, double* __restrict__ h,
    int batch) {
  extern __shared__ __attribute__((aligned(16))) double sw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long inst = blockIdx.x;
  if (inst >= batch) return;
  const int n = p.sw_n, m = p.sw_m, N = p.sw_horizon, naxes = p.sw_naxes, no = p.no, nc = p.nc;
  const int nn = n * n, nm = n * m, abw = nn + nm;
  const SweepLds L = sweep_lds(p);
  double* AB = sw + L.ab;       // [N][n n + n m]: A_k row major, then B_k
  double* xbar = sw + L.xbar;   // [naxes][N][n]
  double* gv = sw + L.gv;       // [naxes][m][N][n]
  double* qs = sw + L.qs;       // [no]
  double* par = sw + L.par;     // [nparams + 1], the last 0.0
  double* scr = sw + L.scratch;
  double* cvec = sw + L.cvec;   // [ncvec][SW_NMAX]
  int* itb = reinterpret_cast<int*>(sw + L.ints);
  const int* terms = itb + L.i_term;
  const int* lims = itb + L.i_lim;
  const int* cptr = itb + L.i_cptr;
  const int* cent = itb + L.i_cent;
  const int* gptr = itb + L.i_gptr;
  const int2* gent = reinterpret_cast<const int2*>(itb + L.i_gent);
  const int* axis = itb + L.i_axis;
  const double* pb = params + (size_t)inst * p.nparams;

  // ---- set-up: every step's (A_k, B_k), the parameters, the plan's tables -----------------------------
  {
    const double* A = sysA + inst * strideA;
    const double* Bm = sysB + inst * strideB;
    for (int e = tid; e < N * nn; e += SW_BLOCK) AB[(e / nn) * abw + e % nn] = A[e];
    for (int e = tid; e < N * nm; e += SW_BLOCK) AB[(e / nm) * abw + nn + e % nm] = Bm[e];
    for (int e = tid; e <= p.nparams; e += SW_BLOCK) par[e] = e < p.nparams ? pb[e] : 0.0;
    for (int e = tid; e < p.sw_ncvec * SW_NMAX; e += SW_BLOCK) cvec[e] = (p.dtab + p.sw_doff_cvec)[e];
    auto copy = [&](int dst, int src_off, int count) {
      for (int e = tid; e < count; e += SW_BLOCK) itb[dst + e] = (p.itab + src_off)[e];
    };
    copy(L.i_term, p.off_sw_term, p.sw_nterm * SW_TERM_WORDS);
    copy(L.i_lim, p.off_sw_lim, p.sw_nlim * LIMW);
    copy(L.i_cptr, p.off_sw_cptr, N + 1);
    copy(L.i_cent, p.off_sw_cent, p.sw_ncent);
    copy(L.i_gptr, p.off_sw_gptr, N + 1);
    copy(L.i_gent, p.off_sw_gent, 2 * p.sw_ngent);
    copy(L.i_axis, p.off_sw_axis, naxes * SW_AXIS_WORDS);
  }
  // this thread's columns: axis, input, step; the diagonal terms on them (a cost on the input itself)
  int ca[CPT], cj[CPT], cl[CPT];
  double dPc[CPT], dqc[CPT];
  const int32_t* colw = p.itab + p.off_sw_col;
#pragma unroll
  for (int t = 0; t < CPT; ++t) {
    const int c = tid + t * SW_BLOCK;
    ca[t] = -1;
    cj[t] = cl[t] = 0;
    dPc[t] = dqc[t] = 0.0;
    if (c < no) {
      const int w = colw[c];
      ca[t] = w & 255;
      cj[t] = (w >> 8) & 255;
      cl[t] = w >> 16;
      diagonal_of_column(p, pb, c, dPc[t], dqc[t]);
    }
  }
  __syncthreads();

  // ---- the free response of every axis: x_k = A_k x_{k-1} (a thread per axis) ---------------------
  if (tid < naxes) {
    double x[SW_NMAX];
#pragma unroll
    for (int i = 0; i < SW_NMAX; ++i) x[i] = i < n ? given[(size_t)inst * p.ng + axis[tid * SW_AXIS_WORDS] + i] : 0.0;
    for (int k = 0; k < N; ++k) {
      const double* Ak = AB + k * abw;
      double y[SW_NMAX];
#pragma unroll
      for (int i = 0; i < SW_NMAX; ++i) {
        y[i] = 0.0;
        if (i < n) {
#pragma unroll
          for (int t = 0; t < SW_NMAX; ++t)
            if (t < n) y[i] = fma(Ak[i * n + t], x[t], y[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < SW_NMAX; ++i) {
        x[i] = y[i];
        if (i < n) xbar[(tid * N + k) * n + i] = y[i];
      }
    }
  }
  __syncthreads();

  // ---- h: (extreme + arrow . center) - arrow . (c . x of the line's step) ---------------------------
  if (G != nullptr)
    for (int e = tid; e < p.sw_ngent; e += SW_BLOCK) {
      const int2 ge = gent[e];
      const int* rec = lims + ge.x * LIMW;
      const int i = ge.y, nax = rec[SL_NAXES];
      double ac = 0.0, ad = 0.0;
      for (int ax = 0; ax < nax; ++ax) {
        const int* xr = rec + SW_LIM_WORDS + ax * SW_LAX_WORDS;
        const double ar = par[xr[SX_ARROW] + i * xr[SX_ARROW_STEP]];
        ac += ar * par[xr[SX_CENTER] + i * xr[SX_CENTER_STEP]];
        const double* xb = xbar + (xr[SX_AXIS] * N + xr[SX_K0] + i * xr[SX_KSTEP]) * n;
        const double* cv = cvec + xr[SX_CVEC];
        double d = 0.0;
        for (int s = 0; s < n; ++s) d = fma(cv[s], xb[s], d);
        ad = fma(ar, d, ad);
      }
      h[(size_t)inst * nc + rec[SL_OUT0] + i] = (par[rec[SL_EXTREME] + i * rec[SL_EXTREME_STEP]] + ac) - ad;
    }

  // ---- backward: Psi_l, lam_l -> gv[a][j][l] = Psi_l B_l[:, j], qs[(a, j, l)] = B_l[:, j] . lam_l --------
  // wavefront 0: lane (a, i, jj) holds Psi[i][jj] of axis a; the lanes jj == 0 also lam[i]; products
  // through the scratch (a wavefront's LDS operations complete in order: no barrier)
  if (wave == 0 && P != nullptr) {
    const int a = lane / nn, e = lane - a * nn, i = e / n, jj = e - i * n;
    const bool live = a < naxes;
    double* sP = scr + (live ? a : 0) * 2 * SW_NMAX * SW_NMAX;   // Psi of this axis, then T = Psi A
    double* sT = sP + SW_NMAX * SW_NMAX;
    double* sL = scr + 2 * SW_AXMAX * SW_NMAX * SW_NMAX + (live ? a : 0) * SW_NMAX;
    double psi = 0.0, lam = 0.0;
    for (int l = N - 1; l >= 0; --l) {
      if (l + 1 < N) {  // Psi <- A_{l+1}^T Psi A_{l+1}, lam <- A_{l+1}^T lam
        const double* An = AB + (l + 1) * abw;
        if (live) {
          sP[i * n + jj] = psi;
          if (jj == 0) sL[i] = lam;
        }
        asm volatile("" ::: "memory");
        double t = 0.0, lnew = 0.0;
        if (live) {
          for (int s = 0; s < n; ++s) t = fma(sP[i * n + s], An[s * n + jj], t);
          sT[i * n + jj] = t;
          if (jj == 0)
            for (int s = 0; s < n; ++s) lnew = fma(An[s * n + i], sL[s], lnew);
        }
        asm volatile("" ::: "memory");
        psi = 0.0;
        if (live)
          for (int s = 0; s < n; ++s) psi = fma(An[s * n + i], sT[s * n + jj], psi);
        lam = lnew;
      }
      // + W_l, rho_l: the cost rows of step l on this lane's axis
      const int e1 = cptr[l + 1];
      for (int ce = cptr[l]; ce < e1; ++ce) {
        const int* tr = terms + cent[ce] * SW_TERM_WORDS;
        if (!live || tr[ST_AXIS] != a) continue;
        const double* cv = cvec + tr[ST_CVEC];
        const double w = par[tr[ST_WPARAM]];
        psi = fma(w * cv[i], cv[jj], psi);
        if (jj == 0) {
          const double* xb = xbar + (a * N + l) * n;
          double d = 0.0;
          for (int s = 0; s < n; ++s) d = fma(cv[s], xb[s], d);
          lam = fma(w * (d - par[tr[ST_AIMPARAM]]), cv[i], lam);
        }
      }
      // gv, qs of this step
      const double* Bl = AB + l * abw + nn;
      if (live) {
        sP[i * n + jj] = psi;
        if (jj == 0) sL[i] = lam;
      }
      asm volatile("" ::: "memory");
      if (live)
        for (int j = jj; j < m; j += n) {  // lane (a, i, j): (Psi B_l[:, j])[i]; the lanes i == 0 also q
          double gsum = 0.0;
          for (int s = 0; s < n; ++s) gsum = fma(sP[i * n + s], Bl[s * m + j], gsum);
          gv[((a * m + j) * N + l) * n + i] = gsum;
          if (i == 0) {
            double qsum = 0.0;
            for (int s = 0; s < n; ++s) qsum = fma(Bl[s * m + j], sL[s], qsum);
            qs[axis[a * SW_AXIS_WORDS + 1 + j] + l] = qsum;
          }
        }
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  if (P != nullptr)
#pragma unroll
    for (int t = 0; t < CPT; ++t) {
      const int c = tid + t * SW_BLOCK;
      if (c < no) q[(size_t)inst * no + c] = qs[c] + dqc[t];
    }

  // ---- forward sweep: u = Phi(l, l'+1) B_l' per column; rows of G of step l, P at and below the diagonal ----
  double* Pb = P + (size_t)inst * no * no;
  double* Gb = G + (size_t)inst * nc * no;
  double u[CPT][SW_NMAX];
#pragma unroll
  for (int t = 0; t < CPT; ++t)
#pragma unroll
    for (int i = 0; i < SW_NMAX; ++i) u[t][i] = 0.0;
  for (int l = 0; l < N; ++l) {
    const double* Al = AB + l * abw;
    const double* Bl = Al + nn;
#pragma unroll
    for (int t = 0; t < CPT; ++t) {
      if (ca[t] < 0) continue;
      if (cl[t] == l) {
#pragma unroll
        for (int i = 0; i < SW_NMAX; ++i) u[t][i] = i < n ? Bl[i * m + cj[t]] : 0.0;
      } else if (cl[t] < l) {
        double y[SW_NMAX];
#pragma unroll
        for (int i = 0; i < SW_NMAX; ++i) {
          y[i] = 0.0;
          if (i < n) {
#pragma unroll
            for (int s = 0; s < SW_NMAX; ++s)
              if (s < n) y[i] = fma(Al[i * n + s], u[t][s], y[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < SW_NMAX; ++i) u[t][i] = y[i];
      }
    }
    if (P != nullptr)
      for (int a = 0; a < naxes; ++a)
        for (int j = 0; j < m; ++j) {
          const double* g = gv + ((a * m + j) * N + l) * n;
          const int r = axis[a * SW_AXIS_WORDS + 1 + j] + l;
#pragma unroll
          for (int t = 0; t < CPT; ++t) {
            const int c = tid + t * SW_BLOCK;
            if (ca[t] < 0) continue;
            if (ca[t] != a) {
              Pb[(size_t)r * no + c] = 0.0;
            } else if (cl[t] <= l) {
              double v = 0.0;
#pragma unroll
              for (int s = 0; s < SW_NMAX; ++s)
                if (s < n) v = fma(g[s], u[t][s], v);
              if (c == r) v += dPc[t];
              Pb[(size_t)r * no + c] = v;
            }
          }
        }
    if (G != nullptr) {
      const int e1 = gptr[l + 1];
      for (int ge = gptr[l]; ge < e1; ++ge) {
        const int2 line = gent[ge];
        const int* rec = lims + line.x * LIMW;
        const int i = line.y, nax = rec[SL_NAXES];
        double val[CPT];
#pragma unroll
        for (int t = 0; t < CPT; ++t) val[t] = 0.0;
        for (int ax = 0; ax < nax; ++ax) {
          const int* xr = rec + SW_LIM_WORDS + ax * SW_LAX_WORDS;
          const double ar = par[xr[SX_ARROW] + i * xr[SX_ARROW_STEP]];
          const double* cv = cvec + xr[SX_CVEC];
#pragma unroll
          for (int t = 0; t < CPT; ++t) {
            double d = 0.0;
#pragma unroll
            for (int s = 0; s < SW_NMAX; ++s)
              if (s < n) d = fma(cv[s], u[t][s], d);
            val[t] += ca[t] == xr[SX_AXIS] ? ar * d : 0.0;
          }
        }
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
          const int c = tid + t * SW_BLOCK;
          if (c < no) Gb[(size_t)(rec[SL_OUT0] + i) * no + c] = val[t];
        }
      }
    }
  }
  if (P == nullptr) return;

  // ---- backward sweep: z = Phi(l', l+1)^T Psi_l' B_l' per column; P above the diagonal ---------------
  double z[CPT][SW_NMAX];
#pragma unroll
  for (int t = 0; t < CPT; ++t)
#pragma unroll
    for (int i = 0; i < SW_NMAX; ++i) z[t][i] = 0.0;
  for (int l = N - 1; l >= 0; --l) {
    const double* An = AB + (l + 1 < N ? l + 1 : l) * abw;
    const double* Bl = AB + l * abw + nn;
#pragma unroll
    for (int t = 0; t < CPT; ++t) {
      if (ca[t] < 0) continue;
      if (cl[t] == l) {
        const double* g = gv + ((ca[t] * m + cj[t]) * N + l) * n;
#pragma unroll
        for (int i = 0; i < SW_NMAX; ++i) z[t][i] = i < n ? g[i] : 0.0;
      } else if (cl[t] > l) {
        double y[SW_NMAX];
#pragma unroll
        for (int i = 0; i < SW_NMAX; ++i) {
          y[i] = 0.0;
          if (i < n) {
#pragma unroll
            for (int s = 0; s < SW_NMAX; ++s)
              if (s < n) y[i] = fma(An[s * n + i], z[t][s], y[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < SW_NMAX; ++i) z[t][i] = y[i];
      }
    }
    for (int a = 0; a < naxes; ++a)
      for (int j = 0; j < m; ++j) {
        const int r = axis[a * SW_AXIS_WORDS + 1 + j] + l;
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
          const int c = tid + t * SW_BLOCK;
          if (ca[t] != a || cl[t] <= l) continue;
          double v = 0.0;
#pragma unroll
          for (int s = 0; s < SW_NMAX; ++s)
            if (s < n) v = fma(Bl[s * m + j], z[t][s], v);
          Pb[(size_t)r * no + c] = v;
        }
      }
  }
}

}  // namespace

bool sweep_eligible(const PlanDev& p) { return p.sw_ok != 0; }

int launch_assemble_sweep(const PlanDev& p, const SrcTable& src, const double* params,
                          const double* given, double* P, double* q, double* G, double* h, int batch,
                          hipStream_t stream, hipError_t* err) {
  const int n = p.sw_n, m = p.sw_m, naxes = p.sw_naxes;
  if (n < 1 || n > SW_NMAX || m < 1 || m > SW_MMAX || naxes < 1 || naxes > SW_AXMAX ||
      naxes * n * n > 64 || p.no > SW_BLOCK * 4)
    return MPCASM_ERR_LIMIT;
  const size_t lds = (size_t)sweep_lds(p).total * sizeof(double);
  if (lds > (size_t)RESIDENT_LDS_LIMIT) return MPCASM_ERR_LIMIT;
  const double* A = src.ptr[p.sw_src_a];
  const double* Bm = src.ptr[p.sw_src_b];
  const long long sa = src.stride[p.sw_src_a], sb = src.stride[p.sw_src_b];
#define MPCASM_SWEEP_CASE(CPT)                                                                         \
  if (p.no <= SW_BLOCK * CPT) {                                                                        \
    auto kernel = ltv_sweep_kernel<CPT>;                                                               \
    if (lds > 64 * 1024) {                                                                             \
      *err = allow_whole_lds(reinterpret_cast<const void*>(kernel));                                   \
      if (*err != hipSuccess) return MPCASM_ERR_HIP;                                                   \
    }                                                                                                  \
    hipLaunchKernelGGL(kernel, dim3((unsigned)batch), dim3(SW_BLOCK), lds, stream, p, A, sa, Bm, sb,   \
                       params, given, P, q, G, h, batch);                                              \
    *err = hipGetLastError();                                                                          \
    return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;                                            \
  }
  MPCASM_SWEEP_CASE(1)
  MPCASM_SWEEP_CASE(2)
  MPCASM_SWEEP_CASE(4)
#undef MPCASM_SWEEP_CASE
  return MPCASM_ERR_LIMIT;
}

}  // namespace mpcasm
