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
// column (an axis, an input, a step) -- two neighbouring ones where the width allows (PAIR: its two elements
// of a row leave as one 16-byte store, which is what this kernel's time follows: the width of a store and
// where the results lie, not its instruction count) -- and carries its u, then its z, in registers; it
// writes the rows of its OWN axis (the blocks of P between axes are zeros, written as such); all steps'
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

// LDS of one workgroup (doubles): all steps' [A_k | B_k], the free responses, Psi_l B_l[:, j], lam_l, the
// parameters, the combinations c, the weights arrow * c of the lines of G -- per limit and axis when no
// limit has an arrow of its own per line (`per_line` 0: 0.5 KB for C5; 24 KB in all, six workgroups per CU), else
// per line and axis (26 KB) --, then small integer tables: the terms, the per-step list of G's lines
// (one word per line: row of G | limit << 20), the axes
struct SweepLds {
  int ab, xbar, gv, lam, par, cvec, lw, ints, total;
  int i_term, i_gptr, i_axis, i_lword, nints;
};
constexpr int LIMW = SW_LIM_WORDS + SW_AXMAX * SW_LAX_WORDS;
constexpr int SW_TERMS_REG = 8;   // cost terms the recursion waves keep in registers
constexpr int SW_LINES_REG = 8;   // lines of G of one step whose words are fetched a step ahead
// [A_k | B_k] of a step: n n + n m doubles, rounded up to whole 16-byte pieces; the vectors the sweeps read
// per step (Psi_l B_l[:, j], the weights of a line of G) are padded to SW_NMAX = 4 doubles: two 16-byte reads
__host__ __device__ inline int sweep_abw(int n, int m) { return (n * n + n * m + 1) & ~1; }
__host__ __device__ inline SweepLds sweep_lds(const PlanDev& p, int per_line) {
  const int n = p.sw_n, m = p.sw_m, N = p.sw_horizon, naxes = p.sw_naxes;
  SweepLds x;
  x.ab = 0;
  x.gv = x.ab + N * sweep_abw(n, m);
  x.lw = x.gv + naxes * m * N * SW_NMAX;
  x.xbar = x.lw + (per_line ? p.sw_ngent : p.sw_nlim) * naxes * SW_NMAX;
  x.lam = x.xbar;   // (rho_l, then lam_l, take x_l's place once h has been written: one wavefront does all three)
  x.par = x.lam + naxes * N * n;
  x.cvec = x.par + p.nparams + 1;
  x.cvec += x.cvec & 1;
  x.ints = x.cvec + p.sw_ncvec * SW_NMAX;
  x.ints += x.ints & 1;
  x.i_term = 0;
  x.i_gptr = x.i_term + p.sw_nterm * SW_TERM_WORDS;
  x.i_axis = x.i_gptr + N + 2;
  x.i_lword = x.i_axis + naxes * SW_AXIS_WORDS;
  x.nints = x.i_lword + p.sw_ngent + SW_LINES_REG;   // (read ahead behind the last line)
  x.total = x.ints + (x.nints + 1) / 2;
  x.total += x.total & 1;
  return x;
}

// value of lane `src` (same for a double's two halves)
__device__ __forceinline__ double lane_value(double v, int src) { return __shfl(v, src, 64); }

// CPT: columns per thread (no <= SW_BLOCK * CPT); NS, MS, AS: the system's states, inputs and the axes
// as constants (0: read from the plan) -- the loops over them unroll without guards, which is a
// third of the instructions of a step for the LIPM family (n = 3, m = 1, two axes: C5)
// PAIR (with CPT = 2): a thread's two columns are NEIGHBOURS (2 tid, 2 tid + 1) instead of SW_BLOCK apart: the two
// elements it computes of a row of G or P leave as ONE 16-byte store.
template <int CPT, int NS, int MS, int AS, bool PAIR = false>
__global__ __launch_bounds__(SW_BLOCK) void ltv_sweep_kernel(
    PlanDev p, const double* __restrict__ sysA, long long strideA, const double* __restrict__ sysB,
    long long strideB, const double* __restrict__ params, const double* __restrict__ given,
    double* __restrict__ P, double* __restrict__ q, double* __restrict__ G, double* __restrict__ h,
    int batch, int per_line, int reg_lines) {
  extern __shared__ __attribute__((aligned(16))) double sw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long inst = blockIdx.x;
  if (inst >= batch) return;
  const int n = NS ? NS : p.sw_n, m = MS ? MS : p.sw_m, naxes = AS ? AS : p.sw_naxes;
  const int N = p.sw_horizon, no = p.no, nc = p.nc;
  const int nn = n * n, nm = n * m, abw = sweep_abw(n, m);
  const SweepLds L = sweep_lds(p, per_line);
  double* AB = sw + L.ab;       // [N][n n + n m]: A_k row major, then B_k
  double* xbar = sw + L.xbar;   // [naxes][N][n]
  double* gv = sw + L.gv;       // [N][naxes][m][SW_NMAX]: what a step's sweeps read lies side by side
  double* lamT = sw + L.lam;    // [naxes][N][n]
  double* par = sw + L.par;     // [nparams + 1], the last 0.0
  double* cvec = sw + L.cvec;   // [ncvec][SW_NMAX]
  double* lw = sw + L.lw;       // [lines or limits][naxes][SW_NMAX]
  int* itb = reinterpret_cast<int*>(sw + L.ints);
  const int* terms = itb + L.i_term;
  const int* gptr = itb + L.i_gptr;
  const int* axis = itb + L.i_axis;
  int* lword = itb + L.i_lword;  // [lines + SW_LINES_REG] row of G | limit << 20
  // (the limits' records and the list of lines are read where threads work side by side: from the plan)
  const int32_t* lims = p.itab + p.off_sw_lim;
  const int2* gent = reinterpret_cast<const int2*>(p.itab + p.off_sw_gent);
  const double* pb = params + (size_t)inst * p.nparams;
  const int nlines = p.sw_ngent;

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
    copy(L.i_gptr, p.off_sw_gptr, N + 1);
    if (tid == 0) itb[L.i_gptr + N + 1] = nlines;   // (read a step ahead)
    copy(L.i_axis, p.off_sw_axis, naxes * SW_AXIS_WORDS);
  }
  // this thread's columns: axis, input, step; the diagonal terms on them (a cost on the input itself)
  int ca[CPT], cj[CPT], cl[CPT];
  double dPc[CPT], dqc[CPT];
  const int32_t* colw = p.itab + p.off_sw_col;
#pragma unroll
  for (int t = 0; t < CPT; ++t) {
    const int c = PAIR ? tid * CPT + t : tid + t * SW_BLOCK;
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

  // ---- the recursions, side by side on two wavefronts; the others lay out the lines of G -------------
  // Lane (a, i, jj) = a 16 + i 4 + jj: a quad holds a row i of an n x n matrix of axis a, so "row i times
  // a column" is four quad broadcasts (DPP) and "column i of A^T times ..." four lane reads -- no trip
  // through LDS in the dependent chain; the step's A comes out of LDS a step ahead.
  const int ra = lane >> 4, ri = (lane >> 2) & 3, rj = lane & 3;
  const bool rlive = ra < naxes && ri < n && rj < n;
  if (wave == 0 && P != nullptr) {
    // Psi_l = W_l + A_{l+1}^T Psi_{l+1} A_{l+1};  gv[a][j][l] = Psi_l B_l[:, j]
    // W_l: the terms that cover the whole horizon are one constant per lane (w c[i] c[jj] summed); the
    // others (a schedule) are walked downwards with a counter -- `tnext` the next step that has a row,
    // `tleft` how many are left: no search and no division per step.
    double wall = 0.0, wcc[SW_TERMS_REG];
    int tnext[SW_TERMS_REG], tks[SW_TERMS_REG], tleft[SW_TERMS_REG];
    int npart = 0;
    for (int t = 0; t < p.sw_nterm; ++t) {
      const int* tr = terms + t * SW_TERM_WORDS;
      const bool mine = rlive && tr[ST_AXIS] == ra;
      const double* cv = cvec + tr[ST_CVEC];
      const double v = mine ? (par[tr[ST_WPARAM]] * cv[ri]) * cv[rj] : 0.0;
      const bool whole = tr[ST_K0] == 0 && tr[ST_KSTEP] == 1 && tr[ST_COUNT] == N;   // (wave-uniform)
      if (whole) {
        wall += v;
      } else {
#pragma unroll
        for (int k = 0; k < SW_TERMS_REG; ++k)
          if (k == npart) {
            wcc[k] = v;
            tks[k] = tr[ST_KSTEP];
            tleft[k] = tr[ST_COUNT];
            tnext[k] = tr[ST_K0] + (tr[ST_COUNT] - 1) * tr[ST_KSTEP];
          }
        ++npart;
      }
    }
    const bool in_regs = npart <= SW_TERMS_REG;
#pragma unroll
    for (int k = 0; k < SW_TERMS_REG; ++k)
      if (k >= npart) {
        wcc[k] = 0.0;
        tks[k] = tleft[k] = tnext[k] = 0;
      }
    double psi = 0.0;
    for (int l = N - 1; l >= 0; --l) {
      if (l + 1 < N) {
        const double* An = AB + (l + 1) * abw;
        double acol[SW_NMAX], arow[SW_NMAX];   // A[s][jj], A[s][i]
#pragma unroll
        for (int s_ = 0; s_ < SW_NMAX; ++s_) {
          acol[s_] = (rlive && s_ < n) ? An[s_ * n + rj] : 0.0;
          arow[s_] = (rlive && s_ < n) ? An[s_ * n + ri] : 0.0;
        }
        // T[i][jj] = sum_s Psi[i][s] A[s][jj]
        double t_ = quad_broadcast<0>(psi) * acol[0];
        t_ = fma(quad_broadcast<1>(psi), acol[1], t_);
        t_ = fma(quad_broadcast<2>(psi), acol[2], t_);
        t_ = fma(quad_broadcast<3>(psi), acol[3], t_);
        // Psi'[i][jj] = sum_s A[s][i] T[s][jj]
        const int base = (lane & ~15) | rj;
        double nw = arow[0] * lane_value(t_, base);
        nw = fma(arow[1], lane_value(t_, base + 4), nw);
        nw = fma(arow[2], lane_value(t_, base + 8), nw);
        nw = fma(arow[3], lane_value(t_, base + 12), nw);
        psi = rlive ? nw : 0.0;
      }
      psi += wall;
      if (in_regs) {
#pragma unroll
        for (int t = 0; t < SW_TERMS_REG; ++t) {
          if (t >= npart) break;
          const bool on = tleft[t] > 0 && l == tnext[t];
          const int take = !on ? 0 : (tks[t] == 0 ? tleft[t] : 1);   // (rows at one step: all of them)
          psi = fma(wcc[t], (double)take, psi);
          tleft[t] -= take;
          tnext[t] -= on ? tks[t] : 0;
        }
      } else {
        for (int t = 0; t < p.sw_nterm; ++t) {
          const int* tr = terms + t * SW_TERM_WORDS;
          const int dk = l - tr[ST_K0], ks = tr[ST_KSTEP], cnt = tr[ST_COUNT];
          if (tr[ST_K0] == 0 && ks == 1 && cnt == N) continue;       // (in `wall`)
          bool on;
          if (ks <= 1)
            on = ks == 1 ? (dk >= 0 && dk < cnt) : dk == 0;
          else
            on = dk >= 0 && dk % ks == 0 && dk / ks < cnt;
          if (!on || !rlive || tr[ST_AXIS] != ra) continue;
          const double* cv = cvec + tr[ST_CVEC];
          psi = fma((par[tr[ST_WPARAM]] * cv[ri]) * cv[rj], ks == 0 ? (double)cnt : 1.0, psi);
        }
      }
      // gv[a][j][l][i] = sum_s Psi[i][s] B_l[s][j]: lane (a, i, jj) takes the inputs j = jj, jj + n, ...
      const double* Bl = AB + l * abw + nn;
      const double p0 = quad_broadcast<0>(psi), p1 = quad_broadcast<1>(psi), p2 = quad_broadcast<2>(psi),
                   p3 = quad_broadcast<3>(psi);
      if (rlive)
        for (int j = rj; j < m; j += n) {
          double gsum = p0 * Bl[j];
          if (n > 1) gsum = fma(p1, Bl[m + j], gsum);
          if (n > 2) gsum = fma(p2, Bl[2 * m + j], gsum);
          if (n > 3) gsum = fma(p3, Bl[3 * m + j], gsum);
          gv[((l * naxes + ra) * m + j) * SW_NMAX + ri] = gsum;
        }
    }
  } else if (wave == 1) {
    // lane (a, i, 0): the free response x_k = A_k x_{k-1}; then rho_l for every step at once (lanes over
    // (axis, step)); then lam_l = rho_l + A_{l+1}^T lam_{l+1}, step by step on lane (a, i, 0)
    const bool xl = rlive && rj == 0;
    const int base = lane & ~15;   // x[s], lam[s] live in lane (a, s, 0)
    double x = xl ? given[(size_t)inst * p.ng + axis[ra * SW_AXIS_WORDS] + ri] : 0.0;
    for (int k = 0; k < N; ++k) {
      const double* Ak = AB + k * abw;
      double arow[SW_NMAX];   // A[i][s]
#pragma unroll
      for (int s_ = 0; s_ < SW_NMAX; ++s_) arow[s_] = (xl && s_ < n) ? Ak[ri * n + s_] : 0.0;
      double y = arow[0] * lane_value(x, base);
      y = fma(arow[1], lane_value(x, base + 4), y);
      y = fma(arow[2], lane_value(x, base + 8), y);
      y = fma(arow[3], lane_value(x, base + 12), y);
      x = xl ? y : 0.0;
      if (xl) xbar[(ra * N + k) * n + ri] = x;
    }
    asm volatile("" ::: "memory");   // (a wavefront's LDS operations complete in order: xbar is there)
    // h: (extreme + arrow . center) - arrow . (c . x of the line's step) -- here, by this wavefront, because
    // rho and lam below overwrite x (one buffer for the three: 4.8 KB of LDS less for C5, a sixth workgroup
    // per CU)
    if (G != nullptr)
      for (int e = lane; e < nlines; e += 64) {
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
          for (int s_ = 0; s_ < n; ++s_) d = fma(cv[s_], xb[s_], d);
          ad = fma(ar, d, ad);
        }
        h[(size_t)inst * nc + rec[SL_OUT0] + i] = (par[rec[SL_EXTREME] + i * rec[SL_EXTREME_STEP]] + ac) - ad;
      }
    if (P != nullptr) {
      asm volatile("" ::: "memory");
      // rho_l[i] = sum over the cost rows of step l on the axis of w (c . x_l - aim) c[i] -> lamT, in x_l's place
      // (a lane reads the x of its own (axis, step) and nothing else's)
      for (int e = lane; e < naxes * N; e += 64) {
        const int a = e / N, l = e - a * N;
        const double* xb = xbar + e * n;
        double rho[SW_NMAX];
#pragma unroll
        for (int i = 0; i < SW_NMAX; ++i) rho[i] = 0.0;
        for (int t = 0; t < p.sw_nterm; ++t) {
          const int* tr = terms + t * SW_TERM_WORDS;
          const int dk = l - tr[ST_K0], ks = tr[ST_KSTEP], cnt = tr[ST_COUNT];
          bool on;
          if (ks <= 1)                      // (the usual schedules: no division)
            on = ks == 1 ? (dk >= 0 && dk < cnt) : dk == 0;
          else
            on = dk >= 0 && dk % ks == 0 && dk / ks < cnt;
          if (!on || tr[ST_AXIS] != a) continue;
          const double* cv = cvec + tr[ST_CVEC];
          double d = 0.0;
          for (int s_ = 0; s_ < n; ++s_) d = fma(cv[s_], xb[s_], d);
          const double wr = par[tr[ST_WPARAM]] * (ks == 0 ? (double)cnt : 1.0) * (d - par[tr[ST_AIMPARAM]]);
#pragma unroll
          for (int i = 0; i < SW_NMAX; ++i)
            if (i < n) rho[i] = fma(wr, cv[i], rho[i]);
        }
#pragma unroll
        for (int i = 0; i < SW_NMAX; ++i)
          if (i < n) lamT[e * n + i] = rho[i];
      }
      asm volatile("" ::: "memory");
      double lam = 0.0;
      for (int l = N - 1; l >= 0; --l) {
        double acc = xl ? lamT[(ra * N + l) * n + ri] : 0.0;
        if (l + 1 < N) {
          const double* An = AB + (l + 1) * abw;
          double acl[SW_NMAX];   // A[s][i]
#pragma unroll
          for (int s_ = 0; s_ < SW_NMAX; ++s_) acl[s_] = (xl && s_ < n) ? An[s_ * n + ri] : 0.0;
          acc = fma(acl[0], lane_value(lam, base), acc);
          acc = fma(acl[1], lane_value(lam, base + 4), acc);
          acc = fma(acl[2], lane_value(lam, base + 8), acc);
          acc = fma(acl[3], lane_value(lam, base + 12), acc);
        }
        lam = xl ? acc : 0.0;
        if (xl) lamT[(ra * N + l) * n + ri] = lam;
      }
    }
  }
  if (G != nullptr && wave >= 2) {
    // the lines of G in the order of the steps: a word each (row of G | limit << 20); the weights
    // arrow * c per limit and axis -- or, where a limit's arrow changes from line to line, per line
    const int wt = tid - 128, WT = SW_BLOCK - 128;
    for (int e = wt; e < nlines + SW_LINES_REG; e += WT) {
      const int2 ge = gent[e < nlines ? e : nlines - 1];
      const int* rec = lims + ge.x * LIMW;
      lword[e] = (rec[SL_OUT0] + ge.y) | (ge.x << 20);
      if (per_line && e < nlines) {
        for (int a = 0; a < naxes; ++a)
          for (int s_ = 0; s_ < SW_NMAX; ++s_) lw[(e * naxes + a) * SW_NMAX + s_] = 0.0;
        for (int ax = 0; ax < rec[SL_NAXES]; ++ax) {
          const int* xr = rec + SW_LIM_WORDS + ax * SW_LAX_WORDS;
          const double ar = par[xr[SX_ARROW] + ge.y * xr[SX_ARROW_STEP]];
          const double* cv = cvec + xr[SX_CVEC];
          for (int s_ = 0; s_ < n; ++s_) lw[(e * naxes + xr[SX_AXIS]) * SW_NMAX + s_] += ar * cv[s_];
        }
      }
    }
    if (!per_line)
      for (int li = wt; li < p.sw_nlim; li += WT) {
        const int* rec = lims + li * LIMW;
        for (int a = 0; a < naxes; ++a)
          for (int s_ = 0; s_ < SW_NMAX; ++s_) lw[(li * naxes + a) * SW_NMAX + s_] = 0.0;
        for (int ax = 0; ax < rec[SL_NAXES]; ++ax) {
          const int* xr = rec + SW_LIM_WORDS + ax * SW_LAX_WORDS;
          const double ar = par[xr[SX_ARROW]];
          const double* cv = cvec + xr[SX_CVEC];
          for (int s_ = 0; s_ < n; ++s_) lw[(li * naxes + xr[SX_AXIS]) * SW_NMAX + s_] += ar * cv[s_];
        }
      }
  }
  __syncthreads();

  // ---- q = B_l[:, j] . lam_l (h: written by the wavefront that ran the free response, above) ---------
  if (P != nullptr)
#pragma unroll
    for (int t = 0; t < CPT; ++t) {
      const int c = PAIR ? tid * CPT + t : tid + t * SW_BLOCK;
      if (c < no) {
        const double* Bl = AB + cl[t] * abw + nn;
        const double* lm = lamT + (ca[t] * N + cl[t]) * n;
        double acc = dqc[t];
        for (int s_ = 0; s_ < n; ++s_) acc = fma(Bl[s_ * m + cj[t]], lm[s_], acc);
        q[(size_t)inst * no + c] = acc;
      }
    }

  double* Pb = P + (size_t)inst * no * no;
  double* Gb = G + (size_t)inst * nc * no;
  // ---- the blocks of P between different axes: zeros (no cost couples two axes here), written as such --
  // 16 bytes per store, no arithmetic -- instead of being "computed" element by element in the sweeps
  if (P != nullptr && naxes > 1) {
    const bool wide = (N & 1) == 0 && (no & 1) == 0;   // (pairs of columns stay inside a row, 16-byte aligned)
    for (int a = 0; a < naxes; ++a)
      for (int j = 0; j < m; ++j)
        for (int a2 = 0; a2 < naxes; ++a2) {
          if (a2 == a) continue;
          for (int j2 = 0; j2 < m; ++j2) {
            const int col0 = axis[a2 * SW_AXIS_WORDS + 1 + j2];
            double* blk = Pb + (size_t)axis[a * SW_AXIS_WORDS + 1 + j] * no + col0;
            // (plain stores: a block's rows end inside cache lines the neighbouring block completes)
            if (wide && (col0 & 1) == 0)
              for (int e = tid * 2; e < N * N; e += SW_BLOCK * 2) {
                const int r = e / N, c = e - r * N;
                *reinterpret_cast<double2*>(blk + (size_t)r * no + c) = double2{0.0, 0.0};
              }
            else
              for (int e = tid; e < N * N; e += SW_BLOCK) {
                const int r = e / N, c = e - r * N;
                blk[(size_t)r * no + c] = 0.0;
              }
          }
        }
  }

  // ---- forward sweep: u = Phi(l, l'+1) B_l' per column; rows of G of step l, P at and below the diagonal ----
  // (branch-free per step: a thread's column is "not yet" (u = 0), "now" (u = B_l[:, j]) or "running"
  // (u = A_l u) by selects; threads without a column leave here).  Everything a step needs comes out of
  // LDS in 16-byte pieces up front -- the step's [A | B] (the same for every lane), the lane's own axis'
  // Psi_l B_l[:, j] and the weights of the step's lines of G -- and a thread writes the row of its OWN axis.
  if ((CPT == 1 && tid >= no) || (PAIR && tid * CPT >= no)) return;
  double u[CPT][SW_NMAX];
  double* prow[CPT][SW_MMAX];      // P[(own axis, j, step l)][own column], stepping down a row per step
  int goff[CPT];                   // the lane's axis' slice of a step of gv, of a slot of lw (doubles)
#pragma unroll
  for (int t = 0; t < CPT; ++t) {
    const int c = PAIR ? tid * CPT + t : tid + t * SW_BLOCK;
    const int cat = ca[t] < 0 ? 0 : ca[t];
    goff[t] = cat * m * SW_NMAX;
#pragma unroll
    for (int i = 0; i < SW_NMAX; ++i) u[t][i] = 0.0;
#pragma unroll
    for (int j = 0; j < SW_MMAX; ++j)
      prow[t][j] = Pb + (size_t)(j < m ? axis[cat * SW_AXIS_WORDS + 1 + j] : 0) * no + (c < no ? c : 0);
  }
  // Regular lines (the launcher finds them in the plan's tables): the first `reg_lines` lines of EVERY step belong
  // to the same limits in the same order and sit one row further down G per step -- limits on every step of the
  // horizon, C5's 4 per step.  Their weights (per limit and axis) are the same at every step: in registers for
  // the whole sweep; their rows: pointers stepped by one row -- no word, no weight, no multiplication per step.
  constexpr int RLMAX = 4;
  double wreg[RLMAX][CPT][SW_NMAX];
  double* gp[RLMAX];
#pragma unroll
  for (int x = 0; x < RLMAX; ++x) {
    const int word = (G != nullptr && x < reg_lines) ? __builtin_amdgcn_readfirstlane(lword[x]) : 0;
    gp[x] = Gb + (size_t)(word & 0xFFFFF) * no;
#pragma unroll
    for (int t = 0; t < CPT; ++t) {
      const double* wp = lw + (size_t)(word >> 20) * naxes * SW_NMAX + (goff[t] / m);
#pragma unroll
      for (int s_ = 0; s_ < SW_NMAX; ++s_) wreg[x][t][s_] = (x < reg_lines && s_ < n) ? wp[s_] : 0.0;
    }
  }
  constexpr int LR = CPT == 1 ? SW_LINES_REG : (CPT == 2 ? 4 : 2);   // lines of a step in registers
  int wcur[LR];   // the words of the step's first lines (wave-uniform)
#pragma unroll
  for (int x = 0; x < LR; ++x) wcur[x] = G != nullptr ? __builtin_amdgcn_readfirstlane(lword[x]) : 0;
  int e1 = G != nullptr ? __builtin_amdgcn_readfirstlane(gptr[1]) : 0, e0 = 0;
  for (int l = 0; l < N; ++l) {
    const double* Al = AB + l * abw;
    const double* Bl = Al + nn;
    const int e2 = gptr[l + 2];                 // (the end of the NEXT step's lines: a step ahead)
    double am[SW_NMAX][SW_NMAX];
#pragma unroll
    for (int i = 0; i < SW_NMAX; ++i)
#pragma unroll
      for (int s_ = 0; s_ < SW_NMAX; ++s_) am[i][s_] = (i < n && s_ < n) ? Al[i * n + s_] : 0.0;
#pragma unroll
    for (int t = 0; t < CPT; ++t) {
      double bnow[SW_NMAX], y[SW_NMAX];
#pragma unroll
      for (int i = 0; i < SW_NMAX; ++i) {
        bnow[i] = i < n ? Bl[i * m + cj[t]] : 0.0;
        asm volatile("" : "+v"(bnow[i]));       // (read by every lane, not only by the ones that start now)
      }
#pragma unroll
      for (int i = 0; i < SW_NMAX; ++i) {
        y[i] = 0.0;
        if (i < n) {
#pragma unroll
          for (int s_ = 0; s_ < SW_NMAX; ++s_)
            if (s_ < n) y[i] = fma(am[i][s_], u[t][s_], y[i]);
          y[i] = cl[t] == l ? bnow[i] : y[i];   // (u is 0 before its step: A_l 0 = 0)
        }
      }
#pragma unroll
      for (int i = 0; i < SW_NMAX; ++i) u[t][i] = y[i];
    }
    if (P != nullptr) {
      double pv[CPT][SW_MMAX];
#pragma unroll
      for (int t = 0; t < CPT; ++t) {
        const double* gl = gv + (size_t)l * naxes * m * SW_NMAX + goff[t];
#pragma unroll
        for (int j = 0; j < SW_MMAX; ++j) {
          double v = (cl[t] == l && cj[t] == j) ? dPc[t] : 0.0;
#pragma unroll
          for (int s_ = 0; s_ < SW_NMAX; ++s_)
            if (s_ < n && j < m) v = fma(gl[j * SW_NMAX + s_], u[t][s_], v);
          pv[t][j] = v;
        }
      }
      // this axis: at and below the diagonal now, above it in the backward sweep (plain stores: the
      // backward sweep completes the line)
#pragma unroll
      for (int j = 0; j < SW_MMAX; ++j) {
        if (j >= m) break;
        if constexpr (PAIR) {
          const bool m0 = cl[0] <= l, m1 = cl[1] <= l;
          if (m0 && m1 && prow[1][j] == prow[0][j] + 1) {
            *reinterpret_cast<double2*>(prow[0][j]) = double2{pv[0][j], pv[1][j]};
          } else {
            if (m0) *prow[0][j] = pv[0][j];
            if (m1) *prow[1][j] = pv[1][j];
          }
        } else {
#pragma unroll
          for (int t = 0; t < CPT; ++t) {
            const int c = tid + t * SW_BLOCK;
            if ((CPT == 1 || c < no) && cl[t] <= l) *prow[t][j] = pv[t][j];
          }
        }
#pragma unroll
        for (int t = 0; t < CPT; ++t) prow[t][j] += no;
      }
    }
    if (G != nullptr && reg_lines > 0) {
#pragma unroll
      for (int x = 0; x < RLMAX; ++x) {
        if (x >= reg_lines) break;
        double vv[CPT];
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
          vv[t] = 0.0;
#pragma unroll
          for (int s_ = 0; s_ < SW_NMAX; ++s_)
            if (s_ < n) vv[t] = fma(wreg[x][t][s_], u[t][s_], vv[t]);
        }
        if constexpr (PAIR) {
          // (an even width: the pair lies in the row or not at all, 16-byte aligned)
          store_result(reinterpret_cast<double2*>(gp[x] + tid * 2), double2{vv[0], vv[1]});
        } else {
#pragma unroll
          for (int t = 0; t < CPT; ++t) {
            const int c = tid + t * SW_BLOCK;
            if (CPT == 1 || c < no) store_result(gp[x] + c, vv[t]);
          }
        }
        gp[x] += no;
      }
      // (what a step holds beyond its regular lines: one by one)
      for (int e = e0 + reg_lines; e < e1; ++e) {
        const int word = lword[e];
        double* grow = Gb + (size_t)(word & 0xFFFFF) * no;
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
          const int c = PAIR ? tid * CPT + t : tid + t * SW_BLOCK;
          const double* wp = lw + (size_t)(word >> 20) * naxes * SW_NMAX + (goff[t] / m);
          double v = 0.0;
          for (int s_ = 0; s_ < n; ++s_) v = fma(wp[s_], u[t][s_], v);
          if (CPT == 1 || c < no) store_result(grow + c, v);
        }
      }
    } else if (G != nullptr) {
      // This step's lines: their words came in a step ahead (wcur); now the weights of this thread's axis
      // are requested for all of them at once, and the NEXT step's words -- one trip through LDS per step.
      const int cnt = e1 - e0;
      const int first_next = e1;
      double wv[LR][CPT][SW_NMAX];
#pragma unroll
      for (int x = 0; x < LR; ++x) {
        const int word = wcur[x];
        const int slot = per_line ? e0 + (x < cnt ? x : 0) : (word >> 20);
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
          const double* wp = lw + (size_t)slot * naxes * SW_NMAX + (goff[t] / m);
#pragma unroll
          for (int s_ = 0; s_ < SW_NMAX; ++s_) wv[x][t][s_] = s_ < n ? wp[s_] : 0.0;
        }
      }
      int wnext[LR];
#pragma unroll
      for (int x = 0; x < LR; ++x) wnext[x] = lword[first_next + x];
#pragma unroll
      for (int x = 0; x < LR; ++x) {
        if (x >= cnt) break;
        double* grow = Gb + (size_t)(wcur[x] & 0xFFFFF) * no;
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
          const int c = PAIR ? tid * CPT + t : tid + t * SW_BLOCK;
          double v = 0.0;
#pragma unroll
          for (int s_ = 0; s_ < SW_NMAX; ++s_)
            if (s_ < n) v = fma(wv[x][t][s_], u[t][s_], v);
          if (CPT == 1 || c < no) store_result(grow + c, v);
        }
      }
      // (a step with more lines than are kept in registers: the rest one by one)
      for (int e = e0 + LR; e < e1; ++e) {
        const int word = lword[e];
        const int slot = per_line ? e : (word >> 20);
        double* grow = Gb + (size_t)(word & 0xFFFFF) * no;
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
          const int c = PAIR ? tid * CPT + t : tid + t * SW_BLOCK;
          const double* wp = lw + (size_t)slot * naxes * SW_NMAX + (goff[t] / m);
          double v = 0.0;
          for (int s_ = 0; s_ < n; ++s_) v = fma(wp[s_], u[t][s_], v);
          if (CPT == 1 || c < no) store_result(grow + c, v);
        }
      }
#pragma unroll
      for (int x = 0; x < LR; ++x) wcur[x] = __builtin_amdgcn_readfirstlane(wnext[x]);
    }
    e0 = e1;
    e1 = __builtin_amdgcn_readfirstlane(e2);
  }
  if (P == nullptr) return;

  // ---- backward sweep: z = Phi(l', l+1)^T Psi_l' B_l' per column; P above the diagonal (the own axis' rows) ----
  double z[CPT][SW_NMAX];
#pragma unroll
  for (int t = 0; t < CPT; ++t)
#pragma unroll
    for (int i = 0; i < SW_NMAX; ++i) z[t][i] = 0.0;
  for (int l = N - 1; l >= 0; --l) {
    const double* An = AB + (l + 1 < N ? l + 1 : l) * abw;
    const double* Bl = AB + l * abw + nn;
    double at[SW_NMAX][SW_NMAX], bm[SW_NMAX][SW_MMAX], bv[CPT][SW_MMAX];
#pragma unroll
    for (int i = 0; i < SW_NMAX; ++i)
#pragma unroll
      for (int s_ = 0; s_ < SW_NMAX; ++s_) at[s_][i] = (i < n && s_ < n) ? An[s_ * n + i] : 0.0;
#pragma unroll
    for (int s_ = 0; s_ < SW_NMAX; ++s_)
#pragma unroll
      for (int j = 0; j < SW_MMAX; ++j) bm[s_][j] = (s_ < n && j < m) ? Bl[s_ * m + j] : 0.0;
#pragma unroll
    for (int t = 0; t < CPT; ++t) {
      const int c = PAIR ? tid * CPT + t : tid + t * SW_BLOCK;
      const double* gl = gv + (size_t)l * naxes * m * SW_NMAX + goff[t] + cj[t] * SW_NMAX;
      double gnow[SW_NMAX], y[SW_NMAX];
#pragma unroll
      for (int i = 0; i < SW_NMAX; ++i) {
        gnow[i] = i < n ? gl[i] : 0.0;
        asm volatile("" : "+v"(gnow[i]));
      }
#pragma unroll
      for (int i = 0; i < SW_NMAX; ++i) {
        y[i] = 0.0;
        if (i < n) {
#pragma unroll
          for (int s_ = 0; s_ < SW_NMAX; ++s_)
            if (s_ < n) y[i] = fma(at[s_][i], z[t][s_], y[i]);
          y[i] = cl[t] == l ? gnow[i] : y[i];    // (z is 0 behind its step: A^T 0 = 0)
        }
      }
#pragma unroll
      for (int i = 0; i < SW_NMAX; ++i) z[t][i] = y[i];
#pragma unroll
      for (int j = 0; j < SW_MMAX; ++j) {
        if (j >= m) break;
        prow[t][j] -= no;                       // (the forward sweep left it one row behind the last)
        double v = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < SW_NMAX; ++s_)
          if (s_ < n) v = fma(bm[s_][j], z[t][s_], v);
        bv[t][j] = v;
        if constexpr (!PAIR)
          if ((CPT == 1 || c < no) && cl[t] > l) *prow[t][j] = v;
      }
    }
    if constexpr (PAIR) {
#pragma unroll
      for (int j = 0; j < SW_MMAX; ++j) {
        if (j >= m) break;
        const bool m0 = cl[0] > l, m1 = cl[1] > l;
        if (m0 && m1 && prow[1][j] == prow[0][j] + 1) {
          *reinterpret_cast<double2*>(prow[0][j]) = double2{bv[0][j], bv[1][j]};
        } else {
          if (m0) *prow[0][j] = bv[0][j];
          if (m1) *prow[1][j] = bv[1][j];
        }
      }
    }
  }
}

}  // namespace

bool sweep_eligible(const PlanDev& p) { return p.sw_ok != 0; }

int launch_assemble_sweep(const PlanDev& p, const SrcTable& src, const double* params,
                          const double* given, double* P, double* q, double* G, double* h, int batch,
                          hipStream_t stream, hipError_t* err, const int32_t* h_itab) {
  const int n = p.sw_n, m = p.sw_m, naxes = p.sw_naxes;
  if (n < 1 || n > SW_NMAX || m < 1 || m > SW_MMAX || naxes < 1 || naxes > SW_AXMAX ||
      naxes * n * n > 64 || p.no > SW_BLOCK * 4)
    return MPCASM_ERR_LIMIT;
  // weights per limit unless some limit has an arrow of its own per line (or more than 2^11 limits /
  // 2^20 lines of G: the line word's fields)
  int per_line = p.sw_nlim >= (1 << 11) || p.nc >= (1 << 20);
  if (h_itab == nullptr) return MPCASM_ERR_ARG;
  for (int li = 0; li < p.sw_nlim && !per_line; ++li) {
    const int32_t* rec = h_itab + p.off_sw_lim + li * LIMW;
    for (int ax = 0; ax < rec[SL_NAXES]; ++ax) per_line |= rec[SW_LIM_WORDS + ax * SW_LAX_WORDS + SX_ARROW_STEP] != 0;
  }
  if (p.nc >= (1 << 20)) return MPCASM_ERR_LIMIT;
  // regular lines: the first R lines of every step are lines of the same R limits, a row further down per step
  int reg_lines = 0;
  if (!per_line && p.sw_ngent > 0) {
    const int32_t* gptr = h_itab + p.off_sw_gptr;
    const int32_t* gent = h_itab + p.off_sw_gent;
    const int N = p.sw_horizon;
    int R = 4;
    for (int l = 0; l < N; ++l) R = std::min(R, gptr[l + 1] - gptr[l]);
    for (int x = 0; x < R; ++x) {
      const int lim0 = gent[2 * (gptr[0] + x)], i0 = gent[2 * (gptr[0] + x) + 1];
      bool ok = true;
      for (int l = 1; l < N && ok; ++l) {
        const int e = gptr[l] + x;
        ok = gent[2 * e] == lim0 && gent[2 * e + 1] == i0 + l;
      }
      if (!ok) {
        R = x;
        break;
      }
    }
    reg_lines = R;
  }
  const size_t lds = (size_t)sweep_lds(p, per_line).total * sizeof(double);
  if (lds > (size_t)RESIDENT_LDS_LIMIT) return MPCASM_ERR_LIMIT;
  const double* A = src.ptr[p.sw_src_a];
  const double* Bm = src.ptr[p.sw_src_b];
  const long long sa = src.stride[p.sw_src_a], sb = src.stride[p.sw_src_b];
#define MPCASM_SWEEP_CASE(CPT, NS, MS, AS) MPCASM_SWEEP_CASE_P(CPT, NS, MS, AS, false)
#define MPCASM_SWEEP_CASE_P(CPT, NS, MS, AS, PAIR)                                                     \
  if (p.no <= SW_BLOCK * CPT && (NS == 0 || (n == NS && m == MS && naxes == AS))) {                    \
    auto kernel = ltv_sweep_kernel<CPT, NS, MS, AS, PAIR>;                                             \
    if (lds > 64 * 1024) {                                                                             \
      *err = allow_whole_lds(reinterpret_cast<const void*>(kernel));                                   \
      if (*err != hipSuccess) return MPCASM_ERR_HIP;                                                   \
    }                                                                                                  \
    hipLaunchKernelGGL(kernel, dim3((unsigned)batch), dim3(SW_BLOCK), lds, stream, p, A, sa, Bm, sb,   \
                       params, given, P, q, G, h, batch, per_line, reg_lines);                         \
    *err = hipGetLastError();                                                                          \
    return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;                                            \
  }
  if ((p.no & 1) == 0 && (p.sw_horizon & 1) == 0) { MPCASM_SWEEP_CASE_P(2, 3, 1, 2, true) }
  MPCASM_SWEEP_CASE(1, 3, 1, 2)
  if ((p.no & 1) == 0 && (p.sw_horizon & 1) == 0) { MPCASM_SWEEP_CASE_P(2, 0, 0, 0, true) }
  MPCASM_SWEEP_CASE(1, 0, 0, 0)
  MPCASM_SWEEP_CASE(2, 0, 0, 0)
  MPCASM_SWEEP_CASE(4, 0, 0, 0)
#undef MPCASM_SWEEP_CASE
#undef MPCASM_SWEEP_CASE_P
  return MPCASM_ERR_LIMIT;
}

}  // namespace mpcasm
