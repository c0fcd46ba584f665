// tiled.hip -- K2 + K3 + K4 for wide problems (no >= T_BLOCK: BASELINE config C4, nx=12 nu=6
// N=64, no=384, nc=1536) on gfx950, WITHOUT a workspace in HBM.
//
// Reference semantics (python/mpc_interface/body.py):
//   K2  make_preview_matrices / get_matrices_from_dynamics / _from_definition  :149-193
//   K3  generate_qp_cost :266-302, generate_all_qp_costs :322-329
//   K4  generate_qp_constraint :236-264, generate_all_qp_constraints :304-320
//   K1  tools.extend_matrices, tools.py:14-33 (plans compiled with lti=[...]: the horizon
//       tables come from a pre-pass on the system's own (A, B))
//
// The staged pipeline (assemble.hip) composes every row of the preview matrices into an HBM
// workspace V (2.4 MB per C4 instance), and three more kernels read it back: 3.7 x the
// algorithmic traffic, half of the time outside the matrix core.  Here a workgroup owns one
// T_BLOCK x T_BLOCK block of one instance's P and composes the 16 rows of a *stage* straight
// into the LDS tiles its MFMAs read -- through the plan's column tables (plan_tables.h
// H_T_CI_OK): the value of base row k in column c is stream[offset(c) + k * rs(c)], no segment
// lookup, no branch (columns nothing feeds and identity blocks read small constant tables).
// Workgroups on the diagonal (bi == bj) also turn the same composed rows into the gradient
// (VALU, while the matrix core runs) and compose + stream the rows of G of their column block;
// the one of block (0, 0) writes h.  d = Mg . given (one value per workspace row) comes from a
// small pre-pass.  Per stage the plan states which 16-column tiles can be non-zero at all
// (TS_MASK*): products of structurally zero tiles are skipped -- for the block-lower-triangular
// horizon matrices of an LTI system that is 34 of 64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace mpcasm {

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int TK = 16;              // rows of a stage
constexpr int TLD = T_BLOCK + 16;   // row stride of an LDS tile (doubles): k-rows on disjoint bank halves
constexpr int NSTREAM = T_SID_CONST + 1;

__device__ __forceinline__ int sext24(int x) { return (x << 8) >> 8; }

// the streams of one instance: sources at this instance's slice, then the plan's dtab
__device__ __forceinline__ void stream_bases(const PlanDev& p, const SrcTable& src, long inst,
                                             const double** s_base, int tid) {
  if (tid < NSTREAM) {
    const double* b = p.dtab;
    if (tid < T_SID_CONST) b = tid < p.nsrc ? src.ptr[tid] + inst * src.stride[tid] : p.dtab;
    s_base[tid] = b;
  }
}

// ---------------------------------------------------------------------------
// pre-pass 1 (plans with generated groups): TA = S and TB of every group from (A, B), by the
// reference's own recurrence X_d = A X_{d-1}, X_0 = [B | A] (tools.py:24-29).  One workgroup
// per (instance, group); X lives in LDS, one barrier per step.
// ---------------------------------------------------------------------------
constexpr int LTI_XMAX = 2048;  // n (m + n) doubles of one X

constexpr int LTI_HIST = 16;  // steps of the input columns kept in LDS before they are written out

__global__ __launch_bounds__(BLOCK) void lti_tables_kernel(PlanDev p, SrcTable src,
                                                           double* __restrict__ work,
                                                           long long work_stride) {
  // dynamic LDS, sized for the group at hand: A | X twice | the input columns of LTI_HIST steps
  extern __shared__ __attribute__((aligned(16))) double lti_lds[];
  const int tid = threadIdx.x;
  const long inst = blockIdx.x / p.t_nlti;
  const int g = blockIdx.x - inst * p.t_nlti;
  const int32_t* rec = p.itab + p.off_t_lti + g * T_LTI_WORDS;
  const int n = rec[TL_N], m = rec[TL_M], N = rec[TL_HORIZON];
  double* sA = lti_lds;
  double* sX0 = lti_lds + n * n;
  double* sX1 = sX0 + n * (m + n);
  double* hist = sX1 + n * (m + n);  // [n m][LTI_HIST]
  const int32_t* ids = p.itab + p.off_t_lti_ids + rec[TL_IDS];
  const double* A = src.ptr[ids[0]] + inst * src.stride[ids[0]];
  const double* B = src.ptr[ids[1]] + inst * src.stride[ids[1]];
  double* TA = work + inst * work_stride + rec[TL_TA];
  double* TB = work + inst * work_stride + rec[TL_TB];
  const int w = m + n, elems = n * w, nm = n * m;
  for (int e = tid; e < n * n; e += BLOCK) sA[e] = A[e];
  for (int e = tid; e < elems; e += BLOCK) {
    const int i = e / w, c = e - i * w;
    sX0[e] = c < m ? B[i * m + c] : A[i * n + (c - m)];
  }
  // the zeros in front of every row of TB: whole runs of N doubles
  for (int e = tid; e < nm * N; e += BLOCK) TB[(size_t)(e / N) * 2 * N + (e % N)] = 0.0;
  __syncthreads();
  for (int d = 0; d < N; ++d) {
    const double* X = (d & 1) ? sX1 : sX0;
    double* Xn = (d & 1) ? sX0 : sX1;
    for (int e = tid; e < elems; e += BLOCK) {
      const int i = e / w, c = e - i * w;
      const double v = X[e];
      if (c < m)
        hist[(i * m + c) * LTI_HIST + (d % LTI_HIST)] = v;  // (A^d B)[i][c], written out below
      else
        TA[(size_t)d * n * n + (size_t)(c - m) * n + i] = v;  // S[k][j][i] = (A^{k+1})[i][j]: a contiguous block per step
      double acc = 0.0;
      for (int t = 0; t < n; ++t) acc = fma(sA[i * n + t], X[t * w + c], acc);
      Xn[e] = acc;
    }
    lds_barrier();  // (orders LDS only: the step's stores to the scratch stream on behind it)
    if ((d + 1) % LTI_HIST == 0 || d + 1 == N) {
      // LTI_HIST steps of every row of TB at once: whole cache lines instead of one double per step
      const int d0 = d - d % LTI_HIST, cnt = d + 1 - d0;
      for (int e = tid; e < nm * cnt; e += BLOCK) {
        const int row = e / cnt, k = e - row * cnt;
        TB[(size_t)row * 2 * N + N + d0 + k] = hist[row * LTI_HIST + k];
      }
      lds_barrier();
    }
  }
}

// The same tables for SMALL systems (n (m + n) <= 64 elements, the LIPM family): a wavefront per system -- or
// per several (the biped's 12 elements: four systems of 16 lanes) -- instead of a workgroup; lane e = (i, c) holds X[i][c], the step's operands cross lanes through LDS (one
// wavefront's LDS operations complete in order: no barrier), the tables are collected in LDS and leave in
// whole lines.  Same products in the same order as lti_tables_kernel above, i.e. the same numbers.
// (That kernel keeps 12 of 256 threads busy on the biped's 3 x 4 elements: 0.29 ms per 65 536 systems.)
__global__ __launch_bounds__(BLOCK) void lti_tables_small_kernel(PlanDev p, SrcTable src,
                                                                 double* __restrict__ work,
                                                                 long long work_stride, long jobs,
                                                                 int per_system, int lanes, int systems) {
  // `systems` systems per wavefront, `lanes` (a power of two >= n (m + n)) lanes each
  extern __shared__ __attribute__((aligned(16))) double lti_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / lanes, l = lane - sub * lanes;
  double* mine = lti_lds + ((size_t)wave * systems + sub) * per_system;  // A [64] | X twice [2][64] | TA | TB
  double* sA = mine;
  double* sX = mine + 64;
  double* tab = mine + 192;
  const long per_wg = (long)(BLOCK / 64) * systems;
  for (long first = ((long)blockIdx.x * (BLOCK / 64) + wave) * systems; first < jobs; first += (long)gridDim.x * per_wg) {
    const long job = first + sub;
    const bool live = sub < systems && job < jobs;
    const long jb = live ? job : first;  // (idle lanes follow the wavefront's first system: same loop bounds)
    const long inst = jb / p.t_nlti;
    const int g = (int)(jb - inst * p.t_nlti);
    const int32_t* rec = p.itab + p.off_t_lti + g * T_LTI_WORDS;
    const int n = rec[TL_N], m = rec[TL_M], N = rec[TL_HORIZON];
    const int32_t* ids = p.itab + p.off_t_lti_ids + rec[TL_IDS];
    const double* A = src.ptr[ids[0]] + inst * src.stride[ids[0]];
    const double* B = src.ptr[ids[1]] + inst * src.stride[ids[1]];
    const int w = m + n, elems = n * w, nta = N * n * n, ntb = n * m * 2 * N;
    double* tA = tab;
    double* tB = tab + nta;
    const int i = l / w, c = l - i * w;  // (lanes >= elems idle)
    const bool mine_ok = live && l < elems;
    if (live && l < n * n) sA[l] = A[l];
    if (mine_ok) sX[l] = c < m ? B[i * m + c] : A[i * n + (c - m)];
    if (live)
      for (int e = l; e < ntb; e += lanes) tB[e] = 0.0;  // (the zeros in front of every row of TB, and the rest)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int d = 0; d < N; ++d) {
      const double* X = sX + (d & 1) * 64;
      double* Xn = sX + ((d & 1) ^ 1) * 64;
      if (mine_ok) {
        const double v = X[l];
        if (c < m)
          tB[(i * m + c) * 2 * N + N + d] = v;       // (A^d B)[i][c]
        else
          tA[d * n * n + (c - m) * n + i] = v;       // S[k][j][i] = (A^{k+1})[i][j]
        double acc = 0.0;
        for (int t = 0; t < n; ++t) acc = fma(sA[i * n + t], X[t * w + c], acc);
        Xn[l] = acc;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (live) {
      double* TA = work + inst * work_stride + rec[TL_TA];
      double* TB = work + inst * work_stride + rec[TL_TB];
      for (int e = l; e < nta; e += lanes) TA[e] = tA[e];
      for (int e = l; e < ntb; e += lanes) TB[e] = tB[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (read out before the next systems overwrite it)
  }
}

// ---------------------------------------------------------------------------
// pre-pass 2: d[r] = (Mg . given)[r] for every workspace row: a thread per row (a wavefront per
// row, lanes over the dozen given columns of a wide problem, spent 6 % of a C4 call here), the
// given vector in LDS; the row's entries and the column tables are the same for every instance
// (L2), a thread's loads of one entry independent of each other.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void compose_d_kernel(PlanDev p, SrcTable src,
                                                          const double* __restrict__ given,
                                                          double* __restrict__ work,
                                                          long long work_stride, int nrb) {
  __shared__ const double* s_base[NSTREAM];
  extern __shared__ __attribute__((aligned(16))) double s_given[];
  const int tid = threadIdx.x;
  const long inst = blockIdx.x / nrb;
  const int rb = blockIdx.x - inst * nrb;
  stream_bases(p, src, inst, s_base, tid);
  for (int c = tid; c < p.ng; c += BLOCK) s_given[c] = given[inst * p.ng + c];
  __syncthreads();
  const int r = rb * BLOCK + tid;
  if (r >= p.rtot) return;
  const int32_t* rowptr = p.itab + p.off_rowptr;
  const int32_t* entbase = p.itab + p.off_entbase;
  const int32_t* entk = p.itab + p.off_entk;
  const double* coef = p.dtab + p.doff_entcoef;
  const int2* cig = reinterpret_cast<const int2*>(p.itab + p.off_t_cig);
  double acc = 0.0;
  for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
    const int k = entk[e];
    const int2* row = cig + (size_t)entbase[e] * p.ng;
    double part = 0.0;
    for (int c = 0; c < p.ng; ++c) {
      const int2 ci = row[c];
      part = fma(s_base[(unsigned)ci.y >> 24][(long)ci.x + (long)k * sext24(ci.y)], s_given[c], part);
    }
    acc = fma(coef[e], part, acc);
  }
  work[inst * work_stride + r] = acc;
}

// ---------------------------------------------------------------------------
// the tiled kernel
// ---------------------------------------------------------------------------
// what two adjacent columns of one base variable read (from the column table)
struct ColRef {
  const double* p0;
  const double* p1;
  int rs0, rs1;
};

__device__ __forceinline__ ColRef load_colref(const int32_t* __restrict__ cio, int nop, int base,
                                              int col, const double* const* s_base) {
  const int4 v = *reinterpret_cast<const int4*>(cio + ((size_t)base * nop + col) * 2);
  ColRef r;
  r.p0 = s_base[(unsigned)v.y >> 24] + v.x;
  r.rs0 = sext24(v.y);
  r.p1 = s_base[(unsigned)v.w >> 24] + v.z;
  r.rs1 = sext24(v.w);
  return r;
}

__device__ __forceinline__ double2 colref_at(const ColRef& c, int k) {
  double2 v;
  v.x = c.p0[(long)k * c.rs0];
  v.y = c.p1[(long)k * c.rs1];
  return v;
}

// two adjacent columns of a row of G leave as one 16-byte piece when the rows are 16-byte aligned (an even
// width); an odd width shifts every other row by 8 bytes: two 8-byte stores there, the last column alone
__device__ __forceinline__ void store_pair(double* row, int col, int no, bool odd, double2 v) {
  if (!odd) {
    if (col < no) store_result(reinterpret_cast<double2*>(row + col), v);
  } else {
    if (col < no) store_result(row + col, v.x);
    if (col + 1 < no) store_result(row + col + 1, v.y);
  }
}

struct RowTables {
  const int32_t* rowptr;
  const int32_t* entbase;
  const int32_t* entk;
  const double* coef;
  const int32_t* cio;
  int nop;
};

// two adjacent columns of workspace row r (wave-uniform): sum over the row's entries
__device__ __forceinline__ double2 compose_row2(const RowTables& t, int r, int col,
                                                const double* const* s_base, int& cur_base,
                                                ColRef& cr) {
  double2 acc{0.0, 0.0};
  const int e1 = t.rowptr[r + 1];
  for (int e = t.rowptr[r]; e < e1; ++e) {
    const int b = t.entbase[e];
    if (b != cur_base) {
      cr = load_colref(t.cio, t.nop, b, col, s_base);
      cur_base = b;
    }
    const double cf = t.coef[e];
    const double2 v = colref_at(cr, t.entk[e]);
    acc.x = fma(cf, v.x, acc.x);
    acc.y = fma(cf, v.y, acc.y);
  }
  return acc;
}

// the 8 tiles of block b out of a stage's 64-bit tile mask (tiles beyond 62 fold onto bit 63)
__device__ __forceinline__ unsigned block_tiles(unsigned lo, unsigned hi, int b) {
  const unsigned long long m = ((unsigned long long)hi << 32) | lo;
  if (8 * b + 7 < 63) return (unsigned)(m >> (8 * b)) & 0xFFu;
  unsigned out = 0;
  for (int j = 0; j < 8; ++j) out |= (unsigned)((m >> min(8 * b + j, 63)) & 1ull) << j;
  return out;
}

// the products of the first NA x NB tiles of a wavefront's quadrant over the 16 rows of a stage;
// at / bt: the lane's element of row lk in the A / B tile (lane map of v_mfma_f64_16x16x4_f64:
// A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15])
template <int NA, int NB>
__device__ __forceinline__ void mfma_tiles(f64x4 (&acc)[4][4], const double* at, const double* bt) {
#pragma unroll
  for (int kk = 0; kk < TK; kk += 4) {
    double a[NA], b[NB];
#pragma unroll
    for (int t = 0; t < NA; ++t) a[t] = at[kk * TLD + t * 16];
#pragma unroll
    for (int t = 0; t < NB; ++t) b[t] = bt[kk * TLD + t * 16];
#pragma unroll
    for (int ta = 0; ta < NA; ++ta)
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) acc[ta][tb] = mfma_f64_16x16x4(a[ta], b[tb], acc[ta][tb]);
  }
}

struct Stage {
  int arow, brow, drow, nrows, flags, base_a, base_b;
  int cls;  // 0: no Hessian part; n = 1..4: the first n x n tiles of every wavefront's quadrant
  int wslot, aimslot;
  // rows wave, wave + 4, wave + 8, wave + 12 of the stage (this wavefront's): base row and
  // coefficient of simple A / B rows, the rows of G riding on the A rows
  int ka[TK / WAVES], kb[TK / WAVES];
  double ca[TK / WAVES], cb[TK / WAVES];
  int4 pig[TK / WAVES];
};

// a value every lane loads from the same address, through the vector memory path: it counts in
// vmcnt (in order, waited for where it is used) instead of the scalar cache's lgkmcnt, which every
// LDS wait of the matrix-core phase would have to include
__device__ __forceinline__ double uniform_load(const double* p, int vzero) { return p[vzero]; }

__global__ __launch_bounds__(BLOCK, 2) void tiled_assemble_kernel(
    PlanDev p, SrcTable src, const double* __restrict__ params, const double* __restrict__ work,
    long long work_stride, double* __restrict__ P, double* __restrict__ q, double* __restrict__ G,
    double* __restrict__ h, int nb, int npairs, int sym, int batch) {
  __shared__ __attribute__((aligned(16))) double As[2][TK * TLD];
  __shared__ __attribute__((aligned(16))) double Bs[2][TK * TLD];
  __shared__ const double* s_base[NSTREAM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the blocks of one instance go to ONE XCD (workgroups x, x + 8, x + 16 ... share an L2):
  // they read the same sources and tables
  const unsigned xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const long inst = (long)(slot / npairs) * 8 + xcd;
  int pr = slot % npairs;
  if (inst >= batch) return;
  int bi = 0, bj = 0;
  if (sym) {  // pair index -> (bi <= bj)
    int rowlen = nb;
    while (pr >= rowlen) {
      pr -= rowlen;
      --rowlen;
      ++bi;
    }
    bj = bi + pr;
  } else {
    bi = pr / nb;
    bj = pr - bi * nb;
  }
  const bool diag = bi == bj;
  if (!P && !diag) return;  // only the constraints are wanted: the diagonal workgroups write them
  stream_bases(p, src, inst, s_base, tid);
  const int no = p.no;
  const double* pb = params + (size_t)inst * p.nparams;
  const double* dvec = work + inst * work_stride;
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  // (the instance's parameters are read through the scalar cache below: bring their lines in now)
  if (tid * 8 < p.nparams) asm volatile("" ::"v"(pb[tid * 8]));
  __syncthreads();

  RowTables rt;
  rt.rowptr = p.itab + p.off_rowptr;
  rt.entbase = p.itab + p.off_entbase;
  rt.entk = p.itab + p.off_entk;
  rt.coef = p.dtab + p.doff_entcoef;
  rt.cio = p.itab + p.off_t_cio;
  rt.nop = p.t_nop;
  const int32_t* stages = p.itab + p.off_t_stage;
  const int32_t* srow = p.itab + p.off_t_srow;
  const double* scoef = p.dtab + p.t_doff_scoef;
  const int4* pigs = reinterpret_cast<const int4*>(p.itab + p.off_t_pig);
  const int li = lane & 15, lk = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  // compose role: wavefront `wave` takes rows wave, wave + 4, ... of a stage, lane `lane` two columns
  const int scol = lane * 2;
  const int colA = bi * T_BLOCK + scol, colB = bj * T_BLOCK + scol;
  const bool want_g = G != nullptr && diag;
  const bool odd = no & 1;

  f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  double2 qacc{0.0, 0.0};

  int baseA = -1, baseB = -1;
  ColRef crA, crB;
  crA.p0 = crA.p1 = crB.p0 = crB.p1 = p.dtab;
  crA.rs0 = crA.rs1 = crB.rs0 = crB.rs1 = 0;

  // ---- stages (sorted by class in the plan) ---------------------------------------------
  int s = -1;
  Stage cur;
  auto next_stage = [&]() __attribute__((always_inline)) -> bool {
    for (++s; s < p.t_nstage; ++s) {
      const int32_t* rec = stages + s * T_STAGE_WORDS;
      const int info = rec[TS_INFO];
      const int fl = (info >> 8) & 255;
      const unsigned ta = block_tiles((unsigned)rec[TS_MASKA_LO], (unsigned)rec[TS_MASKA_HI], bi);
      const unsigned tb = block_tiles((unsigned)rec[TS_MASKB_LO], (unsigned)rec[TS_MASKB_HI], bj);
      const bool for_p = P && (fl & TS_FLAG_P) && ta && tb;
      // the diagonal workgroup also takes the stage for the gradient (when its A rows reach this
      // column block at all) and for the rows of G that ride on it (zeros included)
      const bool for_diag = diag && ((P && ta) || (want_g && (fl & TS_FLAG_G)));
      if (!for_p && !for_diag) continue;
      cur.arow = rec[TS_AROW];
      cur.brow = rec[TS_BROW];
      cur.drow = rec[TS_DROW];
      cur.nrows = info & 255;
      cur.flags = for_p ? fl : (fl & ~TS_FLAG_P);
      cur.base_a = rec[TS_BASE] & 0xFFFF;
      cur.base_b = (unsigned)rec[TS_BASE] >> 16;
      cur.cls = info >> 16;  // (the plan's: the stages come sorted by it)
      cur.wslot = rec[TS_WPARAM];
      cur.aimslot = rec[TS_AIMPARAM];
#pragma unroll
      for (int rr = 0; rr < TK / WAVES; ++rr) {
        const int i = wave + WAVES * rr;
        cur.ka[rr] = srow[(s * 2 + 0) * TK + i];
        cur.kb[rr] = srow[(s * 2 + 1) * TK + i];
        cur.ca[rr] = scoef[(s * 2 + 0) * TK + i];
        cur.cb[rr] = scoef[(s * 2 + 1) * TK + i];
        cur.pig[rr] = pigs[s * TK + i];
      }
      return true;
    }
    return false;
  };

  // values of the stage's rows in this thread's columns (simple rows: the entry's coefficient
  // still apart; the loads are in flight while the matrix core works on the stage before), the
  // stage's weight, aim and the d of its rows
  double2 va[TK / WAVES], vb[TK / WAVES];
  auto compose_tile = [&](int row0, int nrows, bool simple, int base, const int* ks, int col,
                          int& cur_base, ColRef& cr, double2* v) __attribute__((always_inline)) {
    if (simple) {
      if (base != cur_base) {
        cr = load_colref(rt.cio, rt.nop, base, col, s_base);
        cur_base = base;
      }
#pragma unroll
      for (int rr = 0; rr < TK / WAVES; ++rr) v[rr] = colref_at(cr, ks[rr]);
    } else {
#pragma unroll
      for (int rr = 0; rr < TK / WAVES; ++rr) {
        const int i = wave + WAVES * rr;
        v[rr] = compose_row2(rt, row0 + (i < nrows ? i : 0), col, s_base, cur_base, cr);
      }
    }
  };
  auto compose_stage = [&]() __attribute__((always_inline)) {
    compose_tile(cur.arow, cur.nrows, cur.flags & TS_FLAG_SIMPLE_A, cur.base_a, cur.ka, colA, baseA, crA, va);
    if ((cur.flags & TS_FLAG_P) && !(diag && (cur.flags & TS_FLAG_SAME)))
      compose_tile(cur.brow, cur.nrows, cur.flags & TS_FLAG_SIMPLE_B, cur.base_b, cur.kb, colB, baseB, crB, vb);
  };
  // registers -> LDS tiles (the A rows weighted); on the way the diagonal workgroup adds the
  // weighted rows into the gradient and writes the rows of G that are arrow * (an A row)
  auto store_stage = [&](int buf) __attribute__((always_inline)) {
    const bool same = diag && (cur.flags & TS_FLAG_SAME);
    const bool simple_a = cur.flags & TS_FLAG_SIMPLE_A, simple_b = cur.flags & TS_FLAG_SIMPLE_B;
    const double scale = (cur.flags & TS_FLAG_HALF) ? 0.5 : 1.0;
    // (wave-uniform, through the scalar cache: the parameters' lines were brought in at the start)
    const double vw = P ? pb[cur.wslot] : 0.0, vaim = pb[cur.aimslot];
    double vd[TK / WAVES];
#pragma unroll
    for (int rr = 0; rr < TK / WAVES; ++rr) {
      const int i = wave + WAVES * rr;
      vd[rr] = (diag && P) ? dvec[cur.drow + (i < cur.nrows ? i : 0)] : 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < TK / WAVES; ++rr) {
      const int i = wave + WAVES * rr;
      const bool valid = i < cur.nrows;
      const double fa = simple_a ? cur.ca[rr] : (valid ? 1.0 : 0.0);
      double2 a = va[rr];
      a.x *= fa;
      a.y *= fa;
      if (want_g && (cur.flags & TS_FLAG_G) && colA < no) {
        const int4 pg = cur.pig[rr];
        if (pg.x >= 0) {
          const double ar = pb[pg.y];
          store_pair(G + ((size_t)inst * p.nc + pg.x) * no, colA, no, odd, double2{ar * a.x, ar * a.y});
        }
        if (pg.z >= 0) {
          const double ar = pb[pg.w];
          store_pair(G + ((size_t)inst * p.nc + pg.z) * no, colA, no, odd, double2{ar * a.x, ar * a.y});
        }
      }
      if (!P) continue;
      double2 b;
      if (same) {
        b = a;
      } else {
        const double fb = simple_b ? cur.cb[rr] : (valid ? 1.0 : 0.0);
        b = vb[rr];
        b.x *= fb;
        b.y *= fb;
      }
      a.x *= vw;
      a.y *= vw;
      if (diag) {
        const double r = scale * (vd[rr] - vaim);
        qacc.x = fma(a.x, r, qacc.x);
        qacc.y = fma(a.y, r, qacc.y);
      }
      if (cur.flags & TS_FLAG_P) {
        *reinterpret_cast<double2*>(As[buf] + i * TLD + scol) = a;
        *reinterpret_cast<double2*>(Bs[buf] + i * TLD + scol) = b;
      }
    }
  };

  bool more = next_stage();
  if (more) {
    compose_stage();
    store_stage(0);
  }
  int buf = 0;
  lds_barrier();
  auto run_class = [&](auto cls_tag) __attribute__((always_inline)) {
    constexpr int N = decltype(cls_tag)::value;
    while (more && cur.cls == N) {
      const bool mult = cur.flags & TS_FLAG_P;  // (not when this block's columns are zero in it)
      const bool have_next = next_stage();      // (the stage in the tiles has class N)
      if (have_next) compose_stage();           // in flight while this stage is multiplied
      if constexpr (N > 0)
        if (mult)
          mfma_tiles<N, N>(acc, As[buf] + lk * TLD + wr * 64 + li, Bs[buf] + lk * TLD + wc * 64 + li);
      if (have_next) store_stage(buf ^ 1);
      if constexpr (N > 0) {
        lds_barrier();
        buf ^= 1;
      } else {
        // (no tile was read: only a stage of a Hessian class that has just been stored needs
        // the barrier before it is multiplied)
        if (have_next && cur.cls > 0) {
          lds_barrier();
          buf ^= 1;
        }
      }
      more = have_next;
    }
  };
  run_class(std::integral_constant<int, 0>{});
  run_class(std::integral_constant<int, 1>{});
  run_class(std::integral_constant<int, 2>{});
  run_class(std::integral_constant<int, 3>{});
  run_class(std::integral_constant<int, 4>{});

  // ---- rows of G that ride on no stage, h ------------------------------------------------
  if (want_g) {
    const int32_t* grow = p.itab + p.off_t_grow;
    const int32_t* rrw = p.itab + p.off_rs_rr;
    const int32_t* grest = p.itab + p.off_t_grest;
    int baseG = -1;
    ColRef crG = crA;
    for (int x0 = wave; x0 < p.t_ngrest; x0 += WAVES) {
      const int R = grest[x0];
      const int32_t* x = rrw + (size_t)R * RS_RR_WORDS;
      const int naxes = x[RR_NAXES];
      double2 out{0.0, 0.0};
      for (int ax = 0; ax < naxes; ++ax) {
        const double ar = pb[x[RR_ARROW + ax]];
        const double2 v = compose_row2(rt, grow[R * RS_AXMAX + ax], colA, s_base, baseG, crG);
        out.x = fma(ar, v.x, out.x);
        out.y = fma(ar, v.y, out.y);
      }
      store_pair(G + ((size_t)inst * p.nc + R) * no, colA, no, odd, out);
    }
    if (bi == 0)  // h: (extreme + arrow . center) - arrow . d
      for (int R = tid; R < p.nc; R += BLOCK) {
        const int32_t* x = rrw + (size_t)R * RS_RR_WORDS;
        double ac = 0.0, ad = 0.0;
        for (int ax = 0; ax < x[RR_NAXES]; ++ax) {
          const double ar = pb[x[RR_ARROW + ax]];
          ac += ar * pb[x[RR_CENTER + ax]];
          ad = fma(ar, dvec[grow[R * RS_AXMAX + ax]], ad);
        }
        h[(size_t)inst * p.nc + R] = (pb[x[RR_EXTREME]] + ac) - ad;
      }
  }
  if (!P) return;

  // ---- results ---------------------------------------------------------------------------
  double* part = As[0];  // (the tiles are no longer needed)
  // the diagonal gterms on column bi * T_BLOCK + tid (diagonal workgroups)
  double dPc = 0.0, dqc = 0.0;
  if (diag && tid < T_BLOCK && bi * T_BLOCK + tid < no)
    diagonal_of_column(p, pb, bi * T_BLOCK + tid, dPc, dqc);
  if (diag) {            // ... and the wavefronts' partial gradients, through LDS
    lds_barrier();
    if (tid < T_BLOCK) part[WAVES * T_BLOCK + tid] = dPc;
    part[wave * T_BLOCK + scol] = qacc.x;
    part[wave * T_BLOCK + scol + 1] = qacc.y;
    lds_barrier();
  }
  double* Pb = P + (size_t)inst * no * no;
  const bool mirror = sym && !diag;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = bi * T_BLOCK + wr * 64 + ta * 16 + lk + 4 * reg;
        const int col = bj * T_BLOCK + wc * 64 + tb * 16 + li;
        if (row < no && col < no) {
          double v = acc[ta][tb][reg];
          if (diag && row == col) v += part[WAVES * T_BLOCK + (row - bi * T_BLOCK)];
          Pb[(size_t)row * no + col] = v;
          if (mirror) Pb[(size_t)col * no + row] = v;
        }
      }
  if (diag && tid < T_BLOCK) {  // the gradient: the four wavefronts hold the sums over their rows
    const int c = bi * T_BLOCK + tid;
    if (c < no) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) sum += part[w * T_BLOCK + tid];
      q[(size_t)inst * no + c] = sum + dqc;
    }
  }
}

// ---------------------------------------------------------------------------
// Toeplitz form (plans whose every stage is TS_FLAG_TOEPLITZ: the rows of every cost are rows
// of the states of ONE system x+ = A x + B u whose horizon tables the pre-pass generated): the
// rows of a stage are windows of the group's table TB[i][j][.] (plan_tables.h TL_*), so the
// workgroup copies TB (n m 2N doubles: 74 KB for C4) into LDS once and the matrix core's
// operands are read straight out of it -- element (row r, column c) of the stage sits at
// TB[colpart(c) + U + r].  No tile is composed, nothing is written to LDS in the loop, no barrier.
// Diagonal workgroups read the same windows once more for the gradient and the riding rows of G.
// LDS: TB | zeros (what columns outside the system's inputs read) | d.
// ---------------------------------------------------------------------------
// One LDS-DMA load (global_load_lds_dwordx4): every active lane's 16 bytes from its own global
// address to lds_base + lane * 16, no register in between.  M0 carries the LDS base; it is
// compiler-reserved, so it is saved, set and restored inside the one statement.  The compiler does
// not count this load: the issuing wave waits for it itself (s_waitcnt vmcnt(0)).
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}

__device__ __forceinline__ double lds_f64(const char* lds, unsigned byte_off) {
  return *reinterpret_cast<const double*>(lds + byte_off);
}

// Which of a wavefront's 4 x 4 tiles a stage of class N multiplies, by the wavefront's role:
//   FULL  every pair of the first N tiles (a block off the diagonal);
//   TRI   pairs ta <= tb only: a quadrant ON the diagonal of a diagonal block of a symmetric P (the
//         other triangle is its mirror image: written, not computed);
//   TOP / BOT  the quadrant above the diagonal of a diagonal block, shared by the wavefront that owns
//         it (tile rows 0, 1) and the one whose own quadrant -- below the diagonal -- is a mirror
//         image (tile rows 2, 3).
// A diagonal block then takes 2 N^2 + N products per four k-steps instead of 4 N^2.
enum Role { ROLE_FULL = 0, ROLE_TRI = 1, ROLE_TOP = 2, ROLE_BOT = 3 };
template <int ROLE, int N>
struct TileRange {
  static constexpr int a0 = ROLE == ROLE_BOT ? (N < 2 ? N : 2) : 0;
  static constexpr int a1 = ROLE == ROLE_TOP ? (N < 2 ? N : 2) : N;
  static constexpr bool tri = ROLE == ROLE_TRI;
  __device__ static constexpr bool has(int ta, int tb) {
    return ta >= a0 && ta < a1 && tb < N && (!tri || ta <= tb);
  }
};

template <int ROLE, int N>
__device__ __forceinline__ void mfma_toeplitz(f64x4 (&acc)[4][4], const char* lds,
                                              const unsigned (&pa)[4], const unsigned (&pb)[4],
                                              const unsigned (&ma)[4], const unsigned (&mb)[4],
                                              unsigned ua, unsigned ub, double sa, double sb, int lk,
                                              int nrows) {
  using R = TileRange<ROLE, N>;
  if constexpr (R::a0 >= R::a1) return;
  unsigned aa[4], ab[4];
#pragma unroll
  for (int t = 0; t < N; ++t) {
    aa[t] = pa[t] + (ua & ma[t]);
    ab[t] = pb[t] + (ub & mb[t]);
  }
#pragma unroll
  for (int kk = 0; kk < TK; kk += 4) {
    const bool live = kk + lk < nrows;  // (a term's last stage may be short: rows behind it are zeros)
    double a[4], b[4];
#pragma unroll
    for (int t = R::a0; t < R::a1; ++t) {
      const double va = lds_f64(lds, aa[t] + kk * 8) * sa;
      a[t] = live ? va : 0.0;
    }
#pragma unroll
    for (int t = 0; t < N; ++t) {
      const double vb = lds_f64(lds, ab[t] + kk * 8) * sb;
      b[t] = live ? vb : 0.0;
    }
#pragma unroll
    for (int ta = R::a0; ta < R::a1; ++ta)
#pragma unroll
      for (int tb = R::tri ? ta : 0; tb < N; ++tb) acc[ta][tb] = mfma_f64_16x16x4(a[ta], b[tb], acc[ta][tb]);
  }
}

__global__ __launch_bounds__(BLOCK, 2) void toeplitz_assemble_kernel(
    PlanDev p, SrcTable src, const double* __restrict__ params, const double* __restrict__ work,
    long long work_stride, double* __restrict__ P, double* __restrict__ q, double* __restrict__ G,
    double* __restrict__ h, int nb, int npairs, int sym, int batch, int tbn, int nzero, int phases) {
  // (phases: MPCASM_OPT_PHASE_MASK, a timing-only ablation for profiles -- bit 1 the matrix-core
  // products, bit 2 the gradient, bit 3 the rows of G, bit 5 the stores of P; all set in production)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ONE workgroup per instance: it takes the blocks of P one after the other, with the table (and
  // everything else of the set-up) in LDS once -- a workgroup per block paid the 74 KB copy, the
  // column look-ups and the launch six times per C4 instance (a fifth of a call)
  const long inst = blockIdx.x;
  if (inst >= batch) return;
  const int no = p.no;
  const double* pb = params + (size_t)inst * p.nparams;
  const double* dvec = work + inst * work_stride;
  const int32_t* stages = p.itab + p.off_t_stage;
  const double* scoef = p.dtab + p.t_doff_scoef;
  const int4* pigs = reinterpret_cast<const int4*>(p.itab + p.off_t_pig);
  const int32_t* grp = p.itab + p.off_t_lti;  // the one generated group
  const int32_t* ids = p.itab + p.off_t_lti_ids + grp[TL_IDS];
  const int first_u = ids[0];
  // ---- the table, the zeros and d into LDS ------------------------------------------------
  {
    // LDS-DMA: the wavefronts take the table's 1 KB chunks in turn, every load in flight at once
    // (copied through registers a thread's 18 loads came back one after the other: 38 us of a
    // workgroup's ~200)
    const char* tb = reinterpret_cast<const char*>(src.ptr[first_u] + inst * src.stride[first_u]);
    const unsigned lds0 = (unsigned)(uintptr_t)lds;  // (the LDS byte address: low half of the flat one)
    const unsigned tb_bytes = (unsigned)tbn * 8u;    // (a multiple of 16; the scratch is 16-byte aligned)
    for (unsigned c0 = (unsigned)wave * 1024u; c0 < tb_bytes; c0 += WAVES * 1024u) {
      const unsigned off = c0 + (unsigned)lane * 16u;
      if (off < tb_bytes) lds_dma16(tb + off, __builtin_amdgcn_readfirstlane(lds0 + c0));
    }
    double* z = reinterpret_cast<double*>(lds) + tbn;
    for (int e = tid; e < nzero; e += BLOCK) z[e] = 0.0;
    if (tid * 8 < p.nparams) asm volatile("" ::"v"(pb[tid * 8]));  // (parameters: lines into L2)
  }
  const unsigned zero_off = (unsigned)tbn * 8u, d_off = (unsigned)(tbn + nzero) * 8u;
  // d (behind the table and the zeros; the same LDS serves the gradient's reduction at the end of
  // a diagonal block, so it is fetched again before the next one)
  auto fetch_d = [&]() __attribute__((always_inline)) {
    const char* dsrc = reinterpret_cast<const char*>(dvec);
    const unsigned d_bytes = (unsigned)(p.rtot + (p.rtot & 1)) * 8u, d0 = (unsigned)(uintptr_t)lds + d_off;
    for (unsigned c0 = (unsigned)wave * 1024u; c0 < d_bytes; c0 += WAVES * 1024u) {
      const unsigned off = c0 + (unsigned)lane * 16u;
      if (off < d_bytes) lds_dma16(dsrc + off, __builtin_amdgcn_readfirstlane(d0 + c0));
    }
  };
  bool d_stale = P != nullptr;
  const int li = lane & 15, lk = lane >> 4;
  const int32_t* cio = p.itab + p.off_t_cio;
  const int base0 = stages[TS_BASE] & 0xFFFF, sboff0 = stages[TS_SBOFFA];
  auto column = [&](int col, unsigned extra, unsigned& part, unsigned& mask) __attribute__((always_inline)) {
    const int2 ci = *reinterpret_cast<const int2*>(cio + ((size_t)base0 * p.t_nop + col) * 2);
    const bool valid = ((unsigned)ci.y >> 24) != (unsigned)T_SID_CONST;
    part = valid ? (unsigned)(ci.x - sboff0) * 8u + extra : zero_off + extra;
    mask = valid ? 0xFFFFFFFFu : 0u;
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's LDS-DMA loads have landed

  for (int job = 0; job < npairs; ++job) {
  int bi = 0, bj = 0;
  if (sym) {  // job -> block pair (bi <= bj)
    int pr = job, rowlen = nb;
    while (pr >= rowlen) {
      pr -= rowlen;
      --rowlen;
      ++bi;
    }
    bj = bi + pr;
  } else {
    bi = job / nb;
    bj = job - bi * nb;
  }
  const bool diag = bi == bj;
  if (!P && !diag) continue;  // only the constraints are wanted: diagonal blocks write them
  if (diag && d_stale) {
    __syncthreads();  // (nobody still reads the partial sums of the diagonal block before)
    fetch_d();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    d_stale = false;
  }
  // the wavefront's role and the 64 x 64 quadrant (wr, wc) of the block its tiles belong to: in a
  // diagonal block of a symmetric P the quadrant below the diagonal is a mirror image, so its
  // wavefront helps with the one above (tile rows 2, 3)
  const bool tri_block = diag && sym;
  const int role = !tri_block ? ROLE_FULL : (wave == 0 || wave == 3 ? ROLE_TRI : (wave == 1 ? ROLE_TOP : ROLE_BOT));
  const int wr = (tri_block && wave == 2) ? 0 : wave >> 1, wc = (tri_block && wave == 2) ? 1 : wave & 1;
  // ---- per lane: where its columns sit in the table (state-independent part) ---------------
  unsigned pa[4], pbt[4], ma[4], mb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    column(bi * T_BLOCK + wr * 64 + t * 16 + li, (unsigned)lk * 8u, pa[t], ma[t]);
    column(bj * T_BLOCK + wc * 64 + t * 16 + li, (unsigned)lk * 8u, pbt[t], mb[t]);
  }
  // the diagonal workgroup's second role: wavefront `wave` takes rows wave, wave + 4, ... of a stage,
  // lane `lane` two columns of block bi
  const int colA = bi * T_BLOCK + lane * 2;
  unsigned pq0, pq1, mq0, mq1;
  column(colA, 0u, pq0, mq0);
  column(colA + 1, 0u, pq1, mq1);
  const bool want_g = G != nullptr && diag;
  __syncthreads();  // (the table and d are in place; the block before is done with the LDS it borrowed)

  f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  double2 qacc{0.0, 0.0};

  // The stage records are read one stage ahead (a record is 64 bytes through the scalar cache: its
  // trip is over by the time the stage before has been multiplied; read where they are used, the
  // 288 records of a C4 instance's six blocks cost a tenth of the call in waits alone).
  struct Rec {
    int4 a, b, c, d;  // the 16 words (TS_*)
    double ca, cb;
  };
  auto load_rec = [&](int sx) __attribute__((always_inline)) {
    const int4* r4 = reinterpret_cast<const int4*>(stages + (size_t)sx * T_STAGE_WORDS);
    Rec r;
    r.a = r4[0];
    r.b = r4[1];
    r.c = r4[2];
    r.d = r4[3];
    r.ca = scoef[(sx * 2 + 0) * TK];
    r.cb = scoef[(sx * 2 + 1) * TK];
    return r;
  };
  int s = 0;
  Rec cur = load_rec(0);
  auto run_class = [&](auto role_tag, auto cls_tag) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_tag)::value;
    constexpr int N = decltype(cls_tag)::value;
    while (s < p.t_nstage) {
      const int info = cur.a.w;  // TS_INFO
      if ((info >> 16) != N) break;
      const Rec rec = cur;
      ++s;
      cur = load_rec(s < p.t_nstage ? s : 0);  // (in flight while this stage is worked on)
      const int sx = s - 1;
      const int fl = (info >> 8) & 255, nrows = info & 255;
      const unsigned ta = block_tiles((unsigned)rec.b.z, (unsigned)rec.b.w, bi);  // TS_MASKA_*
      const unsigned tb = block_tiles((unsigned)rec.c.x, (unsigned)rec.c.y, bj);  // TS_MASKB_*
      const bool for_p = P && (fl & TS_FLAG_P) && ta && tb && (phases & 2);
      const bool for_q = diag && P && ta && (phases & 4), for_g = want_g && (fl & TS_FLAG_G) && (phases & 8);
      if (!for_p && !for_q && !for_g) continue;
      const unsigned ua = (unsigned)rec.d.x * 8u, ub = (unsigned)rec.d.y * 8u;  // TS_UA, TS_UB
      const double ca = rec.ca, cb = rec.cb;
      const double w = P ? pb[rec.b.x] : 0.0;  // TS_WPARAM
      if (for_q || for_g) {
        const double aim = pb[rec.b.y];  // TS_AIMPARAM
        const double scale = (fl & TS_FLAG_HALF) ? 0.5 : 1.0;
        const int drow = rec.a.z;  // TS_DROW
        // this wavefront's four rows: the riding rows of G and their arrows first, all at once
        // (two dependent trips through the scalar cache for the lot, not two per row)
        int4 pg[TK / WAVES];
        double ar0[TK / WAVES], ar1[TK / WAVES];
#pragma unroll
        for (int rr = 0; rr < TK / WAVES; ++rr) {
          const int i = wave + WAVES * rr;
          pg[rr] = (for_g && i < nrows) ? pigs[sx * TK + i] : int4{-1, 0, -1, 0};
        }
#pragma unroll
        for (int rr = 0; rr < TK / WAVES; ++rr) {
          ar0[rr] = pb[pg[rr].x >= 0 ? pg[rr].y : 0];
          ar1[rr] = pb[pg[rr].z >= 0 ? pg[rr].w : 0];
        }
#pragma unroll
        for (int rr = 0; rr < TK / WAVES; ++rr) {
          const int i = wave + WAVES * rr;
          if (i >= nrows) break;
          double2 a;
          a.x = lds_f64(lds, pq0 + (ua & mq0) + i * 8) * ca;
          a.y = lds_f64(lds, pq1 + (ua & mq1) + i * 8) * ca;
          if (colA < no) {
            if (pg[rr].x >= 0)
              store_result(reinterpret_cast<double2*>(G + ((size_t)inst * p.nc + pg[rr].x) * no + colA),
                           double2{ar0[rr] * a.x, ar0[rr] * a.y});
            if (pg[rr].z >= 0)
              store_result(reinterpret_cast<double2*>(G + ((size_t)inst * p.nc + pg[rr].z) * no + colA),
                           double2{ar1[rr] * a.x, ar1[rr] * a.y});
          }
          if (for_q) {
            const double r = scale * (lds_f64(lds, d_off + (unsigned)(drow + i) * 8u) - aim);
            qacc.x = fma(w * a.x, r, qacc.x);
            qacc.y = fma(w * a.y, r, qacc.y);
          }
        }
      }
      if constexpr (N > 0)
        if (for_p) mfma_toeplitz<ROLE, N>(acc, lds, pa, pbt, ma, mb, ua, ub, w * ca, cb, lk, nrows);
    }
  };
  // which tiles of acc hold results, by role (what the epilogue writes): every class's union
  unsigned computed = 0;  // bit ta * 4 + tb
  auto run_role = [&](auto role_tag) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_tag)::value;
    run_class(role_tag, std::integral_constant<int, 0>{});
    run_class(role_tag, std::integral_constant<int, 1>{});
    run_class(role_tag, std::integral_constant<int, 2>{});
    run_class(role_tag, std::integral_constant<int, 3>{});
    run_class(role_tag, std::integral_constant<int, 4>{});
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
        if (TileRange<ROLE, 4>::has(ta, tb)) computed |= 1u << (ta * 4 + tb);
  };
  if (!(phases & 1))
    ;  // (profiling: no stage loop at all)
  else if (role == ROLE_FULL)
    run_role(std::integral_constant<int, ROLE_FULL>{});
  else if (role == ROLE_TRI)
    run_role(std::integral_constant<int, ROLE_TRI>{});
  else if (role == ROLE_TOP)
    run_role(std::integral_constant<int, ROLE_TOP>{});
  else
    run_role(std::integral_constant<int, ROLE_BOT>{});

  // ---- rows of G that ride on no stage, h ------------------------------------------------
  if (want_g) {
    const int32_t* grow = p.itab + p.off_t_grow;
    const int32_t* rrw = p.itab + p.off_rs_rr;
    if (p.t_ngrest > 0) {  // (through the column tables, from the scratch)
      __shared__ const double* s_base[NSTREAM];
      stream_bases(p, src, inst, s_base, tid);
      __syncthreads();
      RowTables rt;
      rt.rowptr = p.itab + p.off_rowptr;
      rt.entbase = p.itab + p.off_entbase;
      rt.entk = p.itab + p.off_entk;
      rt.coef = p.dtab + p.doff_entcoef;
      rt.cio = cio;
      rt.nop = p.t_nop;
      const int32_t* grest = p.itab + p.off_t_grest;
      int baseG = -1;
      ColRef crG;
      crG.p0 = crG.p1 = p.dtab;
      crG.rs0 = crG.rs1 = 0;
      for (int x0 = wave; x0 < p.t_ngrest; x0 += WAVES) {
        const int R = grest[x0];
        const int32_t* x = rrw + (size_t)R * RS_RR_WORDS;
        const int naxes = x[RR_NAXES];
        double2 out{0.0, 0.0};
        for (int ax = 0; ax < naxes; ++ax) {
          const double ar = pb[x[RR_ARROW + ax]];
          const double2 v = compose_row2(rt, grow[R * RS_AXMAX + ax], colA, s_base, baseG, crG);
          out.x = fma(ar, v.x, out.x);
          out.y = fma(ar, v.y, out.y);
        }
        if (colA < no)
          store_result(reinterpret_cast<double2*>(G + ((size_t)inst * p.nc + R) * no + colA), out);
      }
    }
    if (bi == 0)  // h: (extreme + arrow . center) - arrow . d
      for (int R = tid; R < p.nc; R += BLOCK) {
        const int32_t* x = rrw + (size_t)R * RS_RR_WORDS;
        double ac = 0.0, ad = 0.0;
        for (int ax = 0; ax < x[RR_NAXES]; ++ax) {
          const double ar = pb[x[RR_ARROW + ax]];
          ac += ar * pb[x[RR_CENTER + ax]];
          ad = fma(ar, dvec[grow[R * RS_AXMAX + ax]], ad);
        }
        h[(size_t)inst * p.nc + R] = (pb[x[RR_EXTREME]] + ac) - ad;
      }
  }
  if (!P) continue;

  double* part = reinterpret_cast<double*>(lds + d_off);  // (d is no longer needed by this block)
  double dPc = 0.0, dqc = 0.0;  // the diagonal gterms on column bi * T_BLOCK + tid
  if (diag && tid < T_BLOCK && bi * T_BLOCK + tid < no)
    diagonal_of_column(p, pb, bi * T_BLOCK + tid, dPc, dqc);
  if (diag) {  // ... and the wavefronts' partial gradients, through LDS
    __syncthreads();
    d_stale = true;
    if (tid < T_BLOCK) part[WAVES * T_BLOCK + tid] = dPc;
    part[wave * T_BLOCK + lane * 2] = qacc.x;
    part[wave * T_BLOCK + lane * 2 + 1] = qacc.y;
    __syncthreads();
  }
  double* Pb = P + (size_t)inst * no * no;
  // mirror images: everything of a block off the diagonal of a symmetric P; in a diagonal block
  // the tiles above the tile diagonal (a tile ON it holds both of its triangles)
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      if (!((computed >> (ta * 4 + tb)) & 1u) || !(phases & 32)) continue;
      const bool mirror = sym && (!diag || wr != wc || ta != tb);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = bi * T_BLOCK + wr * 64 + ta * 16 + lk + 4 * reg;
        const int col = bj * T_BLOCK + wc * 64 + tb * 16 + li;
        if (row < no && col < no) {
          double v = acc[ta][tb][reg];
          if (diag && row == col) v += part[WAVES * T_BLOCK + (row - bi * T_BLOCK)];
          Pb[(size_t)row * no + col] = v;
          if (mirror) Pb[(size_t)col * no + row] = v;
        }
      }
    }
  if (diag && tid < T_BLOCK) {  // the gradient: the four wavefronts hold the sums over their rows
    const int c = bi * T_BLOCK + tid;
    if (c < no) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) sum += part[w * T_BLOCK + tid];
      q[(size_t)inst * no + c] = sum + dqc;
    }
  }
  }  // (the next block of this instance)
}

// ---------------------------------------------------------------------------
// Scan form (plans with H_T_SCAN: every Hessian term is w (c M_i)^T (c M_i) over ALL N rows of one
// state i of the generated group).  M_i[k][(j, l)] = T_ij[k - l] (T_ij[d] = (A^d B)[i][j], d >= 0,
// tools.py:27-31) makes the block of P on the columns of two inputs a sum along diagonals:
//     P[(j,l)][(j',l')] = C[(j,l)][(j',l')] + P[(j,l+1)][(j',l'+1)]     (nothing behind l = N - 1),
//     C[r][c] = sum_g (w_g c_g^2) M_g[N-1][r] M_g[N-1][c]                 (the LAST row of every state)
// -- K multiply-adds per element of P instead of the K N of the product (C4: 4 % of the matrix
// core's work in the Toeplitz form above) -- and the gradient an adjoint recursion,
//     q[(j,l)] = B[:,j] . lam_l,   lam_l = rho_l + A^T lam_{l+1},   rho_l[i] = sum_g (w_g c_g)(d_g[l] - aim_g)
// (O(N n^2) instead of a correlation of the table with d - aim), which leaves a kernel that only has to
// WRITE: 5.9 MB per C4 instance, 80 % of it rows of G that are windows of the same table.  One workgroup
// per instance; LDS: d | parameters | lam | tickets | Tc, the table WITHOUT the zero halves of TB
// (Tc[(i m + j) N + d] = T_ij[d]: 37 KB for C4 instead of 74; a column l > k of row k is masked by a
// compare instead of read as a zero); no barrier behind the set-up.  Wavefront 0 runs the recursion
// and writes q; then all draw tickets:
//   * a row block of P (an input's N rows, from l = N - 1 down): lane = column of an input's block,
//     its K weighted last-row values for every column block in registers, the row's K values by
//     broadcast reads, the running sums shifted one lane per row (DPP) -- 8 bytes per lane, 512-byte
//     runs, a whole row of P per step;
//   * four rows of G: 16 bytes per lane straight out of the table times the row's arrow.
// h comes first, out of LDS; rows of G that are no single state row and unknowns outside the input
// blocks are composed behind the tickets through the column tables (cold paths).
// ---------------------------------------------------------------------------
constexpr int SCAN_GROUP = 8;     // rows of G per ticket, at most (MPCASM_SCAN_GROUP: 1 .. 8)
constexpr int SCAN_GCH_MAX = 8;   // 128-column chunks of a row of G: no <= 1024
constexpr int SCAN_AREG = 16;     // states up to which wavefront 0 keeps its column of A in registers

#ifndef MPCASM_SCAN_DPP
#define MPCASM_SCAN_DPP 1   // (tools/microbench/dpp_wave_shift.hip: wave_shl:1 is lane i <- lane i + 1, 0 into lane 63)
#endif
// lane i <- lane i + 1, 0 into lane 63
__device__ __forceinline__ double lane_from_next(double v) {
#if MPCASM_SCAN_DPP
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xF, 0xF, true);  // wave_shl:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
#else
  const double x = __shfl_down(v, 1, 64);
  return (threadIdx.x & 63) == 63 ? 0.0 : x;
#endif
}

// where things sit in the scan kernel's dynamic LDS (bytes); the table last, at least N doubles in:
// a window that starts before its row's first step reads what lies in front (and is masked)
struct ScanLds {  // (byte offsets)
  unsigned d, par, lam, ticket, rows, tb, pw, total;
};
// `rows_in_lds`: the records of the rows of G (16 bytes each: table row, arrow slot, coefficient) ride
// along -- read per row out of LDS, in order with the table's own reads, instead of through the scalar
// cache (whose returns every LDS wait would have to include)
__host__ __device__ inline ScanLds scan_lds(int dlen, int nparams, int n, int m, int N, int nc, bool rows_in_lds) {
  ScanLds x;
  x.d = 0;
  x.par = x.d + (unsigned)dlen * 8u;
  x.lam = x.par + (unsigned)((nparams + 2) & ~1) * 8u;
  x.ticket = x.lam + (unsigned)((n * N + 1) & ~1) * 8u;
  x.rows = x.ticket + 16u;
  x.tb = x.rows + (rows_in_lds ? (unsigned)nc * 16u : 0u);
  if (x.tb < (unsigned)N * 8u) x.tb = (unsigned)((N + 1) & ~1) * 8u;
  x.pw = x.tb + (unsigned)(n * m * N) * 8u;                   // two n x n powers of A (fused set-up)
  x.total = x.pw + (unsigned)(2 * n * n) * 8u;
  return x;
}
// two workgroups per CU as long as an instance stays below this
constexpr unsigned SCAN_HALF_CU = 80 * 1024;

template <int KP, int CB>
__global__ __launch_bounds__(BLOCK, 2) void toeplitz_scan_kernel(
    PlanDev p, SrcTable src, const double* __restrict__ sysA, long long strideA,
    const double* __restrict__ sysB, long long strideB, const double* __restrict__ params,
    const double* __restrict__ work, long long work_stride, double* __restrict__ P,
    double* __restrict__ q, double* __restrict__ G, double* __restrict__ h, int batch, int dlen,
    int whole_lines, int group, int rows_in_lds, int phases, const double* __restrict__ given, int fused) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long inst = blockIdx.x;
  if (inst >= batch) return;
  const int no = p.no, nc = p.nc, K = p.t_scan, nblk = p.t_scan_nblk;
  const double* pb = params + (size_t)inst * p.nparams;
  const int32_t* grp = p.itab + p.off_t_lti;
  const int n = grp[TL_N], m = grp[TL_M], N = grp[TL_HORIZON];
  const int32_t* ids = p.itab + p.off_t_lti_ids + grp[TL_IDS];
  const int first_u = ids[0];
  const int2* blk = reinterpret_cast<const int2*>(p.itab + p.off_t_scan_blk);
  const int4* gts = reinterpret_cast<const int4*>(p.itab + p.off_t_scan_gt);
  const double* gcs = p.dtab + p.t_doff_scan_gc;
  const int32_t* colblk = p.itab + p.off_t_scan_colblk;
  const ScanLds L = scan_lds(dlen, p.nparams, n, m, N, nc, rows_in_lds != 0);
  int* ticket = reinterpret_cast<int*>(lds + L.ticket);
  // ---- set-up: the table (the value halves of TB's rows), d, the parameters into LDS -------------
  if (fused) {
    // Fused set-up (H_T_SCAN_FUSED): no pre-pass has run.  The whole workgroup makes the table where it is needed,
    // Tc[(i m + j) N + d] = (A^d B)[i][j], and the free response x_k = A^{k+1} given, by DOUBLING: with A^h at
    // hand, X_{d+h} = A^h X_d for every d < h at once -- log2(N) rounds of independent 12-term products
    // instead of N dependent steps (the reference's recurrence X_d = A X_{d-1}, tools.py:24-29, as one
    // wavefront's chain of LDS round trips took 85 us per instance; the association differs from it by a few
    // ulp, as in the persistent kernel's on-chip tables).  Then d[first row of the term + k] = c x_k[state]
    // for every term: what compose_d_kernel computes through the column tables when every workspace row
    // is such a row.
    const double* A = sysA + inst * strideA;
    const double* Bm = sysB + inst * strideB;
    double* Tc = reinterpret_cast<double*>(lds + L.tb);
    double* dl = reinterpret_cast<double*>(lds + L.d);
    double* xb = reinterpret_cast<double*>(lds + L.lam);   // [N][n] (the gradient's buffer, not yet in use)
    double* pw = reinterpret_cast<double*>(lds + L.pw);    // A^have, and the next power beside it
    const int nn = n * n, nm = n * m;
    for (int e = tid; e < nn; e += BLOCK) pw[e] = A[e];
    for (int e = tid; e < nm; e += BLOCK) Tc[e * N] = Bm[e];
    if (tid < n) {
      double acc = 0.0;
      for (int t = 0; t < n; ++t) acc = fma(A[tid * n + t], given[inst * p.ng + t], acc);
      xb[tid] = acc;                                       // x_0 = A given
    }
    __syncthreads();
    int cur = 0;
    for (int have = 1; have < N;) {
      const int cnt = have < N - have ? have : N - have;
      const double* Ph = pw + cur * nn;
      double* Pn = pw + (cur ^ 1) * nn;
      const int e1 = cnt * nm, e2 = e1 + cnt * n, e3 = e2 + (have + cnt < N ? nn : 0);
      for (int e = tid; e < e3; e += BLOCK) {
        double acc = 0.0;
        if (e < e1) {                    // X_{have + d} = A^have X_d: element (i, j), d fastest
          const int r = e / cnt, d = e - r * cnt, i = r / m, j = r - i * m;
          for (int t = 0; t < n; ++t) acc = fma(Ph[i * n + t], Tc[(t * m + j) * N + d], acc);
          Tc[r * N + have + d] = acc;
        } else if (e < e2) {             // x_{have + d} = A^have x_d
          const int f = e - e1, i = f / cnt, d = f - i * cnt;
          for (int t = 0; t < n; ++t) acc = fma(Ph[i * n + t], xb[d * n + t], acc);
          xb[(have + d) * n + i] = acc;
        } else {                         // A^{2 have}
          const int f = e - e2, i = f / n, c = f - i * n;
          for (int t = 0; t < n; ++t) acc = fma(Ph[i * n + t], Ph[t * n + c], acc);
          Pn[f] = acc;
        }
      }
      __syncthreads();
      have += cnt;
      cur ^= 1;
    }
    const unsigned per_state = (unsigned)(m * N);
    for (int e = tid; e < K * N; e += BLOCK) {
      const int g = e / N, k = e - g * N;
      const int4 tg = gts[g];
      dl[tg.w + k] = gcs[g] * xb[k * n + (int)((unsigned)tg.x / per_state)];
    }
    if (tid == 0 && dlen > p.rtot) dl[p.rtot] = 0.0;
  } else {
    const char* tb = reinterpret_cast<const char*>(src.ptr[first_u] + inst * src.stride[first_u]);
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const unsigned half = (unsigned)N >> 1;                    // 16-byte pieces of a row (N is even)
    const unsigned pieces = (unsigned)(n * m) * half;
    for (unsigned e0 = (unsigned)wave * 64u; e0 < pieces; e0 += WAVES * 64u) {
      const unsigned e = e0 + (unsigned)lane;
      if (e < pieces) {
        const unsigned r = e / half, x = e - r * half;
        lds_dma16(tb + ((size_t)r * 2 * N + N) * 8 + x * 16u, __builtin_amdgcn_readfirstlane(lds0 + L.tb + e0 * 16u));
      }
    }
    const char* dsrc = reinterpret_cast<const char*>(work + inst * work_stride);
    const unsigned d_bytes = (unsigned)dlen * 8u;
    for (unsigned c0 = (unsigned)wave * 1024u; c0 < d_bytes; c0 += WAVES * 1024u) {
      const unsigned off = c0 + (unsigned)lane * 16u;
      if (off < d_bytes) lds_dma16(dsrc + off, __builtin_amdgcn_readfirstlane(lds0 + L.d + c0));
    }
  }
  {
    double* par = reinterpret_cast<double*>(lds + L.par);
    for (int e = tid; e <= p.nparams; e += BLOCK) par[e] = e < p.nparams ? pb[e] : 0.0;  // ([nparams] reads 0.0)
    if (tid < 2) ticket[tid] = 0;  // (row blocks of P, groups of rows of G)
    if (rows_in_lds && G != nullptr) {
      const int2* sg = reinterpret_cast<const int2*>(p.itab + p.off_t_scan_grow);
      const double* sc = p.dtab + p.t_doff_scan_gcoef;
      for (int R = tid; R < nc; R += BLOCK) {
        const int2 r = sg[R];
        *reinterpret_cast<int2*>(lds + L.rows + (unsigned)R * 16u) = r;
        *reinterpret_cast<double*>(lds + L.rows + (unsigned)R * 16u + 8u) = sc[R];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's LDS-DMA loads have landed
  __syncthreads();

  // ---- h: (extreme + arrow . center) - arrow . d ---------------------------------------------------
  if (G != nullptr && (phases & 8)) {
    const int32_t* grow = p.itab + p.off_t_grow;
    const int32_t* rrw = p.itab + p.off_rs_rr;
    for (int R = tid; R < nc; R += BLOCK) {
      const int32_t* x = rrw + (size_t)R * RS_RR_WORDS;
      double ac = 0.0, ad = 0.0;
      for (int ax = 0; ax < x[RR_NAXES]; ++ax) {
        const double ar = pb[x[RR_ARROW + ax]];
        ac += ar * pb[x[RR_CENTER + ax]];
        ad = fma(ar, lds_f64(lds, L.d + (unsigned)grow[R * RS_AXMAX + ax] * 8u), ad);
      }
      h[(size_t)inst * nc + R] = (pb[x[RR_EXTREME]] + ac) - ad;
    }
  }
  // ---- the gradient, by wavefront 0 while the others already draw tickets --------------------------
  if (P != nullptr && (phases & 4) && wave == 0) {
    const double* A = sysA + inst * strideA;
    const double* Bm = sysB + inst * strideB;
    // lam[i][l] in LDS: first rho_l[i] = sum over the terms on state i of (w c)(d[l] - aim) -- lanes over
    // the steps, the terms one after the other (their constants are wave-uniform) --, then, step by
    // step from the last, lam_l[i] += sum_t A[t][i] lam_{l+1}[t] on lane i < n
    double* lam = reinterpret_cast<double*>(lds + L.lam);
    for (int e = lane; e < n * N; e += 64) lam[e] = 0.0;
    const unsigned per_state = (unsigned)(m * N);
    for (int g = 0; g < K; ++g) {
      const int4 x = gts[g];
      const double wc = pb[x.y] * gcs[g], aim = pb[x.z];
      const unsigned st = (unsigned)x.x / per_state;
      for (int l = lane; l < N; l += 64)
        lam[st * N + l] += wc * (lds_f64(lds, L.d + (unsigned)(x.w + l) * 8u) - aim);
    }
    double acol[SCAN_AREG];
    const bool in_regs = n <= SCAN_AREG;
#pragma unroll
    for (int t = 0; t < SCAN_AREG; ++t) acol[t] = (in_regs && t < n && lane < n) ? A[t * n + lane] : 0.0;
    asm volatile("" ::: "memory");
    for (int l = N - 2; l >= 0; --l) {
      double acc = lane < n ? lam[lane * N + l] : 0.0;
      if (in_regs) {
#pragma unroll
        for (int t = 0; t < SCAN_AREG; ++t)
          if (t < n) acc = fma(acol[t], lam[t * N + l + 1], acc);
      } else {
        for (int t = 0; t < n; ++t) acc = fma(lane < n ? A[t * n + lane] : 0.0, lam[t * N + l + 1], acc);
      }
      if (lane < n) lam[lane * N + l] = acc;
      asm volatile("" ::: "memory");  // (a wavefront's LDS operations complete in order: the next step reads these)
    }
    for (int c = lane; c < no; c += 64) {   // q[(j, l)] = B[:, j] . lam_l + the diagonal terms
      const int b = colblk[c];
      double dP, dq;
      diagonal_of_column(p, pb, c, dP, dq);
      double acc = 0.0;
      if (b >= 0) {
        const int j = blk[b].y / N, l = c - blk[b].x;
        for (int i = 0; i < n; ++i) acc = fma(Bm[i * m + j], lds_f64(lds, L.lam + (unsigned)(i * N + l) * 8u), acc);
      }
      q[(size_t)inst * no + c] = acc + dq;
    }
  }

  // ---- tickets, first round: the row blocks of P -------------------------------------------------
  double* Pb = P + (size_t)inst * no * no;
  if (P != nullptr && (phases & 2)) {
    // where every term's state starts in Tc (wave-uniform), and the lane's K weighted last-row values
    // (w c^2) Tc[state, column block, N - 1 - lane] in every column block (0 beyond its N columns)
    unsigned sb8[KP];
    double Tl[CB][KP];
#pragma unroll
    for (int g = 0; g < KP; ++g) {
      const int4 x = gts[g < K ? g : 0];
      const double cg = gcs[g < K ? g : 0];
      sb8[g] = L.tb + (unsigned)x.x * 8u;
      const double wcc = (pb[x.y] * cg) * cg;
#pragma unroll
      for (int s = 0; s < CB; ++s) {
        const bool live = s < nblk && g < K && lane < N;
        const unsigned a = live ? sb8[g] + (unsigned)(blk[s < nblk ? s : 0].y + N - 1 - lane) * 8u : L.tb;
        Tl[s][g] = live ? wcc * lds_f64(lds, a) : 0.0;
      }
    }
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nblk) break;
      const int bi = t;
      const int r0 = blk[bi].x;
      // the diagonal terms on this block's own columns (a cost on the input itself)
      double dP = 0.0, dq = 0.0;
      if (lane < N) diagonal_of_column(p, pb, r0 + lane, dP, dq);
      double run[CB];
#pragma unroll
      for (int s = 0; s < CB; ++s) run[s] = 0.0;
      const unsigned rowpart = (unsigned)(blk[bi].y + N - 1) * 8u;
      for (int l = N - 1; l >= 0; --l) {
        double beta[KP];
#pragma unroll
        for (int g = 0; g < KP; ++g)
          beta[g] = g < K ? lds_f64(lds, sb8[g] + rowpart - (unsigned)l * 8u) : 0.0;
        double* prow = Pb + (size_t)(r0 + l) * no;
#pragma unroll
        for (int s = 0; s < CB; ++s) {
          if (s >= nblk) break;
          double C = 0.0;
#pragma unroll
          for (int g = 0; g < KP; ++g) C = fma(beta[g], Tl[s][g], C);
          run[s] = C + lane_from_next(run[s]);
          const double v = (s == bi && lane == l) ? run[s] + dP : run[s];
          if (lane < N && (phases & 32)) {
            if (whole_lines)
              store_result(prow + blk[s].x + lane, v);
            else
              prow[blk[s].x + lane] = v;
          }
        }
      }
    }
  }

  // ---- tickets, second round: rows of G, SCAN_GROUP at a time ------------------------------------
  double* Gb = G + (size_t)inst * nc * no;
  if (G != nullptr && (phases & 8)) {
    // per lane: its two columns of every 128-column chunk of a row -- (j N - l) 8 + where Tc starts, and
    // the column's step l (a row of step k holds zeros in l > k; a column of no input block: always)
    const int nch = (no + 127) >> 7;
    int gp0[SCAN_GCH_MAX], gp1[SCAN_GCH_MAX], gl0[SCAN_GCH_MAX], gl1[SCAN_GCH_MAX];
#pragma unroll
    for (int ch = 0; ch < SCAN_GCH_MAX; ++ch) {
      gp0[ch] = gp1[ch] = (int)L.tb;
      gl0[ch] = gl1[ch] = 0x7FFFFFFF;
      if (ch < nch) {
        const int c = ch * 128 + lane * 2;
        if (c < no) {
          const int b = colblk[c];
          if (b >= 0) {
            gl0[ch] = c - blk[b].x;
            gp0[ch] = (int)L.tb + (blk[b].y - gl0[ch]) * 8;
          }
        }
        if (c + 1 < no) {
          const int b = colblk[c + 1];
          if (b >= 0) {
            gl1[ch] = c + 1 - blk[b].x;
            gp1[ch] = (int)L.tb + (blk[b].y - gl1[ch]) * 8;
          }
        }
      }
    }
    const int2* sgrow = reinterpret_cast<const int2*>(p.itab + p.off_t_scan_grow);
    const double* sgcoef = p.dtab + p.t_doff_scan_gcoef;
    const int ngroups = (nc + group - 1) / group;
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(ticket + 1, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= ngroups) break;
      const int R0 = t * group;
      int2 rec[SCAN_GROUP];
      double cf[SCAN_GROUP];
#pragma unroll
      for (int rr = 0; rr < SCAN_GROUP; ++rr) {
        const int R = (rr < group && R0 + rr < nc) ? R0 + rr : nc - 1;
        if (rows_in_lds) {
          rec[rr] = *reinterpret_cast<const int2*>(lds + L.rows + (unsigned)R * 16u);
          cf[rr] = lds_f64(lds, L.rows + (unsigned)R * 16u + 8u);
        } else {
          rec[rr] = sgrow[R];
          cf[rr] = sgcoef[R];
        }
      }
#pragma unroll
      for (int rr = 0; rr < SCAN_GROUP; ++rr) {
        const int R = R0 + rr;
        if (rr >= group || R >= nc) break;
        const int ux = __builtin_amdgcn_readfirstlane(rec[rr].x), slot = __builtin_amdgcn_readfirstlane(rec[rr].y);
        if (ux < 0) continue;
        const int u8 = ux * 8;
        const int k = ux % N;   // (the row's step: i m N + k)
        const double ar = lds_f64(lds, L.par + (unsigned)slot * 8u) * cf[rr];
        double* grow_out = Gb + (size_t)R * no + lane * 2;
#pragma unroll
        for (int ch = 0; ch < SCAN_GCH_MAX; ++ch) {
          if (ch >= nch) break;
          const double t0 = lds_f64(lds, (unsigned)(gp0[ch] + u8)), t1 = lds_f64(lds, (unsigned)(gp1[ch] + u8));
          double2 v;
          v.x = gl0[ch] <= k ? ar * t0 : 0.0;
          v.y = gl1[ch] <= k ? ar * t1 : 0.0;
          if (ch * 128 + lane * 2 < no) {
            if (whole_lines)
              store_result(reinterpret_cast<double2*>(grow_out + ch * 128), v);
            else
              *reinterpret_cast<double2*>(grow_out + ch * 128) = v;
          }
        }
      }
    }
  }

  // ---- cold paths --------------------------------------------------------------------------------
  // rows of G that are no single state row: through the column tables, from the scratch
  if (G != nullptr && p.t_scan_ngrest > 0 && (phases & 8)) {
    __shared__ const double* s_base[NSTREAM];
    stream_bases(p, src, inst, s_base, tid);
    __syncthreads();
    RowTables rt;
    rt.rowptr = p.itab + p.off_rowptr;
    rt.entbase = p.itab + p.off_entbase;
    rt.entk = p.itab + p.off_entk;
    rt.coef = p.dtab + p.doff_entcoef;
    rt.cio = p.itab + p.off_t_cio;
    rt.nop = p.t_nop;
    const int32_t* grow = p.itab + p.off_t_grow;
    const int32_t* rrw = p.itab + p.off_rs_rr;
    const int32_t* grest = p.itab + p.off_t_scan_grest;
    for (int x0 = wave; x0 < p.t_scan_ngrest; x0 += WAVES) {
      const int R = grest[x0];
      const int32_t* x = rrw + (size_t)R * RS_RR_WORDS;
      const int naxes = x[RR_NAXES];
      for (int c0 = 0; c0 < p.t_nop; c0 += T_BLOCK) {
        const int colA = c0 + lane * 2;
        int baseG = -1;
        ColRef crG;
        crG.p0 = crG.p1 = p.dtab;
        crG.rs0 = crG.rs1 = 0;
        double2 out{0.0, 0.0};
        for (int ax = 0; ax < naxes; ++ax) {
          const double ar = pb[x[RR_ARROW + ax]];
          const double2 v = compose_row2(rt, grow[R * RS_AXMAX + ax], colA, s_base, baseG, crG);
          out.x = fma(ar, v.x, out.x);
          out.y = fma(ar, v.y, out.y);
        }
        if (colA < no) *reinterpret_cast<double2*>(Gb + (size_t)R * no + colA) = out;
      }
    }
  }
  // unknowns in no input block: their rows and columns of P hold the diagonal terms only
  if (P != nullptr && p.t_scan_nother > 0 && (phases & 2))
    for (int r = wave; r < no; r += WAVES) {
      const bool row_other = colblk[r] < 0;
      for (int c = lane; c < no; c += 64)
        if (row_other || colblk[c] < 0) {
          double dP = 0.0, dq = 0.0;
          if (r == c) diagonal_of_column(p, pb, c, dP, dq);
          Pb[(size_t)r * no + c] = dP;
        }
    }
}

inline unsigned ceil_div(unsigned a, unsigned b) { return (a + b - 1) / b; }
inline long ceil_div(long a, long b) { return (a + b - 1) / b; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }


// ---------------------------------------------------------------------------
// Shared-model form (round 4): every source of the launch is ONE tensor for the whole batch (stride 0:
// one model, many states -- a fleet on the same robot; C4 with its S, U read from memory).  Then
//   * the composed rows V[r][c] are the same for every instance: composed ONCE (shared_rows_kernel);
//   * P is linear in the weights, P_b = sum_g w_b[g] K_g with K_g the Hessian of the plan at "weight g
//     = 1, all others 0": the T <= SH_TMAX matrices K_g come out of the general kernel itself, run on T
//     pseudo-instances whose parameters are those unit vectors -- the same arithmetic, once per launch
//     instead of once per instance;
//   * a row of G is sum_ax arrow_b[ax] V[row_ax]: a scaled copy.
// What is left per instance is streaming: 8 no^2 + 8 nc no bytes of weighted sums written in flat
// chunks of 1024 doubles (whole lines, nontemporal), the operands of a chunk in registers for all the
// instances a workgroup walks -- and q, h (a GEMV with the rows of V against w s (d - aim)).
// The general kernel multiplies the same tiles again for every instance: 29 ms per 8192 C4 instances.
// ---------------------------------------------------------------------------
constexpr int SH_TMAX = 32;     // weight parameters with a Hessian term
constexpr int SH_QB = 4;        // instances a workgroup of the q / h kernel takes together

struct SharedSlots {
  int n;
  int slot[SH_TMAX];
};

// scratch behind the per-instance part of the workspace (doubles)
struct SharedScratch {
  size_t v, k, pseudo, zero, qdummy, total;
};
__host__ __device__ inline SharedScratch shared_scratch(const PlanDev& p) {
  SharedScratch x;
  x.v = 0;
  x.k = x.v + (size_t)p.rtot * p.t_nop;
  x.pseudo = x.k + (size_t)SH_TMAX * p.no * p.no;
  x.zero = x.pseudo + (size_t)SH_TMAX * (p.nparams + 2);
  x.qdummy = x.zero + (size_t)p.rtot + 2;
  x.total = x.qdummy + (size_t)SH_TMAX * p.no + 2;
  x.total += x.total & 1;
  return x;
}

// pseudo-instance g: weight slot[g] = 1, every other parameter 0; the zero d vector
__global__ __launch_bounds__(BLOCK) void shared_pseudo_kernel(SharedSlots sl, int nparams, int rtot,
                                                             double* __restrict__ pseudo,
                                                             double* __restrict__ zero) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < sl.n * nparams) {
    const int g = i / nparams, k = i - g * nparams;
    pseudo[i] = k == sl.slot[g] ? 1.0 : 0.0;
  }
  if (i < rtot) zero[i] = 0.0;
}

// V[r][c], c < nop: every workspace row once, through the column tables (the sources are shared)
__global__ __launch_bounds__(BLOCK) void shared_rows_kernel(PlanDev p, SrcTable src,
                                                           double* __restrict__ V) {
  __shared__ const double* s_base[NSTREAM];
  const int tid = threadIdx.x;
  stream_bases(p, src, 0, s_base, tid);
  __syncthreads();
  const int pairs = p.t_nop / 2;
  const long e = (long)blockIdx.x * BLOCK + tid;
  if (e >= (long)p.rtot * pairs) return;
  const int r = (int)(e / pairs), col = (int)(e - (long)r * pairs) * 2;
  RowTables rt;
  rt.rowptr = p.itab + p.off_rowptr;
  rt.entbase = p.itab + p.off_entbase;
  rt.entk = p.itab + p.off_entk;
  rt.coef = p.dtab + p.doff_entcoef;
  rt.cio = p.itab + p.off_t_cio;
  rt.nop = p.t_nop;
  int cur = -1;
  ColRef cr;
  cr.p0 = cr.p1 = p.dtab;
  cr.rs0 = cr.rs1 = 0;
  const double2 v = compose_row2(rt, r, col, s_base, cur, cr);
  *reinterpret_cast<double2*>(V + (size_t)r * p.t_nop + col) = v;
}

// P_b = sum_g w_b[g] K_g.  A workgroup takes a flat range of SH_PRANGE elements of P for SH_PIB instances and
// walks it from front to back (every instance's stream of stores consecutive in memory, as for G below);
// per step a thread reads its 16 bytes of every K_g out of L2 -- TG / SH_PIB bytes read per byte written --
// and the weights of its instances out of LDS.
constexpr int SH_PIB = 16;           // instances of a workgroup
constexpr int SH_PRANGE = 4096;      // elements of P per instance and workgroup: 8 steps of 512

template <int TG>
__global__ __launch_bounds__(BLOCK) void shared_p_kernel(SharedSlots sl, long nn,
                                                        const double* __restrict__ K,
                                                        const double* __restrict__ params, int nparams,
                                                        double* __restrict__ P, int batch) {
  // (the weights in LDS before the first store: a load in the loop waits for the stores in flight in front
  // of it -- one counter counts both)
  __shared__ double sh_w[TG][SH_PIB];
  const long r0 = (long)blockIdx.x * SH_PRANGE;
  const long r1 = r0 + SH_PRANGE < nn ? r0 + SH_PRANGE : nn;
  const long b0 = (long)blockIdx.y * SH_PIB;
  const int nb = (int)(b0 + SH_PIB <= batch ? SH_PIB : batch - b0);
  for (int i = threadIdx.x; i < TG * SH_PIB; i += BLOCK) {
    const int g = i / SH_PIB, bb = i - g * SH_PIB;
    sh_w[g][bb] = (g < sl.n && bb < nb) ? params[(size_t)(b0 + bb) * nparams + sl.slot[g]] : 0.0;
  }
  __syncthreads();
  for (long e = r0 + threadIdx.x * 2; e < r1; e += BLOCK * 2) {   // (nn is even: pairs never straddle the end)
    double2 k[TG];
#pragma unroll
    for (int g = 0; g < TG; ++g)
      k[g] = g < sl.n ? *reinterpret_cast<const double2*>(K + (size_t)g * nn + e) : double2{0.0, 0.0};
#pragma unroll 4
    for (int bb = 0; bb < SH_PIB; ++bb) {
      if (bb >= nb) break;
      double2 acc{0.0, 0.0};
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        const double w = sh_w[g][bb];
        acc.x = fma(w, k[g].x, acc.x);
        acc.y = fma(w, k[g].y, acc.y);
      }
      store_result(reinterpret_cast<double2*>(P + (size_t)(b0 + bb) * nn + e), acc);
    }
  }
}

// G_b[R][c] = sum_ax arrow_b[R][ax] V[row(R, ax)][c].  A workgroup takes a flat range of SH_GRANGE elements of
// G for SH_GIB instances and walks it from front to back: every instance's stream of stores is
// consecutive in memory (a first version kept a chunk's operands in registers and walked the instances --
// every store of a workgroup 4.7 MB from the one before: 2.2 TB/s, however few loads the loop held), and
// the rows of V come out of L2 once per SH_GIB instances, requested a step ahead.
constexpr int SH_GIB = 8;            // instances of a workgroup
constexpr int SH_GRANGE = 8192;      // elements of G per instance and workgroup: 16 steps of 512
constexpr int SH_GSTEP = BLOCK * 2;  // a thread's 16 bytes per step
constexpr int SH_GRMAX = SH_GRANGE / T_BLOCK + 2;   // rows a range can touch (no >= T_BLOCK)

__global__ __launch_bounds__(BLOCK) void shared_g_kernel(PlanDev p, const double* __restrict__ V,
                                                        const double* __restrict__ params,
                                                        double* __restrict__ G, int batch) {
  __shared__ double sh_ar[SH_GIB][SH_GRMAX][2];   // the arrows of the range's rows (a free axis: 0)
  const long total = (long)p.nc * p.no;
  const long r0 = (long)blockIdx.x * SH_GRANGE;
  const long r1 = r0 + SH_GRANGE < total ? r0 + SH_GRANGE : total;
  const long b0 = (long)blockIdx.y * SH_GIB;
  const int nb = (int)(b0 + SH_GIB <= batch ? SH_GIB : batch - b0);
  const int32_t* grow = p.itab + p.off_t_grow;
  const int32_t* rrw = p.itab + p.off_rs_rr;
  const int R0 = (int)(r0 / p.no), nrl = (int)((r1 - 1) / p.no) - R0 + 1;
  for (int i = threadIdx.x; i < SH_GIB * nrl * 2; i += BLOCK) {
    const int bb = i / (nrl * 2), rem = i - bb * (nrl * 2), rl = rem >> 1, ax = rem & 1;
    const int32_t* rec = rrw + (size_t)(R0 + rl) * RS_RR_WORDS;
    sh_ar[bb][rl][ax] = (bb < nb && ax < rec[RR_NAXES])
                            ? params[(size_t)(b0 + bb) * p.nparams + rec[RR_ARROW + ax]] : 0.0;
  }
  __syncthreads();
  // (rows of one axis only -- wave-uniform, told by the plan's records -- skip the second operand)
  bool two = false;
  for (int rl = 0; rl < nrl; ++rl) two = two || rrw[(size_t)(R0 + rl) * RS_RR_WORDS + RR_NAXES] > 1;
  auto fetch = [&](long e, double2& a, double2& c, int& rl) {
    // the two elements e, e + 1 lie in one row (an even width); a thread behind the range reads the
    // range's first pair and stores nothing
    const long ee = e < r1 ? e : r0;
    const int R = (int)(ee / p.no), col = (int)(ee - (long)R * p.no);
    rl = R - R0;
    const int naxes = rrw[(size_t)R * RS_RR_WORDS + RR_NAXES];
    a = naxes > 0 ? *reinterpret_cast<const double2*>(V + (size_t)grow[R * RS_AXMAX + 0] * p.t_nop + col)
                  : double2{0.0, 0.0};
    c = (two && naxes > 1) ? *reinterpret_cast<const double2*>(V + (size_t)grow[R * RS_AXMAX + 1] * p.t_nop + col)
                           : double2{0.0, 0.0};
  };
  long e = r0 + threadIdx.x * 2;
  double2 va, vc;
  int rl;
  fetch(e, va, vc, rl);
  for (; e - threadIdx.x * 2 < r1; e += SH_GSTEP) {
    double2 na, nc2;
    int nrl2;
    fetch(e + SH_GSTEP, na, nc2, nrl2);       // the next step's operands: in flight behind this step's stores
    if (e < r1) {
#pragma unroll
      for (int bb = 0; bb < SH_GIB; ++bb) {
        if (bb >= nb) break;
        const double a0 = sh_ar[bb][rl][0], a1 = sh_ar[bb][rl][1];
        store_result(reinterpret_cast<double2*>(G + (size_t)(b0 + bb) * total + e),
                     double2{fma(a1, vc.x, a0 * va.x), fma(a1, vc.y, a0 * va.y)});
      }
    }
    va = na, vc = nc2, rl = nrl2;
  }
}

// q_b[c] = sum over the stages' rows of V[row][c] w s (d_b[drow] - aim) + the diagonal terms;
// h_b[R] = (extreme + arrow . center) - arrow . d_b: SH_QB instances per workgroup, their residuals
// rho_b[k] = w s (d - aim) in LDS, a thread per column
__global__ __launch_bounds__(BLOCK) void shared_qh_kernel(PlanDev p, const double* __restrict__ V,
                                                         const double* __restrict__ params,
                                                         const double* __restrict__ work,
                                                         long long work_stride, double* __restrict__ q,
                                                         double* __restrict__ h, int batch, int nrho) {
  extern __shared__ __attribute__((aligned(16))) double sh_rho[];   // [nrho][SH_QB]
  int* sh_row = reinterpret_cast<int*>(sh_rho + (size_t)nrho * SH_QB);   // [nrho] row of V
  const int tid = threadIdx.x;
  const long b0 = (long)blockIdx.x * SH_QB;
  const int32_t* stages = p.itab + p.off_t_stage;
  for (int k = tid; k < nrho; k += BLOCK) {
    const int st = k / TK, i = k - st * TK;
    const int32_t* rec = stages + st * T_STAGE_WORDS;
    const int info = rec[TS_INFO], nrows = info & 255, fl = (info >> 8) & 255;
    const bool live = i < nrows;
    sh_row[k] = live ? rec[TS_AROW] + i : 0;
    const double scale = (fl & TS_FLAG_HALF) ? 0.5 : 1.0;
#pragma unroll
    for (int x = 0; x < SH_QB; ++x) {
      const long b = b0 + x;
      double r = 0.0;
      if (live && b < batch) {
        const double* pb = params + (size_t)b * p.nparams;
        const double d = work[b * work_stride + rec[TS_DROW] + i];
        r = pb[rec[TS_WPARAM]] * (scale * (d - pb[rec[TS_AIMPARAM]]));
      }
      sh_rho[(size_t)k * SH_QB + x] = r;
    }
  }
  __syncthreads();
  if (q != nullptr)
    for (int c = tid; c < p.no; c += BLOCK) {
      double acc[SH_QB];
#pragma unroll
      for (int x = 0; x < SH_QB; ++x) acc[x] = 0.0;
      for (int k = 0; k < nrho; ++k) {
        const double v = V[(size_t)sh_row[k] * p.t_nop + c];
#pragma unroll
        for (int x = 0; x < SH_QB; ++x) acc[x] = fma(v, sh_rho[(size_t)k * SH_QB + x], acc[x]);
      }
#pragma unroll
      for (int x = 0; x < SH_QB; ++x) {
        const long b = b0 + x;
        if (b >= batch) break;
        double dP, dq;
        diagonal_of_column(p, params + (size_t)b * p.nparams, c, dP, dq);
        q[(size_t)b * p.no + c] = acc[x] + dq;
      }
    }
  if (h != nullptr) {
    const int32_t* grow = p.itab + p.off_t_grow;
    const int32_t* rrw = p.itab + p.off_rs_rr;
    for (int R = tid; R < p.nc; R += BLOCK) {
      const int32_t* x = rrw + (size_t)R * RS_RR_WORDS;
      for (int y = 0; y < SH_QB; ++y) {
        const long b = b0 + y;
        if (b >= batch) break;
        const double* pb = params + (size_t)b * p.nparams;
        const double* dvec = work + b * work_stride;
        double ac = 0.0, ad = 0.0;
        for (int ax = 0; ax < x[RR_NAXES]; ++ax) {
          const double ar = pb[x[RR_ARROW + ax]];
          ac += ar * pb[x[RR_CENTER + ax]];
          ad = fma(ar, dvec[grow[R * RS_AXMAX + ax]], ad);
        }
        h[(size_t)b * p.nc + R] = (pb[x[RR_EXTREME]] + ac) - ad;
      }
    }
  }
}

}  // namespace

extern int g_phase_mask;  // fused.hip (MPCASM_OPT_PHASE_MASK)

bool tiled_eligible(const PlanDev& p) { return p.t_ok != 0 && p.no >= T_BLOCK; }

// the shared-model form may run on plans without generated groups and with rows of G in 16-byte pieces
static bool shared_form_plan(const PlanDev& p) { return p.t_nlti == 0 && (p.no & 1) == 0 && p.t_nop % 2 == 0; }

size_t tiled_workspace_bytes(const PlanDev& p, int batch) {
  // per instance: d and the generated tables; behind them, once: the scratch of the shared-model form
  return ((size_t)batch * p.t_work + (shared_form_plan(p) ? shared_scratch(p).total : 0)) * sizeof(double);
}

// The horizon tables of the plan's generated groups, from the (A, B) in the slots of each group's
// first two sources, into the scratch (t_work doubles per instance); `eff`: the launch's sources
// with those tables in the places of the groups' U_j and S -- what the column tables address.
int launch_lti_tables(const PlanDev& p, const SrcTable& src, double* w, int batch,
                      const int32_t* h_itab, SrcTable* eff, hipStream_t stream) {
  *eff = src;
  if (p.t_nlti == 0) return MPCASM_OK;
  if (h_itab == nullptr || w == nullptr) return MPCASM_ERR_ARG;
  const long long stride = p.t_work;
  for (int g = 0; g < p.t_nlti; ++g) {
    const int32_t* rec = h_itab + p.off_t_lti + g * T_LTI_WORDS;
    const int n = rec[TL_N], m = rec[TL_M];
    if (n * (m + n) > LTI_XMAX || n > 64) return MPCASM_ERR_LIMIT;
    const int32_t* ids = h_itab + p.off_t_lti_ids + rec[TL_IDS];
    for (int j = 0; j < m; ++j) {
      eff->ptr[ids[j]] = w + rec[TL_TB];
      eff->stride[ids[j]] = stride;
    }
    eff->ptr[ids[m]] = w + rec[TL_TA];
    eff->stride[ids[m]] = stride;
  }
  size_t lds = 0, small = 0;
  bool all_small = true;
  for (int g = 0; g < p.t_nlti; ++g) {
    const int32_t* rec = h_itab + p.off_t_lti + g * T_LTI_WORDS;
    const size_t n = rec[TL_N], m = rec[TL_M], N = rec[TL_HORIZON];
    lds = std::max(lds, (n * n + 2 * n * (m + n) + n * m * LTI_HIST) * sizeof(double));
    const size_t tables = N * n * n + n * m * 2 * N;
    all_small = all_small && n * (m + n) <= 64 && tables <= 1800;  // (four wavefronts: under 64 KB of LDS)
    small = std::max(small, 192 + tables + (tables & 1));
  }
  if (all_small) {  // a wavefront per `systems` systems
    size_t elems = 0;
    for (int g = 0; g < p.t_nlti; ++g) {
      const int32_t* rec = h_itab + p.off_t_lti + g * T_LTI_WORDS;
      elems = std::max<size_t>(elems, (size_t)rec[TL_N] * (rec[TL_M] + rec[TL_N]));
    }
    int lanes = 1;
    while ((size_t)lanes < elems) lanes *= 2;
    // (as many systems per wavefront as lanes and 64 KB of LDS per workgroup allow)
    const int systems = (int)std::max<size_t>(1, std::min<size_t>(64 / lanes, (64 * 1024 / sizeof(double) / (BLOCK / 64)) / small));
    const long jobs = (long)batch * p.t_nlti;
    const long per_wg = (long)(BLOCK / 64) * systems;
    const unsigned grid = (unsigned)std::min<long>((jobs + per_wg - 1) / per_wg, 256L * 8);
    hipLaunchKernelGGL(lti_tables_small_kernel, dim3(grid), dim3(BLOCK),
                       small * systems * (BLOCK / 64) * sizeof(double), stream, p, src, w, stride, jobs, (int)small,
                       lanes, systems);
    return MPCASM_OK;
  }
  hipLaunchKernelGGL(lti_tables_kernel, dim3((unsigned)batch * p.t_nlti), dim3(BLOCK), lds, stream, p,
                     src, w, stride);
  return MPCASM_OK;
}

namespace {

// The shared-model form; MPCASM_ERR_LIMIT when the launch is not one (a source of an instance's own, more
// weights than SH_TMAX, a row of G with more than two axes, an odd width): the general kernel then.
int launch_shared_form(const PlanDev& p, const SrcTable& eff, const double* params, double* w,
                       long long stride, double* P, double* q, double* G, double* h, int batch, int nb,
                       int npairs, int sym, const int32_t* h_itab, hipStream_t stream, hipError_t* err) {
  if (!shared_form_plan(p) || w == nullptr) return MPCASM_ERR_LIMIT;
  for (int i = 0; i < p.nsrc; ++i)
    if (eff.stride[i] != 0) return MPCASM_ERR_LIMIT;
  // the weight parameters: of every stage and of every diagonal term
  SharedSlots sl;
  sl.n = 0;
  for (int g = 0; g < SH_TMAX; ++g) sl.slot[g] = 0;
  auto add = [&](int slot) {
    for (int g = 0; g < sl.n; ++g)
      if (sl.slot[g] == slot) return true;
    if (sl.n == SH_TMAX) return false;
    sl.slot[sl.n++] = slot;
    return true;
  };
  for (int st = 0; st < p.t_nstage; ++st)
    if (!add(h_itab[p.off_t_stage + st * T_STAGE_WORDS + TS_WPARAM])) return MPCASM_ERR_LIMIT;
  for (int g = 0; g < p.ngterm; ++g) {
    const int32_t* rec = h_itab + p.off_gterm + g * GT_WORDS;
    if ((rec[GT_FLAGS] & GT_FLAG_DIAG) && !add(rec[GT_WPARAM])) return MPCASM_ERR_LIMIT;
  }
  if (batch < 2 * sl.n) return MPCASM_ERR_LIMIT;   // (the K_g cost a launch of sl.n instances themselves)
  if (G != nullptr)
    for (int R = 0; R < p.nc; ++R)
      if (h_itab[p.off_rs_rr + R * RS_RR_WORDS + RR_NAXES] > 2) return MPCASM_ERR_LIMIT;
  const int nrho = p.t_nstage * TK;
  const size_t qh_lds = (size_t)nrho * SH_QB * sizeof(double) + (size_t)nrho * sizeof(int);
  if (qh_lds > 64 * 1024) return MPCASM_ERR_LIMIT;
  const SharedScratch S = shared_scratch(p);
  double* base = w + (size_t)batch * stride;
  double* V = base + S.v;
  double* K = base + S.k;
  double* pseudo = base + S.pseudo;
  double* zero = base + S.zero;
  double* qdummy = base + S.qdummy;
  const long pairs = (long)p.rtot * (p.t_nop / 2);
  hipLaunchKernelGGL(shared_rows_kernel, dim3((unsigned)ceil_div(pairs, (long)BLOCK)), dim3(BLOCK), 0, stream, p,
                     eff, V);
  const long nn = (long)p.no * p.no;
  if (P != nullptr && sl.n > 0) {
    const int fill = std::max(sl.n * p.nparams, p.rtot);
    hipLaunchKernelGGL(shared_pseudo_kernel, dim3((unsigned)ceil_div(fill, BLOCK)), dim3(BLOCK), 0, stream, sl,
                       p.nparams, p.rtot, pseudo, zero);
    // K_g: the Hessian of pseudo-instance g, by the general kernel (d = 0, no constraints)
    const unsigned groups = ceil_div((unsigned)sl.n, 8u);
    hipLaunchKernelGGL(tiled_assemble_kernel, dim3(groups * 8 * (unsigned)npairs), dim3(BLOCK), 0, stream, p,
                       eff, pseudo, zero, 0ll, K, qdummy, static_cast<double*>(nullptr),
                       static_cast<double*>(nullptr), nb, npairs, sym, sl.n);
  }
  if (P != nullptr) {
    const dim3 pgrid((unsigned)ceil_div(nn, (long)SH_PRANGE), (unsigned)ceil_div(batch, SH_PIB));
    // (the kernel multiplies by every one of its TG matrices, the ones behind the last being zeros: the
    // instantiation just above the number of weights -- at 32 for C4's 18 the kernel was bound by its FMAs)
#define MPCASM_SHARED_P(TG)                                                                                \
  hipLaunchKernelGGL((shared_p_kernel<TG>), pgrid, dim3(BLOCK), 0, stream, sl, nn, K, params, p.nparams, P, batch)
    if (sl.n > 24) MPCASM_SHARED_P(32);
    else if (sl.n > 20) MPCASM_SHARED_P(24);
    else if (sl.n > 16) MPCASM_SHARED_P(20);
    else if (sl.n > 12) MPCASM_SHARED_P(16);
    else if (sl.n > 8) MPCASM_SHARED_P(12);
    else if (sl.n > 4) MPCASM_SHARED_P(8);
    else if (sl.n > 0) MPCASM_SHARED_P(4);
#undef MPCASM_SHARED_P
    else
      (void)hipMemsetAsync(P, 0, sizeof(double) * (size_t)batch * nn, stream);
  }
  if (G != nullptr && p.nc > 0)
    hipLaunchKernelGGL(shared_g_kernel,
                       dim3((unsigned)ceil_div((long)p.nc * p.no, (long)SH_GRANGE), (unsigned)ceil_div(batch, SH_GIB)),
                       dim3(BLOCK), 0, stream, p, V, params, G, batch);
  if (P != nullptr || (G != nullptr && p.nc > 0))
    hipLaunchKernelGGL(shared_qh_kernel, dim3((unsigned)ceil_div(batch, SH_QB)), dim3(BLOCK), qh_lds, stream, p,
                       V, params, w, stride, P != nullptr ? q : nullptr,
                       (G != nullptr && p.nc > 0) ? h : nullptr, batch, nrho);
  *err = hipGetLastError();
  t_last_kernel = MPCASM_KERNEL_TILED_SHARED;
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

template <int KP, int CB>
int launch_scan_as(const PlanDev& p, const SrcTable& eff, const double* A, long long strideA,
                   const double* Bm, long long strideB, const double* params, const double* w,
                   long long stride, double* P, double* q, double* G, double* h, int batch, int dlen,
                   int whole_lines, int rows_in_lds, size_t lds, const double* given, int fused,
                   hipStream_t stream, hipError_t* err) {
  auto kernel = toeplitz_scan_kernel<KP, CB>;
  if (lds > 64 * 1024) {
    *err = allow_whole_lds(reinterpret_cast<const void*>(kernel));
    if (*err != hipSuccess) return MPCASM_ERR_HIP;
  }
  // rows of G per ticket (tuning aid MPCASM_SCAN_GROUP; tools/ab_scan.sh): fewer = the wavefronts of a
  // workgroup write closer together, more = fewer tickets
  static const int group = [] {
    const char* e = getenv("MPCASM_SCAN_GROUP");
    const int v = e ? atoi(e) : SCAN_GROUP;   // (r04: 8 rows per ticket 8.6 ms per 8192 C4 instances, 4: 9.3, 1: 9.7)
    return v < 1 ? 1 : (v > SCAN_GROUP ? SCAN_GROUP : v);
  }();
  hipLaunchKernelGGL(kernel, dim3((unsigned)batch), dim3(BLOCK), lds, stream, p, eff, A, strideA, Bm, strideB,
                     params, w, stride, P, q, G, h, batch, dlen, whole_lines, group, rows_in_lds, g_phase_mask,
                     given, fused);
  *err = hipGetLastError();
  if (*err == hipSuccess) t_last_kernel = MPCASM_KERNEL_TILED_SCAN;
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

// the scan form for this plan, or MPCASM_ERR_LIMIT: no instantiation holds its K terms and column
// blocks in registers, or an instance does not fit in LDS (the Toeplitz form takes it then).
// `src`: the launch's own sources (the group's (A, B) in the slots of its first two), `eff`: with the
// generated tables in the places of the group's U_j and S.
// `fused`: no pre-pass has run -- the kernel makes its table and d itself (H_T_SCAN_FUSED); `eff` and `w` unused.
int launch_scan(const PlanDev& p, const SrcTable& src, const SrcTable& eff, const double* params,
                const double* w, long long stride, double* P, double* q, double* G, double* h, int batch,
                const int32_t* h_itab, hipStream_t stream, hipError_t* err, const double* given = nullptr,
                int fused = 0) {
  const int K = p.t_scan, nblk = p.t_scan_nblk;
  const int32_t* rec = h_itab + p.off_t_lti;
  const int n = rec[TL_N], m = rec[TL_M], N = rec[TL_HORIZON];
  if (p.no > 128 * SCAN_GCH_MAX || n > 64 || (N & 1)) return MPCASM_ERR_LIMIT;
  if (fused && given == nullptr) return MPCASM_ERR_LIMIT;
  const int dlen = p.rtot + (p.rtot & 1);
  // the records of G's rows in LDS while that leaves room for two workgroups per CU
  const int rows_in_lds = scan_lds(dlen, p.nparams, n, m, N, p.nc, true).total + NSTREAM * sizeof(double*) <= SCAN_HALF_CU;
  const size_t lds = scan_lds(dlen, p.nparams, n, m, N, p.nc, rows_in_lds != 0).total;
  if (lds + NSTREAM * sizeof(double*) > (size_t)RESIDENT_LDS_LIMIT) return MPCASM_ERR_LIMIT;
  const int32_t* ids = h_itab + p.off_t_lti_ids + rec[TL_IDS];
  // results leave as whole 128-byte lines when every run of a wavefront's store starts and ends on one
  int whole = p.no % 16 == 0 && N % 16 == 0;
  const int32_t* blk = h_itab + p.off_t_scan_blk;
  for (int b = 0; b < nblk; ++b) whole = whole && blk[2 * b] % 16 == 0;
#define MPCASM_SCAN_CASE(KP, CB)                                                                          \
  if (K <= KP && nblk <= CB)                                                                             \
    return launch_scan_as<KP, CB>(p, eff, src.ptr[ids[0]], src.stride[ids[0]], src.ptr[ids[1]],           \
                                  src.stride[ids[1]], params, w, stride, P, q, G, h, batch, dlen, whole, \
                                  rows_in_lds, lds, given, fused, stream, err);
  MPCASM_SCAN_CASE(4, 4)
  MPCASM_SCAN_CASE(8, 4)
  MPCASM_SCAN_CASE(4, 8)
  MPCASM_SCAN_CASE(8, 8)
  MPCASM_SCAN_CASE(12, 4)
  MPCASM_SCAN_CASE(12, 6)
  MPCASM_SCAN_CASE(16, 4)
#undef MPCASM_SCAN_CASE
  return MPCASM_ERR_LIMIT;
}

}  // namespace

int launch_assemble_tiled(const PlanDev& p, const SrcTable& src, const double* params,
                          const double* given, double* P, double* q, double* G, double* h,
                          void* work, int batch, hipStream_t stream, hipError_t* err,
                          const int32_t* h_itab) {
  double* w = static_cast<double*>(work);
  const long long stride = p.t_work;
  // the scan form with its set-up fused (H_T_SCAN_FUSED): one kernel, no scratch (MPCASM_OPT_PATH 1 keeps the
  // pre-passes: the A/B of the two)
  if (p.t_scan_fused && p.t_scan > 0 && p.t_toeplitz && p.t_nlti == 1 && h_itab != nullptr && t_path == 0 &&
      (p.no & 1) == 0) {
    const int rc = launch_scan(p, src, src, params, nullptr, 0, P, q, G, h, batch, h_itab, stream, err, given, 1);
    if (rc != MPCASM_ERR_LIMIT) return rc;
  }
  SrcTable eff;
  {
    const int rc = launch_lti_tables(p, src, w, batch, h_itab, &eff, stream);
    if (rc != MPCASM_OK) return rc;
  }
  if (p.rtot > 0) {
    const unsigned nrb = ceil_div(p.rtot, BLOCK);
    if ((size_t)p.ng * sizeof(double) > 48 * 1024) return MPCASM_ERR_LIMIT;
    hipLaunchKernelGGL(compose_d_kernel, dim3(nrb * batch), dim3(BLOCK), (size_t)p.ng * sizeof(double),
                       stream, p, eff, given, w, stride, (int)nrb);
  }
  const int nb = (int)ceil_div(p.no, T_BLOCK);
  const int sym = p.rs_sym_any;
  const int npairs = sym ? nb * (nb + 1) / 2 : nb * nb;
  const unsigned groups = ceil_div((unsigned)batch, 8u);
  // (the Toeplitz forms write rows of G in 16-byte pieces: an even width)
  if (p.t_toeplitz && p.t_nlti == 1 && h_itab != nullptr && t_path != 3 && (p.no & 1) == 0) {
    // every stage is a window of the one generated group's table: operands straight out of LDS
    const int32_t* rec = h_itab + p.off_t_lti;
    const int tbn = rec[TL_N] * rec[TL_M] * 2 * rec[TL_HORIZON];
    if (p.t_scan > 0 && t_path != 4 && (rec[TL_TB] & 1) == 0 && (stride & 1) == 0) {
      // scan form: every Hessian term is the full horizon of one state -- P is a sum along diagonals
      const int rc = launch_scan(p, src, eff, params, w, stride, P, q, G, h, batch, h_itab, stream, err);
      if (rc != MPCASM_ERR_LIMIT) return rc;
    }
    const int nzero = (std::max(rec[TL_HORIZON], 16) + 16 + 1) & ~1;  // (even: what follows stays 16-byte aligned)
    // (behind the table and the zeros: d, and at the end of a diagonal block the gradient's partial
    // sums of four wavefronts + the diagonal gterms of its T_BLOCK columns)
    const size_t lds = ((size_t)tbn + nzero + std::max(p.rtot + (p.rtot & 1), (WAVES + 1) * T_BLOCK)) * sizeof(double);
    if (lds <= (size_t)RESIDENT_LDS_LIMIT && (rec[TL_TB] & 1) == 0 && (stride & 1) == 0) {
      *err = allow_whole_lds(reinterpret_cast<const void*>(toeplitz_assemble_kernel));
      if (*err != hipSuccess) return MPCASM_ERR_HIP;
      hipLaunchKernelGGL(toeplitz_assemble_kernel, dim3((unsigned)batch), dim3(BLOCK), lds,
                         stream, p, eff, params, w, stride, P, q, G, h, nb, npairs, sym, batch, tbn, nzero,
                         g_phase_mask);
      *err = hipGetLastError();
      return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
    }
  }
  // one model for the whole batch (every source shared): P and G are weighted sums of matrices that are
  // the same for every instance -- the shared-model form (MPCASM_OPT_PATH 3 keeps the general kernel)
  if (t_path != 3 && h_itab != nullptr && batch >= 8) {
    const int rc = launch_shared_form(p, eff, params, w, stride, P, q, G, h, batch, nb, npairs, sym, h_itab,
                                      stream, err);
    if (rc != MPCASM_ERR_LIMIT) return rc;
  }
  t_last_kernel = MPCASM_KERNEL_TILED;
  hipLaunchKernelGGL(tiled_assemble_kernel, dim3(groups * 8 * (unsigned)npairs), dim3(BLOCK), 0,
                     stream, p, eff, params, w, stride, P, q, G, h, nb, npairs, sym, batch);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

}  // namespace mpcasm
