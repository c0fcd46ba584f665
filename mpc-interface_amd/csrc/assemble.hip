// assemble.hip -- K2 (preview composition), K3 (Hessian / gradient, fp64 MFMA)
// and K4 (constraint stack) for gfx950: the staged pipeline that works for any
// problem size (workspace in HBM).  Small problems take the fused single-launch
// path of fused.hip instead.
//
// Reference semantics (python/mpc_interface/body.py):
//   K2  make_preview_matrices / get_matrices_from_dynamics / _from_definition  :149-193
//   K3  generate_qp_cost :266-302, generate_all_qp_costs :322-329
//   K4  generate_qp_constraint :236-264, generate_all_qp_constraints :304-320
//       with Constraint.matrices()/bound() (restrictions.py:175-199) folded in.
//
// Workspace per instance: V[rtot][ldv] doubles.  Row r of V holds, for one row
// of a row-set, the optim part Mo (columns 0..no-1) and in column `no` the dot
// product d = Mg . given -- the given part itself is never stored.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace mpcasm {

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

// ---------------------------------------------------------------------------
// K2: one wavefront per row, lanes over columns
// ---------------------------------------------------------------------------
constexpr int K2_ROWS_PER_WAVE = 4;

__global__ __launch_bounds__(BLOCK) void compose_rowsets_kernel(PlanDev p, SrcTable src,
                                                                const double* __restrict__ given,
                                                                double* __restrict__ V, int nrb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long inst = blockIdx.x / nrb;
  const int rb = blockIdx.x - inst * nrb;
  const int W = p.ng + p.no;
  const int32_t* rowptr = p.itab + p.off_rowptr;
  const int32_t* entbase = p.itab + p.off_entbase;
  const int32_t* entk = p.itab + p.off_entk;
  const double* coef = p.dtab + p.doff_entcoef;
  const double* g = given + inst * p.ng;
  double* Vb = V + (size_t)inst * p.rtot * p.ldv;

  const int r0 = (rb * WAVES + wave) * K2_ROWS_PER_WAVE;
  for (int r = r0; r < min(r0 + K2_ROWS_PER_WAVE, p.rtot); ++r) {
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    double dpart = 0.0;
    for (int c = lane; c < W; c += 64) {
      const double v = compose_element(p, src, inst, W, c, e0, e1, entbase, entk, coef);
      if (c < p.ng)
        dpart = fma(v, g[c], dpart);
      else
        Vb[(size_t)r * p.ldv + (c - p.ng)] = v;
    }
    dpart = wave_sum(dpart);
    if (lane == 0) Vb[(size_t)r * p.ldv + p.no] = dpart;
  }
}

// all definitions, full [Mg | Mo] rows (Formulation.PM)
__global__ __launch_bounds__(BLOCK) void compose_preview_kernel(PlanDev p, SrcTable src,
                                                                double* __restrict__ PM, int nrb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long inst = blockIdx.x / nrb;
  const int rb = blockIdx.x - inst * nrb;
  const int W = p.ng + p.no;
  const int32_t* rowptr = p.itab + p.off_pm_rowptr;
  const int32_t* entbase = p.itab + p.off_pm_entbase;
  const int32_t* entk = p.itab + p.off_pm_entk;
  const double* coef = p.dtab + p.doff_pm_entcoef;
  double* out = PM + (size_t)inst * p.pmrows * W;

  const int r0 = (rb * WAVES + wave) * K2_ROWS_PER_WAVE;
  for (int r = r0; r < min(r0 + K2_ROWS_PER_WAVE, p.pmrows); ++r) {
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    for (int c = lane; c < W; c += 64)
      out[(size_t)r * W + c] = compose_element(p, src, inst, W, c, e0, e1, entbase, entk, coef);
  }
}

// The same matrices from the element program (plan_tables.h H_PM_*): a thread per element
// of [Mg | Mo]; most are structural zeros -- one 4-byte read of the map, one coalesced store
// -- and a non-zero one is a short list of coef * source value.  One workgroup takes
// PM_CHUNK consecutive elements of one instance; the source table goes through LDS once
// (a by-value table indexed per lane would travel through scratch memory).
constexpr int PM_CHUNK = 4 * BLOCK;
__global__ __launch_bounds__(BLOCK) void preview_elements_kernel(PlanDev p, SrcTable src,
                                                                 double* __restrict__ PM,
                                                                 int chunks) {
  __shared__ const double* s_ptr[MAX_SOURCES];
  __shared__ long long s_stride[MAX_SOURCES];
  const int tid = threadIdx.x;
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < MAX_SOURCES; ++k) {
      s_ptr[k] = src.ptr[k];
      s_stride[k] = src.stride[k];
    }
  }
  __syncthreads();
  const long inst = blockIdx.x / chunks;
  const int chunk = blockIdx.x - inst * chunks;
  const int elems = p.pmrows * (p.ng + p.no);
  const int32_t* map = p.itab + p.off_pm_map;
  const int32_t* fdp = p.itab + p.off_pm_fdptr;
  const uint2* ops = reinterpret_cast<const uint2*>(p.itab + p.off_pm_op);
  const double* pool = p.dtab + p.doff_pm_pool;
  double* out = PM + (size_t)inst * elems;
  auto element = [&](int m) {
    double acc = 0.0;
    if (m >= 0) {
      for (int o = fdp[m]; o < fdp[m + 1]; ++o) {
        const uint2 op = ops[o];
        const unsigned sid = op.y & 255u;
        const double val = sid == 255u ? 1.0 : s_ptr[sid][inst * s_stride[sid] + op.x];
        acc = fma(pool[op.y >> 8], val, acc);
      }
    }
    return acc;
  };
  // (pairs of elements per thread with 16-byte stores were no faster)
#pragma unroll
  for (int k = 0; k < PM_CHUNK / BLOCK; ++k) {
    const int e = chunk * PM_CHUNK + k * BLOCK + tid;
    if (e >= elems) break;
    out[e] = element(map[e]);
  }
}

// ---------------------------------------------------------------------------
// K3: [P | q] = sum_gterms  (w A)^T [B | r]   with v_mfma_f64_16x16x4_f64.
// One wavefront owns a 32x32 block of [P | q] (2x2 MFMA tiles) and walks every
// gterm, four workspace rows per MFMA.  Operand lane map (cdna guide section 3):
// A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15];
// D: col = lane&15, row = (lane>>4) + 4*reg.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void hessian_kernel(PlanDev p,
                                                        const double* __restrict__ params,
                                                        const double* __restrict__ V,
                                                        double* __restrict__ P,
                                                        double* __restrict__ q, int nrb,
                                                        int q_only) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long inst = blockIdx.x / nrb;
  const int rb = blockIdx.x - inst * nrb;
  const int no = p.no;
  const int nbr = (no + 31) / 32;      // block rows
  const int nbc = (no + 1 + 31) / 32;  // block columns (q is column `no`)
  const int blk = rb * WAVES + wave;
  if (blk >= (q_only ? nbr : nbr * nbc)) return;
  // q_only: P comes from hessian_gemm_kernel; only the block column holding q is computed
  const int bi = q_only ? blk : blk / nbc, bj = q_only ? no / 32 : blk - (blk / nbc) * nbc;
  const bool has_qcol = (bj == no / 32);
  const double* Vb = V + (size_t)inst * p.rtot * p.ldv;
  const double* pb = params + (size_t)inst * p.nparams;
  const int32_t* gt = p.itab + p.off_gterm;
  const int li = lane & 15, lk = lane >> 4;
  const int ldv = p.ldv;

  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

  const int arow0 = bi * 32 + li, arow1 = arow0 + 16;  // rows of P == columns of A's source
  const int bcol0 = bj * 32 + li, bcol1 = bcol0 + 16;

  for (int g = 0; g < p.ngterm; ++g) {
    const int32_t* rec = gt + g * GT_WORDS;
    const int flags = rec[GT_FLAGS];
    if (flags & GT_FLAG_DIAG) continue;  // added analytically at the store
    // structurally-zero 16-column tiles of the operands (plan tile masks): a term only
    // reaches P inside this 32x32 block when A has a tile in its rows and B in its columns
    const unsigned ma = (unsigned)rec[GT_MASKA], mb = (unsigned)rec[GT_MASKB];
    const bool a_here = (ma >> min(2 * bi, 30)) & 3u || 2 * bi >= 30;
    const bool b_here = (mb >> min(2 * bj, 30)) & 3u || 2 * bj >= 30;
    const bool hasP = (flags & GT_FLAG_P) && a_here && b_here && !q_only;
    if (!a_here || (!hasP && !has_qcol)) continue;
    const int aoff = rec[GT_AOFF], boff = rec[GT_BOFF], doff = rec[GT_DOFF];
    const int nrows = rec[GT_NROWS];
    const double w = pb[rec[GT_WPARAM]];
    const double aim = pb[rec[GT_AIMPARAM]];
    const double scale = (flags & GT_FLAG_HALF) ? 0.5 : 1.0;
    for (int k0 = 0; k0 < nrows; k0 += 4) {
      const int k = k0 + lk;
      const bool valid = k < nrows;
      const double* arow = Vb + (size_t)(aoff + k) * ldv;
      const double* brow = Vb + (size_t)(boff + k) * ldv;
      double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
      if (valid) {
        if (arow0 < no) a0 = w * arow[arow0];
        if (arow1 < no) a1 = w * arow[arow1];
        if (hasP) {
          if (bcol0 < no) b0 = brow[bcol0];
          if (bcol1 < no) b1 = brow[bcol1];
        }
        if (has_qcol) {
          const double r = scale * (Vb[(size_t)(doff + k) * ldv + no] - aim);
          if (bcol0 == no) b0 = r;
          if (bcol1 == no) b1 = r;
        }
      }
      acc[0][0] = mfma_f64_16x16x4(a0, b0, acc[0][0]);
      acc[0][1] = mfma_f64_16x16x4(a0, b1, acc[0][1]);
      acc[1][0] = mfma_f64_16x16x4(a1, b0, acc[1][0]);
      acc[1][1] = mfma_f64_16x16x4(a1, b1, acc[1][1]);
    }
  }

  double* Pb = P + (size_t)inst * no * no;
  double* qb = q + (size_t)inst * no;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = bi * 32 + ti * 16 + lk + 4 * reg;
        const int col = bj * 32 + tj * 16 + li;
        if (row < no) {
          double dP = 0.0, dq = 0.0;
          if (row == col || col == no) diagonal_terms(p, pb, row, dP, dq);
          if (col < no) {
            if (!q_only) Pb[(size_t)row * no + col] = acc[ti][tj][reg] + (row == col ? dP : 0.0);
          } else if (col == no)
            qb[row] = acc[ti][tj][reg] + dq;
        }
      }
}

// ---------------------------------------------------------------------------
// K3 for wide problems (no >= 128, BASELINE config C4): P = sum_terms (w A)^T B as an
// LDS-tiled batched GEMM on the fp64 matrix core.  A workgroup owns a 128 x 128 block
// of one instance's P (four wavefronts, 64 x 64 = 4 x 4 MFMA tiles each, accumulators
// in registers for the whole K loop); the rows of the workspace are staged 16 at a
// time through LDS with coalesced 16-byte loads (row stride padded to 144 doubles so
// that the fragment reads of the four k-rows fall on disjoint bank halves).  When
// every term is symmetric only blocks bi <= bj are computed and mirrored on store.
// The gradient is computed by gradient_kernel.
// ---------------------------------------------------------------------------
constexpr int GB = 128, GK = 16, GLD = GB + 16;

__global__ __launch_bounds__(BLOCK) void hessian_gemm_kernel(PlanDev p,
                                                             const double* __restrict__ params,
                                                             const double* __restrict__ V,
                                                             double* __restrict__ P, int nb,
                                                             int npairs, int sym) {
  __shared__ __attribute__((aligned(16))) double As[2][GK * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[2][GK * GLD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long inst = blockIdx.x / npairs;
  int pr = blockIdx.x - inst * npairs;
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
  const int no = p.no, ldv = p.ldv;
  const double* Vb = V + (size_t)inst * p.rtot * ldv;
  const double* pb = params + (size_t)inst * p.nparams;
  const int32_t* gt = p.itab + p.off_gterm;
  const int li = lane & 15, lk = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

  // staging role: thread -> (row within a group of 4, 16-byte column piece)
  const int srow = tid >> 6, scol = (tid & 63) * 2;
  const int acol = bi * GB + scol, bcol = bj * GB + scol;
  const unsigned amask = 0xFFu << min(8 * bi, 24), bmask = 0xFFu << min(8 * bj, 24);

  // the K loop runs over stages (term g, rows k0 .. k0 + GK); terms whose tile masks
  // show no structural non-zero in this block's rows / columns are skipped (blocks beyond
  // tile 30 are never skipped)
  int g = -1, k0 = 0, nrows = 0, aoff = 0, boff = 0;
  double w = 0.0;
  auto next_stage = [&]() -> bool {
    k0 += GK;
    while (k0 >= nrows) {
      ++g;
      if (g >= p.ngterm) return false;
      const int32_t* rec = gt + g * GT_WORDS;
      if (!(rec[GT_FLAGS] & GT_FLAG_P)) continue;  // (diagonal terms carry no P flag)
      if (8 * bi < 30 && !((unsigned)rec[GT_MASKA] & amask)) continue;
      if (8 * bj < 30 && !((unsigned)rec[GT_MASKB] & bmask)) continue;
      aoff = rec[GT_AOFF];
      boff = rec[GT_BOFF];
      nrows = rec[GT_NROWS];
      w = pb[rec[GT_WPARAM]];
      k0 = 0;
    }
    return true;
  };
  double2 ra[GK / 4], rb[GK / 4];
  auto load_stage = [&]() {  // HBM/L2 -> registers
#pragma unroll
    for (int rr = 0; rr < GK / 4; ++rr) {
      const int k = k0 + 4 * rr + srow;
      double2 a{0.0, 0.0}, b{0.0, 0.0};
      if (k < nrows) {
        const double* ar = Vb + (size_t)(aoff + k) * ldv;
        const double* br = Vb + (size_t)(boff + k) * ldv;
        if (acol + 1 < no) {
          a = *reinterpret_cast<const double2*>(ar + acol);
        } else if (acol < no) {
          a.x = ar[acol];
        }
        if (bcol + 1 < no) {
          b = *reinterpret_cast<const double2*>(br + bcol);
        } else if (bcol < no) {
          b.x = br[bcol];
        }
      }
      a.x *= w;
      a.y *= w;
      ra[rr] = a;
      rb[rr] = b;
    }
  };
  auto store_stage = [&](int buf) {  // registers -> LDS
#pragma unroll
    for (int rr = 0; rr < GK / 4; ++rr) {
      *reinterpret_cast<double2*>(As[buf] + (4 * rr + srow) * GLD + scol) = ra[rr];
      *reinterpret_cast<double2*>(Bs[buf] + (4 * rr + srow) * GLD + scol) = rb[rr];
    }
  };

  bool more = next_stage();
  if (more) {
    load_stage();
    store_stage(0);
  }
  int buf = 0;
  __syncthreads();
  while (more) {
    const bool have_next = next_stage();
    if (have_next) load_stage();  // in flight while this stage is multiplied
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      double a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = As[buf][(kk + lk) * GLD + wr * 64 + t * 16 + li];
        b[t] = Bs[buf][(kk + lk) * GLD + wc * 64 + t * 16 + li];
      }
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = mfma_f64_16x16x4(a[ta], b[tb], acc[ta][tb]);
    }
    if (have_next) store_stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    more = have_next;
  }

  double* Pb = P + (size_t)inst * no * no;
  const bool mirror = sym && bi != bj;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = bi * GB + wr * 64 + ta * 16 + lk + 4 * reg;
        const int col = bj * GB + wc * 64 + tb * 16 + li;
        if (row < no && col < no) {
          double v = acc[ta][tb][reg];
          if (row == col) {
            double dP, dq;
            diagonal_terms(p, pb, row, dP, dq);
            v += dP;
          }
          Pb[(size_t)row * no + col] = v;
          if (mirror) Pb[(size_t)col * no + row] = v;
        }
      }
}

// Gradient for wide problems: q[c] = sum_terms w s sum_k V[a + k][c] (V[d + k][no] - aim).
// A workgroup takes 64 columns of one instance; its four wavefronts split the rows of
// every term (coalesced 512-byte row pieces) and reduce through LDS.
__global__ __launch_bounds__(BLOCK) void gradient_kernel(PlanDev p,
                                                         const double* __restrict__ params,
                                                         const double* __restrict__ V,
                                                         double* __restrict__ q, int ncb) {
  __shared__ double part[WAVES][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long inst = blockIdx.x / ncb;
  const int cb = blockIdx.x - inst * ncb;
  const int no = p.no, ldv = p.ldv;
  const int c = cb * 64 + lane;
  const double* Vb = V + (size_t)inst * p.rtot * ldv;
  const double* pb = params + (size_t)inst * p.nparams;
  const int32_t* gt = p.itab + p.off_gterm;
  double acc = 0.0;
  for (int g = 0; g < p.ngterm; ++g) {
    const int32_t* rec = gt + g * GT_WORDS;
    if (rec[GT_FLAGS] & GT_FLAG_DIAG) continue;
    const double w = pb[rec[GT_WPARAM]];
    const double aim = pb[rec[GT_AIMPARAM]];
    const double scale = (rec[GT_FLAGS] & GT_FLAG_HALF) ? 0.5 : 1.0;
    const int aoff = rec[GT_AOFF], doff = rec[GT_DOFF], nrows = rec[GT_NROWS];
    for (int k = wave; k < nrows; k += WAVES) {
      const double r = scale * (Vb[(size_t)(doff + k) * ldv + no] - aim);
      if (c < no) acc = fma(w * Vb[(size_t)(aoff + k) * ldv + c], r, acc);
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < no) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += part[w][lane];
    double dP, dq;
    diagonal_terms(p, pb, c, dP, dq);
    q[(size_t)inst * no + c] = s + dq;
  }
}

// ---------------------------------------------------------------------------
// K4: rows of the stacked G and h.  A workgroup takes K4_ROWS consecutive rows of
// one instance: the first threads resolve the row records (limit, axes, arrows,
// workspace rows) into LDS and write h, then 16 threads per row stream the row as
// 16-byte pieces (sum over axes of arrow * workspace row), coalesced.
// ---------------------------------------------------------------------------
constexpr int K4_ROWS = 16;
constexpr int K4_AXMAX = 8;

__global__ __launch_bounds__(BLOCK) void constraints_kernel(PlanDev p,
                                                            const double* __restrict__ params,
                                                            const double* __restrict__ V,
                                                            double* __restrict__ G,
                                                            double* __restrict__ h, int nrb,
                                                            int batch) {
  __shared__ double s_arrow[K4_ROWS][K4_AXMAX];
  __shared__ int s_voff[K4_ROWS][K4_AXMAX];
  __shared__ int s_nax[K4_ROWS];
  const int tid = threadIdx.x;
  // Workgroups go to the 8 XCDs round robin and every XCD has its own L2.  The row blocks
  // of one instance read the same workspace rows several times over (every facet of a box
  // reads the rows of its variable), so they are dealt to ONE XCD, one after the other:
  // workgroups x, x + 8, x + 16 ... walk the row blocks of instance 8 s + x.
  const unsigned xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const long inst = (long)(slot / nrb) * 8 + xcd;
  const int rb = slot % nrb;
  if (inst >= batch) return;
  const int no = p.no, ldv = p.ldv;
  const double* Vb = V + (size_t)inst * p.rtot * ldv;
  const double* pb = params + (size_t)inst * p.nparams;
  const int R0 = rb * K4_ROWS;
  const int nrows = min(K4_ROWS, p.nc - R0);

  if (tid < nrows) {
    const int R = R0 + tid;
    const int32_t* lm = p.itab + p.off_limit + (p.itab + p.off_rowlimit)[R] * LM_WORDS;
    const int r = R - lm[LM_OUT0];
    const int naxes = lm[LM_NAXES];
    const int32_t* lx = p.itab + p.off_lax + lm[LM_LAX0] * LX_WORDS;
    const double* arrow = pb + lm[LM_ARROW_P] + (lm[LM_ARROW_ROWS] == 1 ? 0 : r) * naxes;
    const double* center = pb + lm[LM_CENTER_P] + (lm[LM_CENTER_ROWS] == 1 ? 0 : r) * naxes;
    const double extreme = pb[lm[LM_EXTREME_P] + (lm[LM_EXTREME_ROWS] == 1 ? 0 : r)];
    double ac = 0.0, ad = 0.0;
    s_nax[tid] = naxes;
    for (int ax = 0; ax < naxes; ++ax) {
      const int rr = lx[ax * LX_WORDS + LX_ROWS] == 1 ? 0 : r;
      const int voff = (lx[ax * LX_WORDS + LX_ROWOFF] + rr) * ldv;
      if (ax < K4_AXMAX) {
        s_arrow[tid][ax] = arrow[ax];
        s_voff[tid][ax] = voff;
      }
      ac += arrow[ax] * center[ax];
      ad = fma(arrow[ax], Vb[voff + no], ad);
    }
    h[(size_t)inst * p.nc + R] = (extreme + ac) - ad;
  }
  __syncthreads();

  constexpr int TPR = BLOCK / K4_ROWS;  // threads per row
  const int row = tid / TPR, lr = tid - row * TPR;
  if (row >= nrows) return;
  const int naxes = s_nax[row];
  double* Grow = G + ((size_t)inst * p.nc + R0 + row) * no;
  // rows of G start 16-byte aligned only when no is even
  if ((no & 1) == 0 && naxes <= 2) {
    // The kernel waits for memory, not for arithmetic (94 % of its wave cycles on C3), so
    // all loads of four pieces of the row are issued before the first is used.
    double2* G2 = reinterpret_cast<double2*>(Grow);
    const int npair = no >> 1;
    const bool whole_lines = (no & 15) == 0;
    const double a0 = s_arrow[row][0], a1 = naxes > 1 ? s_arrow[row][1] : 0.0;
    const double* v0p = Vb + s_voff[row][0];
    const double* v1p = Vb + s_voff[row][naxes > 1 ? 1 : 0];
    for (int cp0 = lr; cp0 < npair; cp0 += 4 * TPR) {
      double2 v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cp = cp0 + u * TPR < npair ? cp0 + u * TPR : cp0;
        v0[u] = *reinterpret_cast<const double2*>(v0p + 2 * cp);
        v1[u] = *reinterpret_cast<const double2*>(v1p + 2 * cp);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cp = cp0 + u * TPR;
        double2 acc;
        acc.x = fma(a1, v1[u].x, a0 * v0[u].x);
        acc.y = fma(a1, v1[u].y, a0 * v0[u].y);
        // (rows that are whole cache lines leave nontemporal: written once, not read by this
        // launch -- device_prims.h store_result)
        if (cp < npair) {
          if (whole_lines)
            store_result(&G2[cp], acc);
          else
            G2[cp] = acc;
        }
      }
    }
  } else if ((no & 1) == 0) {
    double2* G2 = reinterpret_cast<double2*>(Grow);
    for (int cp = lr; cp < (no >> 1); cp += TPR) {
      double2 acc{0.0, 0.0};
      for (int ax = 0; ax < naxes; ++ax) {
        const double a = s_arrow[row][ax];
        const double2 v = *reinterpret_cast<const double2*>(Vb + s_voff[row][ax] + 2 * cp);
        acc.x = fma(a, v.x, acc.x);
        acc.y = fma(a, v.y, acc.y);
      }
      G2[cp] = acc;
    }
  } else {
    for (int c = lr; c < no; c += TPR) {
      double acc = 0.0;
      for (int ax = 0; ax < naxes; ++ax) acc = fma(s_arrow[row][ax], Vb[s_voff[row][ax] + c], acc);
      Grow[c] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// f2: out[b][r] = PM[b][r][:] . [given ; optim]    (body.py:209-219)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void preview_kernel(const double* __restrict__ PM,
                                                        const double* __restrict__ given,
                                                        const double* __restrict__ optim,
                                                        double* __restrict__ out, int rows, int ng,
                                                        int no, int nrb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long inst = blockIdx.x / nrb;
  const int rb = blockIdx.x - inst * nrb;
  const int W = ng + no;
  const int r = rb * WAVES + wave;
  if (r >= rows) return;
  const double* row = PM + ((size_t)inst * rows + r) * W;
  const double* g = given + (size_t)inst * ng;
  const double* x = optim + (size_t)inst * no;
  double s = 0.0;
  for (int c = lane; c < W; c += 64) s = fma(row[c], c < ng ? g[c] : x[c - ng], s);
  s = wave_sum(s);
  if (lane == 0) out[(size_t)inst * rows + r] = s;
}

inline unsigned ceil_div(unsigned a, unsigned b) { return (a + b - 1) / b; }

}  // namespace

size_t assemble_workspace_bytes(const PlanDev& p, int batch) {
  return (size_t)batch * p.rtot * p.ldv * sizeof(double);
}

int launch_assemble_staged(const PlanDev& p, const SrcTable& src, const double* params,
                           const double* given, double* P, double* q, double* G, double* h,
                           void* work, int batch, hipStream_t stream, hipError_t* err) {
  double* V = static_cast<double*>(work);
  if (p.rtot > 0) {
    const unsigned nrb = ceil_div(p.rtot, WAVES * K2_ROWS_PER_WAVE);
    hipLaunchKernelGGL(compose_rowsets_kernel, dim3(nrb * batch), dim3(BLOCK), 0, stream, p, src,
                       given, V, (int)nrb);
  }
  if (P && p.no >= GB) {
    // wide problems: P by the LDS-tiled GEMM, q by the block kernel on its own column
    const int nb = (int)ceil_div(p.no, GB);
    const int sym = p.rs_sym_any;
    const int npairs = sym ? nb * (nb + 1) / 2 : nb * nb;
    hipLaunchKernelGGL(hessian_gemm_kernel, dim3((unsigned)npairs * batch), dim3(BLOCK), 0, stream,
                       p, params, V, P, nb, npairs, sym);
    const unsigned ncb = ceil_div(p.no, 64);
    hipLaunchKernelGGL(gradient_kernel, dim3(ncb * batch), dim3(BLOCK), 0, stream, p, params, V, q,
                       (int)ncb);
  } else if (P && p.no > 0) {
    const unsigned nblk = ceil_div(p.no, 32) * ceil_div(p.no + 1, 32);
    const unsigned nrb = ceil_div(nblk, WAVES);
    hipLaunchKernelGGL(hessian_kernel, dim3(nrb * batch), dim3(BLOCK), 0, stream, p, params, V, P,
                       q, (int)nrb, 0);
  }
  if (G && p.nc > 0) {
    if (p.max_axes > K4_AXMAX) return MPCASM_ERR_LIMIT;
    const unsigned nrb = ceil_div(p.nc, K4_ROWS);
    const unsigned groups = ceil_div((unsigned)batch, 8u);
    hipLaunchKernelGGL(constraints_kernel, dim3(nrb * groups * 8), dim3(BLOCK), 0, stream, p, params,
                       V, G, h, (int)nrb, batch);
  }
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

int launch_preview_matrices(const PlanDev& p, const SrcTable& src, double* PM, int batch,
                            hipStream_t stream, hipError_t* err) {
  *err = hipSuccess;
  if (p.pmrows > 0 && p.pm_nfd > 0) {
    const unsigned chunks = ceil_div((unsigned)(p.pmrows * (p.ng + p.no)), (unsigned)PM_CHUNK);
    hipLaunchKernelGGL(preview_elements_kernel, dim3(chunks * batch), dim3(BLOCK), 0, stream, p, src,
                       PM, (int)chunks);
    *err = hipGetLastError();
  } else if (p.pmrows > 0) {
    const unsigned nrb = ceil_div(p.pmrows, WAVES * K2_ROWS_PER_WAVE);
    hipLaunchKernelGGL(compose_preview_kernel, dim3(nrb * batch), dim3(BLOCK), 0, stream, p, src,
                       PM, (int)nrb);
    *err = hipGetLastError();
  }
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

int launch_preview(const double* PM, const double* given, const double* optim, double* out,
                   int batch, int rows, int ng, int no, hipStream_t stream, hipError_t* err) {
  const unsigned nrb = ceil_div(rows, WAVES);
  hipLaunchKernelGGL(preview_kernel, dim3(nrb * batch), dim3(BLOCK), 0, stream, PM, given, optim,
                     out, rows, ng, no, (int)nrb);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

// f3: values of a fixed sparsity pattern out of a batch of dense matrices
// (biped_mpc_loop.py:57-58 does this per instance with scipy.sparse.csc_matrix).
// One workgroup walks the pattern of one instance: the index list is read coalesced
// (L2-resident after the first instance), the stores are consecutive; the gathered reads
// stay inside the instance's few KB.
__global__ __launch_bounds__(BLOCK) void gather_kernel(const double* __restrict__ src,
                                                       long long src_stride,
                                                       const int32_t* __restrict__ index, int nnz,
                                                       double* __restrict__ dst, int per_block,
                                                       int batch) {
  const long b0 = (long)blockIdx.x * per_block;
  for (int s = 0; s < per_block && b0 + s < batch; ++s) {
    const double* in = src + (b0 + s) * src_stride;
    double* out = dst + (b0 + s) * (long)nnz;
    for (int k = threadIdx.x; k < nnz; k += BLOCK) out[k] = in[index[k]];
  }
}

int launch_gather(const double* src, long long src_stride, const int32_t* index, int nnz,
                  double* dst, int batch, hipStream_t stream, hipError_t* err) {
  // small patterns: several instances per workgroup, so that a launch has a few
  // thousand workgroups at most and each of them a few KB to move
  int per_block = 1;
  while (per_block < 64 && (long)per_block * nnz < 4096 && batch / (per_block * 2) >= 2048)
    per_block *= 2;
  const unsigned blocks = (unsigned)((batch + per_block - 1) / per_block);
  hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(BLOCK), 0, stream, src, src_stride, index,
                     nnz, dst, per_block, batch);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

// f4: the geometry updates of a Box (restrictions.py:380-486) on the facets' parameters of
// every instance.  One thread per (instance, facet): a facet is a handful of rows and axes.
__global__ __launch_bounds__(BLOCK) void box_transform_kernel(double* __restrict__ params,
                                                              long long nparams, int batch,
                                                              const int32_t* __restrict__ facets,
                                                              int nfacets, int op,
                                                              const double* __restrict__ arg,
                                                              long long arg_stride) {
  const long t = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (t >= (long)batch * nfacets) return;
  const long inst = t / nfacets;
  const int32_t* f = facets + (t - inst * nfacets) * 7;
  double* p = params + inst * nparams;
  double* arrow = p + f[0];
  double* center = p + f[2];
  double* extreme = p + f[4];
  const int arows = f[1], crows = f[3], erows = f[5], axes = f[6];
  const double* a = arg + inst * arg_stride;
  if (op == MPCASM_BOX_RECENTER || op == MPCASM_BOX_TRANSLATE) {
    for (int r = 0; r < crows; ++r)
      for (int x = 0; x < axes; ++x)
        center[r * axes + x] = (op == MPCASM_BOX_TRANSLATE ? center[r * axes + x] : 0.0) + a[x];
    return;
  }
  if (op == MPCASM_BOX_ROTATE) {  // arrow_r <- arrow_r . R^T (axes <= 4 per facet row)
    for (int r = 0; r < arows; ++r) {
      double old[4];
      for (int x = 0; x < axes; ++x) old[x] = arrow[r * axes + x];
      for (int x = 0; x < axes; ++x) {
        double v = 0.0;
        for (int y = 0; y < axes; ++y) v += old[y] * a[x * axes + y];
        arrow[r * axes + x] = v;
      }
    }
  } else if (op == MPCASM_BOX_SCALE) {
    for (int r = 0; r < erows; ++r) extreme[r] *= a[0];
  } else {  // MPCASM_BOX_MARGIN: the Frobenius norm of the whole arrow field, as numpy's norm
    double s = 0.0;
    for (int i = 0; i < arows * axes; ++i) s += arrow[i] * arrow[i];
    const double nrm = sqrt(s);
    for (int r = 0; r < erows; ++r) extreme[r] -= a[0] * nrm;
  }
  // Constraint.normalize(): a negative extreme flips its row (rows match, or both are single)
  for (int r = 0; r < erows; ++r)
    if (extreme[r] < 0.0) {
      extreme[r] = -extreme[r];
      const int ar = arows == 1 ? 0 : r;
      for (int x = 0; x < axes; ++x) arrow[ar * axes + x] = -arrow[ar * axes + x];
    }
}

// f4, state space (restrictions.py:390-404, 417-431 through SS_to_TS, :240-250): the centre of
// facet f becomes (or moves by) L_f,a . p[:, a] per task-space axis a; one thread per
// (instance, facet).
__global__ __launch_bounds__(BLOCK) void box_transform_ss_kernel(
    double* __restrict__ params, long long nparams, int batch, const int32_t* __restrict__ facets,
    int nfacets, int op, const double* __restrict__ L, int lrows, int ss_dim,
    const double* __restrict__ arg, long long arg_stride) {
  const long t = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (t >= (long)batch * nfacets) return;
  const long inst = t / nfacets;
  const int fi = (int)(t - inst * nfacets);
  const int32_t* f = facets + fi * 7;
  double* center = params + inst * nparams + f[2];
  const int axes = f[6];
  const double* p = arg + inst * arg_stride;               // [ss_dim][axes]
  const double* Lf = L + (size_t)fi * axes * lrows * ss_dim;  // [axes][lrows][ss_dim]
  for (int r = 0; r < lrows; ++r)
    for (int x = 0; x < axes; ++x) {
      double v = 0.0;
      for (int k = 0; k < ss_dim; ++k) v = fma(Lf[((size_t)x * lrows + r) * ss_dim + k], p[k * axes + x], v);
      center[r * axes + x] = (op == MPCASM_BOX_TRANSLATE ? center[r * axes + x] : 0.0) + v;
    }
}

int launch_box_transform_ss(double* params, long long nparams, int batch, const int32_t* facets,
                            int nfacets, int op, const double* L, int lrows, int ss_dim,
                            const double* arg, long long arg_stride, hipStream_t stream,
                            hipError_t* err) {
  const long total = (long)batch * nfacets;
  hipLaunchKernelGGL(box_transform_ss_kernel, dim3((unsigned)((total + BLOCK - 1) / BLOCK)),
                     dim3(BLOCK), 0, stream, params, nparams, batch, facets, nfacets, op, L, lrows,
                     ss_dim, arg, arg_stride);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

int launch_box_transform(double* params, long long nparams, int batch, const int32_t* facets,
                         int nfacets, int op, const double* arg, long long arg_stride,
                         hipStream_t stream, hipError_t* err) {
  const long total = (long)batch * nfacets;
  hipLaunchKernelGGL(box_transform_kernel, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK),
                     0, stream, params, nparams, batch, facets, nfacets, op, arg, arg_stride);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

}  // namespace mpcasm
