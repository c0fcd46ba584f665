// fill.hip -- K1: horizon extension (Toeplitz fill) for gfx950.
//
// Replaces tools.extend_matrices (reference python/mpc_interface/tools.py:14-33,
// C++ twin cpp/src/tools.cc:83-144) for a batch of independent systems.
//
//   S[b][k][j][i]    = (A^{k+1})[i][j]
//   U[b][j][k][l][i] = (A^{k-l} B)[i][j]  for l <= k, 0 above the diagonal
//
// The kernel is HBM-write-bound: per system it reads 8(n^2+nm) bytes and
// writes 8(N n^2 + m N^2 n) bytes, zeros included.  Design:
//   * one wavefront (small systems) or one 256-thread workgroup (large) per
//     system; the N blocks A^d B are computed ONCE by the reference's own
//     recurrence X_d = A X_{d-1} (no repeated squaring) and kept in LDS as a
//     block-reversed table R_j[(N-1-d) n + i] = (A^d B)[i][j];
//   * every output row U[j][k][:][:] (N n contiguous doubles) is then a
//     contiguous window of R_j followed by zeros, so the write phase is a pure
//     LDS -> HBM stream of 16-byte stores with consecutive lanes on consecutive
//     addresses and no integer division in the loop;
//   * S is written from registers during the recurrence, in (j, i) order so
//     that consecutive lanes store consecutive doubles.
// The LTV variant (per-step A_k, B_k; no Toeplitz structure) keeps the current
// block row in LDS and advances it with U[k][l] = A_k U[k-1][l].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include "device_common.h"
#include "kernels.h"

namespace mpcasm {

namespace {

constexpr int BLOCK = 256;
constexpr int EPT = 4;  // recurrence elements per thread (n (m+n) <= TPI * EPT)

__host__ __device__ inline size_t even_up(size_t x) { return (x + 1) & ~size_t(1); }

// doubles of LDS one system needs in the LTI kernel
__host__ __device__ inline size_t lti_lds_doubles(int N, int n, int m) {
  return even_up((size_t)m * N * n) + 2 * even_up((size_t)n * (m + n)) + even_up((size_t)n * n);
}

// `count2` sixteen-byte words from LDS (8-byte aligned: ds_read2_b64) to HBM, lane q handles
// words q, q + 64, ...; UNR reads go out before the first store waits for them.  `src` and `dst`
// already carry the lane's own offset (2 lane doubles / lane words).
template <int UNR>
__device__ __forceinline__ void stream_words(const double* __restrict__ src,
                                             double2* __restrict__ dst, int count2, int lane) {
  const int last = count2 - 1 - lane;   // (clamped reads stay inside the run; stores are masked)
  for (int base = 0; base < count2; base += 64 * UNR) {
    double2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int q = min(base + 64 * u, last);
      v[u].x = src[2 * q];
      v[u].y = src[2 * q + 1];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (base + 64 * u <= last) dst[base + 64 * u] = v[u];
  }
}

// PAD (one system per workgroup, N n even): every input's table is followed by N n zeros, so
// that a row of U is one window of it (see fill_lti_quad_kernel) and the wavefronts stream whole
// rows with a constant-address loop instead of walking (row, position) counters with selects.
template <int TPI, bool GENERIC, bool PAD = false>
__global__ __launch_bounds__(BLOCK) void fill_lti_kernel(const double* __restrict__ A,
                                                         const double* __restrict__ B,
                                                         double* __restrict__ S,
                                                         double* __restrict__ U, int batch, int N,
                                                         int n, int m) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int IPB = BLOCK / TPI;
  const int tid = threadIdx.x % TPI;
  const int slot = threadIdx.x / TPI;
  const long inst = (long)blockIdx.x * IPB + slot;
  const bool live = inst < batch;

  const int rl = N * n;        // doubles in one output row U[j][k][:][:]
  const int xw = m + n;        // columns of X = [A^d B | A^{d+1}]
  const int xsz = n * xw;
  const int rs = PAD ? 2 * rl : rl;  // doubles from one input's table to the next
  double* R = lds + (size_t)slot * lti_lds_doubles(N, n, m);  // [m][rs] block-reversed A^d B (| zeros)
  double* X = R + even_up((size_t)m * rs);                    // [2][xw][n] column-major
  double* Am = X + 2 * even_up((size_t)xsz);                  // [n][n] row-major
  const size_t xstep = even_up((size_t)xsz);

  const double* Ab = A + (size_t)inst * n * n;
  const double* Bb = B + (size_t)inst * n * m;
  double* Sb = S + (size_t)inst * N * n * n;
  double* Ub = U + (size_t)inst * m * N * rl;

  // element e of X: column c = e / n (c < m: input c, else state column c-m), row i = e % n.
  // Each thread owns the same few elements in every step, so (c, i) are divided out once;
  // GENERIC (n (m+n) > TPI * EPT) falls back to a strided loop with the division inside.
  int ec[EPT], ei[EPT];
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int e = tid + u * TPI;
    ec[u] = e / n;
    ei[u] = e - ec[u] * n;
  }
  auto for_each_element = [&](auto&& body) {
    if constexpr (!GENERIC) {
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        const int e = tid + u * TPI;
        if (e < xsz) body(e, ec[u], ei[u]);
      }
    } else {
      for (int e = tid; e < xsz; e += TPI) {
        const int c = e / n;
        body(e, c, e - c * n);
      }
    }
  };

  if (PAD) {  // the upper halves of the tables: zeros for the whole launch of this workgroup
    for (int e = tid; e < m * rl; e += TPI) R[(size_t)(e / rl) * rs + rl + (e % rl)] = 0.0;
  }
  if (live) {
    for (int e = tid; e < n * n; e += TPI) Am[e] = Ab[e];
    for_each_element([&](int e, int c, int i) {
      if (c < m) {
        const double v = Bb[i * m + c];
        X[e] = v;
        R[(size_t)c * rs + (size_t)(N - 1) * n + i] = v;  // d = 0
      } else {
        const double v = Ab[i * n + (c - m)];
        X[e] = v;
        Sb[e - n * m] = v;  // S[0][j][i], (j, i) order == e order
      }
    });
  }
  __syncthreads();

  // recurrence X_d = A X_{d-1}  (tools.py:24-29: left-multiply the previous block row)
  for (int d = 1; d < N; ++d) {
    const double* Xp = X + ((d - 1) & 1) * xstep;
    double* Xc = X + (d & 1) * xstep;
    if (live) {
      for_each_element([&](int e, int c, int i) {
        double v = 0.0;
        for (int t = 0; t < n; ++t) v = fma(Am[i * n + t], Xp[c * n + t], v);
        Xc[e] = v;
        if (c < m)
          R[(size_t)c * rs + (size_t)(N - 1 - d) * n + i] = v;
        else
          Sb[(size_t)d * n * n + (e - n * m)] = v;
      });
    }
    // (LDS only: __syncthreads() would also wait, in every step, for the step's stores of S to be
    // acknowledged by memory -- N - 1 round trips per system before the first row of U leaves)
    lds_barrier();
  }

  if (!live) return;

  if (PAD) {
    // rows (j, k) dealt to the wavefronts: row = window of [R_j | zeros] at (N-1-k) n
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rl2 = rl >> 1;
    for (int row = wave; row < m * N; row += TPI / 64) {
      const int j = row / N, k = row - j * N;
      stream_words<6>(R + (size_t)j * rs + (size_t)(N - 1 - k) * n + 2 * lane,
                      reinterpret_cast<double2*>(Ub + (size_t)row * rl) + lane, rl2, lane);
    }
    return;
  }
  // write phase: row (j, k) = window of R_j shifted by (N-1-k) n, zeros after (k+1) n
  if ((rl & 1) == 0) {
    const int rl2 = rl >> 1;
    const long total2 = (long)m * N * rl2;
    const int dr = TPI / rl2, dp = TPI - dr * rl2;
    long q = tid;
    int row = tid / rl2;
    int pos = tid - row * rl2;
    int j = row / N, k = row - j * N;
    double2* __restrict__ out = reinterpret_cast<double2*>(Ub);
    while (q < total2) {
      const int lim = (k + 1) * n, sh = (N - 1 - k) * n;
      const double* Rj = R + (size_t)j * rl;
      const int p0 = 2 * pos;
      const int i0 = min(p0 + sh, rl - 1), i1 = min(p0 + 1 + sh, rl - 1);
      double2 v;
      v.x = p0 < lim ? Rj[i0] : 0.0;
      v.y = p0 + 1 < lim ? Rj[i1] : 0.0;
      out[q] = v;
      q += TPI;
      pos += dp;
      k += dr;
      if (pos >= rl2) {
        pos -= rl2;
        ++k;
      }
      if (k >= N) {
        j += k / N;
        k = k % N;
      }
    }
  } else {
    const long total = (long)m * N * rl;
    const int dr = TPI / rl, dp = TPI - dr * rl;
    long q = tid;
    int row = tid / rl;
    int pos = tid - row * rl;
    int j = row / N, k = row - j * N;
    while (q < total) {
      const int lim = (k + 1) * n, sh = (N - 1 - k) * n;
      const double* Rj = R + (size_t)j * rl;
      Ub[q] = pos < lim ? Rj[min(pos + sh, rl - 1)] : 0.0;
      q += TPI;
      pos += dp;
      k += dr;
      if (pos >= rl) {
        pos -= rl;
        ++k;
      }
      if (k >= N) {
        j += k / N;
        k = k % N;
      }
    }
  }
}

// Tiny systems (n (m+n) <= 32, e.g. the LIPM: n=3, m=1): the recurrence occupies only
// n (m+n) lanes, so one wavefront runs SPW = 64 / (n (m+n)) systems side by side and needs
// no workgroup barrier at all (a wavefront's LDS operations complete in order); S is staged
// in LDS as well, and the wavefront then streams S and U of its systems with 16-byte stores.
__host__ __device__ inline size_t tiny_lds_doubles(int N, int n, int m) {
  return even_up((size_t)m * N * n) + 2 * even_up((size_t)n * (m + n)) + even_up((size_t)n * n) +
         even_up((size_t)N * n * n);
}

__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(BLOCK) void fill_lti_tiny_kernel(const double* __restrict__ A,
                                                              const double* __restrict__ B,
                                                              double* __restrict__ S,
                                                              double* __restrict__ U, int batch,
                                                              int N, int n, int m, int spw) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xw = m + n, xsz = n * xw, rl = N * n, nn = n * n;
  const size_t per = tiny_lds_doubles(N, n, m);
  const long sys0 = ((long)blockIdx.x * (BLOCK / 64) + wave) * spw;  // first system of this wave

  // recurrence role: lane -> (system sub, element e = c n + i)
  const int sub = lane / xsz, e = lane - sub * xsz;
  const int c = e / n, i = e - c * n;
  const bool worker = sub < spw && sys0 + sub < batch;
  double* base = lds + ((size_t)wave * spw + (sub < spw ? sub : 0)) * per;
  double* R = base;                                  // [m][rl] block-reversed A^d B
  double* X = R + even_up((size_t)m * rl);           // [2][xw][n]
  double* Am = X + 2 * even_up((size_t)xsz);         // [n][n]
  double* Sl = Am + even_up((size_t)nn);             // [N][n][n] as stored in S
  const size_t xstep = even_up((size_t)xsz);
  if (worker) {
    const double* Ab = A + (size_t)(sys0 + sub) * nn;
    const double* Bb = B + (size_t)(sys0 + sub) * n * m;
    if (e < nn) Am[e] = Ab[e];
    double v;
    if (c < m) {
      v = Bb[i * m + c];
      R[(size_t)c * rl + (size_t)(N - 1) * n + i] = v;
    } else {
      v = Ab[i * n + (c - m)];
      Sl[e - n * m] = v;
    }
    X[e] = v;
  }
  wave_lds_sync();
  for (int d = 1; d < N; ++d) {
    const double* Xp = X + ((d - 1) & 1) * xstep;
    double* Xc = X + (d & 1) * xstep;
    if (worker) {
      double v = 0.0;
      for (int t = 0; t < n; ++t) v = fma(Am[i * n + t], Xp[c * n + t], v);
      Xc[e] = v;
      if (c < m)
        R[(size_t)c * rl + (size_t)(N - 1 - d) * n + i] = v;
      else
        Sl[(size_t)d * nn + (e - n * m)] = v;
    }
    wave_lds_sync();
  }

  // write phase: all 64 lanes stream the wave's systems one after the other
  const bool vec = (rl & 1) == 0 && ((N * nn) & 1) == 0;
  for (int s = 0; s < spw; ++s) {
    const long sys = sys0 + s;
    if (sys >= batch) break;
    const double* Rs = lds + ((size_t)wave * spw + s) * per;
    const double* Ss = Rs + even_up((size_t)m * rl) + 2 * even_up((size_t)xsz) + even_up((size_t)nn);
    double* Sb = S + (size_t)sys * N * nn;
    double* Ub = U + (size_t)sys * m * N * rl;
    if (vec) {
      const int s2 = (N * nn) >> 1;
      double2* S2 = reinterpret_cast<double2*>(Sb);
      const double2* Sl2 = reinterpret_cast<const double2*>(Ss);
      for (int q = lane; q < s2; q += 64) S2[q] = Sl2[q];
      const int rl2 = rl >> 1;
      const long total2 = (long)m * N * rl2;
      const int dr = 64 / rl2, dp = 64 - dr * rl2;
      long q = lane;
      int row = lane / rl2;
      int pos = lane - row * rl2;
      int j = row / N, k = row - j * N;
      double2* out = reinterpret_cast<double2*>(Ub);
      while (q < total2) {
        const int lim = (k + 1) * n, sh = (N - 1 - k) * n;
        const double* Rj = Rs + (size_t)j * rl;
        const int p0 = 2 * pos;
        const int i0 = min(p0 + sh, rl - 1), i1 = min(p0 + 1 + sh, rl - 1);
        double2 v;
        v.x = p0 < lim ? Rj[i0] : 0.0;
        v.y = p0 + 1 < lim ? Rj[i1] : 0.0;
        out[q] = v;
        q += 64;
        pos += dp;
        k += dr;
        if (pos >= rl2) {
          pos -= rl2;
          ++k;
        }
        if (k >= N) {
          j += k / N;
          k = k % N;
        }
      }
    } else {
      for (int q = lane; q < N * nn; q += 64) Sb[q] = Ss[q];
      const long total = (long)m * N * rl;
      for (long q = lane; q < total; q += 64) {
        const int row = (int)(q / rl), pos = (int)(q - (long)row * rl);
        const int j = row / N, k = row - j * N;
        Ub[q] = pos < (k + 1) * n ? Rs[(size_t)j * rl + pos + (N - 1 - k) * n] : 0.0;
      }
    }
  }
}

// doubles of LDS one system needs in the LTV kernel
__host__ __device__ inline size_t ltv_lds_doubles(int N, int n, int m) {
  return 2 * even_up((size_t)m * N * n) + 2 * even_up((size_t)n * n) + even_up((size_t)n * n) +
         even_up((size_t)n * m);
}

// LTV: x_{k+1} = A_k x_k + B_k u_k.  Row k of U is A_k times row k-1 plus B_k on
// the diagonal; rows are streamed to HBM as they are produced.
template <int TPI>
__global__ __launch_bounds__(BLOCK) void fill_ltv_kernel(const double* __restrict__ A,
                                                         const double* __restrict__ B,
                                                         double* __restrict__ S,
                                                         double* __restrict__ U, int batch, int N,
                                                         int n, int m) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int IPB = BLOCK / TPI;
  const int tid = threadIdx.x % TPI;
  const int slot = threadIdx.x / TPI;
  const long inst = (long)blockIdx.x * IPB + slot;
  const bool live = inst < batch;

  const int rl = N * n;
  const size_t rstep = even_up((size_t)m * rl), pstep = even_up((size_t)n * n);
  double* Rw = lds + (size_t)slot * ltv_lds_doubles(N, n, m);  // [2][m][rl] current block row
  double* Pk = Rw + 2 * rstep;                                 // [2][n][n] as [j][i]
  double* Ak = Pk + 2 * pstep;                                 // [n][n] row-major
  double* Bk = Ak + pstep;                                     // [n][m] row-major

  const double* Ab = A + (size_t)inst * N * n * n;
  const double* Bb = B + (size_t)inst * N * n * m;
  double* Sb = S + (size_t)inst * N * n * n;
  double* Ub = U + (size_t)inst * m * N * rl;

  // thread <-> fixed (l-lane, i): no division inside the loops
  const int LQ = TPI / n;  // l values covered per pass
  const int lq = tid / n, li = tid - lq * n;
  const bool worker = lq < LQ;

  for (int k = 0; k < N; ++k) {
    const double* Rp = Rw + ((k + 1) & 1) * rstep;  // row k-1
    double* Rc = Rw + (k & 1) * rstep;              // row k
    const double* Pp = Pk + ((k + 1) & 1) * pstep;
    double* Pc = Pk + (k & 1) * pstep;
    if (live) {
      for (int e = tid; e < n * n; e += TPI) Ak[e] = Ab[(size_t)k * n * n + e];
      for (int e = tid; e < n * m; e += TPI) Bk[e] = Bb[(size_t)k * n * m + e];
    }
    __syncthreads();
    if (live) {
      // S[k] = (A_k P_{k-1})^T, P_{-1} = I ; Pc[j][i] = P[i][j]
      for (int e = tid; e < n * n; e += TPI) {
        const int j = e / n, i = e - j * n;
        double v;
        if (k == 0) {
          v = Ak[i * n + j];
        } else {
          v = 0.0;
          for (int t = 0; t < n; ++t) v = fma(Ak[i * n + t], Pp[j * n + t], v);
        }
        Pc[e] = v;
        Sb[(size_t)k * n * n + e] = v;
      }
      if (worker) {
        for (int j = 0; j < m; ++j) {
          const double* Rpj = Rp + (size_t)j * rl;
          double* Rcj = Rc + (size_t)j * rl;
          double* out = Ub + ((size_t)j * N + k) * rl;
          for (int l = lq; l < N; l += LQ) {
            double v = 0.0;
            if (l < k) {
              for (int t = 0; t < n; ++t) v = fma(Ak[li * n + t], Rpj[l * n + t], v);
            } else if (l == k) {
              v = Bk[li * m + j];
            }
            if (l <= k) Rcj[l * n + li] = v;
            out[l * n + li] = v;
          }
        }
      }
    }
    __syncthreads();
  }
}

// LTV, one 256-thread workgroup per system, no HBM read inside the step loop: every
// step's (A_k, B_k) is copied to LDS up front (one wait, before the first store), so the
// loop only computes and stores and its barriers order LDS only -- a barrier or a load wait
// that drained vmcnt would cost one HBM write round trip per step.  Step k: thread (l, i)
// makes element i of block l <= k of row k from row k-1, while row k-1 -- complete, zeros
// right of the diagonal included -- streams to HBM as 16-byte stores.  LDS per system: two
// rows + N (n^2 + n m) doubles: a few KB, so eight workgroups share a CU and the store
// stream of one hides the recurrence of the others.
__host__ __device__ inline size_t ltv_block_lds_doubles(int N, int n, int m) {
  return 2 * even_up((size_t)m * N * n) + 2 * even_up((size_t)n * n) +
         (size_t)N * (even_up((size_t)n * n) + even_up((size_t)n * m));
}

__device__ __forceinline__ void block_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// NS: the number of states when it is small (the products unroll, A_k's row lives in
// registers), 0 = any
template <int NS>
__global__ __launch_bounds__(BLOCK) void fill_ltv_block_kernel(const double* __restrict__ A,
                                                               const double* __restrict__ B,
                                                               double* __restrict__ S,
                                                               double* __restrict__ U, int N,
                                                               int n_any, int m) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int n = NS ? NS : n_any;
  const int tid = threadIdx.x;
  const long inst = blockIdx.x;
  const int rl = N * n, nn = n * n, nm = n * m;
  const size_t rstep = even_up((size_t)m * rl), pstep = even_up((size_t)nn);
  const size_t abstep = even_up((size_t)nn) + even_up((size_t)nm);
  double* Rw = lds;               // [2][m][rl] rows k-1 / k of every input's block row
  double* Pk = Rw + 2 * rstep;    // [2][n][n] running product as [j][i]
  double* AB = Pk + 2 * pstep;    // [N]{A_k [n][n], B_k [n][m]}

  const double* Ab = A + (size_t)inst * N * nn;
  const double* Bb = B + (size_t)inst * N * nm;
  double* Sb = S + (size_t)inst * N * nn;
  double* Ub = U + (size_t)inst * m * N * rl;

  for (size_t e = tid; e < 2 * rstep; e += BLOCK) Rw[e] = 0.0;  // right of the diagonal: zeros
  for (int i = tid; i < N * nn; i += BLOCK) {
    const int k = i / nn;
    AB[k * abstep + (i - k * nn)] = Ab[i];
  }
  for (int i = tid; i < N * nm; i += BLOCK) {
    const int k = i / nm;
    AB[k * abstep + even_up((size_t)nn) + (i - k * nm)] = Bb[i];
  }
  block_lds_barrier();

  // thread <-> fixed (l-lane, i): no division inside the loops
  const int LQ = BLOCK / n;
  const int lq = tid / n, li = tid - lq * n;
  const bool worker = lq < LQ;
  const int sj = tid / n, si = tid - sj * n;  // S element (j, i) of the first pass

  auto stream_row = [&](const double* Rrow, int k) {
    for (int j = 0; j < m; ++j) {
      const double* Rj = Rrow + (size_t)j * rl;
      double* out = Ub + ((size_t)j * N + k) * rl;
      if ((rl & 1) == 0) {
        const double2* r2 = reinterpret_cast<const double2*>(Rj);
        double2* o2 = reinterpret_cast<double2*>(out);
        for (int q = tid; q < (rl >> 1); q += BLOCK) o2[q] = r2[q];
      } else {
        for (int q = tid; q < rl; q += BLOCK) out[q] = Rj[q];
      }
    }
  };

  for (int k = 0; k < N; ++k) {
    const double* Ak = AB + (size_t)k * abstep;
    const double* Bk = Ak + even_up((size_t)nn);
    const double* Rp = Rw + ((k + 1) & 1) * rstep;  // row k-1
    double* Rc = Rw + (k & 1) * rstep;              // row k
    const double* Pp = Pk + ((k + 1) & 1) * pstep;
    double* Pc = Pk + (k & 1) * pstep;

    // S[k] = (A_k P_{k-1})^T, P_{-1} = I ; Pc[j][i] = P[i][j]
    for (int e = tid; e < nn; e += BLOCK) {
      const int j = e == tid ? sj : e / n, i = e == tid ? si : e - (e / n) * n;
      double v;
      if (k == 0) {
        v = Ak[i * n + j];
      } else {
        v = 0.0;
        for (int t = 0; t < n; ++t) v = fma(Ak[i * n + t], Pp[j * n + t], v);
      }
      Pc[e] = v;
      Sb[(size_t)k * nn + e] = v;
    }
    // row k: blocks l < k are A_k times row k-1, block k is B_k, blocks l > k stay zero
    if (worker) {
      double arow[NS ? NS : 1];
      if (NS) {
#pragma unroll
        for (int t = 0; t < NS; ++t) arow[t] = Ak[li * NS + t];
      }
      for (int j = 0; j < m; ++j) {
        const double* Rpj = Rp + (size_t)j * rl;
        double* Rcj = Rc + (size_t)j * rl;
        for (int l = lq; l <= k; l += LQ) {
          double v;
          if (l < k) {
            v = 0.0;
            if (NS) {
#pragma unroll
              for (int t = 0; t < NS; ++t) v = fma(arow[t], Rpj[l * NS + t], v);
            } else {
              for (int t = 0; t < n; ++t) v = fma(Ak[li * n + t], Rpj[l * n + t], v);
            }
          } else {
            v = Bk[li * m + j];
          }
          Rcj[l * n + li] = v;
        }
      }
    }
    // row k-1 (complete since the last barrier) goes to HBM while row k is being made
    if (k > 0) stream_row(Rp, k - 1);
    block_lds_barrier();
  }
  stream_row(Rw + ((N - 1) & 1) * rstep, N - 1);
}

// LTV, one wavefront per system (n <= 64, N n <= 1024): no workgroup barrier -- the
// wavefront double-buffers the block row AND the step matrices in its own LDS slice,
// loads (A_{k+1}, B_{k+1}) into registers while row k is being produced, and needs one
// wave-level LDS sync per step.
// ALL: every step's (A_k, B_k) of the system sits in LDS from the start (one wait, before the
// first store); otherwise two slots, refilled from HBM a step ahead.  A wavefront that must
// wait for a load while its row stores are in flight waits for the stores too (one vmcnt
// counts both), i.e. one HBM write round trip per step: ALL is several times faster whenever
// the N (n^2 + n m) doubles fit.
__host__ __device__ inline size_t ltv_wave_lds_doubles(int N, int n, int m, bool all = false) {
  return 2 * even_up((size_t)m * N * n) + 2 * even_up((size_t)n * n) +
         (all ? (size_t)N : 2) * (even_up((size_t)n * n) + even_up((size_t)n * m));
}

template <bool ALL>
__global__ __launch_bounds__(BLOCK) void fill_ltv_wave_kernel(const double* __restrict__ A,
                                                              const double* __restrict__ B,
                                                              double* __restrict__ S,
                                                              double* __restrict__ U, int batch,
                                                              int N, int n, int m) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long inst = (long)blockIdx.x * (BLOCK / 64) + wave;
  if (inst >= batch) return;  // whole wavefront; no barrier below

  const int rl = N * n, nn = n * n, nm = n * m;
  const size_t rstep = even_up((size_t)m * rl), pstep = even_up((size_t)nn);
  const size_t abstep = even_up((size_t)nn) + even_up((size_t)nm);
  double* Rw = lds + (size_t)wave * ltv_wave_lds_doubles(N, n, m, ALL);  // [2][m][rl]
  double* Pk = Rw + 2 * rstep;                                      // [2][n][n] as [j][i]
  double* AB = Pk + 2 * pstep;                                      // [2]{A_k [n][n], B_k [n][m]}

  const double* Ab = A + (size_t)inst * N * nn;
  const double* Bb = B + (size_t)inst * N * nm;
  double* Sb = S + (size_t)inst * N * nn;
  double* Ub = U + (size_t)inst * m * N * rl;

  const int LQ = 64 / n;
  const int lq = lane / n, li = lane - lq * n;
  const bool worker = lq < LQ;
  const int ej = lane / n, ei = lane - ej * n;  // S element (j, i) for lane < n n (first pass)

  // both row buffers start as zeros: the blocks right of the diagonal are never written
  for (size_t e = lane; e < 2 * rstep; e += 64) Rw[e] = 0.0;
  // step 0 matrices (ALL: those of every step)
  {  // (flat, coalesced copies: the steps' matrices are contiguous in HBM)
    const int na = (ALL ? N : 1) * nn, nb = (ALL ? N : 1) * nm;
    for (int i = lane; i < na; i += 64) {
      const int k = i / nn;
      AB[k * abstep + (i - k * nn)] = Ab[i];
    }
    for (int i = lane; i < nb; i += 64) {
      const int k = i / nm;
      AB[k * abstep + even_up((size_t)nn) + (i - k * nm)] = Bb[i];
    }
  }
  wave_lds_sync();

  auto stream_row = [&](const double* Rrow, int k) {
    for (int j = 0; j < m; ++j) {
      const double* Rj = Rrow + (size_t)j * rl;
      double* out = Ub + ((size_t)j * N + k) * rl;
      if ((rl & 1) == 0) {
        const double2* r2 = reinterpret_cast<const double2*>(Rj);
        double2* o2 = reinterpret_cast<double2*>(out);
        for (int q = lane; q < (rl >> 1); q += 64) o2[q] = r2[q];
      } else {
        for (int q = lane; q < rl; q += 64) out[q] = Rj[q];
      }
    }
  };

  for (int k = 0; k < N; ++k) {
    const double* Ak = AB + (ALL ? k : (k & 1)) * abstep;
    const double* Bk = Ak + even_up((size_t)nn);
    double* ABn = AB + ((k + 1) & 1) * abstep;
    const double* Rp = Rw + ((k + 1) & 1) * rstep;  // row k-1
    double* Rc = Rw + (k & 1) * rstep;              // row k
    const double* Pp = Pk + ((k + 1) & 1) * pstep;
    double* Pc = Pk + (k & 1) * pstep;

    // next step's matrices leave HBM now (register staged; at most 2 + 1 values per lane
    // on the small shapes this kernel is dispatched for)
    double an[2] = {0.0, 0.0}, bn = 0.0;
    const bool more = !ALL && k + 1 < N;
    if (more) {
      if (lane < nn) an[0] = Ab[(size_t)(k + 1) * nn + lane];
      if (lane + 64 < nn) an[1] = Ab[(size_t)(k + 1) * nn + lane + 64];
      if (lane < nm) bn = Bb[(size_t)(k + 1) * nm + lane];
    }

    // S[k] = (A_k P_{k-1})^T, P_{-1} = I ; Pc[j][i] = P[i][j]
    for (int e = lane; e < nn; e += 64) {
      const int j = e == lane ? ej : e / n, i = e == lane ? ei : e - (e / n) * n;
      double v;
      if (k == 0) {
        v = Ak[i * n + j];
      } else {
        v = 0.0;
        for (int t = 0; t < n; ++t) v = fma(Ak[i * n + t], Pp[j * n + t], v);
      }
      Pc[e] = v;
      Sb[(size_t)k * nn + e] = v;
    }
    // row k of every input: blocks l < k are A_k times row k-1, block k is B_k; blocks
    // l > k stay zero in the buffer (zeroed once, never written)
    if (worker) {
      for (int j = 0; j < m; ++j) {
        const double* Rpj = Rp + (size_t)j * rl;
        double* Rcj = Rc + (size_t)j * rl;
        for (int l = lq; l <= k; l += LQ) {
          double v;
          if (l < k) {
            v = 0.0;
            for (int t = 0; t < n; ++t) v = fma(Ak[li * n + t], Rpj[l * n + t], v);
          } else {
            v = Bk[li * m + j];
          }
          Rcj[l * n + li] = v;
        }
      }
    }
    // stream the PREVIOUS row (complete since the last sync; zeros included) to HBM with
    // 16-byte stores while this one is being produced: both only read row k-1
    if (k > 0) stream_row(Rp, k - 1);
    if (more) {
      if (lane < nn) ABn[lane] = an[0];
      if (lane + 64 < nn) ABn[lane + 64] = an[1];
      if (lane < nm) ABn[even_up((size_t)nn) + lane] = bn;
    }
    wave_lds_sync();
  }
  stream_row(Rw + ((N - 1) & 1) * rstep, N - 1);
}

// ---------------------------------------------------------------------------------------
// Few states (n <= 4, n + m <= 16: the LIPM family of C2 / C3 / C5), one wavefront per
// workgroup.  Both kernels below are written against ONE budget: vector instructions per
// stored kilobyte.  A wave64 instruction occupies its SIMD for ~4 cycles whatever it does, so
// a write loop that spends 40 instructions of index arithmetic per 1 KiB store (the kernels
// above: running (row, position) counters, clamps, selects) is bound by instruction issue,
// not by HBM, as soon as every CU holds a few waves.  Here the per-lane part of every address
// is a constant computed once per wave and the per-row part is wave-uniform (scalar unit):
// a store costs one LDS read (ds_read2_b64) and the store itself.
//
// LTI: the recurrence runs in registers -- a system occupies m + n quads of lanes, lane
// (c, i) holds X[i][c] of X_d = [A^d B | A^{d+1}], A's row i sits in registers and the n
// values X[t][c] come from the lane's own quad with DPP quad broadcasts: no LDS round trip in
// the dependent chain (~40 cycles a step instead of ~150).  Every step drops its values into
// LDS: the state columns in S's own order, the input columns into a block-reversed table
// R_j[(N-1-d) n + i] that is FOLLOWED BY N n ZEROS -- row k of U_j is then the window
// T_j[(N-1-k) n ...][0 .. N n), structural zeros included, with no select and no clamp.
// Rows shorter than a wavefront (N n / 2 sixteen-byte words <= 32) go out 64 / L rows per
// instruction, L the next power of two.
__host__ __device__ inline size_t quad_lds_doubles(int N, int n, int m, int spw) {
  return (size_t)spw * ((size_t)N * n * n + (size_t)m * 2 * N * n) + 64;   // + idle lanes' scratch
}

template <int NS>
__global__ __launch_bounds__(64) void fill_lti_quad_kernel(const double* __restrict__ A,
                                                           const double* __restrict__ B,
                                                           double* __restrict__ S,
                                                           double* __restrict__ U, int batch, int N,
                                                           int m, int spw, int lshift, int whole_lines) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int n = NS, nn = NS * NS;
  // rows of U that are whole cache lines (N n a multiple of 16) leave with nontemporal stores:
  // 0.68 -> 0.77 of HBM at 524 288 C2 systems, no difference below 65 536; rows that end
  // inside a line (N n = 300) lose a fifth with them -- the halves of a line must meet in L2
  auto put = [whole_lines](double2* dst, double2 v) {
    if (whole_lines)
      store_result(dst, v);
    else
      *dst = v;
  };
  const int lane = threadIdx.x;
  const int rl = N * n, rl2 = rl >> 1;
  const long sys0 = (long)blockIdx.x * spw;
  const int nsys = (int)min((long)spw, (long)batch - sys0);
  double* Sl = lds;                          // [spw][N][n][n], S's own layout
  double* T = lds + (size_t)spw * N * nn;    // [spw][m]{R_j (rl) | zeros (rl)}

  // recurrence role: quad c of system `sub`, row i
  const int lps = 4 * (m + n);               // lanes per system
  const int sub = lane / lps, within = lane - sub * lps;
  const int c = within >> 2, i = within & 3;
  const bool worker = sub < nsys && i < n;
  double a[NS], x = 0.0;
#pragma unroll
  for (int t = 0; t < NS; ++t) a[t] = 0.0;
  if (worker) {
    const double* Ab = A + (size_t)(sys0 + sub) * nn;
    const double* Bb = B + (size_t)(sys0 + sub) * n * m;
#pragma unroll
    for (int t = 0; t < NS; ++t) a[t] = Ab[i * n + t];
    x = c < m ? Bb[i * m + c] : Ab[i * n + (c - m)];
  }
  {  // the tables start as zeros (the upper halves stay that way); the loads are in flight
    double2* T2 = reinterpret_cast<double2*>(T);
    const int count2 = spw * m * rl;         // (2 rl doubles per table = rl sixteen-byte words)
    const double2 zero = {0.0, 0.0};
    for (int e = lane; e < count2; e += 64) T2[e] = zero;
  }
  // where this lane's values go, step after step; idle lanes write their own scratch word
  double* w = T + (size_t)spw * m * 2 * rl + lane;
  int wstep = 0;
  if (worker) {
    if (c < m) {
      w = T + (size_t)(sub * m + c) * 2 * rl + (size_t)(N - 1) * n + i;
      wstep = -n;
    } else {
      w = Sl + (size_t)sub * N * nn + (c - m) * n + i;
      wstep = nn;
    }
  }
  // X_d = A X_{d-1}  (tools.py:24-29: left-multiply the previous block row)
  for (int d = 0; d < N; ++d) {
    *w = x;
    w += wstep;
    double y = a[0] * quad_broadcast<0>(x);
    if (NS > 1) y = fma(a[NS > 1 ? 1 : 0], quad_broadcast<1>(x), y);
    if (NS > 2) y = fma(a[NS > 2 ? 2 : 0], quad_broadcast<2>(x), y);
    if (NS > 3) y = fma(a[NS > 3 ? 3 : 0], quad_broadcast<3>(x), y);
    x = y;
  }
  wave_lds_sync();

  // S of the wave's systems: contiguous in HBM and in LDS, one flat copy
  stream_words<4>(Sl + 2 * lane, reinterpret_cast<double2*>(S + (size_t)sys0 * N * nn) + lane,
                  (nsys * N * nn) >> 1, lane);
  if (rl2 > 64) {  // long rows: a row is ceil(rl2 / 64) instructions
    for (int sj = 0; sj < nsys * m; ++sj) {
      const double* Tj = T + (size_t)sj * 2 * rl + 2 * lane;
      double2* out = reinterpret_cast<double2*>(U + ((size_t)sys0 * m + sj) * N * rl) + lane;
      for (int k = 0; k < N; ++k)
        stream_words<4>(Tj + (N - 1 - k) * n, out + (size_t)k * rl2, rl2, lane);
    }
  } else {
    // short rows: R = 64 >> lshift rows per instruction, lane = (row r, word p) -- constants
    const int L = 1 << lshift, R = 64 >> lshift;
    const int r = lane >> lshift, p = lane & (L - 1);
    const int full = N / R;                    // row groups with R valid rows
    const bool tail = full * R + r < N;        // this lane's row of the last, partial group
    const int lds_step = R * n, out_step = R * rl2;
    for (int sj = 0; sj < nsys * m; ++sj) {
      // group g: row k = g R + r, window T_j[(N-1-k) n + 2 p ...]
      const double* win = T + (size_t)sj * 2 * rl + (N - 1 - r) * n + 2 * p;
      double2* dst = reinterpret_cast<double2*>(U + ((size_t)sys0 * m + sj) * N * rl) +
                     (size_t)r * rl2 + p;
      if (p < rl2) {
        int g = 0;
        for (; g + 4 <= full; g += 4) {
          double2 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            v[u].x = win[-(g + u) * lds_step];
            v[u].y = win[-(g + u) * lds_step + 1];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) put(&dst[(size_t)(g + u) * out_step], v[u]);
        }
        for (; g < full; ++g) {
          double2 v;
          v.x = win[-g * lds_step];
          v.y = win[-g * lds_step + 1];
          put(&dst[(size_t)g * out_step], v);
        }
        if (tail) {
          double2 v;
          v.x = win[-full * lds_step];
          v.y = win[-full * lds_step + 1];
          put(&dst[(size_t)full * out_step], v);
        }
      }
    }
  }
}

// `count` doubles from HBM to LDS, eight loads per lane in flight (16-byte loads when the
// source allows)
__device__ __forceinline__ void copy_to_lds(double* __restrict__ dst, const double* __restrict__ src,
                                            int count, int lane) {
  if (((count & 1) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
    const double2* s2 = reinterpret_cast<const double2*>(src);
    double2* d2 = reinterpret_cast<double2*>(dst);
    const int c2 = count >> 1;
    for (int base = 0; base < c2; base += 64 * 8) {
      double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s2[min(base + 64 * u + lane, c2 - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + 64 * u + lane < c2) d2[base + 64 * u + lane] = v[u];
    }
  } else {
    for (int base = 0; base < count; base += 64 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[min(base + 64 * u + lane, count - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + 64 * u + lane < count) dst[base + 64 * u + lane] = v[u];
    }
  }
}

// LTV, one wavefront per system, the whole block row advanced in place of the reference's
// per-block products: the row buffer holds [P_k (n columns of the running product) | block
// l = 0 .. N-1 of every input], every block is "A_k times the same block one step ago", so
// ONE unrolled pass structure serves S and U, lanes are (block, i) with 64 / n blocks per
// pass, and blocks right of the diagonal are zeros times A_k: computed like the others, no
// select; the diagonal block B_k is written after the passes (LDS operations of a wavefront
// complete in order).  A pass is two LDS reads, n FMAs with A_k's row in registers and one LDS
// write at immediate offsets.  Row k-1 streams to HBM (ds_read2_b64 + 16-byte stores) while
// row k is being made.  Every step's (A_k, B_k) is copied to LDS before the first store (a
// load inside the step loop would wait for the row stores in flight: one vmcnt counts both).
template <int NS>
struct LtvRow {
  static constexpr int LQ = 64 / NS;            // blocks per pass
  static constexpr int PASS = LQ * NS;          // doubles per pass
  __host__ __device__ static int passes(int N, int m) { return (NS + m * N + LQ - 1) / LQ; }
  __host__ __device__ static size_t row_doubles(int N, int m) {
    return even_up((size_t)passes(N, m) * PASS + 2);
  }
  __host__ __device__ static size_t a_doubles(int N) { return even_up((size_t)N * NS * NS); }
  __host__ __device__ static size_t lds_doubles(int N, int m) {
    return 2 * row_doubles(N, m) + a_doubles(N) + even_up((size_t)N * NS * m);
  }
};

template <int NS, int P>
__device__ __forceinline__ void ltv_passes(const double* __restrict__ rd, double* __restrict__ wr,
                                           const double (&a)[NS], bool worker) {
  constexpr int PASS = LtvRow<NS>::PASS;
  double x[P][NS];
#pragma unroll
  for (int q = 0; q < P; ++q)
#pragma unroll
    for (int t = 0; t < NS; ++t) x[q][t] = rd[q * PASS + t];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    double y = a[0] * x[q][0];
#pragma unroll
    for (int t = 1; t < NS; ++t) y = fma(a[t], x[q][t], y);
    if (worker) wr[q * PASS] = y;
  }
}

template <int NS>
__global__ __launch_bounds__(64) void fill_ltv_row_kernel(const double* __restrict__ A,
                                                          const double* __restrict__ B,
                                                          double* __restrict__ S,
                                                          double* __restrict__ U, int N, int m) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using Lay = LtvRow<NS>;
  constexpr int n = NS, nn = NS * NS, LQ = Lay::LQ, PASS = Lay::PASS;
  const int lane = threadIdx.x;
  const long inst = blockIdx.x;
  const int rl = N * n, rl2 = rl >> 1, nm = n * m;
  const size_t rstep = Lay::row_doubles(N, m);
  double* Row = lds;                     // [2][rstep]
  double* At = lds + 2 * rstep;          // [N][n][n], as in HBM
  double* Bt = At + Lay::a_doubles(N);   // [N][n][m], as in HBM

  const double* Ab = A + (size_t)inst * N * nn;
  const double* Bb = B + (size_t)inst * N * nm;
  double* Sb = S + (size_t)inst * N * nn;
  double* Ub = U + (size_t)inst * m * N * rl;

  // every step's matrices: flat copies, all loads in flight before the first LDS write waits
  // for one (a load-wait-write loop would pay one HBM latency per 64 elements)
  copy_to_lds(At, Ab, N * nn, lane);
  copy_to_lds(Bt, Bb, N * nm, lane);
  {  // zeros in both rows, P_{-1} = I in row "-1"
    double2* Z = reinterpret_cast<double2*>(Row);
    const double2 zero = {0.0, 0.0};
    for (int e = lane; e < (int)rstep; e += 64) Z[e] = zero;
    wave_lds_sync();
    if (lane < n) Row[rstep + lane * n + lane] = 1.0;
  }

  const int lq = lane / n, li = lane - lq * n;
  const bool worker = lq < LQ;
  const int rd_off = worker ? lq * n : 0, wr_off = worker ? lq * n + li : 0;
  const int arow = worker ? li * n : 0;
  const int bj = lane / n, bi = lane - bj * n;   // B_k element (i, j) of lane < n m
  const int boff = lane < nm ? bi * m + bj : 0;
  const int bdiag = nn + bj * N * n + bi;        // (+ k n) where B_k[i][j] lands in row k

  auto stream_row = [&](const double* Rrow, int k) {
    const double sv = Rrow[lane < nn ? lane : 0];
    for (int j = 0; j < m; ++j)
      stream_words<4>(Rrow + nn + (size_t)j * rl + 2 * lane,
                      reinterpret_cast<double2*>(Ub + ((size_t)j * N + k) * rl) + lane, rl2, lane);
    if (lane < nn) Sb[(size_t)k * nn + lane] = sv;
  };

  // One LDS round trip per step: everything step k reads -- the NEXT step's A row and B element,
  // row k-1 for the stream (S and up to four 1 KiB store instructions of U) and row k-1 for
  // the passes -- is issued before the first result is waited for; then FMAs, LDS writes and
  // the global stores, none of which anything waits for.
  const bool fast = m == 1 && rl2 <= 256;
  const int last = rl2 - 1 - lane;
  const double* ak = At + arow;    // A_k's row of this lane
  const double* bp = Bt + boff;    // B_k's element of this lane
  double a[NS], bk = *bp;
#pragma unroll
  for (int t = 0; t < NS; ++t) a[t] = ak[t];
  int cur = 0;
  double* sdst = Sb + (lane < nn ? lane : 0) - nn;            // S[k-1]
  double2* udst = reinterpret_cast<double2*>(Ub) + lane - rl2;  // U_0[k-1]
  for (int k = 0; k < N; ++k) {
    const double* Rp = Row + (cur ^ 1) * rstep;  // row k-1
    double* Rc = Row + cur * rstep;              // row k
    if (k + 1 < N) {
      ak += nn;
      bp += nm;
    }
    double an[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) an[t] = ak[t];
    const double bn = *bp;
    const double sv = Rp[lane < nn ? lane : 0];
    double2 v[4];
    if (fast) {
      const double* src = Rp + nn + 2 * lane;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = min(64 * u, last);
        v[u].x = src[2 * q];
        v[u].y = src[2 * q + 1];
      }
    }
    // blocks that can be non-zero in row k-1: P and, of the LAST input, l < k
    const int live = n + (m - 1) * N + k;
    const int np = (live + LQ - 1) / LQ;
    const double* rd = Rp + rd_off;
    double* wr = Rc + wr_off;
    int done = 0;
    while (np - done >= 4) {
      ltv_passes<NS, 4>(rd + done * PASS, wr + done * PASS, a, worker);
      done += 4;
    }
    switch (np - done) {
      case 3: ltv_passes<NS, 3>(rd + done * PASS, wr + done * PASS, a, worker); break;
      case 2: ltv_passes<NS, 2>(rd + done * PASS, wr + done * PASS, a, worker); break;
      case 1: ltv_passes<NS, 1>(rd + done * PASS, wr + done * PASS, a, worker); break;
      default: break;
    }
    if (lane < nm) Rc[bdiag + k * n] = bk;
    // row k-1 (complete, zeros right of the diagonal included) goes to HBM meanwhile
    if (k > 0) {
      if (fast) {
        if (lane < nn) *sdst = sv;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (64 * u <= last) udst[64 * u] = v[u];
      } else {
        stream_row(Rp, k - 1);
      }
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) a[t] = an[t];
    bk = bn;
    cur ^= 1;
    sdst += nn;
    udst += rl2;
  }
  stream_row(Row + (cur ^ 1) * rstep, N - 1);
}

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return allow_whole_lds(reinterpret_cast<const void*>(kernel));
}

}  // namespace

int launch_fill_su(const double* A, const double* B, double* S, double* U, int batch, int N, int n,
                   int m, int ltv, hipStream_t stream, hipError_t* err) {
  *err = hipSuccess;
  constexpr size_t LDS_MAX = 160 * 1024;
  const bool aligned16 = (((uintptr_t)S | (uintptr_t)U) & 15) == 0;
  const bool pairs = ((N * n) & 1) == 0 && aligned16;   // rows of U are whole 16-byte words
  if (!ltv && n <= 4 && m + n <= 16 && pairs && ((N * n * n) & 1) == 0) {
    // few states: registers + DPP recurrence, constant-address write loop
    int spw = 16 / (m + n);
    while (spw > 1 && quad_lds_doubles(N, n, m, spw) * sizeof(double) > 40 * 1024) --spw;
    // (fewer systems per wavefront while the launch has fewer wavefronts than this: 8192 systems
    // as 8192 wavefronts of one reach 0.66 of HBM, as 2048 of four 0.59; tuning aid: the variable)
    static const int SPW_MIN_WAVES = getenv("MPCASM_FILL_MIN_WAVES") ? atoi(getenv("MPCASM_FILL_MIN_WAVES")) : 8192;
    while (spw > 1 && (batch + spw - 1) / spw < SPW_MIN_WAVES) --spw;
    const size_t bytes = quad_lds_doubles(N, n, m, spw) * sizeof(double);
    if (bytes <= LDS_MAX) {
      auto kernel = n == 1   ? fill_lti_quad_kernel<1>
                    : n == 2 ? fill_lti_quad_kernel<2>
                    : n == 3 ? fill_lti_quad_kernel<3>
                             : fill_lti_quad_kernel<4>;
      if ((*err = allow_lds(kernel, bytes)) != hipSuccess) return MPCASM_ERR_HIP;
      int lshift = 0;
      while (lshift < 6 && (1 << lshift) < (N * n) / 2) ++lshift;
      hipLaunchKernelGGL(kernel, dim3((batch + spw - 1) / spw), dim3(64), bytes, stream, A, B, S, U,
                         batch, N, m, spw, lshift, (N * n) % 16 == 0 ? 1 : 0);
      *err = hipGetLastError();
      return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
    }
  }
  if (ltv && n <= 4 && n * m <= 64 && pairs) {
    const size_t bytes = (n == 1   ? LtvRow<1>::lds_doubles(N, m)
                          : n == 2 ? LtvRow<2>::lds_doubles(N, m)
                          : n == 3 ? LtvRow<3>::lds_doubles(N, m)
                                   : LtvRow<4>::lds_doubles(N, m)) * sizeof(double);
    if (bytes <= 64 * 1024) {
      auto kernel = n == 1   ? fill_ltv_row_kernel<1>
                    : n == 2 ? fill_ltv_row_kernel<2>
                    : n == 3 ? fill_ltv_row_kernel<3>
                             : fill_ltv_row_kernel<4>;
      hipLaunchKernelGGL(kernel, dim3(batch), dim3(64), bytes, stream, A, B, S, U, N, m);
      *err = hipGetLastError();
      return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
    }
  }
  if (!ltv) {
    const size_t per = lti_lds_doubles(N, n, m) * sizeof(double);
    const int xsz = n * (m + n);
    // one wavefront per system when it fits comfortably, else one workgroup
    const bool small = xsz <= 64 * EPT && per * 4 <= 48 * 1024;
    // systems per wavefront of the tiny kernel: as many as fit its lanes, but fewer when
    // the batch is too small to give every CU a couple of workgroups
    int spw = xsz <= 32 ? 64 / xsz : 0;
    while (spw > 2 && (batch + 4 * spw - 1) / (4 * spw) < 512) --spw;
    const size_t tiny = tiny_lds_doubles(N, n, m) * sizeof(double) * 4 * (spw > 0 ? spw : 1);
    if (spw >= 2 && tiny <= 64 * 1024) {
      const int per_block = 4 * spw;
      const int blocks = (batch + per_block - 1) / per_block;
      hipLaunchKernelGGL(fill_lti_tiny_kernel, dim3(blocks), dim3(BLOCK), tiny, stream, A, B, S, U,
                         batch, N, n, m, spw);
    } else if (small) {
      const size_t bytes = per * 4;
      const int blocks = (batch + 3) / 4;
      hipLaunchKernelGGL((fill_lti_kernel<64, false>), dim3(blocks), dim3(BLOCK), bytes, stream, A,
                         B, S, U, batch, N, n, m);
    } else if (per > LDS_MAX) {
      return MPCASM_ERR_LIMIT;  // the A^d B table of one system must fit in LDS
    } else if (xsz <= BLOCK * EPT) {
      const size_t padded = per + (size_t)m * N * n * sizeof(double);
      if (pairs && padded <= 78 * 1024) {  // (two workgroups still share a CU)
        if ((*err = allow_lds(fill_lti_kernel<BLOCK, false, true>, padded)) != hipSuccess)
          return MPCASM_ERR_HIP;
        hipLaunchKernelGGL((fill_lti_kernel<BLOCK, false, true>), dim3(batch), dim3(BLOCK), padded,
                           stream, A, B, S, U, batch, N, n, m);
        *err = hipGetLastError();
        return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
      }
      if ((*err = allow_lds(fill_lti_kernel<BLOCK, false>, per)) != hipSuccess)
        return MPCASM_ERR_HIP;
      hipLaunchKernelGGL((fill_lti_kernel<BLOCK, false>), dim3(batch), dim3(BLOCK), per, stream, A,
                         B, S, U, batch, N, n, m);
    } else {
      if ((*err = allow_lds(fill_lti_kernel<BLOCK, true>, per)) != hipSuccess)
        return MPCASM_ERR_HIP;
      hipLaunchKernelGGL((fill_lti_kernel<BLOCK, true>), dim3(batch), dim3(BLOCK), per, stream, A,
                         B, S, U, batch, N, n, m);
    }
  } else {
    const size_t per = ltv_lds_doubles(N, n, m) * sizeof(double);
    const bool small = n <= 64 && per * 4 <= 64 * 1024 && N * n <= 1024;
    const size_t wper = ltv_wave_lds_doubles(N, n, m) * sizeof(double) * 4;
    const size_t wall = ltv_wave_lds_doubles(N, n, m, true) * sizeof(double) * 4;
    const size_t bper = ltv_block_lds_doubles(N, n, m) * sizeof(double);
    if (n <= BLOCK && bper <= 20 * 1024) {
      // a workgroup per system, all step matrices resident, eight workgroups per CU
      auto kernel = n == 2   ? fill_ltv_block_kernel<2>
                    : n == 3 ? fill_ltv_block_kernel<3>
                    : n == 4 ? fill_ltv_block_kernel<4>
                             : fill_ltv_block_kernel<0>;
      hipLaunchKernelGGL(kernel, dim3(batch), dim3(BLOCK), bper, stream, A, B, S, U, N, n, m);
    } else if (n * n <= 128 && n * m <= 64 && wall <= 80 * 1024 && batch < 8192) {
      // every step's (A_k, B_k) resident (two workgroups per CU still fit): measured faster
      // while the batch leaves the CUs a single round of workgroups (C5: 0.43 against 0.35
      // of HBM peak at 2 048 systems), slower beyond (0.43 against 0.53 at 16 384)
      if ((*err = allow_lds(fill_ltv_wave_kernel<true>, wall)) != hipSuccess) return MPCASM_ERR_HIP;
      const int blocks = (batch + 3) / 4;
      hipLaunchKernelGGL(fill_ltv_wave_kernel<true>, dim3(blocks), dim3(BLOCK), wall, stream, A, B,
                         S, U, batch, N, n, m);
    } else if (n * n <= 128 && n * m <= 64 && wper <= 64 * 1024) {
      const int blocks = (batch + 3) / 4;
      hipLaunchKernelGGL(fill_ltv_wave_kernel<false>, dim3(blocks), dim3(BLOCK), wper, stream, A, B,
                         S, U, batch, N, n, m);
    } else if (small) {
      const size_t bytes = per * 4;
      const int blocks = (batch + 3) / 4;
      hipLaunchKernelGGL(fill_ltv_kernel<64>, dim3(blocks), dim3(BLOCK), bytes, stream, A, B, S, U,
                         batch, N, n, m);
    } else {
      if (n > BLOCK || per > LDS_MAX) return MPCASM_ERR_LIMIT;
      if ((*err = allow_lds(fill_ltv_kernel<BLOCK>, per)) != hipSuccess) return MPCASM_ERR_HIP;
      hipLaunchKernelGGL(fill_ltv_kernel<BLOCK>, dim3(batch), dim3(BLOCK), per, stream, A, B, S, U,
                         batch, N, n, m);
    }
  }
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

}  // namespace mpcasm
