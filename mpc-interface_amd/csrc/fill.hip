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

template <int TPI, bool GENERIC>
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
  double* R = lds + (size_t)slot * lti_lds_doubles(N, n, m);  // [m][rl] block-reversed A^d B
  double* X = R + even_up((size_t)m * rl);                    // [2][xw][n] column-major
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

  if (live) {
    for (int e = tid; e < n * n; e += TPI) Am[e] = Ab[e];
    for_each_element([&](int e, int c, int i) {
      if (c < m) {
        const double v = Bb[i * m + c];
        X[e] = v;
        R[(size_t)c * rl + (size_t)(N - 1) * n + i] = v;  // d = 0
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
          R[(size_t)c * rl + (size_t)(N - 1 - d) * n + i] = v;
        else
          Sb[(size_t)d * n * n + (e - n * m)] = v;
      });
    }
    __syncthreads();
  }

  if (!live) return;

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

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

int launch_fill_su(const double* A, const double* B, double* S, double* U, int batch, int N, int n,
                   int m, int ltv, hipStream_t stream, hipError_t* err) {
  *err = hipSuccess;
  constexpr size_t LDS_MAX = 160 * 1024;
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
