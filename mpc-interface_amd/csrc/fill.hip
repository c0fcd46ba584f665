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
    if (small) {
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
    if (small) {
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
