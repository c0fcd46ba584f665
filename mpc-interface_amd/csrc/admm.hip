// admm.hip -- K5: the step after the assembly, on the device (the "or"): a batch
// of dense QPs  min 1/2 x'Px + q'x  s.t.  Gx <= h  straight out of mpcasm_assemble's buffers, iterated
// with OSQP's ADMM -- the solve the reference's loop hands to qpsolvers.osqp_solve_qp
// (use_examples/simple_functional_example/biped_mpc_loop.py:57-60).  osqp is a third-party dependency
// of the example, absent from the reference tree; the iteration is the published one (Stellato,
// Banjac, Goulart, Bemporad, Boyd, Math. Prog. Comp. 12 (2020), Algorithm 1, reduced KKT form):
//     (P + sigma I + rho G'G) xt = sigma x - q + G'(rho z - y);   zt = G xt
//     x+ = alpha xt + (1 - alpha) x
//     z+ = min(alpha zt + (1 - alpha) z + y / rho, h);   y+ = y + rho (alpha zt + (1 - alpha) z - z+)
// restated on the CPU by oracle/admm_oracle.py, which the tests hold this kernel to.
//
// One workgroup of four wavefronts per instance, everything in LDS: the matrix K = P + sigma I + rho G'G is
// factored once (Cholesky, in place), inverted once (L^-1 column by column, every thread its own right-hand
// side; then K^-1 = L^-T L^-1), and an iteration is three matrix-vector products -- G'v, K^-1 r and G xt --
// with no dependent chain longer than a row: the two triangular solves per iteration that OSQP's LDL' does on
// the host would be 2 no dependent steps of a wavefront each.  K is symmetric positive definite by
// construction (sigma > 0), well conditioned for the steps OSQP uses; the explicit inverse changes the
// iterates by rounding only.  LDS per instance: no (no|1) doubles for K^-1, max(nc, no) (no|1) for G (and,
// before G is needed, L^-1), the vectors and the partial sums: 40 KB for the biped at N = 16 (no = 36,
// nc = 76): four instances per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_prims.h"
#include "kernels.h"

namespace mpcasm {

namespace {

constexpr int ADMM_BLOCK = 256;
constexpr int ADMM_WAVES = ADMM_BLOCK / 64;

struct AdmmLds {
  int mi, gs, x, q, z, y, h, v, r, xt, pa, pb, total, ld, lp;
};
__host__ __device__ inline AdmmLds admm_lds(int no, int nc) {
  AdmmLds L;
  L.ld = no | 1;   // (an odd leading dimension: a column of a row-major matrix is conflict-free)
  L.lp = no > nc ? no : nc;
  L.mi = 0;
  L.gs = L.mi + no * L.ld;
  L.x = L.gs + (nc > no ? nc : no) * L.ld;
  L.q = L.x + no;
  L.z = L.q + no;
  L.y = L.z + nc;
  L.h = L.y + nc;
  L.v = L.h + nc;                     // rho z - y
  L.r = L.v + nc;                     // the right-hand side
  L.xt = L.r + no;
  L.pa = L.xt + no;                   // [ADMM_WAVES][lp] partial sums of G'v, then of G xt
  L.pb = L.pa + ADMM_WAVES * L.lp;    // [ADMM_WAVES][no] partial sums of K^-1 r
  L.total = L.pb + ADMM_WAVES * no;
  L.total += L.total & 1;
  return L;
}

// sum_i a[i sa] b[i sb], i < n: four sums side by side, eight pairs of reads in flight -- the loop must
// not be one dependent chain of read -> fma
__device__ __forceinline__ double dot4(const double* a, int sa, const double* b, int sb, int n, double init) {
  double s0 = init, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    double av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) av[u] = a[(i + u) * sa], bv[u] = b[(i + u) * sb];
    s0 = fma(av[0], bv[0], s0);
    s1 = fma(av[1], bv[1], s1);
    s2 = fma(av[2], bv[2], s2);
    s3 = fma(av[3], bv[3], s3);
    s0 = fma(av[4], bv[4], s0);
    s1 = fma(av[5], bv[5], s1);
    s2 = fma(av[6], bv[6], s2);
    s3 = fma(av[7], bv[7], s3);
  }
  for (; i + 4 <= n; i += 4) {
    const double a0 = a[i * sa], a1 = a[(i + 1) * sa], a2 = a[(i + 2) * sa], a3 = a[(i + 3) * sa];
    const double b0 = b[i * sb], b1 = b[(i + 1) * sb], b2 = b[(i + 2) * sb], b3 = b[(i + 3) * sb];
    s0 = fma(a0, b0, s0);
    s1 = fma(a1, b1, s1);
    s2 = fma(a2, b2, s2);
    s3 = fma(a3, b3, s3);
  }
  for (; i < n; ++i) s0 = fma(a[i * sa], b[i * sb], s0);
  return (s0 + s1) + (s2 + s3);
}

// One workgroup of four wavefronts per instance.  Everything that is a sum over rows or columns is cut in
// four -- wavefront w takes the terms i = w, w + 4, ... -- and the partial sums meet in LDS behind a barrier:
// one wavefront per instance (the first version) had nothing to hide an LDS round trip behind and spent
// 10 000 cycles per iteration on three matrix-vector products of 36 x 76.
__global__ __launch_bounds__(ADMM_BLOCK) void admm_kernel(
    int no, int nc, const double* __restrict__ P, const double* __restrict__ q,
    const double* __restrict__ G, const double* __restrict__ h, double* __restrict__ X,
    double* __restrict__ Y, double* __restrict__ Z, double* __restrict__ res, double rho, double sigma,
    double alpha, int iters, int warm, int batch, double* __restrict__ Kinv, int kinv_valid) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long inst = blockIdx.x;
  if (inst >= batch) return;
  const AdmmLds L = admm_lds(no, nc);
  const int ld = L.ld, lp = L.lp;
  double* Mi = sm + L.mi;   // K, then its Cholesky factor (lower), then K^-1
  double* Gs = sm + L.gs;   // G [nc][ld]; in between: T = L^-1 [no][ld]
  double* xs = sm + L.x;
  double* qs = sm + L.q;
  double* zs = sm + L.z;
  double* ys = sm + L.y;
  double* hs = sm + L.h;
  double* vs = sm + L.v;
  double* rv = sm + L.r;
  double* xt = sm + L.xt;
  double* pa = sm + L.pa;
  double* pbuf = sm + L.pb;
  const double* Pb = P + (size_t)inst * no * no;
  const double* Gb = G + (size_t)inst * nc * no;
  auto load_g = [&]() {
    for (int e = tid; e < nc * no; e += ADMM_BLOCK) {
      const int r = e / no, c = e - r * no;
      Gs[r * ld + c] = Gb[e];
    }
  };

  // ---- K = P + sigma I + rho G'G ------------------------------------------------------------------
  // (a caller whose P and G did not change since the last call -- the same model and structure, a new
  // `given` -- hands K^-1 back in: no factorisation, the larger part of a call of 25 iterations)
  const bool reuse = Kinv != nullptr && kinv_valid != 0;
  load_g();
  for (int e = tid; e < no; e += ADMM_BLOCK) {
    qs[e] = q[(size_t)inst * no + e];
    xs[e] = warm ? X[(size_t)inst * no + e] : 0.0;
  }
  for (int e = tid; e < nc; e += ADMM_BLOCK) {
    const double hv = h[(size_t)inst * nc + e];
    hs[e] = hv;
    ys[e] = warm ? Y[(size_t)inst * nc + e] : 0.0;
    zs[e] = warm ? Z[(size_t)inst * nc + e] : fmin(0.0, hv);   // (a cold start: z = min(G x, h) with x = 0)
  }
  __syncthreads();
  for (int e = tid; e < no * no; e += ADMM_BLOCK) {
    const int a = e / no, b = e - a * no;
    if (reuse) {
      Mi[a * ld + b] = Kinv[(size_t)inst * no * no + e];
      continue;
    }
    const double acc = dot4(Gs + a, ld, Gs + b, ld, nc, 0.0);
    Mi[a * ld + b] = fma(rho, acc, Pb[e]) + (a == b ? sigma : 0.0);
  }
  __syncthreads();

  bool ok = true;
  if (!reuse) {
    // ---- Cholesky, in place (the lower triangle): K = L L' ----------------------------------------
    for (int k = 0; k < no; ++k) {
      const double dkk = Mi[k * ld + k];
      ok = ok && dkk > 0.0;
      const double d = sqrt(dkk > 0.0 ? dkk : 1.0);
      __syncthreads();
      for (int i = k + tid; i < no; i += ADMM_BLOCK) Mi[i * ld + k] = i == k ? d : Mi[i * ld + k] / d;
      __syncthreads();
      // the trailing block: wavefront w takes the rows k + 1 + w, + 4, ..., a lane the column k + 1 + lane
      // (+ 64, ... for more unknowns than lanes) up to the diagonal
      for (int j = k + 1 + lane; j < no; j += 64) {
        const double ljk = Mi[j * ld + k];
        for (int i = k + 1 + wave + (j > k + 1 + wave ? ((j - k - 1 - wave + ADMM_WAVES - 1) / ADMM_WAVES) * ADMM_WAVES : 0);
             i < no; i += ADMM_WAVES)
          Mi[i * ld + j] = fma(-Mi[i * ld + k], ljk, Mi[i * ld + j]);
      }
      __syncthreads();
    }
    // ---- T = L^-1: thread c solves L t = e_c (rows above c stay zero) --------------------------------
    double* T = Gs;
    for (int c = tid; c < no; c += ADMM_BLOCK)
      for (int i = 0; i < no; ++i) {
        const double sv = (i == c ? 1.0 : 0.0) - (i > c ? dot4(Mi + i * ld + c, 1, T + c * ld + c, ld, i - c, 0.0) : 0.0);
        T[i * ld + c] = i < c ? 0.0 : sv / Mi[i * ld + i];
      }
    __syncthreads();
    // ---- K^-1 = T' T ---------------------------------------------------------------------------------
    for (int e = tid; e < no * no; e += ADMM_BLOCK) {
      const int a = e / no, b = e - a * no;
      const int i0 = a > b ? a : b;
      Mi[a * ld + b] = dot4(T + i0 * ld + a, ld, T + i0 * ld + b, ld, no - i0, 0.0);
    }
    __syncthreads();
    if (Kinv != nullptr)
      for (int e = tid; e < no * no; e += ADMM_BLOCK)
        Kinv[(size_t)inst * no * no + e] = ok ? Mi[(e / no) * ld + e % no] : __builtin_nan("");
    load_g();   // (the factorisation used G's place for L^-1)
    __syncthreads();
  }

  // ---- the iterations: six short phases, a barrier behind each; every sum a strided inner product with
  // several reads in flight (dot4) -------------------------------------------------------------------------
  const double inv_rho = 1.0 / rho;
  // wavefront w's share of a sum over `count` terms: i = w, w + 4, ...
  const int rows_w = nc > wave ? (nc - wave + ADMM_WAVES - 1) / ADMM_WAVES : 0;
  const int cols_w = no > wave ? (no - wave + ADMM_WAVES - 1) / ADMM_WAVES : 0;
  for (int r = tid; r < nc; r += ADMM_BLOCK) vs[r] = fma(rho, zs[r], -ys[r]);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    // pa[w][c] = sum over this wavefront's rows of G[r][c] v[r]
    for (int c = lane; c < no; c += 64)
      pa[wave * lp + c] = dot4(Gs + wave * ld + c, ADMM_WAVES * ld, vs + wave, ADMM_WAVES, rows_w, 0.0);
    __syncthreads();
    for (int b = tid; b < no; b += ADMM_BLOCK)
      rv[b] = fma(sigma, xs[b], -qs[b]) + ((pa[b] + pa[lp + b]) + (pa[2 * lp + b] + pa[3 * lp + b]));
    __syncthreads();
    // pb[w][c] = sum over this wavefront's b of K^-1[c][b] r[b]
    for (int c = lane; c < no; c += 64)
      pbuf[wave * no + c] = dot4(Mi + c * ld + wave, ADMM_WAVES, rv + wave, ADMM_WAVES, cols_w, 0.0);
    __syncthreads();
    for (int c = tid; c < no; c += ADMM_BLOCK) {
      const double xtc = (pbuf[c] + pbuf[no + c]) + (pbuf[2 * no + c] + pbuf[3 * no + c]);
      xt[c] = xtc;
      xs[c] = fma(alpha, xtc, (1.0 - alpha) * xs[c]);
    }
    __syncthreads();
    // pa[w][r] = sum over this wavefront's c of G[r][c] xt[c]
    for (int r = lane; r < nc; r += 64)
      pa[wave * lp + r] = dot4(Gs + r * ld + wave, ADMM_WAVES, xt + wave, ADMM_WAVES, cols_w, 0.0);
    __syncthreads();
    for (int r = tid; r < nc; r += ADMM_BLOCK) {
      const double zt = (pa[r] + pa[lp + r]) + (pa[2 * lp + r] + pa[3 * lp + r]);
      const double zr = fma(alpha, zt, (1.0 - alpha) * zs[r]);
      const double zn = fmin(fma(ys[r], inv_rho, zr), hs[r]);
      const double yn = fma(rho, zr - zn, ys[r]);
      ys[r] = yn;
      zs[r] = zn;
      vs[r] = fma(rho, zn, -yn);
    }
    __syncthreads();
  }

  // ---- results; OSQP's residuals |Gx - z|_inf, |Px + q + G'y|_inf -------------------------------
  const double bad = __builtin_nan("");
  for (int c = tid; c < no; c += ADMM_BLOCK) X[(size_t)inst * no + c] = ok ? xs[c] : bad;
  for (int r = tid; r < nc; r += ADMM_BLOCK) {
    Y[(size_t)inst * nc + r] = ok ? ys[r] : bad;
    Z[(size_t)inst * nc + r] = ok ? zs[r] : bad;
  }
  if (res != nullptr) {
    double rp = 0.0, rd = 0.0;
    for (int r = tid; r < nc; r += ADMM_BLOCK) rp = fmax(rp, fabs(dot4(Gs + r * ld, 1, xs, 1, no, 0.0) - zs[r]));
    for (int c = tid; c < no; c += ADMM_BLOCK) {
      double acc = qs[c];
      for (int b = 0; b < no; ++b) acc = fma(Pb[(size_t)c * no + b], xs[b], acc);
      rd = fmax(rd, fabs(dot4(Gs + c, ld, ys, 1, nc, acc)));
    }
    for (int off = 32; off > 0; off >>= 1) {
      rp = fmax(rp, __shfl_xor(rp, off, 64));
      rd = fmax(rd, __shfl_xor(rd, off, 64));
    }
    if (lane == 0) {
      pbuf[wave] = rp;
      pbuf[ADMM_WAVES + wave] = rd;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < ADMM_WAVES; ++w) {
        rp = fmax(rp, pbuf[w]);
        rd = fmax(rd, pbuf[ADMM_WAVES + w]);
      }
      res[inst * 2 + 0] = ok ? rp : bad;
      res[inst * 2 + 1] = ok ? rd : bad;
    }
  }
}

}  // namespace

size_t admm_lds_bytes(int no, int nc) { return (size_t)admm_lds(no, nc).total * sizeof(double); }

int launch_admm(int no, int nc, const double* P, const double* q, const double* G, const double* h,
                double* x, double* y, double* z, double* res, double rho, double sigma, double alpha,
                int iters, int warm, int batch, double* kinv, int kinv_valid, hipStream_t stream,
                hipError_t* err) {
  const size_t lds = admm_lds_bytes(no, nc);
  if (lds > (size_t)RESIDENT_LDS_LIMIT) return MPCASM_ERR_LIMIT;
  if (lds > 64 * 1024) {
    *err = allow_whole_lds(reinterpret_cast<const void*>(admm_kernel));
    if (*err != hipSuccess) return MPCASM_ERR_HIP;
  }
  hipLaunchKernelGGL(admm_kernel, dim3((unsigned)batch), dim3(ADMM_BLOCK), lds, stream, no, nc, P, q, G,
                     h, x, y, z, res, rho, sigma, alpha, iters, warm, batch, kinv, kinv_valid);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

}  // namespace mpcasm
