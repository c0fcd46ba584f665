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
// One wavefront per instance, everything in LDS: the matrix K = P + sigma I + rho G'G is factored once
// (Cholesky, in place), inverted once (L^-1 column by column, every lane its own right-hand side; then
// K^-1 = L^-T L^-1), and an iteration is three matrix-vector products -- G'v (lanes over the unknowns),
// K^-1 r (the same) and G xt (lanes over the rows) -- with no dependent chain longer than a row: the
// two triangular solves per iteration that OSQP's LDL' does on the host would be 2 no dependent steps
// of a wavefront each.  K is symmetric positive definite by construction (sigma > 0), well conditioned
// for the steps OSQP uses; the explicit inverse changes the iterates by rounding only.
// LDS per instance: no (no|1) doubles for K^-1, max(nc, no) (no|1) for G (and, before G is needed,
// L^-1), nine vectors: 34 KB for the biped at N = 16 (no = 36, nc = 76): four instances per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_prims.h"
#include "kernels.h"

namespace mpcasm {

namespace {

constexpr int ADMM_BLOCK = 64;

struct AdmmLds {
  int mi, gs, x, xt, rhs, q, z, y, v, h, total, ld;
};
__host__ __device__ inline AdmmLds admm_lds(int no, int nc) {
  AdmmLds L;
  L.ld = no | 1;   // (an odd leading dimension: a column of a row-major matrix is conflict-free)
  L.mi = 0;
  L.gs = L.mi + no * L.ld;
  L.x = L.gs + (nc > no ? nc : no) * L.ld;
  L.xt = L.x + no;
  L.rhs = L.xt + no;
  L.q = L.rhs + no;
  L.z = L.q + no;
  L.y = L.z + nc;
  L.v = L.y + nc;
  L.h = L.v + nc;
  L.total = L.h + nc;
  L.total += L.total & 1;
  return L;
}

// sum_i a[i sa] b[i sb], i < n: four sums side by side -- one wavefront per SIMD has nothing to hide an LDS
// read behind but its own other reads, so the loop must not be one dependent chain of read -> fma
__device__ __forceinline__ double dot4(const double* a, int sa, const double* b, int sb, int n, double init) {
  double s0 = init, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = 0;
  for (; i + 8 <= n; i += 8) {   // (eight pairs of reads in flight)
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

__global__ __launch_bounds__(ADMM_BLOCK) void admm_kernel(
    int no, int nc, const double* __restrict__ P, const double* __restrict__ q,
    const double* __restrict__ G, const double* __restrict__ h, double* __restrict__ X,
    double* __restrict__ Y, double* __restrict__ Z, double* __restrict__ res, double rho, double sigma,
    double alpha, int iters, int warm, int batch, double* __restrict__ Kinv, int kinv_valid) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  const long inst = blockIdx.x;
  if (inst >= batch) return;
  const AdmmLds L = admm_lds(no, nc);
  const int ld = L.ld;
  double* Mi = sm + L.mi;   // K, then its Cholesky factor (lower), then K^-1
  double* Gs = sm + L.gs;   // G [nc][ld]; in between: T = L^-1 [no][ld]
  double* xs = sm + L.x;
  double* xt = sm + L.xt;
  double* rhs = sm + L.rhs;
  double* qs = sm + L.q;
  double* zs = sm + L.z;
  double* ys = sm + L.y;
  double* vs = sm + L.v;
  double* hs = sm + L.h;
  const double* Pb = P + (size_t)inst * no * no;
  const double* Gb = G + (size_t)inst * nc * no;
  auto load_g = [&]() {
    for (int e = lane; e < nc * no; e += ADMM_BLOCK) {
      const int r = e / no, c = e - r * no;
      Gs[r * ld + c] = Gb[e];
    }
  };

  // ---- K = P + sigma I + rho G'G ------------------------------------------------------------------
  // (a caller whose P and G did not change since the last call -- the same model and structure, a new
  // `given` -- hands K^-1 back in: no factorisation, the larger part of a call of 25 iterations)
  const bool reuse = Kinv != nullptr && kinv_valid != 0;
  load_g();
  for (int e = lane; e < no; e += ADMM_BLOCK) {
    qs[e] = q[(size_t)inst * no + e];
    xs[e] = warm ? X[(size_t)inst * no + e] : 0.0;
  }
  for (int e = lane; e < nc; e += ADMM_BLOCK) {
    hs[e] = h[(size_t)inst * nc + e];
    ys[e] = warm ? Y[(size_t)inst * nc + e] : 0.0;
    zs[e] = warm ? Z[(size_t)inst * nc + e] : 0.0;
  }
  __syncthreads();
  for (int e = lane; e < no * no; e += ADMM_BLOCK) {
    const int a = e / no, b = e - a * no;
    if (reuse) {
      Mi[a * ld + b] = Kinv[(size_t)inst * no * no + e];
      continue;
    }
    const double acc = dot4(Gs + a, ld, Gs + b, ld, nc, 0.0);
    Mi[a * ld + b] = fma(rho, acc, Pb[e]) + (a == b ? sigma : 0.0);
  }
  if (!warm) {   // (a cold start: z = min(G x, h) with x = 0)
    for (int e = lane; e < nc; e += ADMM_BLOCK) zs[e] = fmin(0.0, hs[e]);
  }
  __syncthreads();

  // ---- Cholesky, in place (the lower triangle): K = L L' ------------------------------------------
  bool ok = true;
  if (!reuse) {
  for (int k = 0; k < no; ++k) {
    const double dkk = Mi[k * ld + k];
    ok = ok && dkk > 0.0;
    const double d = sqrt(dkk > 0.0 ? dkk : 1.0);
    __syncthreads();
    for (int i = k + lane; i < no; i += ADMM_BLOCK) Mi[i * ld + k] = i == k ? d : Mi[i * ld + k] / d;
    __syncthreads();
    // (rows of the trailing block in turn, lanes over the columns j <= i of a row: no division; a row
    // of at most 64 columns is one step, and the rows do not depend on each other)
    // (the lane's own L[j][k] once per column k, for the first 64 columns behind k; four rows on their way at a time)
    const int j0 = k + 1 + lane;
    const double ljk = j0 < no ? Mi[j0 * ld + k] : 0.0;
    int i = k + 1;
    for (; i + 4 <= no; i += 4) {
      double lik[4], mij[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        lik[u] = Mi[(i + u) * ld + k];
        mij[u] = j0 <= i + u ? Mi[(i + u) * ld + j0] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 <= i + u) Mi[(i + u) * ld + j0] = fma(-lik[u], ljk, mij[u]);
    }
    for (; i < no; ++i)
      if (j0 <= i) Mi[i * ld + j0] = fma(-Mi[i * ld + k], ljk, Mi[i * ld + j0]);
    for (int ii = k + 1; ii < no; ++ii) {   // (columns beyond the first 64: more unknowns than lanes)
      const double lik = Mi[ii * ld + k];
      for (int j = j0 + ADMM_BLOCK; j <= ii; j += ADMM_BLOCK) Mi[ii * ld + j] = fma(-lik, Mi[j * ld + k], Mi[ii * ld + j]);
    }
    __syncthreads();
  }
  // ---- T = L^-1: lane c solves L t = e_c (rows above c stay zero) -----------------------------------
  double* T = Gs;
  for (int c = lane; c < no; c += ADMM_BLOCK)
    for (int i = 0; i < no; ++i) {
      const double s = (i == c ? 1.0 : 0.0) - (i > c ? dot4(Mi + i * ld + c, 1, T + c * ld + c, ld, i - c, 0.0) : 0.0);
      T[i * ld + c] = i < c ? 0.0 : s / Mi[i * ld + i];
    }
  __syncthreads();
  // ---- K^-1 = T' T ---------------------------------------------------------------------------------------
  for (int e = lane; e < no * no; e += ADMM_BLOCK) {
    const int a = e / no, b = e - a * no;
    const int i0 = a > b ? a : b;
    Mi[a * ld + b] = dot4(T + i0 * ld + a, ld, T + i0 * ld + b, ld, no - i0, 0.0);
  }
  __syncthreads();
  if (Kinv != nullptr)
    for (int e = lane; e < no * no; e += ADMM_BLOCK) Kinv[(size_t)inst * no * no + e] = ok ? Mi[(e / no) * ld + e % no] : __builtin_nan("");
  }
  if (!reuse) load_g();   // (the factorisation used G's place for L^-1)
  __syncthreads();

  // ---- the iterations -------------------------------------------------------------------------------------
  const double inv_rho = 1.0 / rho;
  for (int it = 0; it < iters; ++it) {
    for (int r = lane; r < nc; r += ADMM_BLOCK) vs[r] = fma(rho, zs[r], -ys[r]);
    __syncthreads();
    for (int c = lane; c < no; c += ADMM_BLOCK) {
      rhs[c] = dot4(Gs + c, ld, vs, 1, nc, fma(sigma, xs[c], -qs[c]));
    }
    __syncthreads();
    for (int c = lane; c < no; c += ADMM_BLOCK) {
      xt[c] = dot4(Mi + c * ld, 1, rhs, 1, no, 0.0);
    }
    __syncthreads();
    for (int r = lane; r < nc; r += ADMM_BLOCK) {
      const double acc = dot4(Gs + r * ld, 1, xt, 1, no, 0.0);
      const double zr = fma(alpha, acc, (1.0 - alpha) * zs[r]);
      const double zn = fmin(fma(ys[r], inv_rho, zr), hs[r]);
      ys[r] = fma(rho, zr - zn, ys[r]);
      zs[r] = zn;
    }
    for (int c = lane; c < no; c += ADMM_BLOCK) xs[c] = fma(alpha, xt[c], (1.0 - alpha) * xs[c]);
    __syncthreads();
  }

  // ---- results; OSQP's residuals |Gx - z|_inf, |Px + q + G'y|_inf -------------------------------
  const double bad = __builtin_nan("");
  for (int c = lane; c < no; c += ADMM_BLOCK) X[(size_t)inst * no + c] = ok ? xs[c] : bad;
  for (int r = lane; r < nc; r += ADMM_BLOCK) {
    Y[(size_t)inst * nc + r] = ok ? ys[r] : bad;
    Z[(size_t)inst * nc + r] = ok ? zs[r] : bad;
  }
  if (res != nullptr) {
    double rp = 0.0, rd = 0.0;
    for (int r = lane; r < nc; r += ADMM_BLOCK) {
      double acc = 0.0;
      for (int c = 0; c < no; ++c) acc = fma(Gs[r * ld + c], xs[c], acc);
      rp = fmax(rp, fabs(acc - zs[r]));
    }
    for (int c = lane; c < no; c += ADMM_BLOCK) {
      double acc = qs[c];
      for (int b = 0; b < no; ++b) acc = fma(Pb[(size_t)c * no + b], xs[b], acc);
      for (int r = 0; r < nc; ++r) acc = fma(Gs[r * ld + c], ys[r], acc);
      rd = fmax(rd, fabs(acc));
    }
    for (int off = 32; off > 0; off >>= 1) {
      rp = fmax(rp, __shfl_xor(rp, off, 64));
      rd = fmax(rd, __shfl_xor(rd, off, 64));
    }
    if (lane == 0) {
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
