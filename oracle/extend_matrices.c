/* extend_matrices.c -- plain C restatement of the horizon extension  --  TEST
 * INFRASTRUCTURE ONLY (the compiled CPU twin of oracle/qp_oracle.py:extend_matrices).
 *
 * Follows the reference's recurrence, python/mpc_interface/tools.py:14-33 (C++ twin
 * cpp/src/tools.cc:83-144): the block column [B | A] is left-multiplied by A once per
 * horizon step (no repeated squaring), with the same summation order t = 0..n-1 as a
 * row-times-column dot product:
 *     S[k][j][i]     = (A^{k+1})[i][j]
 *     U[j][k][l][i]  = (A^{k-l} B)[i][j]   for l <= k, else 0
 * Layouts as include/mpcasm.h documents them for mpcasm_fill_su (one system, row-major).
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this.
 *
 * Pinned: tests/test_oracle_golden.py checks it against the golden vectors of the real
 * reference (fixture g1_extend, incl. the reference's own LIP_matrices) and against the
 * numpy oracle.  The LTV variant has no reference counterpart: parity unpinned beyond the
 * degenerate LTI case.
 */
#include <stdlib.h>
#include <string.h>

/* x: n x w, y = a (n x n) . x */
static void left_multiply(const double* a, const double* x, double* y, int n, int w) {
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < w; ++c) {
      double acc = 0.0;
      for (int t = 0; t < n; ++t) acc += a[i * n + t] * x[t * w + c];
      y[i * w + c] = acc;
    }
}

/* ltv == 0: A [n][n], B [n][m];  ltv == 1: A [N][n][n], B [N][n][m] (x_{k+1} = A_k x_k + B_k u_k).
 * S [N][n][n], U [m][N][N][n].  Returns 0, or -1 when out of memory. */
int oracle_extend_matrices(const double* A, const double* B, double* S, double* U, int N, int n,
                           int m, int ltv) {
  const size_t nn = (size_t)n * n, nm = (size_t)n * m;
  memset(U, 0, sizeof(double) * (size_t)m * N * N * n);
  if (!ltv) {
    /* blocks[d] = A^d B, power = A^{k+1} */
    double* blocks = (double*)malloc(sizeof(double) * (size_t)N * nm);
    double* power = (double*)malloc(sizeof(double) * 2 * nn);
    if (!blocks || !power) {
      free(blocks);
      free(power);
      return -1;
    }
    memcpy(blocks, B, sizeof(double) * nm);
    memcpy(power, A, sizeof(double) * nn);
    for (int k = 0; k < N; ++k) {
      double* cur = power + (size_t)(k & 1) * nn;
      if (k > 0) {
        left_multiply(A, power + (size_t)((k - 1) & 1) * nn, cur, n, n);       /* tools.py:29 */
        left_multiply(A, blocks + (size_t)(k - 1) * nm, blocks + (size_t)k * nm, n, m);
      }
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) S[((size_t)k * n + j) * n + i] = cur[i * n + j];
      for (int l = 0; l <= k; ++l)
        for (int j = 0; j < m; ++j)
          for (int i = 0; i < n; ++i)
            U[(((size_t)j * N + k) * N + l) * n + i] = blocks[(size_t)(k - l) * nm + i * m + j];
    }
    free(blocks);
    free(power);
    return 0;
  }
  /* LTV: s_k = A_k s_{k-1}, row_k[l] = A_k row_{k-1}[l] (l < k), row_k[k] = B_k */
  double* row = (double*)malloc(sizeof(double) * 2 * (size_t)N * nm);
  double* s = (double*)malloc(sizeof(double) * 2 * nn);
  if (!row || !s) {
    free(row);
    free(s);
    return -1;
  }
  for (int k = 0; k < N; ++k) {
    const double* Ak = A + (size_t)k * nn;
    const double* Bk = B + (size_t)k * nm;
    double* scur = s + (size_t)(k & 1) * nn;
    double* rcur = row + (size_t)(k & 1) * N * nm;
    const double* rprev = row + (size_t)((k - 1) & 1) * N * nm;
    if (k == 0)
      memcpy(scur, Ak, sizeof(double) * nn);
    else
      left_multiply(Ak, s + (size_t)((k - 1) & 1) * nn, scur, n, n);
    for (int l = 0; l < k; ++l) left_multiply(Ak, rprev + (size_t)l * nm, rcur + (size_t)l * nm, n, m);
    memcpy(rcur + (size_t)k * nm, Bk, sizeof(double) * nm);
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) S[((size_t)k * n + j) * n + i] = scur[i * n + j];
    for (int l = 0; l <= k; ++l)
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i)
          U[(((size_t)j * N + k) * N + l) * n + i] = rcur[(size_t)l * nm + i * m + j];
  }
  free(row);
  free(s);
  return 0;
}

/* `count` systems one after the other (the compiled single-core baseline of K1) */
int oracle_extend_matrices_batch(const double* A, const double* B, double* S, double* U, int count,
                                 int N, int n, int m, int ltv) {
  const size_t sa = (size_t)(ltv ? N : 1) * n * n, sb = (size_t)(ltv ? N : 1) * n * m;
  const size_t ss = (size_t)N * n * n, su = (size_t)m * N * N * n;
  for (int b = 0; b < count; ++b) {
    const int rc = oracle_extend_matrices(A + b * sa, B + b * sb, S + b * ss, U + b * su, N, n, m, ltv);
    if (rc) return rc;
  }
  return 0;
}
