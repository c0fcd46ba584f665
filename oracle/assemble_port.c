/* assemble_port.c -- plain C restatement of the reference's per-tick QP assembly  --  TEST
 * INFRASTRUCTURE ONLY (the compiled CPU twin of oracle/qp_oracle.py: preview_matrices + assemble).
 *
 * One instance = what the reference does per tick for one formulation
 * (python/mpc_interface/body.py:142-348 with tools.py:14-33 in front):
 *   1. tools.extend_matrices(N, A, B)                         tools.py:14-33 (extend_matrices.c)
 *   2. the ExtendedSystem's definitions are views into S, U   dynamics.py:277-298
 *   3. make_preview_matrices: PM[var] = (Mg, Mo) in definition order; a variable of a dynamics
 *      scatters its coefficient blocks into the given / optim columns (body.py:158-177), a
 *      derived one is the sum of coef . PM[dep] over its combination (body.py:179-193)
 *   4. generate_qp_constraint per limit (body.py:236-264, row rule restrictions.py:147-199),
 *      stacked (body.py:304-320)
 *   5. generate_qp_cost per cost (body.py:266-302), summed (body.py:322-329)
 * with the same dense matrices, the same loop structure and the same order of the sums as the numpy
 * oracle -- no flattening, no structural shortcuts: this is the path a C++ port of the reference
 * would run (cpp/ is a non-working skeleton of exactly that), used as the compiled CPU baseline.
 *
 * What differs between formulations is data: oracle/c_port.py walks a formulation the way
 * oracle/qp_oracle.py does (duck-typed on the reference's attribute names) and writes it down as
 * a *recipe* -- a stream of integer records (`ir`) and a pool of doubles (`dr`) -- that the
 * interpreter below executes per instance.  Numbers that may change from tick to tick (weights,
 * aims, arrows, centers, extremes) live in an array of constants of their own, per instance.
 *
 * Pinned: tests/test_oracle_golden.py runs it on every golden case of the real reference (body
 * case, biped ticks of both widths, C3, reduced C4) next to the numpy oracle.
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this.
 */
#include <stdlib.h>
#include <string.h>

int oracle_extend_matrices(const double* A, const double* B, double* S, double* U, int N, int n,
                           int m, int ltv);

enum { OP_END = 0, OP_BASE = 1, OP_DERIVED = 2, OP_LIMIT = 3, OP_COST = 4 };
/* header of ir: ng, no, nvar, pm_doubles, nc, nconst, horizon doubles (0: no per-instance system),
 * N, n, m, then per variable (rows, offset into PM), then the ops */
enum { R_NG = 0, R_NO, R_NVAR, R_PM, R_NC, R_NCONST, R_HLEN, R_N, R_NS, R_NM, R_HEADER };

typedef struct {
  int ng, no, w;           /* w = ng + no: a row of PM[var] is [Mg row | Mo row] */
  const int* vrows;
  const int* voff;
  double* pm;
} Ctx;

/* rows `start + i step`, i < count, of variable v */
static const double* pm_row(const Ctx* c, int v, int r) { return c->pm + c->voff[v] + (size_t)r * c->w; }

/* out[count_out][w] = the picked rows of variable v, through L (count_out x count) when there is one */
static void picked_rows(const Ctx* c, int v, int start, int step, int count, const double* L,
                        int count_out, double* out) {
  if (!L) {
    for (int i = 0; i < count; ++i) memcpy(out + (size_t)i * c->w, pm_row(c, v, start + i * step), sizeof(double) * c->w);
    return;
  }
  memset(out, 0, sizeof(double) * (size_t)count_out * c->w);
  for (int i = 0; i < count_out; ++i)
    for (int k = 0; k < count; ++k) {
      const double l = L[(size_t)i * count + k];
      if (l == 0.0) continue;
      const double* row = pm_row(c, v, start + k * step);
      double* o = out + (size_t)i * c->w;
      for (int x = 0; x < c->w; ++x) o[x] += l * row[x];
    }
}

/* one instance: horizon buffer `hz` (the dynamics' matrices, list order, C order), constants, given.
 * scratch: pm (R_PM doubles) + 4 * maxrows * w doubles.  Q [no][no], q [no], G [nc][no], h [nc]. */
static void run_recipe(const int* ir, const double* dr, const double* hz, const double* cst,
                       const double* given, double* pm, double* scratch, int maxrows, double* Q,
                       double* q, double* G, double* h) {
  Ctx c;
  c.ng = ir[R_NG], c.no = ir[R_NO], c.w = c.ng + c.no;
  const int nvar = ir[R_NVAR], ng = c.ng, no = c.no, w = c.w;
  c.vrows = ir + R_HEADER;
  c.voff = ir + R_HEADER + nvar;
  c.pm = pm;
  memset(pm, 0, sizeof(double) * (size_t)ir[R_PM]);
  memset(Q, 0, sizeof(double) * (size_t)no * no);
  memset(q, 0, sizeof(double) * (size_t)no);
  double* va = scratch;                          /* v rows  [maxrows][w] */
  double* vc = va + (size_t)maxrows * w;         /* cross rows */
  double* rv = vc + (size_t)maxrows * w;         /* residual vectors [2][maxrows] */
  const int* op = ir + R_HEADER + 2 * nvar;
  for (;;) {
    const int code = *op++;
    if (code == OP_END) break;
    if (code == OP_BASE) {
      /* body.py:158-177: Mg[:, given_ID[dep]] = coef / Mo[:, optim_ID[dep]] = coef, entry by entry:
       * constants of the formulation, then elements of this instance's horizon matrices */
      const int v = *op++, ntmpl = *op++, doff = *op++, nsrc = *op++;
      double* M = pm + c.voff[v];
      for (int e = 0; e < ntmpl; ++e) M[op[e]] = dr[doff + e];
      op += ntmpl;
      for (int e = 0; e < nsrc; ++e) M[op[2 * e]] = hz[op[2 * e + 1]];
      op += 2 * nsrc;
    } else if (code == OP_DERIVED) {
      /* body.py:179-193: sum over the combination of coef . PM[dep], in its order */
      const int v = *op++, ndep = *op++;
      double* M = pm + c.voff[v];
      const int rows = c.vrows[v];
      for (int d = 0; d < ndep; ++d) {
        const int dep = *op++, kind = *op++, coff = *op++;
        const int drows = c.vrows[dep];
        const double* D = pm + c.voff[dep];
        if (kind == 0) {               /* a scalar */
          const double s = dr[coff];
          for (size_t x = 0; x < (size_t)rows * w; ++x) M[x] += s * D[x];
        } else {                       /* a matrix [rows][drows] */
          for (int i = 0; i < rows; ++i)
            for (int k = 0; k < drows; ++k) {
              const double s = dr[coff + (size_t)i * drows + k];
              if (s == 0.0) continue;
              for (int x = 0; x < w; ++x) M[(size_t)i * w + x] += s * D[(size_t)k * w + x];
            }
        }
      }
    } else if (code == OP_LIMIT) {
      /* body.py:236-264: cM = sum over the axes of coefs_i (*) M_axis[picked]; A = cMo,
       * h = (extreme + sum arrow center) - cMg given (restrictions.py:185-199) */
      const int out0 = *op++, nlines = *op++, naxes = *op++, carrow = *op++, ccenter = *op++, cext = *op++;
      double* cM = va;
      memset(cM, 0, sizeof(double) * (size_t)nlines * w);
      for (int a = 0; a < naxes; ++a) {
        const int v = *op++, start = *op++, step = *op++, count = *op++, loff = *op++;
        for (int i = 0; i < nlines; ++i) {
          const double ar = cst[carrow + i * naxes + a];
          double* o = cM + (size_t)i * w;
          if (loff < 0) {              /* coefs[i] * M[picked]: row i scaled by its arrow */
            const double* row = pm_row(&c, v, start + i * step);
            for (int x = 0; x < w; ++x) o[x] += ar * row[x];
          } else {                     /* (arrow column * L) @ M[picked] */
            for (int k = 0; k < count; ++k) {
              const double l = ar * dr[loff + (size_t)i * count + k];
              if (l == 0.0) continue;
              const double* row = pm_row(&c, v, start + k * step);
              for (int x = 0; x < w; ++x) o[x] += l * row[x];
            }
          }
        }
      }
      for (int i = 0; i < nlines; ++i) {
        const double* row = cM + (size_t)i * w;
        memcpy(G + (size_t)(out0 + i) * no, row + ng, sizeof(double) * no);
        double ac = 0.0, dg = 0.0;
        for (int a = 0; a < naxes; ++a) ac += cst[carrow + i * naxes + a] * cst[ccenter + i * naxes + a];
        for (int x = 0; x < ng; ++x) dg += row[x] * given[x];
        h[out0 + i] = (cst[cext + i] + ac) - dg;
      }
    } else if (code == OP_COST) {
      /* body.py:266-302: per axis  Q += w vMo^T cMo;
       * q += w (vMo^T (cMg given - cross_aim) + cMo^T (vMg given - aim)) / 2 */
      const int naxes = *op++, cweight = *op++, caim = *op++, ccross = *op++;
      const double wgt = cst[cweight];
      for (int a = 0; a < naxes; ++a) {
        const int v = *op++, x = *op++, start = *op++, step = *op++, count = *op++;
        const int lv = *op++, lx = *op++, rows = *op++;   /* rows: after L */
        picked_rows(&c, v, start, step, count, lv < 0 ? NULL : dr + lv, rows, va);
        const double* C = va;
        if (x != v || lx != lv) {
          picked_rows(&c, x, start, step, count, lx < 0 ? NULL : dr + lx, rows, vc);
          C = vc;
        }
        const double aim = cst[caim + a], cross_aim = cst[ccross + a];
        for (int i = 0; i < rows; ++i) {
          double dv = 0.0, dc = 0.0;
          for (int g = 0; g < ng; ++g) {
            dv += va[(size_t)i * w + g] * given[g];
            dc += C[(size_t)i * w + g] * given[g];
          }
          rv[i] = dc - cross_aim;            /* multiplies vMo^T */
          rv[maxrows + i] = dv - aim;        /* multiplies cMo^T */
        }
        for (int r = 0; r < no; ++r) {
          double* Qr = Q + (size_t)r * no;
          double acc = 0.0;
          for (int i = 0; i < rows; ++i) {
            const double vr = va[(size_t)i * w + ng + r];
            acc += vr * rv[i] + C[(size_t)i * w + ng + r] * rv[maxrows + i];
            if (vr == 0.0) continue;
            const double wv = wgt * vr;
            const double* crow = C + (size_t)i * w + ng;
            for (int s = 0; s < no; ++s) Qr[s] += wv * crow[s];
          }
          q[r] += wgt * acc / 2;
        }
      }
    } else {
      return;   /* (a malformed recipe: c_port.py never writes one) */
    }
  }
}

/* `count` instances.  A [count][n][n], B [count][n][m] (ignored without a per-instance system);
 * consts [count or 1][nconst] (const_stride 0: shared); given [count][ng].  Outputs [keep ? count : 1][...]:
 * with keep == 0 every instance overwrites the first slot (timing).  Returns 0, -1 out of memory. */
int oracle_assemble_batch(const int* ir, const double* dr, const double* A, const double* B,
                          const double* consts, long const_stride, const double* given, int count,
                          int maxrows, int keep, double* Q, double* q, double* G, double* h) {
  const int ng = ir[R_NG], no = ir[R_NO], nc = ir[R_NC], w = ng + no;
  const int hlen = ir[R_HLEN], N = ir[R_N], n = ir[R_NS], m = ir[R_NM];
  double* pm = (double*)malloc(sizeof(double) * ((size_t)ir[R_PM] + 1));
  double* scratch = (double*)malloc(sizeof(double) * ((size_t)2 * maxrows * w + 2 * maxrows + 1));
  double* hz = (double*)malloc(sizeof(double) * ((size_t)hlen + 1));
  int rc = (pm && scratch && hz) ? 0 : -1;
  for (int b = 0; b < count && rc == 0; ++b) {
    if (hlen > 0) {
      /* the dynamics' matrices in the reference's list order: U_0 .. U_{m-1}, then S (tools.py:33) */
      double* U = hz;
      double* S = hz + (size_t)m * N * N * n;
      rc = oracle_extend_matrices(A + (size_t)b * n * n, B + (size_t)b * n * m, S, U, N, n, m, 0);
      if (rc) break;
    }
    const size_t o = keep ? (size_t)b : 0;
    run_recipe(ir, dr, hz, consts + (size_t)b * const_stride, given + (size_t)b * ng, pm, scratch,
               maxrows, Q + o * no * no, q + o * no, G + o * nc * no, h + o * nc);
  }
  free(pm);
  free(scratch);
  free(hz);
  return rc;
}
