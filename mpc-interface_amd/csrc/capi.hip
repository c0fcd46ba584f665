// capi.hip -- extern "C" boundary of libmpcasm.so (declared in include/mpcasm.h).
// Argument checking, plan validation / upload and kernel dispatch; no kernel code.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "jit.h"
#include "kernels.h"

using namespace mpcasm;

namespace mpcasm {
int g_p_direct = 0;  // MPCASM_OPT_P_DIRECT (read when a plan is created)
extern int g_path;
extern int g_phase_mask;
extern int g_resident_per_cu, g_resident_grid;
}

struct mpcasm_plan {
  PlanDev dev;
  int32_t* d_itab;
  double* d_dtab;
  int device;
  int num_cus;
  std::vector<int32_t> h_itab;  // host copy: what a specialised kernel is generated from (jit.hip)
  // mpcasm_plan_set_option: this plan's own choice of path / per-plan compilation / workgroups
  // per CU (-1: the process-wide value of mpcasm_set_option)
  int opt_path = -1, opt_jit = -1, opt_per_cu = -1, opt_grid = -1;
  mutable int last_kernel = 0;  // mpcasm_plan_last_kernel: what the latest mpcasm_assemble launched
};

namespace {

thread_local int g_last_hip = 0;

int hip_fail(hipError_t e) {
  g_last_hip = (int)e;
  return MPCASM_ERR_HIP;
}

// section [off, off + len) must lie inside [H_WORDS, n)
bool in_range(int64_t off, int64_t len, int64_t n, int64_t lo) {
  return off >= lo && len >= 0 && off + len <= n;
}

// the sweep kernel's tables (H_SW_OK): every column, parameter slot, step and combination it reads
int validate_sweep(const int32_t* it, int64_t n, int64_t nd) {
  if (it[H_SW_OK] & ~1) return MPCASM_ERR_PLAN;
  if (!it[H_SW_OK]) return MPCASM_OK;
  const int64_t sn = it[H_SW_N], sm = it[H_SW_M], N = it[H_SW_HORIZON], naxes = it[H_SW_NAXES];
  const int64_t ng = it[H_NG], no = it[H_NO], nc = it[H_NC], nparams = it[H_NPARAMS], nsrc = it[H_NSRC];
  const int64_t nterm = it[H_SW_NTERM], nlim = it[H_SW_NLIM], ncv = it[H_SW_NCVEC];
  constexpr int64_t LIMW = SW_LIM_WORDS + SW_AXMAX * SW_LAX_WORDS;
  if (sn < 1 || sn > SW_NMAX || sm < 1 || sm > SW_MMAX || N < 1 || N > 32767 || naxes < 1 || naxes > SW_AXMAX ||
      it[H_SW_SRC_A] < 0 || it[H_SW_SRC_A] >= nsrc || it[H_SW_SRC_B] < 0 || it[H_SW_SRC_B] >= nsrc ||
      it[H_SW_SRC_A] == it[H_SW_SRC_B] || nterm < 0 || nlim < 0 || ncv < 0 ||
      !in_range(it[H_OFF_SW_AXIS], naxes * SW_AXIS_WORDS, n, H_WORDS) ||
      !in_range(it[H_OFF_SW_TERM], nterm * SW_TERM_WORDS, n, H_WORDS) ||
      !in_range(it[H_OFF_SW_LIM], nlim * LIMW, n, H_WORDS) || !in_range(it[H_OFF_SW_COL], no, n, H_WORDS) ||
      !in_range(it[H_SW_DOFF_CVEC], ncv * SW_NMAX, nd, 0) ||
      !in_range(it[H_OFF_RS_DPAR], no * 2 * RS_DIAG_MAX, n, H_WORDS) || it[H_OFF_RS_DPAR] % 4 ||
      !in_range(it[H_DOFF_RS_DCOEF], no * RS_DIAG_MAX, nd, 0) || it[H_DOFF_RS_DCOEF] % 2)
    return MPCASM_ERR_PLAN;
  const int32_t* ax = it + it[H_OFF_SW_AXIS];
  const int32_t* col = it + it[H_OFF_SW_COL];
  std::vector<char> seen(std::max<int64_t>(no, 1), 0);
  for (int64_t a = 0; a < naxes; ++a) {
    if (ax[a * SW_AXIS_WORDS] < 0 || ax[a * SW_AXIS_WORDS] + sn > ng) return MPCASM_ERR_PLAN;
    for (int64_t j = 0; j < sm; ++j) {
      const int64_t c0 = ax[a * SW_AXIS_WORDS + 1 + j];
      if (c0 < 0 || c0 + N > no) return MPCASM_ERR_PLAN;
      for (int64_t l = 0; l < N; ++l) {
        if (seen[c0 + l] || col[c0 + l] != (int32_t)(a | (j << 8) | (l << 16))) return MPCASM_ERR_PLAN;
        seen[c0 + l] = 1;
      }
    }
  }
  for (int64_t c = 0; c < no; ++c)
    if (!seen[c]) return MPCASM_ERR_PLAN;  // every unknown is an input of the system
  auto steps_ok = [&](int64_t k0, int64_t ks, int64_t cnt) {
    return cnt >= 1 && k0 >= 0 && k0 < N && k0 + (cnt - 1) * ks >= 0 && k0 + (cnt - 1) * ks < N;
  };
  auto slot_ok = [&](int64_t slot, int64_t step, int64_t cnt) {
    return slot >= 0 && step >= 0 && slot + (cnt - 1) * step < nparams;
  };
  const int32_t* tr = it + it[H_OFF_SW_TERM];
  for (int64_t t = 0; t < nterm; ++t, tr += SW_TERM_WORDS)
    if (tr[ST_AXIS] < 0 || tr[ST_AXIS] >= naxes || !steps_ok(tr[ST_K0], tr[ST_KSTEP], tr[ST_COUNT]) ||
        tr[ST_KSTEP] < 0 ||  // (the kernel walks a term's steps downwards from its last)
        tr[ST_WPARAM] < 0 || tr[ST_WPARAM] >= nparams || tr[ST_AIMPARAM] < 0 || tr[ST_AIMPARAM] >= nparams ||
        tr[ST_CVEC] < 0 || tr[ST_CVEC] % SW_NMAX || tr[ST_CVEC] / SW_NMAX >= ncv)
      return MPCASM_ERR_PLAN;
  std::vector<char> line(std::max<int64_t>(nc, 1), 0);
  const int32_t* lm = it + it[H_OFF_SW_LIM];
  for (int64_t t = 0; t < nlim; ++t, lm += LIMW) {
    const int64_t out0 = lm[SL_OUT0], cnt = lm[SL_COUNT], nax = lm[SL_NAXES];
    if (out0 < 0 || cnt < 1 || out0 + cnt > nc || nax < 1 || nax > SW_AXMAX ||
        !slot_ok(lm[SL_EXTREME], lm[SL_EXTREME_STEP], cnt))
      return MPCASM_ERR_PLAN;
    for (int64_t i = 0; i < cnt; ++i) {
      if (line[out0 + i]) return MPCASM_ERR_PLAN;
      line[out0 + i] = 1;
    }
    for (int64_t a = 0; a < nax; ++a) {
      const int32_t* x = lm + SW_LIM_WORDS + a * SW_LAX_WORDS;
      const int32_t* x0 = lm + SW_LIM_WORDS;
      if (x[SX_AXIS] < 0 || x[SX_AXIS] >= naxes || !steps_ok(x[SX_K0], x[SX_KSTEP], cnt) ||
          x[SX_K0] != x0[SX_K0] || x[SX_KSTEP] != x0[SX_KSTEP] ||       // (a line's axes: one step)
          x[SX_CVEC] < 0 || x[SX_CVEC] % SW_NMAX || x[SX_CVEC] / SW_NMAX >= ncv ||
          !slot_ok(x[SX_ARROW], x[SX_ARROW_STEP], cnt) || !slot_ok(x[SX_CENTER], x[SX_CENTER_STEP], cnt))
        return MPCASM_ERR_PLAN;
    }
  }
  for (int64_t R = 0; R < nc; ++R)
    if (!line[R]) return MPCASM_ERR_PLAN;  // every line of G, h is written
  {  // the per-step lists say what the records say: every cost row and every line once, at its step
    const int64_t ncent = it[H_SW_NCENT], ngent = it[H_SW_NGENT];
    if (ncent < 0 || ngent != nc || !in_range(it[H_OFF_SW_CPTR], N + 1, n, H_WORDS) ||
        !in_range(it[H_OFF_SW_CENT], ncent, n, H_WORDS) || !in_range(it[H_OFF_SW_GPTR], N + 1, n, H_WORDS) ||
        !in_range(it[H_OFF_SW_GENT], ngent * 2, n, H_WORDS))
      return MPCASM_ERR_PLAN;
    const int32_t* cptr = it + it[H_OFF_SW_CPTR];
    const int32_t* cent = it + it[H_OFF_SW_CENT];
    const int32_t* gptr = it + it[H_OFF_SW_GPTR];
    const int32_t* gent = it + it[H_OFF_SW_GENT];
    if (cptr[0] != 0 || cptr[N] != ncent || gptr[0] != 0 || gptr[N] != ngent) return MPCASM_ERR_PLAN;
    std::vector<int64_t> rows(std::max<int64_t>(nterm, 1), 0);
    std::fill(line.begin(), line.end(), 0);
    for (int64_t l = 0; l < N; ++l) {
      if (cptr[l + 1] < cptr[l] || gptr[l + 1] < gptr[l]) return MPCASM_ERR_PLAN;
      for (int64_t e = cptr[l]; e < cptr[l + 1]; ++e) {
        const int64_t ti = cent[e];
        if (ti < 0 || ti >= nterm) return MPCASM_ERR_PLAN;
        const int32_t* x = it + it[H_OFF_SW_TERM] + ti * SW_TERM_WORDS;
        const int64_t ks = x[ST_KSTEP], dk = l - x[ST_K0];
        ++rows[ti];
        if (ks == 0 ? dk != 0 : (dk % ks != 0 || dk / ks < 0 || dk / ks >= x[ST_COUNT])) return MPCASM_ERR_PLAN;
      }
      for (int64_t e = gptr[l]; e < gptr[l + 1]; ++e) {
        const int64_t li = gent[2 * e], i = gent[2 * e + 1];
        if (li < 0 || li >= nlim) return MPCASM_ERR_PLAN;
        const int32_t* x = it + it[H_OFF_SW_LIM] + li * LIMW;
        if (i < 0 || i >= x[SL_COUNT] || x[SW_LIM_WORDS + SX_K0] + i * x[SW_LIM_WORDS + SX_KSTEP] != l ||
            line[x[SL_OUT0] + i])
          return MPCASM_ERR_PLAN;
        line[x[SL_OUT0] + i] = 1;
      }
    }
    for (int64_t t = 0; t < nterm; ++t)
      if (rows[t] != (it + it[H_OFF_SW_TERM] + t * SW_TERM_WORDS)[ST_COUNT]) return MPCASM_ERR_PLAN;
  }
  const int32_t* dp = it + it[H_OFF_RS_DPAR];
  for (int64_t i = 0; i < no * 2 * RS_DIAG_MAX; ++i)
    if (dp[i] < 0 || dp[i] > nparams) return MPCASM_ERR_PLAN;
  return MPCASM_OK;
}

// the column tables (H_T_CI_OK) and the tiled program (H_T_OK): every stream, offset and row a
// kernel of tiled.hip derives from them stays inside its table
int validate_tiled(const int32_t* it, const double* h_dtab, int64_t n, int64_t nd) {
  const int64_t ng = it[H_NG], no = it[H_NO], nc = it[H_NC], nbase = it[H_NBASE];
  if ((it[H_T_CI_OK] & ~1) || (it[H_T_OK] & ~1)) return MPCASM_ERR_PLAN;
  if (!it[H_T_CI_OK]) return it[H_T_OK] ? MPCASM_ERR_PLAN : MPCASM_OK;
  const int64_t nop = it[H_T_NOP], ndelta = it[H_T_NDELTA], ddelta = it[H_T_DOFF_DELTA];
  if (nop < no || nop < T_BLOCK || nop % T_BLOCK || ndelta < 2 ||
      !in_range(ddelta, ndelta, nd, 0) || h_dtab[ddelta] != 0.0 ||
      !in_range(it[H_OFF_T_CIG], nbase * ng * 2, n, H_WORDS) ||
      !in_range(it[H_OFF_T_CIO], nbase * nop * 2, n, H_WORDS) || it[H_OFF_T_CIO] % 4 ||
      !in_range(it[H_OFF_ARENA], (int64_t)it[H_NSRC] * 2, n, H_WORDS) ||
      !in_range(it[H_OFF_ENTBASE], it[H_NENT], n, H_WORDS) ||
      !in_range(it[H_OFF_ENTK], it[H_NENT], n, H_WORDS) ||
      !in_range(it[H_OFF_PM_ENTBASE], it[H_PM_NENT], n, H_WORDS) ||
      !in_range(it[H_OFF_PM_ENTK], it[H_PM_NENT], n, H_WORDS))
    return MPCASM_ERR_PLAN;
  // generated groups: their sources are tables in the scratch
  const int64_t nlti = it[H_T_NLTI], twork = it[H_T_WORK];
  if (nlti < 0 || nlti > RS_LTI_MAX || twork < it[H_RTOT] ||
      !in_range(it[H_OFF_T_LTI], nlti * T_LTI_WORDS, n, H_WORDS))
    return MPCASM_ERR_PLAN;
  std::vector<int64_t> size(T_SID_CONST + 1, 0);
  for (int sx = 0; sx < it[H_NSRC]; ++sx) size[sx] = (it + it[H_OFF_ARENA])[2 * sx + 1];
  size[T_SID_CONST] = nd;
  for (int64_t g = 0; g < nlti; ++g) {
    const int32_t* x = it + it[H_OFF_T_LTI] + g * T_LTI_WORDS;
    const int64_t gn = x[TL_N], gm = x[TL_M], gN = x[TL_HORIZON];
    if (gn < 1 || gm < 1 || gN < 1 || gn > 64 || gm > 64 || gN > 4096 ||
        !in_range(it[H_OFF_T_LTI_IDS] + (int64_t)x[TL_IDS], gm + 1, n, H_WORDS) || x[TL_IDS] < 0 ||
        !in_range(x[TL_TA], gN * gn * gn, twork, it[H_RTOT]) ||
        !in_range(x[TL_TB], gn * gm * 2 * gN, twork, it[H_RTOT]))
      return MPCASM_ERR_PLAN;
    const int32_t* ids = it + it[H_OFF_T_LTI_IDS] + x[TL_IDS];
    for (int64_t j = 0; j <= gm; ++j) {
      if (ids[j] < 0 || ids[j] >= it[H_NSRC]) return MPCASM_ERR_PLAN;
      size[ids[j]] = j < gm ? gn * gm * 2 * gN : gN * gn * gn;
    }
  }
  // rows of every base variable some program reads
  std::vector<int64_t> kmax(std::max<int64_t>(nbase, 1), 0);
  auto scan = [&](int off_b, int off_k, int64_t cnt) {
    for (int64_t e = 0; e < cnt; ++e) {
      const int64_t b = (it + off_b)[e], k = (it + off_k)[e];
      if (b < 0 || b >= nbase || k < 0) return false;
      kmax[b] = std::max(kmax[b], k);
    }
    return true;
  };
  if (!scan(it[H_OFF_ENTBASE], it[H_OFF_ENTK], it[H_NENT]) ||
      !scan(it[H_OFF_PM_ENTBASE], it[H_OFF_PM_ENTK], it[H_PM_NENT]))
    return MPCASM_ERR_PLAN;
  {  // first rows of the base variables: ascending, every row some program reads lies inside
    if (!in_range(it[H_OFF_T_BROW0], nbase + 1, n, H_WORDS)) return MPCASM_ERR_PLAN;
    const int32_t* r0 = it + it[H_OFF_T_BROW0];
    if (r0[0] != 0) return MPCASM_ERR_PLAN;
    for (int64_t b = 0; b < nbase; ++b)
      if (r0[b + 1] < r0[b] || (r0[b + 1] > r0[b] ? kmax[b] >= r0[b + 1] - r0[b] : kmax[b] != 0))
        return MPCASM_ERR_PLAN;
  }
  {  // the covered columns of every base variable: ascending ranges of one list, inside [0, ng + no)
    if (!in_range(it[H_OFF_T_BCOLPTR], nbase + 1, n, H_WORDS)) return MPCASM_ERR_PLAN;
    const int32_t* cp = it + it[H_OFF_T_BCOLPTR];
    if (cp[0] != 0 || cp[nbase] < 0 || !in_range(it[H_OFF_T_BCOLS], cp[nbase], n, H_WORDS))
      return MPCASM_ERR_PLAN;
    const int32_t* cl = it + it[H_OFF_T_BCOLS];
    for (int64_t b = 0; b < nbase; ++b) {
      if (cp[b + 1] < cp[b]) return MPCASM_ERR_PLAN;
      for (int64_t i = cp[b]; i < cp[b + 1]; ++i)
        if (cl[i] < 0 || cl[i] >= ng + no || (i > cp[b] && cl[i] <= cl[i - 1])) return MPCASM_ERR_PLAN;
    }
  }
  auto table_ok = [&](const int32_t* ci, int64_t cols) {
    for (int64_t b = 0; b < nbase; ++b)
      for (int64_t c = 0; c < cols; ++c) {
        const int64_t off = (uint32_t)ci[(b * cols + c) * 2];
        const int32_t meta = ci[(b * cols + c) * 2 + 1];
        const int64_t sid = (uint32_t)meta >> 24, rs = (meta << 8) >> 8;
        if (sid > T_SID_CONST || (sid < T_SID_CONST && sid >= it[H_NSRC])) return false;
        const int64_t lo = sid == T_SID_CONST ? ddelta : 0;
        const int64_t hi = sid == T_SID_CONST ? ddelta + ndelta : size[sid];
        const int64_t last = off + kmax[b] * rs;
        if (off < lo || off >= hi || last < lo || last >= hi) return false;
      }
    return true;
  };
  if (!table_ok(it + it[H_OFF_T_CIG], ng) || !table_ok(it + it[H_OFF_T_CIO], nop))
    return MPCASM_ERR_PLAN;
  if (it[H_T_NP1] >= 0) {  // f2's unrolled tables: every entry inside its stream, every index inside its array
    const int64_t np1 = it[H_T_NP1], nbrow = (it + it[H_OFF_T_BROW0])[nbase], pment = it[H_PM_NENT];
    if (!in_range(it[H_OFF_T_P1PTR], nbrow + 1, n, H_WORDS) || !in_range(it[H_OFF_T_P1ENT], np1 * 2, n, H_WORDS) ||
        it[H_OFF_T_P1ENT] % 2 || !in_range(it[H_OFF_T_P2Y], pment, n, H_WORDS))
      return MPCASM_ERR_PLAN;
    const int32_t* pp = it + it[H_OFF_T_P1PTR];
    if (pp[0] != 0 || pp[nbrow] != np1) return MPCASM_ERR_PLAN;
    for (int64_t t = 0; t < nbrow; ++t)
      if (pp[t + 1] < pp[t]) return MPCASM_ERR_PLAN;
    const int32_t* pe = it + it[H_OFF_T_P1ENT];
    for (int64_t e = 0; e < np1; ++e) {
      const int64_t off = (uint32_t)pe[2 * e], sid = (uint32_t)pe[2 * e + 1] >> 24, c = pe[2 * e + 1] & 0xFFFFFF;
      if (sid > T_SID_CONST || (sid < T_SID_CONST && sid >= it[H_NSRC]) || c >= ng + no) return MPCASM_ERR_PLAN;
      const int64_t lo = sid == T_SID_CONST ? ddelta : 0;
      const int64_t hi = sid == T_SID_CONST ? ddelta + ndelta : size[sid];
      if (off < lo || off >= hi) return MPCASM_ERR_PLAN;
    }
    const int32_t* py = it + it[H_OFF_T_P2Y];
    for (int64_t e = 0; e < pment; ++e)
      if (py[e] < 0 || py[e] >= nbrow) return MPCASM_ERR_PLAN;
  } else if (it[H_T_NP1] != -1) {
    return MPCASM_ERR_PLAN;
  }
  if (!it[H_T_OK]) return MPCASM_OK;
  // the tiled program
  const int64_t nstage = it[H_T_NSTAGE], rtot = it[H_RTOT];
  if (no < T_BLOCK || nstage < 0 ||
      !in_range(it[H_OFF_T_STAGE], nstage * T_STAGE_WORDS, n, H_WORDS) || it[H_OFF_T_STAGE] % 4 ||
      !in_range(it[H_OFF_T_GROW], nc * RS_AXMAX, n, H_WORDS) ||
      !in_range(it[H_OFF_RS_RR], nc * RS_RR_WORDS, n, H_WORDS) ||
      !in_range(it[H_OFF_ROWPTR], rtot + 1, n, H_WORDS))
    return MPCASM_ERR_PLAN;
  {  // the per-column table of the diagonal gterms (read when no column has more than RS_DIAG_MAX)
    if (!in_range(it[H_OFF_RS_DPAR], no * 2 * RS_DIAG_MAX, n, H_WORDS) || it[H_OFF_RS_DPAR] % 4 ||
        !in_range(it[H_DOFF_RS_DCOEF], no * RS_DIAG_MAX, nd, 0) || it[H_DOFF_RS_DCOEF] % 2)
      return MPCASM_ERR_PLAN;
    const int32_t* dp = it + it[H_OFF_RS_DPAR];
    for (int64_t i = 0; i < no * 2 * RS_DIAG_MAX; ++i)
      if (dp[i] < 0 || dp[i] > it[H_NPARAMS]) return MPCASM_ERR_PLAN;
  }
  const int64_t ngrest = it[H_T_NGREST];
  if (!in_range(it[H_OFF_T_SROW], nstage * 2 * 16, n, H_WORDS) ||
      !in_range(it[H_T_DOFF_SCOEF], nstage * 2 * 16, nd, 0) ||
      !in_range(it[H_OFF_T_PIG], nstage * 16 * T_PIG_MAX * 2, n, H_WORDS) || it[H_OFF_T_PIG] % 4 ||
      ngrest < 0 || ngrest > nc || !in_range(it[H_OFF_T_GREST], ngrest, n, H_WORDS))
    return MPCASM_ERR_PLAN;
  const int32_t* rowptr = it + it[H_OFF_ROWPTR];
  const int32_t* entbase = it + it[H_OFF_ENTBASE];
  const int32_t* entk = it + it[H_OFF_ENTK];
  const int32_t* srow = it + it[H_OFF_T_SROW];
  const int32_t* pig = it + it[H_OFF_T_PIG];
  const int32_t* grow = it + it[H_OFF_T_GROW];
  const int32_t* rrw = it + it[H_OFF_RS_RR];
  std::vector<char> g_written(std::max<int64_t>(nc, 1), 0);
  int last_cls = 0;
  for (int64_t sx = 0; sx < nstage; ++sx) {
    const int32_t* x = it + it[H_OFF_T_STAGE] + sx * T_STAGE_WORDS;
    const int rows = x[TS_INFO] & 255, fl = (x[TS_INFO] >> 8) & 255, cls = x[TS_INFO] >> 16;
    // sorted by class; class 0 exactly for the stages without a Hessian part
    if (x[TS_INFO] < 0 || cls < last_cls || cls > 4 || (cls > 0) != ((fl & TS_FLAG_P) != 0))
      return MPCASM_ERR_PLAN;
    last_cls = cls;
    // the class covers the masks: in every 64-column quarter only tiles 0 .. cls - 1
    if (fl & TS_FLAG_P)
      for (int side = 0; side < 2; ++side) {
        uint64_t m = ((uint64_t)(uint32_t)x[side ? TS_MASKB_HI : TS_MASKA_HI] << 32) |
                     (uint32_t)x[side ? TS_MASKB_LO : TS_MASKA_LO];
        if ((m >> 63) && cls < 4) return MPCASM_ERR_PLAN;  // (folded tiles: anything)
        for (; m; m >>= 4)
          if ((m & 15u) >> cls) return MPCASM_ERR_PLAN;
      }
    for (int side = 0; side < 2; ++side) {  // base rows of simple stages: the rows' own entries
      if (!(fl & (side ? TS_FLAG_SIMPLE_B : TS_FLAG_SIMPLE_A))) continue;
      const int row0 = side ? x[TS_BROW] : x[TS_AROW];
      if (row0 < 0 || row0 + rows > rtot) return MPCASM_ERR_PLAN;
      for (int i = 0; i < 16; ++i) {
        const int k = srow[(sx * 2 + side) * 16 + i];
        if (i < rows ? (rowptr[row0 + i] < 0 || rowptr[row0 + i] >= it[H_NENT] ||
                        k != entk[rowptr[row0 + i]])
                     : k != 0)
          return MPCASM_ERR_PLAN;
      }
    }
    for (int i = 0; i < 16; ++i)  // riding rows of G: one axis, this very workspace row, once
      for (int t = 0; t < T_PIG_MAX; ++t) {
        const int R = pig[((sx * 16 + i) * T_PIG_MAX + t) * 2], slot = pig[((sx * 16 + i) * T_PIG_MAX + t) * 2 + 1];
        if (R == -1) continue;
        if (R < 0 || R >= nc || i >= rows || !(fl & TS_FLAG_G) || g_written[R] ||
            rrw[R * RS_RR_WORDS + RR_NAXES] != 1 || grow[R * RS_AXMAX] != x[TS_AROW] + i ||
            slot != rrw[R * RS_RR_WORDS + RR_ARROW])
          return MPCASM_ERR_PLAN;
        g_written[R] = 1;
      }
    if (rows < 1 || rows > 16 || (fl & ~127) || x[TS_AROW] < 0 ||
        x[TS_AROW] + rows > rtot || x[TS_BROW] < 0 || x[TS_BROW] + rows > rtot || x[TS_DROW] < 0 ||
        x[TS_DROW] + rows > rtot || x[TS_WPARAM] < 0 || x[TS_WPARAM] >= it[H_NPARAMS] ||
        x[TS_AIMPARAM] < 0 || x[TS_AIMPARAM] >= it[H_NPARAMS] ||
        ((fl & TS_FLAG_SAME) != 0) != (x[TS_AROW] == x[TS_BROW]))
      return MPCASM_ERR_PLAN;
    // a stage flagged simple: one entry per row, all of the stated base
    for (int side = 0; side < 2; ++side) {
      if (!(fl & (side ? TS_FLAG_SIMPLE_B : TS_FLAG_SIMPLE_A))) continue;
      const int row0 = side ? x[TS_BROW] : x[TS_AROW];
      const int base = side ? (int)((uint32_t)x[TS_BASE] >> 16) : (x[TS_BASE] & 0xFFFF);
      if (base >= nbase) return MPCASM_ERR_PLAN;
      for (int r = row0; r < row0 + rows; ++r)
        if (rowptr[r + 1] - rowptr[r] != 1 || rowptr[r] < 0 || rowptr[r] >= it[H_NENT] ||
            entbase[rowptr[r]] != base)
          return MPCASM_ERR_PLAN;
    }
  }
  // Toeplitz form: all stages or none; one generated group; every stage's rows are consecutive
  // rows of one of its states with one coefficient, and -- what lets the kernel keep ONE set of
  // per-lane column offsets -- a column's table offset less the state's own part is the same for
  // every state, inside the group's TB table together with the window
  if (it[H_T_TOEPLITZ] & ~1) return MPCASM_ERR_PLAN;
  {
    int64_t ntoep = 0;
    for (int64_t sx = 0; sx < nstage; ++sx)
      ntoep += ((it + it[H_OFF_T_STAGE] + sx * T_STAGE_WORDS)[TS_INFO] >> 8) & TS_FLAG_TOEPLITZ ? 1 : 0;
    if (it[H_T_TOEPLITZ] ? (ntoep != nstage || nstage == 0 || nlti != 1) : false) return MPCASM_ERR_PLAN;
  }
  if (it[H_T_TOEPLITZ]) {
    const int32_t* g = it + it[H_OFF_T_LTI];
    const int64_t gn = g[TL_N], gm = g[TL_M], gN = g[TL_HORIZON], tbn = gn * gm * 2 * gN;
    const int32_t* ids = it + it[H_OFF_T_LTI_IDS] + g[TL_IDS];
    const double* sc = h_dtab + it[H_T_DOFF_SCOEF];
    const int32_t* cio = it + it[H_OFF_T_CIO];
    const int32_t* x0 = it + it[H_OFF_T_STAGE];
    const int64_t base0 = x0[TS_BASE] & 0xFFFF, sb0 = x0[TS_SBOFFA];
    if (base0 >= nbase) return MPCASM_ERR_PLAN;
    auto is_u = [&](int64_t sid) {
      for (int64_t j = 0; j < gm; ++j)
        if (ids[j] == sid) return true;
      return false;
    };
    for (int64_t sx = 0; sx < nstage; ++sx) {
      const int32_t* x = it + it[H_OFF_T_STAGE] + sx * T_STAGE_WORDS;
      const int rows = x[TS_INFO] & 255, fl = (x[TS_INFO] >> 8) & 255;
      for (int side = 0; side < ((fl & TS_FLAG_P) ? 2 : 1); ++side) {
        if (!(fl & (side ? TS_FLAG_SIMPLE_B : TS_FLAG_SIMPLE_A))) return MPCASM_ERR_PLAN;
        const int64_t base = side ? (int64_t)((uint32_t)x[TS_BASE] >> 16) : (x[TS_BASE] & 0xFFFF);
        const int64_t sb = x[side ? TS_SBOFFB : TS_SBOFFA], U = x[side ? TS_UB : TS_UA];
        const int32_t* ks = srow + (sx * 2 + side) * 16;
        if (base >= nbase || sb < 0 || sb % (gm * 2 * gN) || sb / (gm * 2 * gN) >= gn || U != sb + ks[0])
          return MPCASM_ERR_PLAN;
        for (int i = 0; i < rows; ++i)
          if (ks[i] != ks[0] + i || sc[(sx * 2 + side) * 16 + i] != sc[(sx * 2 + side) * 16])
            return MPCASM_ERR_PLAN;
        for (int64_t c = 0; c < nop; ++c) {
          const int32_t* e = cio + (base * nop + c) * 2;
          const int32_t* e0 = cio + (base0 * nop + c) * 2;
          const int64_t sid = (uint32_t)e[1] >> 24, sid0 = (uint32_t)e0[1] >> 24;
          const bool valid = sid != T_SID_CONST;
          if (valid != (sid0 != T_SID_CONST)) return MPCASM_ERR_PLAN;
          if (!valid) continue;
          const int64_t part = (int64_t)(uint32_t)e[0] - sb;
          if (!is_u(sid) || ((e[1] << 8) >> 8) != 1 || part != (int64_t)(uint32_t)e0[0] - sb0 ||
              part + U < 0 || part + U + 15 + 3 >= tbn + 16)
            return MPCASM_ERR_PLAN;
        }
      }
    }
  }
  // scan form (toeplitz_scan_kernel): only beside the Toeplitz form; every index the kernel forms
  // from these tables stays inside TB, d, the parameters and the results
  {
    const int64_t K = it[H_T_SCAN], nblk = it[H_T_SCAN_NBLK], nrest = it[H_T_SCAN_NGREST];
    if (K < 0 || K > T_SCAN_KMAX || (K > 0 && !it[H_T_TOEPLITZ])) return MPCASM_ERR_PLAN;
    if (K > 0) {
      const int32_t* g = it + it[H_OFF_T_LTI];
      const int64_t gn = g[TL_N], gm = g[TL_M], gN = g[TL_HORIZON], per_state = gm * gN;
      const int64_t nparams = it[H_NPARAMS];
      if (gN < 1 || gN > T_SCAN_NMAX || nblk < 1 || nblk > T_SCAN_BLKMAX || nrest < 0 || nrest > nc ||
          !in_range(it[H_OFF_T_SCAN_BLK], nblk * 2, n, H_WORDS) ||
          !in_range(it[H_OFF_T_SCAN_GT], K * T_SCAN_GT_WORDS, n, H_WORDS) ||
          !in_range(it[H_OFF_T_SCAN_GROW], nc * 2, n, H_WORDS) || it[H_OFF_T_SCAN_GROW] % 2 ||
          !in_range(it[H_OFF_T_SCAN_GREST], nrest, n, H_WORDS) ||
          !in_range(it[H_OFF_T_SCAN_COLBLK], no, n, H_WORDS) ||
          !in_range(it[H_T_DOFF_SCAN_GC], K, nd, 0) || !in_range(it[H_T_DOFF_SCAN_GCOEF], nc, nd, 0))
        return MPCASM_ERR_PLAN;
      const int32_t* blk = it + it[H_OFF_T_SCAN_BLK];
      const int32_t* colblk = it + it[H_OFF_T_SCAN_COLBLK];
      int64_t covered = 0;
      for (int64_t b = 0; b < nblk; ++b) {
        const int64_t c0 = blk[2 * b], pbase = blk[2 * b + 1];
        if (c0 < 0 || c0 + gN > no || pbase < 0 || pbase % gN || pbase / gN >= gm)
          return MPCASM_ERR_PLAN;
        for (int64_t l = 0; l < gN; ++l)
          if (colblk[c0 + l] != b) return MPCASM_ERR_PLAN;  // (hence disjoint)
        covered += gN;
      }
      int64_t other = 0;
      for (int64_t c = 0; c < no; ++c) {
        if (colblk[c] < -1 || colblk[c] >= nblk) return MPCASM_ERR_PLAN;
        other += colblk[c] < 0;
      }
      if (other != it[H_T_SCAN_NOTHER] || covered + other != no) return MPCASM_ERR_PLAN;
      const int32_t* gt = it + it[H_OFF_T_SCAN_GT];
      for (int64_t k = 0; k < K; ++k) {
        const int32_t* x = gt + k * T_SCAN_GT_WORDS;
        if (x[SG_SBOFF] < 0 || x[SG_SBOFF] % per_state || x[SG_SBOFF] / per_state >= gn ||
            x[SG_WPARAM] < 0 || x[SG_WPARAM] >= nparams || x[SG_AIMPARAM] < 0 || x[SG_AIMPARAM] >= nparams ||
            x[SG_DROW] < 0 || x[SG_DROW] + gN > rtot)
          return MPCASM_ERR_PLAN;
      }
      const int32_t* sg = it + it[H_OFF_T_SCAN_GROW];
      const int32_t* rest = it + it[H_OFF_T_SCAN_GREST];
      int64_t r = 0;
      for (int64_t R = 0; R < nc; ++R) {
        const int64_t u = sg[2 * R], slot = sg[2 * R + 1];
        if (u == -1) {
          if (slot != -1 || r >= nrest || rest[r] != R) return MPCASM_ERR_PLAN;
          ++r;
          continue;
        }
        if (u < 0 || u / per_state >= gn || u % per_state >= gN || slot < 0 || slot >= nparams ||
            rrw[R * RS_RR_WORDS + RR_NAXES] != 1 || rrw[R * RS_RR_WORDS + RR_ARROW] != slot)
          return MPCASM_ERR_PLAN;
      }
      if (r != nrest) return MPCASM_ERR_PLAN;
      // fused set-up: `given` is the initial state, the K terms' rows tile the workspace (the kernel writes
      // d[first row + k] for every term and reads nothing else), no row of G through the column tables,
      // every input an unknown
      if (it[H_T_SCAN_FUSED] != 0) {
        if (it[H_T_SCAN_FUSED] != 1 || it[H_NG] != gn || nrest != 0 || rtot != K * gN || nblk != gm)
          return MPCASM_ERR_PLAN;
        std::vector<char> seen((size_t)K, 0);
        for (int64_t k = 0; k < K; ++k) {
          const int64_t dr = gt[k * T_SCAN_GT_WORDS + SG_DROW];
          if (dr % gN || seen[(size_t)(dr / gN)]) return MPCASM_ERR_PLAN;
          seen[(size_t)(dr / gN)] = 1;
        }
      }
    } else if (it[H_T_SCAN_FUSED] != 0) {
      return MPCASM_ERR_PLAN;
    }
  }
  {  // every row of G is written exactly once: riding on a stage or listed in the rest
    const int32_t* grest = it + it[H_OFF_T_GREST];
    for (int64_t i = 0; i < ngrest; ++i) {
      if (grest[i] < 0 || grest[i] >= nc || g_written[grest[i]]) return MPCASM_ERR_PLAN;
      g_written[grest[i]] = 1;
    }
    for (int64_t R = 0; R < nc; ++R)
      if (!g_written[R]) return MPCASM_ERR_PLAN;
  }
  for (int64_t R = 0; R < nc; ++R) {
    const int32_t* x = rrw + R * RS_RR_WORDS;
    if (x[RR_NAXES] < 0 || x[RR_NAXES] > RS_AXMAX || x[RR_EXTREME] < 0 ||
        x[RR_EXTREME] >= it[H_NPARAMS])
      return MPCASM_ERR_PLAN;
    for (int a = 0; a < RS_AXMAX; ++a) {
      const int row = grow[R * RS_AXMAX + a];
      if (a < x[RR_NAXES] ? (row < 0 || row >= rtot) : row != -1) return MPCASM_ERR_PLAN;
      if (x[RR_ARROW + a] < 0 || x[RR_ARROW + a] > it[H_NPARAMS] || x[RR_CENTER + a] < 0 ||
          x[RR_CENTER + a] > it[H_NPARAMS])
        return MPCASM_ERR_PLAN;
      if (a < x[RR_NAXES] && (x[RR_ARROW + a] >= it[H_NPARAMS] || x[RR_CENTER + a] >= it[H_NPARAMS]))
        return MPCASM_ERR_PLAN;
    }
  }
  return MPCASM_OK;
}

int validate_plan(const int32_t* it, const double* h_dtab, size_t n_itab, size_t n_dtab) {
  if (n_itab < (size_t)H_WORDS) return MPCASM_ERR_PLAN;
  if (it[H_MAGIC] != PLAN_MAGIC || it[H_VERSION] != PLAN_VERSION) return MPCASM_ERR_PLAN;
  if ((size_t)it[H_NITAB] != n_itab || (size_t)it[H_NDTAB] != n_dtab) return MPCASM_ERR_PLAN;
  const int64_t n = (int64_t)n_itab, nd = (int64_t)n_dtab;
  const int64_t ng = it[H_NG], no = it[H_NO], nc = it[H_NC];
  if (ng < 0 || no < 0 || nc < 0 || it[H_NPARAMS] < 0) return MPCASM_ERR_PLAN;
  if (it[H_NSRC] < 0 || it[H_NSRC] > MAX_SOURCES) return MPCASM_ERR_LIMIT;
  if (it[H_LDV] < no + 1 || (it[H_LDV] & 1)) return MPCASM_ERR_PLAN;
  const int64_t W = ng + no;
  bool ok = true;
  ok = ok && in_range(it[H_OFF_SEG], (int64_t)it[H_NSEG] * SEG_WORDS, n, H_WORDS);
  ok = ok && in_range(it[H_OFF_COLSEG], (int64_t)it[H_NBASE] * W, n, H_WORDS);
  ok = ok && in_range(it[H_OFF_ROWPTR], (int64_t)it[H_RTOT] + 1, n, H_WORDS);
  ok = ok && in_range(it[H_OFF_ENTBASE], it[H_NENT], n, H_WORDS);
  ok = ok && in_range(it[H_OFF_ENTK], it[H_NENT], n, H_WORDS);
  ok = ok && in_range(it[H_OFF_GTERM], (int64_t)it[H_NGTERM] * GT_WORDS, n, H_WORDS);
  ok = ok && in_range(it[H_OFF_LIMIT], (int64_t)it[H_NLIMIT] * LM_WORDS, n, H_WORDS);
  ok = ok && in_range(it[H_OFF_LAX], (int64_t)it[H_NLAX] * LX_WORDS, n, H_WORDS);
  ok = ok && in_range(it[H_OFF_ROWLIMIT], nc, n, H_WORDS);
  ok = ok && in_range(it[H_OFF_PM_ROWPTR], (int64_t)it[H_PMROWS] + 1, n, H_WORDS);
  ok = ok && in_range(it[H_OFF_PM_ENTBASE], it[H_PM_NENT], n, H_WORDS);
  ok = ok && in_range(it[H_OFF_PM_ENTK], it[H_PM_NENT], n, H_WORDS);
  ok = ok && in_range(it[H_DOFF_ENTCOEF], it[H_NENT], nd, 0);
  ok = ok && in_range(it[H_DOFF_PM_ENTCOEF], it[H_PM_NENT], nd, 0);
  ok = ok && in_range(it[H_DOFF_DIAGCOEF], it[H_NDIAGCOEF], nd, 0);
  if (it[H_FUSED_OK] != 0 && it[H_FUSED_OK] != 1) return MPCASM_ERR_PLAN;
  if (it[H_FUSED_OK]) {
    ok = ok && it[H_ARENA_TOTAL] >= 1 && it[H_NFD] >= 0 && it[H_NOPS] >= 0 && it[H_NCOEF] >= 0;
    ok = ok && in_range(it[H_OFF_ARENA], (int64_t)it[H_NSRC] * 2, n, H_WORDS);
    ok = ok && in_range(it[H_OFF_FD_IDX], it[H_NFD], n, H_WORDS);
    ok = ok && in_range(it[H_OFF_FD_PTR], (int64_t)it[H_NFD] + 1, n, H_WORDS);
    ok = ok && in_range(it[H_OFF_OP], (int64_t)it[H_NOPS] * 2, n, H_WORDS);
      ok = ok && in_range(it[H_DOFF_COEFPOOL], it[H_NCOEF], nd, 0);
  }
  if (!ok) return MPCASM_ERR_PLAN;
  if (it[H_FUSED_OK]) {
    const int32_t* ar = it + it[H_OFF_ARENA];
    for (int s = 0; s < it[H_NSRC]; ++s)
      if (ar[2 * s] < 1 || ar[2 * s + 1] < 0 ||
          (int64_t)ar[2 * s] + ar[2 * s + 1] > it[H_ARENA_TOTAL])
        return MPCASM_ERR_PLAN;
    const int64_t vsize = (int64_t)it[H_RTOT] * it[H_LDV];
    const int32_t* fi = it + it[H_OFF_FD_IDX];
    const int32_t* fp = it + it[H_OFF_FD_PTR];
    if (fp[0] != 0 || fp[it[H_NFD]] != it[H_NOPS]) return MPCASM_ERR_PLAN;
    for (int i = 0; i < it[H_NFD]; ++i)
      if (fi[i] < 0 || fi[i] >= vsize || fp[i + 1] < fp[i]) return MPCASM_ERR_PLAN;
    const uint32_t* op = reinterpret_cast<const uint32_t*>(it + it[H_OFF_OP]);
    for (int o = 0; o < it[H_NOPS]; ++o) {
      if (op[2 * o] >= (uint32_t)it[H_ARENA_TOTAL]) return MPCASM_ERR_PLAN;
      if ((op[2 * o + 1] >> 16) > (uint32_t)ng || (op[2 * o + 1] & 0xFFFFu) >= (uint32_t)it[H_NCOEF])
        return MPCASM_ERR_PLAN;
    }
  }

  if (it[H_PM_NFD] < 0) return MPCASM_ERR_PLAN;
  if (it[H_PM_NFD] > 0) {  // the element program of the preview matrices
    const int64_t nfd = it[H_PM_NFD], nops = it[H_PM_NOPS], npool = it[H_PM_NPOOL];
    const int64_t elems = (int64_t)it[H_PMROWS] * (ng + no);
    if (nops < nfd || npool < 1 || elems < nfd ||
        !in_range(it[H_OFF_PM_MAP], elems, n, H_WORDS) ||
        !in_range(it[H_OFF_PM_FDPTR], nfd + 1, n, H_WORDS) ||
        !in_range(it[H_OFF_PM_OP], nops * 2, n, H_WORDS) || it[H_OFF_PM_OP] % 2 ||
        !in_range(it[H_DOFF_PM_POOL], npool, nd, 0) ||
        !in_range(it[H_OFF_ARENA], (int64_t)it[H_NSRC] * 2, n, H_WORDS) || it[H_NSRC] > 254)
      return MPCASM_ERR_PLAN;
    const int32_t* mp = it + it[H_OFF_PM_MAP];
    for (int64_t e = 0; e < elems; ++e)
      if (mp[e] < -1 || mp[e] >= nfd) return MPCASM_ERR_PLAN;
    const int32_t* fp = it + it[H_OFF_PM_FDPTR];
    if (fp[0] != 0 || fp[nfd] != nops) return MPCASM_ERR_PLAN;
    for (int64_t i = 0; i < nfd; ++i)
      if (fp[i + 1] < fp[i]) return MPCASM_ERR_PLAN;
    const uint32_t* op = reinterpret_cast<const uint32_t*>(it + it[H_OFF_PM_OP]);
    const int32_t* ar = it + it[H_OFF_ARENA];
    for (int64_t o = 0; o < nops; ++o) {
      const uint32_t sid = op[2 * o + 1] & 255u, cid = op[2 * o + 1] >> 8;
      if (cid >= (uint64_t)npool) return MPCASM_ERR_PLAN;
      if (sid == 255u) {
        if (op[2 * o] != 0) return MPCASM_ERR_PLAN;
      } else if (sid >= (uint32_t)it[H_NSRC] || ar[2 * sid + 1] < 0 ||
                 op[2 * o] >= (uint32_t)ar[2 * sid + 1]) {
        return MPCASM_ERR_PLAN;
      }
    }
  }
  if (it[H_RS_OK] != 0 && it[H_RS_OK] != 1) return MPCASM_ERR_PLAN;
  if (it[H_RS_OK]) {
    if (!it[H_FUSED_OK]) return MPCASM_ERR_PLAN;
    const int64_t jc = it[H_RS_JC], slots = jc * RS_NT;
    const int64_t img = it[H_RS_IMG], unit = it[H_RS_UNIT], nchunk = it[H_RS_NCHUNK];
    if (jc < 1 || jc > RS_JC_MAX || it[H_RS_NTRIP] < 0 || it[H_RS_NSPLIT] < 0)
      return MPCASM_ERR_PLAN;
    const int64_t dma = it[H_RS_IMG_DMA], nlti = it[H_RS_NLTI], nab = it[H_RS_AB];
    if ((unit != 4 && unit != 16) || dma < 128 || dma % 128 || img < dma || (img & 1) ||
        img > 65535 || nchunk * 64 * unit != dma * 8 || it[H_NPARAMS] > 65535 || nlti < 0 ||
        nlti > RS_LTI_MAX || nab < 0 || nab % 32 || nab > 4096 || (nlti == 0) != (nab == 0))
      return MPCASM_ERR_PLAN;
    bool r = true;
    r = r && in_range(it[H_OFF_RS_SRC], slots, n, H_WORDS);
    r = r && in_range(it[H_OFF_RS_GIDX], slots, n, H_WORDS);
    r = r && in_range(it[H_OFF_RS_DST], slots, n, H_WORDS);
    r = r && in_range(it[H_DOFF_RS_COEF], slots, nd, 0);
    // (two spare records behind the table: the kernel reads that far ahead)
    r = r && in_range(it[H_OFF_RS_TRIP], ((int64_t)it[H_RS_NTRIP] + 2) * RS_TRIP_WORDS, n, H_WORDS);
    r = r && in_range(it[H_OFF_RS_WTRIP], RS_WAVES * 2, n, H_WORDS);
    r = r && in_range(it[H_OFF_RS_SPLIT], it[H_RS_NSPLIT], n, H_WORDS);
    r = r && it[H_RS_NZBLK] >= 0 && in_range(it[H_OFF_RS_ZBLK], it[H_RS_NZBLK], n, H_WORDS);
    r = r && (it[H_OFF_RS_TRIP] % 8 == 0);
    r = r && in_range(it[H_OFF_RS_RR], nc * RS_RR_WORDS, n, H_WORDS) && it[H_OFF_RS_RR] % 4 == 0;
    r = r && in_range(it[H_OFF_RS_INMETA], nchunk * 64 * 2, n, H_WORDS) &&
        it[H_OFF_RS_INMETA] % 2 == 0;
    r = r && in_range(it[H_DOFF_RS_CONST], 4, nd, 0) && it[H_DOFF_RS_CONST] % 2 == 0;
    r = r && in_range(it[H_RS_IMG_GIVEN], ng + 1, dma, 0);
    r = r && in_range(it[H_RS_IMG_PARAMS], (int64_t)it[H_NPARAMS] + 1, dma, 0);
    r = r && in_range(it[H_OFF_RS_LTI], nlti * RS_LTI_WORDS, n, H_WORDS);
    r = r && in_range(it[H_OFF_RS_ABMETA], nab * 4, n, H_WORDS) && it[H_OFF_RS_ABMETA] % 2 == 0;
    if (!r) return MPCASM_ERR_PLAN;
    {
      const double* c = h_dtab + it[H_DOFF_RS_CONST];
      if (c[0] != 1.0 || c[1] != 1.0 || c[2] != 0.0 || c[3] != 0.0) return MPCASM_ERR_PLAN;
    }
    // the workspace's geometry (plan_tables.h H_RS_COMPACT): dense = the fused kernel's, or compact
    const int64_t vldv = it[H_RS_LDV], vd = it[H_RS_VD], vrow0 = it[H_RS_VROW0];
    const int64_t vrows = vrow0 + it[H_RTOT];
    if (it[H_RS_COMPACT] == 0) {
      if (vldv != it[H_LDV] || vd != no || vrow0 != 0) return MPCASM_ERR_PLAN;
    } else {
      if (it[H_RS_COMPACT] != 1 || vldv < 6 || vldv % 4 != 2 || vd != vldv - 2 || vd > no + 3 || vrow0 < 0 ||
          vrow0 % 4 || vrow0 > 1024 || !it[H_RR_PACKED] || it[H_CSC_PNNZ] != 0 || it[H_CSC_GNNZ] != 0 ||
          !in_range(it[H_OFF_RS_RRWIN], nc, n, H_WORDS))
        return MPCASM_ERR_PLAN;
    }
    const int64_t vsize = vrows * vldv;
    const int32_t* ts = it + it[H_OFF_RS_SRC];
    const int32_t* tg = it + it[H_OFF_RS_GIDX];
    const int32_t* td = it + it[H_OFF_RS_DST];
    for (int64_t i = 0; i < slots; ++i)
      if (ts[i] < 0 || ts[i] >= img || tg[i] < 0 || tg[i] >= img || td[i] < -1 ||
          (td[i] >= 0 && (td[i] & ~RS_DST_ACC) >= vsize))
        return MPCASM_ERR_PLAN;
    {  // the packed copy of the program says what the four tables say
      if (!in_range(it[H_OFF_RS_PROG], slots * 4, n, H_WORDS) || it[H_OFF_RS_PROG] % 4) return MPCASM_ERR_PLAN;
      const int32_t* pr = it + it[H_OFF_RS_PROG];
      const double* tc = h_dtab + it[H_DOFF_RS_COEF];
      for (int64_t i = 0; i < slots; ++i) {
        double c;
        memcpy(&c, pr + 4 * i + 2, sizeof c);
        if ((uint32_t)pr[4 * i] != ((uint32_t)ts[i] | ((uint32_t)tg[i] << 16)) || pr[4 * i + 1] != td[i] ||
            memcmp(&c, &tc[i], sizeof c) != 0)
          return MPCASM_ERR_PLAN;
      }
    }
    const int32_t* sp = it + it[H_OFF_RS_SPLIT];
    for (int i = 0; i < it[H_RS_NSPLIT]; ++i)
      if (sp[i] < 0 || sp[i] >= vsize) return MPCASM_ERR_PLAN;
    const int nb = ((int)no + 3) / 4;  // 4-column blocks of the unknowns
    if (nb > RS_BLOCKS_MAX) return MPCASM_ERR_PLAN;
    for (int i = 0; i < it[H_RS_NZBLK]; ++i) {
      const int z = (it + it[H_OFF_RS_ZBLK])[i];
      if (z < 0 || (z >> 8) >= nb || (z & 255) >= nb) return MPCASM_ERR_PLAN;
    }
    const int32_t* tr = it + it[H_OFF_RS_TRIP];
    for (int i = 0; i < 2 * RS_TRIP_WORDS; ++i)
      if (tr[it[H_RS_NTRIP] * RS_TRIP_WORDS + i] != 0) return MPCASM_ERR_PLAN;
    const int64_t row_bytes = vldv * 8;
    if (it[H_LDV] < no + 2) return MPCASM_ERR_PLAN;  // columns: unknowns, d, ones
    for (int i = 0; i < it[H_RS_NTRIP]; ++i) {
      const int32_t* x = tr + i * RS_TRIP_WORDS;
      const int word = x[RT_WORD], rows = word & 31;
      const int live = (word >> RT_LIVE) & 15, qmask = (word >> RT_QMASK) & 15;
      const bool is_short = (word >> RT_SHORT) & 1;
      if (word < 0 || (word >> 19) != 0 || (rows != 0 && rows != 4 && rows != 16) ||
          (qmask & ~live) || live == 0 || (is_short != (rows == 4)) ||
          (rows == 0 && ((word >> RT_TERM_END) & 1)) ||
          x[RT_W] < 0 || x[RT_W] % 8 || x[RT_W] / 8 >= it[H_NPARAMS] || x[RT_AIM] < 0 ||
          x[RT_AIM] % 8 || x[RT_AIM] / 8 >= it[H_NPARAMS])
        return MPCASM_ERR_PLAN;
      for (int g = 0; g < 4; ++g)  // (all four groups load, live or not)
        if (((x[RT_BI] >> (8 * g)) & 255) >= nb || ((x[RT_BJ] >> (8 * g)) & 255) >= nb)
          return MPCASM_ERR_PLAN;
      // offsets: whole rows on group boundaries (the d offset: column RS_VD of its row); a trip
      // reads `rows` rows from each
      if (it[H_RS_COMPACT] == 0) {
        const int64_t offs[3] = {x[RT_A], x[RT_B], (int64_t)x[RT_D] - (rows > 0 ? no * 8 : 0)};
        for (int k = 0; k < 3; ++k)
          if (offs[k] < 0 || offs[k] % (4 * row_bytes) || offs[k] / row_bytes + rows > it[H_RTOT])
            return MPCASM_ERR_PLAN;
      } else if (rows > 0) {
        // compact: the offsets carry the window's first column; what every group really reads -- 32
        // bytes at its block in each of the rows -- stays inside the workspace
        const int64_t d = (int64_t)x[RT_D] - vd * 8;
        if (d < 0 || d % (4 * row_bytes) || d / row_bytes + rows > vrows) return MPCASM_ERR_PLAN;
        for (int g = 0; g < 4; ++g) {
          const int64_t a = (int64_t)x[RT_A] + 32 * ((x[RT_BI] >> (8 * g)) & 255);
          const int64_t b = (int64_t)x[RT_B] + 32 * ((x[RT_BJ] >> (8 * g)) & 255);
          if (a < 0 || a % 32 || a + (rows - 1) * row_bytes + 32 > vsize * 8) return MPCASM_ERR_PLAN;
          if (!((qmask >> g) & 1) && (b < 0 || b % 32 || b + (rows - 1) * row_bytes + 32 > vsize * 8))
            return MPCASM_ERR_PLAN;
        }
      }
    }
    {  // every wavefront's trips: consecutive, whole packs (first ... last)
      const int32_t* wt = it + it[H_OFF_RS_WTRIP];
      int next = 0;
      for (int w = 0; w < RS_WAVES; ++w) {
        if (wt[2 * w] != next || wt[2 * w + 1] < 0) return MPCASM_ERR_PLAN;
        next += wt[2 * w + 1];
        if (next > it[H_RS_NTRIP]) return MPCASM_ERR_PLAN;
        const int32_t* open = nullptr;
        for (int i = wt[2 * w]; i < next; ++i) {
          const int32_t* x = tr + i * RS_TRIP_WORDS;
          const int word = x[RT_WORD];
          if ((word >> RT_FIRST) & 1) {
            if (open != nullptr) return MPCASM_ERR_PLAN;
            open = x;
          }
          if (open == nullptr || open[RT_BI] != x[RT_BI] || open[RT_BJ] != x[RT_BJ] ||
              (((open[RT_WORD] ^ word) >> RT_LIVE) & 0xFF) != 0)   // live groups, groups of q
            return MPCASM_ERR_PLAN;
          if ((word >> RT_LAST) & 1) open = nullptr;
        }
        if (open != nullptr) return MPCASM_ERR_PLAN;
      }
      if (next != it[H_RS_NTRIP]) return MPCASM_ERR_PLAN;
    }
    const int32_t* rrw = it + it[H_OFF_RS_RR];
    for (int64_t R = 0; R < nc; ++R) {
      const int32_t* x = rrw + R * RS_RR_WORDS;
      if (x[RR_NAXES] < 0 || x[RR_NAXES] > RS_AXMAX || x[RR_EXTREME] < 0 ||
          x[RR_EXTREME] >= it[H_NPARAMS])
        return MPCASM_ERR_PLAN;
      for (int a = 0; a < RS_AXMAX; ++a)
        if (x[RR_VOFF + a] < 0 || x[RR_VOFF + a] % vldv != 0 ||
            x[RR_VOFF + a] / vldv >= std::max<int64_t>(vrows, 1) || x[RR_ARROW + a] < 0 ||
            x[RR_ARROW + a] > it[H_NPARAMS] || x[RR_CENTER + a] < 0 ||
            x[RR_CENTER + a] > it[H_NPARAMS])
          return MPCASM_ERR_PLAN;
      // the packed words the 16-byte-piece path of G reads instead of the fields above
      if (it[H_RR_PACKED] &&
          (x[RR_NAXES] > 2 || x[RR_VOFF] > 65535 || x[RR_VOFF + 1] > 65535 ||
           x[RR_ARROW] > 65535 || x[RR_ARROW + 1] > 65535 ||
           (uint32_t)x[RR_PACKED] != ((uint32_t)x[RR_VOFF] | ((uint32_t)x[RR_VOFF + 1] << 16)) ||
           (uint32_t)x[RR_PACKED + 1] != ((uint32_t)x[RR_ARROW] | ((uint32_t)x[RR_ARROW + 1] << 16))))
        return MPCASM_ERR_PLAN;
    }
    if (it[H_RR_PACKED] != 0 && (it[H_RR_PACKED] != 1 || (no & 1) || nc < 1)) return MPCASM_ERR_PLAN;
    if (it[H_RS_COMPACT]) {  // a window lies inside its row: first + count column pairs <= RS_VD / 2
      const int32_t* win = it + it[H_OFF_RS_RRWIN];
      for (int64_t R = 0; R < nc; ++R)
        for (int a = 0; a < 2; ++a) {
          const uint32_t w = ((uint32_t)win[R] >> (16 * a)) & 0xFFFF;
          if (2 * (int64_t)(w >> 8) > vd) return MPCASM_ERR_PLAN;
        }
    }
    {  // the per-column tables of the diagonal gterms and the piece descriptors of G
      if (!in_range(it[H_OFF_RS_DPAR], no * 2 * RS_DIAG_MAX, n, H_WORDS) || it[H_OFF_RS_DPAR] % 4 ||
          !in_range(it[H_DOFF_RS_DCOEF], no * RS_DIAG_MAX, nd, 0) || it[H_DOFF_RS_DCOEF] % 2)
        return MPCASM_ERR_PLAN;
      const int32_t* dp = it + it[H_OFF_RS_DPAR];
      for (int64_t i = 0; i < no * 2 * RS_DIAG_MAX; ++i)
        if (dp[i] < 0 || dp[i] > it[H_NPARAMS]) return MPCASM_ERR_PLAN;
      const int64_t ngd = it[H_RS_NGDESC];
      if (ngd != 0) {
        const int64_t pieces = nc * (no / 2);
        if (it[H_RS_GSINGLE] < 0 || (it[H_RS_GSINGLE] >> (RS_GDESC_PIECES * (RS_GDESC_THREADS / 64))) != 0)
          return MPCASM_ERR_PLAN;
        if (!it[H_RR_PACKED] || ngd != (int64_t)RS_GDESC_PIECES * RS_GDESC_THREADS || pieces > ngd ||
            !in_range(it[H_OFF_RS_GDESC], ngd * 2, n, H_WORDS) || it[H_OFF_RS_GDESC] % 2)
          return MPCASM_ERR_PLAN;
        const int32_t* gd = it + it[H_OFF_RS_GDESC];
        for (int64_t e = 0; e < ngd; ++e) {  // the same numbers as the row record of the piece
          const int64_t R = e < pieces ? e / (no / 2) : 0, cp = e < pieces ? e % (no / 2) : 0;
          const int32_t* x = rrw + R * RS_RR_WORDS;
          uint32_t v0 = (uint32_t)x[RR_VOFF] + 2 * cp, v1 = (uint32_t)x[RR_VOFF + 1] + 2 * cp;
          uint32_t a0 = (uint32_t)x[RR_ARROW], a1 = (uint32_t)x[RR_ARROW + 1];
          if (it[H_RS_COMPACT]) {  // ... the piece inside the axis' window, or row 0 and the zero arrow
            const uint32_t win = (uint32_t)(it + it[H_OFF_RS_RRWIN])[R];
            const uint32_t d0 = (uint32_t)cp - (win & 255), d1 = (uint32_t)cp - ((win >> 16) & 255);
            const bool in0 = d0 < ((win >> 8) & 255), in1 = d1 < (win >> 24);
            v0 = in0 ? (uint32_t)x[RR_VOFF] + 2 * d0 : 0;
            a0 = in0 ? a0 : (uint32_t)it[H_NPARAMS];
            v1 = in1 ? (uint32_t)x[RR_VOFF + 1] + 2 * d1 : (x[RR_NAXES] >= 2 ? 0 : (uint32_t)x[RR_VOFF + 1]);
            a1 = in1 || x[RR_NAXES] < 2 ? a1 : (uint32_t)it[H_NPARAMS];
          }
          const bool as_is = (uint32_t)gd[2 * e] == (v0 | (v1 << 16)) && (uint32_t)gd[2 * e + 1] == (a0 | (a1 << 16));
          const bool swapped = (uint32_t)gd[2 * e] == (v1 | (v0 << 16)) && (uint32_t)gd[2 * e + 1] == (a1 | (a0 << 16));
          if (!as_is && !swapped) return MPCASM_ERR_PLAN;
        }
        // per thread, the second axis of at most one of its pieces: the same numbers as the
        // second half of that piece's descriptor
        const int64_t nfix = it[H_RS_NGFIX];
        if (nfix < 0 || nfix > RS_GDESC_THREADS) return MPCASM_ERR_PLAN;
        if (nfix != 0) {
          if (!in_range(it[H_OFF_RS_GFIX], RS_GDESC_THREADS * 2, n, H_WORDS) || it[H_OFF_RS_GFIX] % 2)
            return MPCASM_ERR_PLAN;
          const int32_t* gf = it + it[H_OFF_RS_GFIX];
          int64_t seen = 0;
          for (int64_t t = 0; t < RS_GDESC_THREADS; ++t) {
            const uint32_t w = (uint32_t)gf[2 * t], u = w >> 16, arrow = (uint32_t)gf[2 * t + 1];
            if (u == (uint32_t)RS_GFIX_NONE) {
              if ((w & 0xFFFF) != 0 || arrow != (uint32_t)it[H_NPARAMS]) return MPCASM_ERR_PLAN;
              continue;
            }
            const int64_t e = t + (int64_t)u * RS_GDESC_THREADS;
            if (u >= (uint32_t)RS_GDESC_PIECES || e >= pieces || (w & 0xFFFF) != ((uint32_t)gd[2 * e] >> 16) ||
                arrow != ((uint32_t)gd[2 * e + 1] >> 16))
              return MPCASM_ERR_PLAN;
            ++seen;
          }
          if (seen != nfix) return MPCASM_ERR_PLAN;
        }
      } else if (it[H_RS_NGFIX] != 0) {
        return MPCASM_ERR_PLAN;
      }
    }
    if (it[H_CSC_PNNZ] != 0 || it[H_CSC_GNNZ] != 0) {  // the CSC hand-off tables
      const int64_t pn = it[H_CSC_PNNZ], gn = it[H_CSC_GNNZ], ldp = no + (no & 1);
      if (gn * nc > (1ll << 27)) return MPCASM_ERR_PLAN;  // (far beyond what fits on chip)
      if (pn < 0 || gn < 0 || pn > no * no || gn > nc * no || (it[H_CSC_GSINGLE] & ~1) ||
          !in_range(it[H_OFF_CSC_P], pn, n, H_WORDS) || !in_range(it[H_OFF_CSC_G], gn * 2, n, H_WORDS) ||
          it[H_OFF_CSC_G] % 2)
        return MPCASM_ERR_PLAN;
      const int32_t* cp = it + it[H_OFF_CSC_P];
      for (int64_t k = 0; k < pn; ++k)
        if (cp[k] < 0 || cp[k] / ldp >= no || cp[k] % ldp >= no) return MPCASM_ERR_PLAN;
      // an entry of G: the numbers of some row's record, its axes in either order, one column
      const int32_t* cg = it + it[H_OFF_CSC_G];
      for (int64_t k = 0; k < gn; ++k) {
        const uint32_t w0 = (uint32_t)cg[2 * k], w1 = (uint32_t)cg[2 * k + 1];
        bool found = false;
        for (int64_t R = 0; R < nc && !found; ++R) {
          const int32_t* x = rrw + R * RS_RR_WORDS;
          if (x[RR_NAXES] > 2) return MPCASM_ERR_PLAN;
          for (int sw = 0; sw < 2 && !found; ++sw) {
            const uint32_t v0 = (uint32_t)x[RR_VOFF + sw], v1 = (uint32_t)x[RR_VOFF + 1 - sw];
            const uint32_t a0 = (uint32_t)x[RR_ARROW + sw], a1 = (uint32_t)x[RR_ARROW + 1 - sw];
            if (w1 != (a0 | (a1 << 16))) continue;
            const uint32_t c = (w0 & 0xFFFF) - v0;
            found = (w0 & 0xFFFF) >= v0 && c < (uint32_t)no && (w0 >> 16) == v1 + c && v1 + c < 65536;
          }
        }
        if (!found) return MPCASM_ERR_PLAN;
      }
    }
    for (int64_t g = 0; g < nlti; ++g) {  // generated groups: loaded A, B and the tables
      const int32_t* x = it + it[H_OFF_RS_LTI] + g * RS_LTI_WORDS;
      const int64_t gn = x[LT_N], gm = x[LT_M], gN = x[LT_HORIZON];
      if (gn < 1 || gm < 1 || gN < 1 || gn > 64 || gm > 64 || gN > 4096) return MPCASM_ERR_PLAN;
      int stages = 0;
      while ((1ll << stages) < gN) ++stages;
      if (!in_range(x[LT_A], gn * gn, nab, 0) || !in_range(x[LT_B], gn * gm, nab, 0) ||
          !in_range(x[LT_TA], gN * gn * gn, img, dma) || !in_range(x[LT_TB], gN * gn * gm, img, dma) ||
          !in_range(x[LT_TP], (stages + 1) * gn * gn, img, dma))
        return MPCASM_ERR_PLAN;
    }
    // every input load stays inside its stream: sources, given, params, the constants
    auto loads_ok = [&](const int32_t* im, int64_t lanes, int64_t bytes) {
      for (int64_t i = 0; i < lanes; ++i) {
        const int st = im[2 * i];
        const int64_t off = im[2 * i + 1];
        if (st < 0 || st > it[H_NSRC] + 2 || off < 0 || off % bytes) return false;
        const int64_t lim = st < it[H_NSRC]        ? (it + it[H_OFF_ARENA])[2 * st + 1]
                            : st == it[H_NSRC]     ? ng
                            : st == it[H_NSRC] + 1 ? (int64_t)it[H_NPARAMS]
                                                   : 4;
        if (off + bytes > lim * 8) return false;
      }
      return true;
    };
    if (!loads_ok(it + it[H_OFF_RS_INMETA], nchunk * 64, unit) ||
        !loads_ok(it + it[H_OFF_RS_ABMETA], nab * 2, 4))
      return MPCASM_ERR_PLAN;
    for (int c = 0; c < no; ++c) {  // diagonal gterms on one column: RS_DIAG_MAX slots
      int on = 0;
      for (int g = 0; g < it[H_NGTERM]; ++g) {
        const int32_t* r = it + it[H_OFF_GTERM] + g * GT_WORDS;
        if ((r[GT_FLAGS] & GT_FLAG_DIAG) && c >= r[GT_AOFF] && c < r[GT_AOFF] + r[GT_NROWS]) ++on;
      }
      if (on > RS_DIAG_MAX) return MPCASM_ERR_PLAN;
    }
  }

  {
    const int vrc = validate_tiled(it, h_dtab, n, nd);
    if (vrc != MPCASM_OK) return vrc;
    const int src = validate_sweep(it, n, nd);
    if (src != MPCASM_OK) return src;
  }
  // content checks: every index a kernel dereferences stays inside its table
  const int32_t* seg = it + it[H_OFF_SEG];
  for (int s = 0; s < it[H_NSEG]; ++s) {
    const int32_t* r = seg + s * SEG_WORDS;
    if (r[SEG_KIND] != SEG_KIND_GATHER && r[SEG_KIND] != SEG_KIND_IDENTITY) return MPCASM_ERR_PLAN;
    if (r[SEG_KIND] == SEG_KIND_GATHER && (r[SEG_SRC] < 0 || r[SEG_SRC] >= it[H_NSRC]))
      return MPCASM_ERR_PLAN;
    if (r[SEG_DST0] < 0 || r[SEG_LEN] < 0 || r[SEG_DST0] + r[SEG_LEN] > W) return MPCASM_ERR_PLAN;
  }
  const int32_t* colseg = it + it[H_OFF_COLSEG];
  for (int64_t i = 0; i < (int64_t)it[H_NBASE] * W; ++i)
    if (colseg[i] < -1 || colseg[i] >= it[H_NSEG]) return MPCASM_ERR_PLAN;
  auto check_csr = [&](int off_ptr, int rows, int off_base, int nent) {
    const int32_t* rp = it + off_ptr;
    if (rp[0] != 0 || rp[rows] != nent) return false;
    for (int r = 0; r < rows; ++r)
      if (rp[r + 1] < rp[r]) return false;
    const int32_t* eb = it + off_base;
    for (int e = 0; e < nent; ++e)
      if (eb[e] < 0 || eb[e] >= it[H_NBASE]) return false;
    return true;
  };
  if (!check_csr(it[H_OFF_ROWPTR], it[H_RTOT], it[H_OFF_ENTBASE], it[H_NENT])) return MPCASM_ERR_PLAN;
  if (!check_csr(it[H_OFF_PM_ROWPTR], it[H_PMROWS], it[H_OFF_PM_ENTBASE], it[H_PM_NENT]))
    return MPCASM_ERR_PLAN;
  const int32_t* gt = it + it[H_OFF_GTERM];
  for (int g = 0; g < it[H_NGTERM]; ++g) {
    const int32_t* r = gt + g * GT_WORDS;
    const int nr = r[GT_NROWS];
    if (r[GT_WPARAM] < 0 || r[GT_WPARAM] >= it[H_NPARAMS]) return MPCASM_ERR_PLAN;
    if (r[GT_AIMPARAM] < 0 || r[GT_AIMPARAM] >= it[H_NPARAMS]) return MPCASM_ERR_PLAN;
    if (r[GT_FLAGS] & GT_FLAG_DIAG) {  // columns c0 .. c0+nr-1 of the unknowns, coefficient list
      if ((r[GT_FLAGS] & GT_FLAG_P) || nr < 0 || r[GT_AOFF] < 0 || r[GT_AOFF] + nr > no ||
          r[GT_BOFF] < 0 || r[GT_BOFF] + nr > it[H_NDIAGCOEF])
        return MPCASM_ERR_PLAN;
      continue;
    }
    if (nr < 0 || r[GT_AOFF] < 0 || r[GT_AOFF] + nr > it[H_RTOT]) return MPCASM_ERR_PLAN;
    if (r[GT_DOFF] < 0 || r[GT_DOFF] + nr > it[H_RTOT]) return MPCASM_ERR_PLAN;
    if ((r[GT_FLAGS] & GT_FLAG_P) && (r[GT_BOFF] < 0 || r[GT_BOFF] + nr > it[H_RTOT]))
      return MPCASM_ERR_PLAN;
    if (r[GT_WPARAM] < 0 || r[GT_WPARAM] >= it[H_NPARAMS]) return MPCASM_ERR_PLAN;
    if (r[GT_AIMPARAM] < 0 || r[GT_AIMPARAM] >= it[H_NPARAMS]) return MPCASM_ERR_PLAN;
  }
  const int32_t* lim = it + it[H_OFF_LIMIT];
  const int32_t* lax = it + it[H_OFF_LAX];
  const int32_t* rowlimit = it + it[H_OFF_ROWLIMIT];
  for (int l = 0; l < it[H_NLIMIT]; ++l) {
    const int32_t* r = lim + l * LM_WORDS;
    const int nr = r[LM_NROWS], na = r[LM_NAXES];
    if (nr < 0 || na < 0 || r[LM_OUT0] < 0 || r[LM_OUT0] + nr > nc) return MPCASM_ERR_PLAN;
    if (r[LM_LAX0] < 0 || r[LM_LAX0] + na > it[H_NLAX]) return MPCASM_ERR_PLAN;
    const int ar = r[LM_ARROW_ROWS], cr = r[LM_CENTER_ROWS], er = r[LM_EXTREME_ROWS];
    if ((ar != 1 && ar != nr) || (cr != 1 && cr != nr) || (er != 1 && er != nr))
      return MPCASM_ERR_PLAN;
    if (r[LM_ARROW_P] < 0 || r[LM_ARROW_P] + ar * na > it[H_NPARAMS]) return MPCASM_ERR_PLAN;
    if (r[LM_CENTER_P] < 0 || r[LM_CENTER_P] + cr * na > it[H_NPARAMS]) return MPCASM_ERR_PLAN;
    if (r[LM_EXTREME_P] < 0 || r[LM_EXTREME_P] + er > it[H_NPARAMS]) return MPCASM_ERR_PLAN;
    for (int a = 0; a < na; ++a) {
      const int32_t* x = lax + (r[LM_LAX0] + a) * LX_WORDS;
      if ((x[LX_ROWS] != 1 && x[LX_ROWS] != nr) || x[LX_ROWOFF] < 0 ||
          x[LX_ROWOFF] + x[LX_ROWS] > it[H_RTOT])
        return MPCASM_ERR_PLAN;
    }
    for (int k = 0; k < nr; ++k)
      if (rowlimit[r[LM_OUT0] + k] != l) return MPCASM_ERR_PLAN;
  }
  return MPCASM_OK;
}

int make_src_table(const mpcasm_plan* plan, const double* const* h_src,
                   const int64_t* h_src_stride, SrcTable* out) {
  memset(out, 0, sizeof(*out));
  for (int s = 0; s < plan->dev.nsrc; ++s) {
    if (!h_src[s] || h_src_stride[s] < 0) return MPCASM_ERR_ARG;
    out->ptr[s] = h_src[s];
    out->stride[s] = h_src_stride[s];
  }
  return MPCASM_OK;
}

}  // namespace

namespace {
// the int fields of the device-side view, from the (validated) header and tables on the host
void plan_dev_from_tables(const int32_t* it, PlanDev* out) {
  PlanDev& d = *out;
  d.ng = it[H_NG]; d.no = it[H_NO]; d.nc = it[H_NC]; d.nparams = it[H_NPARAMS];
  d.nsrc = it[H_NSRC]; d.nbase = it[H_NBASE]; d.nseg = it[H_NSEG]; d.rtot = it[H_RTOT];
  d.nent = it[H_NENT]; d.ngterm = it[H_NGTERM]; d.nlimit = it[H_NLIMIT]; d.nlax = it[H_NLAX];
  d.pmrows = it[H_PMROWS]; d.pm_nent = it[H_PM_NENT]; d.ldv = it[H_LDV];
  d.off_seg = it[H_OFF_SEG]; d.off_colseg = it[H_OFF_COLSEG]; d.off_rowptr = it[H_OFF_ROWPTR];
  d.off_entbase = it[H_OFF_ENTBASE]; d.off_entk = it[H_OFF_ENTK]; d.off_gterm = it[H_OFF_GTERM];
  d.off_limit = it[H_OFF_LIMIT]; d.off_lax = it[H_OFF_LAX]; d.off_rowlimit = it[H_OFF_ROWLIMIT];
  d.off_pm_rowptr = it[H_OFF_PM_ROWPTR]; d.off_pm_entbase = it[H_OFF_PM_ENTBASE];
  d.off_pm_entk = it[H_OFF_PM_ENTK];
  d.doff_entcoef = it[H_DOFF_ENTCOEF]; d.doff_pm_entcoef = it[H_DOFF_PM_ENTCOEF];
  d.fused_ok = it[H_FUSED_OK]; d.arena_total = it[H_ARENA_TOTAL]; d.off_arena = it[H_OFF_ARENA];
  d.nfd = it[H_NFD]; d.off_fd_idx = it[H_OFF_FD_IDX]; d.off_fd_ptr = it[H_OFF_FD_PTR];
  d.nops = it[H_NOPS]; d.off_op = it[H_OFF_OP]; d.ncoef = it[H_NCOEF];
  d.doff_coefpool = it[H_DOFF_COEFPOOL];
  d.rs_ok = it[H_RS_OK]; d.rs_jc = it[H_RS_JC]; d.rs_sym = it[H_RS_SYM];
  d.rs_ntrip = it[H_RS_NTRIP]; d.off_rs_src = it[H_OFF_RS_SRC]; d.off_rs_gidx = it[H_OFF_RS_GIDX];
  d.off_rs_dst = it[H_OFF_RS_DST]; d.doff_rs_coef = it[H_DOFF_RS_COEF];
  d.off_rs_trip = it[H_OFF_RS_TRIP]; d.off_rs_wtrip = it[H_OFF_RS_WTRIP];
  d.rs_nsplit = it[H_RS_NSPLIT]; d.off_rs_split = it[H_OFF_RS_SPLIT];
  d.off_rs_rr = it[H_OFF_RS_RR]; d.rs_unit = it[H_RS_UNIT]; d.rs_nchunk = it[H_RS_NCHUNK];
  d.off_rs_inmeta = it[H_OFF_RS_INMETA]; d.rs_img = it[H_RS_IMG];
  d.rs_img_given = it[H_RS_IMG_GIVEN]; d.rs_img_params = it[H_RS_IMG_PARAMS];
  d.doff_rs_const = it[H_DOFF_RS_CONST];
  d.rs_nlti = it[H_RS_NLTI]; d.off_rs_lti = it[H_OFF_RS_LTI]; d.rs_img_dma = it[H_RS_IMG_DMA];
  d.rs_ab = it[H_RS_AB]; d.off_rs_abmeta = it[H_OFF_RS_ABMETA];
  d.rr_packed = it[H_RR_PACKED];
  d.off_rs_dpar = it[H_OFF_RS_DPAR]; d.doff_rs_dcoef = it[H_DOFF_RS_DCOEF];
  d.rs_ngdesc = it[H_RS_NGDESC]; d.off_rs_gdesc = it[H_OFF_RS_GDESC];
  d.rs_nzblk = it[H_RS_NZBLK]; d.off_rs_zblk = it[H_OFF_RS_ZBLK];
  d.rs_gsingle = it[H_RS_GSINGLE];
  d.csc_pnnz = it[H_CSC_PNNZ]; d.off_csc_p = it[H_OFF_CSC_P];
  d.csc_gnnz = it[H_CSC_GNNZ]; d.off_csc_g = it[H_OFF_CSC_G];
  d.csc_gsingle = it[H_CSC_GSINGLE];
  d.t_ci_ok = it[H_T_CI_OK]; d.t_nop = it[H_T_NOP]; d.off_t_cig = it[H_OFF_T_CIG];
  d.off_t_cio = it[H_OFF_T_CIO]; d.t_doff_delta = it[H_T_DOFF_DELTA]; d.t_ok = it[H_T_OK];
  d.t_nstage = it[H_T_NSTAGE]; d.off_t_stage = it[H_OFF_T_STAGE]; d.t_nlti = it[H_T_NLTI];
  d.off_t_lti = it[H_OFF_T_LTI]; d.off_t_lti_ids = it[H_OFF_T_LTI_IDS]; d.t_work = it[H_T_WORK];
  d.off_t_grow = it[H_OFF_T_GROW];
  d.off_t_srow = it[H_OFF_T_SROW]; d.t_doff_scoef = it[H_T_DOFF_SCOEF]; d.off_t_pig = it[H_OFF_T_PIG];
  d.t_ngrest = it[H_T_NGREST]; d.off_t_grest = it[H_OFF_T_GREST]; d.off_t_brow0 = it[H_OFF_T_BROW0]; d.t_toeplitz = it[H_T_TOEPLITZ];
  d.off_t_bcolptr = it[H_OFF_T_BCOLPTR]; d.off_t_bcols = it[H_OFF_T_BCOLS];
  d.rs_ngfix = it[H_RS_NGFIX]; d.off_rs_gfix = it[H_OFF_RS_GFIX];
  d.rs_compact = it[H_RS_COMPACT]; d.rs_ldv = it[H_RS_LDV]; d.rs_vd = it[H_RS_VD];
  d.rs_vrow0 = it[H_RS_VROW0]; d.off_rs_rrwin = it[H_OFF_RS_RRWIN];
  d.t_np1 = it[H_T_NP1]; d.off_t_p1ptr = it[H_OFF_T_P1PTR]; d.off_t_p1ent = it[H_OFF_T_P1ENT]; d.off_t_p2y = it[H_OFF_T_P2Y];
  d.t_scan = it[H_T_SCAN]; d.t_scan_nblk = it[H_T_SCAN_NBLK]; d.off_t_scan_blk = it[H_OFF_T_SCAN_BLK];
  d.off_t_scan_gt = it[H_OFF_T_SCAN_GT]; d.t_doff_scan_gc = it[H_T_DOFF_SCAN_GC];
  d.off_t_scan_grow = it[H_OFF_T_SCAN_GROW]; d.t_doff_scan_gcoef = it[H_T_DOFF_SCAN_GCOEF];
  d.t_scan_ngrest = it[H_T_SCAN_NGREST]; d.off_t_scan_grest = it[H_OFF_T_SCAN_GREST];
  d.off_t_scan_colblk = it[H_OFF_T_SCAN_COLBLK]; d.t_scan_nother = it[H_T_SCAN_NOTHER];
  d.sw_ok = it[H_SW_OK]; d.sw_n = it[H_SW_N]; d.sw_m = it[H_SW_M]; d.sw_horizon = it[H_SW_HORIZON];
  d.sw_src_a = it[H_SW_SRC_A]; d.sw_src_b = it[H_SW_SRC_B]; d.sw_naxes = it[H_SW_NAXES];
  d.off_sw_axis = it[H_OFF_SW_AXIS]; d.sw_nterm = it[H_SW_NTERM]; d.off_sw_term = it[H_OFF_SW_TERM];
  d.sw_nlim = it[H_SW_NLIM]; d.off_sw_lim = it[H_OFF_SW_LIM]; d.off_sw_col = it[H_OFF_SW_COL];
  d.sw_doff_cvec = it[H_SW_DOFF_CVEC]; d.sw_ncvec = it[H_SW_NCVEC];
  d.off_sw_cptr = it[H_OFF_SW_CPTR]; d.off_sw_cent = it[H_OFF_SW_CENT]; d.sw_ncent = it[H_SW_NCENT];
  d.off_sw_gptr = it[H_OFF_SW_GPTR]; d.off_sw_gent = it[H_OFF_SW_GENT]; d.sw_ngent = it[H_SW_NGENT];
  d.off_rs_prog = it[H_OFF_RS_PROG];
  d.t_scan_fused = it[H_T_SCAN_FUSED];
  d.t_nbrow = d.t_ci_ok ? it[d.off_t_brow0 + d.nbase] : 0;
  d.pm_nfd = it[H_PM_NFD]; d.off_pm_map = it[H_OFF_PM_MAP]; d.off_pm_fdptr = it[H_OFF_PM_FDPTR];
  d.off_pm_op = it[H_OFF_PM_OP]; d.doff_pm_pool = it[H_DOFF_PM_POOL];
  d.rs_src16 = 0;
  if (d.rs_ok && d.rs_unit == 16)
    for (int64_t i = 0; i < (int64_t)d.rs_nchunk * 64; ++i) {
      const int st = it[d.off_rs_inmeta + 2 * i];
      if (st < d.nsrc) d.rs_src16 |= 1u << st;
    }
  d.doff_diagcoef = it[H_DOFF_DIAGCOEF];
  d.ndiag = 0;
  d.rs_sym_any = 1;
  for (int g = 0; g < it[H_NGTERM]; ++g) {
    const int32_t* r = it + it[H_OFF_GTERM] + g * GT_WORDS;
    if (r[GT_FLAGS] & GT_FLAG_DIAG) ++d.ndiag;
    if ((r[GT_FLAGS] & GT_FLAG_P) && r[GT_AOFF] != r[GT_BOFF]) d.rs_sym_any = 0;
  }
  // the per-column table of the diagonal gterms holds them all when no column carries more than
  // RS_DIAG_MAX (its sections are laid out for every plan)
  d.rs_diag_table = 1;
  {
    std::vector<int> on(std::max(d.no, 1), 0);
    for (int g = 0; g < it[H_NGTERM]; ++g) {
      const int32_t* r = it + it[H_OFF_GTERM] + g * GT_WORDS;
      if (!(r[GT_FLAGS] & GT_FLAG_DIAG)) continue;
      for (int k = 0; k < r[GT_NROWS]; ++k)
        if (++on[r[GT_AOFF] + k] > RS_DIAG_MAX) d.rs_diag_table = 0;
    }
    if (it[H_OFF_RS_DPAR] % 4 || it[H_DOFF_RS_DCOEF] % 2) d.rs_diag_table = 0;
  }
  d.max_axes = 0;
  for (int l = 0; l < it[H_NLIMIT]; ++l) {
    const int na = it[it[H_OFF_LIMIT] + l * LM_WORDS + LM_NAXES];
    if (na > d.max_axes) d.max_axes = na;
  }
  // the op table is read as int2: keep its word offset even (the compiler pads it)
  if (d.fused_ok && (d.off_op & 1)) d.fused_ok = 0;
  d.rs_p_direct = d.rs_ok ? resident_choose_p_direct(d, g_p_direct) : 0;
  // the CSC form of P is read out of its LDS copy: such a plan has no other way
  if (d.csc_pnnz != 0 && d.rs_ok) d.rs_p_direct = resident_choose_p_direct(d, 2);
}

}  // namespace

extern "C" {

int mpcasm_abi_version(void) { return 1000; }

int mpcasm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mpcasm_last_hip(void) { return g_last_hip; }

int mpcasm_set_option(int option, int value) {
  if (option == MPCASM_OPT_PATH) {
    if (value < 0 || value > 4) return MPCASM_ERR_ARG;
    g_path = value;
    return MPCASM_OK;
  }
  if (option == MPCASM_OPT_PHASE_MASK) {
    g_phase_mask = value;
    return MPCASM_OK;
  }
  if (option == MPCASM_OPT_P_DIRECT) {
    if (value < 0 || value > 2) return MPCASM_ERR_ARG;
    g_p_direct = value;
    return MPCASM_OK;
  }
  if (option == MPCASM_OPT_JIT) {
    if (value < 0 || value > 2) return MPCASM_ERR_ARG;
    g_jit = value;
    return MPCASM_OK;
  }
  if (option == MPCASM_OPT_RESIDENT_PER_CU) {
    if (value < 0 || value > 16) return MPCASM_ERR_ARG;
    g_resident_per_cu = value;
    return MPCASM_OK;
  }
  if (option == MPCASM_OPT_RESIDENT_GRID) {
    if (value < 0 || value > (1 << 20)) return MPCASM_ERR_ARG;
    g_resident_grid = value;
    return MPCASM_OK;
  }
  return MPCASM_ERR_ARG;
}

int mpcasm_plan_set_option(mpcasm_plan* plan, int option, int value) {
  if (!plan) return MPCASM_ERR_ARG;
  if (option == MPCASM_OPT_PATH && value >= -1 && value <= 4) plan->opt_path = value;
  else if (option == MPCASM_OPT_JIT && value >= -1 && value <= 2) plan->opt_jit = value;
  else if (option == MPCASM_OPT_RESIDENT_PER_CU && value >= -1 && value <= 16) plan->opt_per_cu = value;
  else if (option == MPCASM_OPT_RESIDENT_GRID && value >= -1 && value <= (1 << 20)) plan->opt_grid = value;
  else return MPCASM_ERR_ARG;
  return MPCASM_OK;
}

const char* mpcasm_status_string(int status) {
  switch (status) {
    case MPCASM_OK: return "ok";
    case MPCASM_ERR_ARG: return "invalid argument";
    case MPCASM_ERR_PLAN: return "malformed plan tables";
    case MPCASM_ERR_HIP: return "HIP runtime error";
    case MPCASM_ERR_NODEVICE: return "no HIP device";
    case MPCASM_ERR_LIMIT: return "problem exceeds a kernel limit";
    default: return "unknown status";
  }
}

int mpcasm_fill_su(const double* d_A, const double* d_B, double* d_S, double* d_U, int batch,
                   int N, int n, int m, int ltv, void* stream) {
  if (!d_A || !d_B || !d_S || !d_U) return MPCASM_ERR_ARG;
  if (batch < 0 || N < 1 || n < 1 || m < 1 || (ltv != 0 && ltv != 1)) return MPCASM_ERR_ARG;
  if (batch == 0) return MPCASM_OK;
  if (mpcasm_device_count() == 0) return MPCASM_ERR_NODEVICE;
  hipError_t err;
  const int rc = launch_fill_su(d_A, d_B, d_S, d_U, batch, N, n, m, ltv,
                                static_cast<hipStream_t>(stream), &err);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_plan_create(const int32_t* h_itab, size_t n_itab, const double* h_dtab, size_t n_dtab,
                       mpcasm_plan** out_plan) {
  if (!h_itab || !out_plan || (n_dtab && !h_dtab)) return MPCASM_ERR_ARG;
  *out_plan = nullptr;
  const int rc = validate_plan(h_itab, h_dtab, n_itab, n_dtab);
  if (rc != MPCASM_OK) return rc;
  {  // a CSC plan runs on the persistent kernel with P in LDS, or not at all
    PlanDev probe;
    memset(&probe, 0, sizeof probe);
    plan_dev_from_tables(h_itab, &probe);
    if ((probe.csc_pnnz != 0 || probe.csc_gnnz != 0) &&
        (probe.rs_p_direct != 0 || resident_lds_bytes(probe) == 0 ||
         resident_lds_bytes(probe) > RESIDENT_LDS_LIMIT))
      return MPCASM_ERR_LIMIT;
  }
  if (mpcasm_device_count() == 0) return MPCASM_ERR_NODEVICE;

  mpcasm_plan* plan = new (std::nothrow) mpcasm_plan();
  if (!plan) return MPCASM_ERR_ARG;
  hipError_t e;
  if ((e = hipGetDevice(&plan->device)) != hipSuccess) {
    delete plan;
    return hip_fail(e);
  }
  plan->d_itab = nullptr;
  plan->d_dtab = nullptr;
  const size_t dbytes = (n_dtab ? n_dtab : 1) * sizeof(double);
  if ((e = hipMalloc(&plan->d_itab, n_itab * sizeof(int32_t))) != hipSuccess ||
      (e = hipMalloc(&plan->d_dtab, dbytes)) != hipSuccess ||
      (e = hipMemcpy(plan->d_itab, h_itab, n_itab * sizeof(int32_t), hipMemcpyHostToDevice)) !=
          hipSuccess ||
      (n_dtab && (e = hipMemcpy(plan->d_dtab, h_dtab, n_dtab * sizeof(double),
                                hipMemcpyHostToDevice)) != hipSuccess)) {
    if (plan->d_itab) (void)hipFree(plan->d_itab);
    if (plan->d_dtab) (void)hipFree(plan->d_dtab);
    delete plan;
    return hip_fail(e);
  }
  PlanDev& d = plan->dev;
  plan_dev_from_tables(h_itab, &d);
  d.itab = plan->d_itab;
  d.dtab = plan->d_dtab;
  plan->h_itab.assign(h_itab, h_itab + n_itab);
  {
    hipDeviceProp_t prop;
    plan->num_cus = 256;
    if (hipGetDeviceProperties(&prop, plan->device) == hipSuccess && prop.multiProcessorCount > 0)
      plan->num_cus = prop.multiProcessorCount;
  }
  *out_plan = plan;
  return MPCASM_OK;
}

int mpcasm_resident_lds_bytes(const int32_t* h_itab, size_t n_itab, const double* h_dtab, size_t n_dtab,
                              int64_t out[2]) {
  if (!h_itab || (n_dtab && !h_dtab) || !out) return MPCASM_ERR_ARG;
  const int rc = validate_plan(h_itab, h_dtab, n_itab, n_dtab);
  if (rc != MPCASM_OK) return rc;
  PlanDev d;
  memset(&d, 0, sizeof d);
  plan_dev_from_tables(h_itab, &d);
  for (int direct = 0; direct < 2; ++direct) {
    d.rs_p_direct = 1 - direct;  // out[0]: P direct, out[1]: P in LDS
    out[direct] = (int64_t)resident_lds_bytes(d);
  }
  return MPCASM_OK;
}

int mpcasm_jit_check(const int32_t* h_itab, size_t n_itab, const double* h_dtab, size_t n_dtab,
                     char* log, size_t log_capacity) {
  if (!h_itab || (n_dtab && !h_dtab)) return MPCASM_ERR_ARG;
  if (log && log_capacity) log[0] = 0;
  const int rc = validate_plan(h_itab, h_dtab, n_itab, n_dtab);
  if (rc != MPCASM_OK) return rc;
  PlanDev d;
  memset(&d, 0, sizeof d);
  plan_dev_from_tables(h_itab, &d);
  if (!d.rs_ok) return MPCASM_ERR_LIMIT;
  std::vector<char> code;
  std::string text;
  // (both forms of the P hand-over when the size of a launch picks one)
  const int small = resident_p_direct_for(d, 1), large = resident_p_direct_for(d, 1 << 30);
  int out = MPCASM_OK;
  if (!jit_available()) {
    text = "libhiprtc.so could not be loaded";
    out = MPCASM_ERR_LIMIT;
  } else {
    for (int form : {small, large}) {
      if (form == large && small != large && out != MPCASM_OK) break;
      d.rs_p_direct = form;
      // (through the disk cache, as a launch would: a second check of the same plan compiles nothing)
      out = jit_code_for(jit_spec_header(d, h_itab), false, 0xBF, &code, &text) ? MPCASM_OK : MPCASM_ERR_HIP;
      if (small == large) break;
    }
  }
  if (log && log_capacity) {
    strncpy(log, text.c_str(), log_capacity - 1);
    log[log_capacity - 1] = 0;
  }
  return out;
}

int mpcasm_jit_stats(int64_t out[3]) {
  if (!out) return MPCASM_ERR_ARG;
  long v[3];
  jit_stats(v);
  out[0] = v[0];
  out[1] = v[1];
  out[2] = v[2];
  return MPCASM_OK;
}

int mpcasm_plan_prepare(const mpcasm_plan* plan, int batch) {
  if (!plan || batch < 0) return MPCASM_ERR_ARG;
  int current = -1;
  if (hipGetDevice(&current) != hipSuccess || current != plan->device) return MPCASM_ERR_ARG;
  PlanDev p = plan->dev;
  if (!p.rs_ok || p.sw_ok) return MPCASM_OK;  // (nothing is compiled per plan for the other kernels)
  p.rs_p_direct = resident_p_direct_for(plan->dev, batch);
  const size_t rs = resident_lds_bytes(p);
  if (rs == 0 || rs > RESIDENT_LDS_LIMIT) return MPCASM_OK;
  t_path = plan->opt_path >= 0 ? plan->opt_path : g_path;
  t_jit = plan->opt_jit >= 0 ? plan->opt_jit : g_jit;
  if (t_path != 0 && p.rs_nlti == 0 && p.csc_pnnz == 0 && p.csc_gnnz == 0) return MPCASM_OK;
  (void)jit_kernel_for(p, plan->h_itab.data(), plan->device, batch, rs);  // (a failure: the ahead-of-time kernel)
  return MPCASM_OK;
}

int mpcasm_plan_destroy(mpcasm_plan* plan) {
  if (!plan) return MPCASM_OK;
  jit_forget(plan->d_itab);
  hipError_t e1 = hipFree(plan->d_itab);
  hipError_t e2 = hipFree(plan->d_dtab);
  delete plan;
  if (e1 != hipSuccess) return hip_fail(e1);
  if (e2 != hipSuccess) return hip_fail(e2);
  return MPCASM_OK;
}

int mpcasm_plan_sizes(const mpcasm_plan* plan, int64_t out[8]) {
  if (!plan || !out) return MPCASM_ERR_ARG;
  const PlanDev& d = plan->dev;
  out[0] = d.ng; out[1] = d.no; out[2] = d.nc; out[3] = d.nparams;
  out[4] = d.nsrc; out[5] = d.rtot; out[6] = d.ldv; out[7] = d.pmrows;
  return MPCASM_OK;
}

int mpcasm_plan_csc_sizes(const mpcasm_plan* plan, int64_t out[2]) {
  if (!plan || !out) return MPCASM_ERR_ARG;
  out[0] = plan->dev.csc_pnnz;
  out[1] = plan->dev.csc_gnnz;
  return MPCASM_OK;
}

int mpcasm_workspace_bytes(const mpcasm_plan* plan, int batch, size_t* out_bytes) {
  if (!plan || !out_bytes || batch < 0) return MPCASM_ERR_ARG;
  // the staged pipeline's workspace; never less than the cycle stamps of the diagnostic
  // persistent kernel take (8 wavefronts x 8 counters per resident workgroup for the phases
  // of the instance loop, as many again for the set-up)
  const size_t groups = (size_t)std::min<long>(batch, (long)plan->num_cus * 8);
  const size_t stamps = 2 * groups * 8 * 8 * sizeof(uint64_t);
  // a wide problem on the tiled kernel needs d and the generated horizon tables only -- unless the
  // options in force send it down the staged pipeline (a test hook): ask again after changing them
  const int path = plan->opt_path >= 0 ? plan->opt_path : g_path;
  if (sweep_eligible(plan->dev)) {  // (everything in LDS)
    *out_bytes = stamps;
    return MPCASM_OK;
  }
  if (tiled_eligible(plan->dev) && path != 2) {
    *out_bytes = std::max(tiled_workspace_bytes(plan->dev, batch), stamps);
    return MPCASM_OK;
  }
  // (the generated horizon tables of mpcasm_preview_direct fit in either case)
  *out_bytes = std::max(std::max(assemble_workspace_bytes(plan->dev, batch),
                                 tiled_workspace_bytes(plan->dev, batch)), stamps);
  return MPCASM_OK;
}

static int assemble_impl(const mpcasm_plan* plan, const double* const* h_src,
                         const int64_t* h_src_stride, const double* d_params, const double* d_given,
                         const int32_t* d_given_index, double* d_P, double* d_q, double* d_G, double* d_h,
                         void* d_work, int batch, void* stream) {
  if (!plan || batch < 0) return MPCASM_ERR_ARG;
  if (batch == 0) return MPCASM_OK;  // nothing to do (empty buffers may be null)
  const PlanDev& d = plan->dev;
  if ((d.nsrc && (!h_src || !h_src_stride)) || (d.nparams && !d_params) || (d.ng && !d_given))
    return MPCASM_ERR_ARG;
  if ((d_P == nullptr) != (d_q == nullptr) || (d_G == nullptr) != (d_h == nullptr))
    return MPCASM_ERR_ARG;
  if (d.rtot && !d_work && !d.sw_ok) return MPCASM_ERR_ARG;
  {  // the plan's tables (and its compiled kernel) live on the device it was created on
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != plan->device) return MPCASM_ERR_ARG;
  }
  // results leave the chip in 16-byte stores
  if ((reinterpret_cast<uintptr_t>(d_P) | reinterpret_cast<uintptr_t>(d_q) |
       reinterpret_cast<uintptr_t>(d_G) | reinterpret_cast<uintptr_t>(d_h)) & 15)
    return MPCASM_ERR_ARG;
  SrcTable src;
  int rc = make_src_table(plan, h_src, h_src_stride, &src);
  if (rc != MPCASM_OK) return rc;
  if (d_given_index != nullptr) {
    // the index of `given`'s rows rides in the last slot of the source table (resident.hip)
    if (d.nsrc >= MAX_SOURCES || d.ng == 0) return MPCASM_ERR_LIMIT;
    src.ptr[MAX_SOURCES - 1] = reinterpret_cast<const double*>(d_given_index);
    src.stride[MAX_SOURCES - 1] = -1;
  }
  hipError_t err;
  // what this launch runs on: the plan's own options where it has them, else the process-wide ones
  t_path = plan->opt_path >= 0 ? plan->opt_path : g_path;
  t_jit = plan->opt_jit >= 0 ? plan->opt_jit : g_jit;
  t_per_cu = plan->opt_per_cu >= 0 ? plan->opt_per_cu : g_resident_per_cu;
  t_grid = plan->opt_grid >= 0 ? plan->opt_grid : g_resident_grid;
  rc = launch_assemble(d, src, d_params, d_given, d_P, d_q, d_G, d_h, d_work, batch,
                       plan->num_cus, static_cast<hipStream_t>(stream), &err, plan->h_itab.data(),
                       plan->device, d_given_index != nullptr);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  if (rc == MPCASM_OK) plan->last_kernel = t_last_kernel;
  return rc;
}

int mpcasm_assemble(const mpcasm_plan* plan, const double* const* h_src,
                    const int64_t* h_src_stride, const double* d_params, const double* d_given,
                    double* d_P, double* d_q, double* d_G, double* d_h, void* d_work, int batch,
                    void* stream) {
  return assemble_impl(plan, h_src, h_src_stride, d_params, d_given, nullptr, d_P, d_q, d_G, d_h, d_work,
                       batch, stream);
}

int mpcasm_assemble_indexed(const mpcasm_plan* plan, const double* const* h_src,
                            const int64_t* h_src_stride, const double* d_params, const double* d_given,
                            const int32_t* d_given_index, double* d_P, double* d_q, double* d_G,
                            double* d_h, void* d_work, int batch, void* stream) {
  if (!d_given_index) return MPCASM_ERR_ARG;
  return assemble_impl(plan, h_src, h_src_stride, d_params, d_given, d_given_index, d_P, d_q, d_G, d_h,
                       d_work, batch, stream);
}

int mpcasm_plan_last_kernel(const mpcasm_plan* plan) { return plan ? plan->last_kernel : MPCASM_ERR_ARG; }

int mpcasm_preview_matrices(const mpcasm_plan* plan, const double* const* h_src,
                            const int64_t* h_src_stride, double* d_PM, int batch, void* stream) {
  if (!plan || !d_PM || batch < 0) return MPCASM_ERR_ARG;
  if (plan->dev.nsrc && (!h_src || !h_src_stride)) return MPCASM_ERR_ARG;
  if (plan->dev.rs_nlti != 0 || plan->dev.sw_ok) return MPCASM_ERR_LIMIT;  // the plan has no S, U to read
  if (batch == 0) return MPCASM_OK;
  SrcTable src;
  int rc = make_src_table(plan, h_src, h_src_stride, &src);
  if (rc != MPCASM_OK) return rc;
  hipError_t err;
  rc = launch_preview_matrices(plan->dev, src, d_PM, batch, static_cast<hipStream_t>(stream), &err);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_preview(const double* d_PM, const double* d_given, const double* d_optim, double* d_out,
                   int batch, int rows, int ng, int no, void* stream) {
  if (!d_PM || !d_out || batch < 0 || rows < 0 || ng < 0 || no < 0) return MPCASM_ERR_ARG;
  if ((ng && !d_given) || (no && !d_optim)) return MPCASM_ERR_ARG;
  if (batch == 0 || rows == 0) return MPCASM_OK;
  hipError_t err;
  const int rc = launch_preview(d_PM, d_given, d_optim, d_out, batch, rows, ng, no,
                                static_cast<hipStream_t>(stream), &err);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_preview_direct(const mpcasm_plan* plan, const double* const* h_src,
                          const int64_t* h_src_stride, const double* d_given, const double* d_optim,
                          double* d_out, void* d_work, int batch, void* stream) {
  if (!plan || !d_out || batch < 0) return MPCASM_ERR_ARG;
  const PlanDev& d = plan->dev;
  if ((d.nsrc && (!h_src || !h_src_stride)) || (d.ng && !d_given) || (d.no && !d_optim))
    return MPCASM_ERR_ARG;
  if (batch == 0 || d.pmrows == 0) return MPCASM_OK;
  if (d.sw_ok) return MPCASM_ERR_LIMIT;  // (per-step dynamics: no horizon tables to preview from)
  if (d.t_nlti != 0 && !d_work) return MPCASM_ERR_ARG;
  {  // the plan's tables live on the device it was created on
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != plan->device) return MPCASM_ERR_ARG;
  }
  SrcTable src, eff;
  int rc = make_src_table(plan, h_src, h_src_stride, &src);
  if (rc != MPCASM_OK) return rc;
  hipError_t err;
  rc = launch_lti_tables(d, src, static_cast<double*>(d_work), batch, plan->h_itab.data(), &eff,
                         static_cast<hipStream_t>(stream));
  if (rc != MPCASM_OK) return rc;
  rc = launch_preview_direct(d, eff, d_given, d_optim, d_out, batch, plan->num_cus,
                             static_cast<hipStream_t>(stream), &err, plan->h_itab.data());
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_preview_goal_distance(const mpcasm_plan* plan, const double* const* h_src,
                                 const int64_t* h_src_stride, const double* d_given,
                                 const double* d_optim, const double* d_params, const int32_t* d_terms,
                                 int nterms, int ngoals, double* d_out, void* d_work, int batch,
                                 void* stream) {
  if (!plan || batch < 0 || nterms < 0 || ngoals < 0) return MPCASM_ERR_ARG;
  const PlanDev& d = plan->dev;
  if (batch == 0 || ngoals == 0) return MPCASM_OK;
  if (!d_out || (d.nsrc && (!h_src || !h_src_stride)) || (d.ng && !d_given) || (d.no && !d_optim) ||
      (nterms && (!d_terms || !d_params)))
    return MPCASM_ERR_ARG;
  if (d.sw_ok) return MPCASM_ERR_LIMIT;
  if (d.t_nlti != 0 && !d_work) return MPCASM_ERR_ARG;
  {
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != plan->device) return MPCASM_ERR_ARG;
  }
  SrcTable src, eff;
  int rc = make_src_table(plan, h_src, h_src_stride, &src);
  if (rc != MPCASM_OK) return rc;
  hipError_t err;
  rc = launch_lti_tables(d, src, static_cast<double*>(d_work), batch, plan->h_itab.data(), &eff,
                         static_cast<hipStream_t>(stream));
  if (rc != MPCASM_OK) return rc;
  rc = launch_preview_goals(d, eff, d_given, d_optim, d_params, d.nparams, d_terms, nterms, ngoals, d_out,
                            batch, plan->num_cus, static_cast<hipStream_t>(stream), &err, plan->h_itab.data());
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_goal_distance(const double* d_preview, int64_t preview_stride, const double* d_params,
                         int64_t n_params, const int32_t* d_terms, int nterms, int ngoals,
                         double* d_out, int batch, void* stream) {
  if (batch < 0 || nterms < 0 || ngoals < 0 || preview_stride < 0 || n_params < 0) return MPCASM_ERR_ARG;
  if (batch == 0 || ngoals == 0) return MPCASM_OK;
  if (!d_preview || !d_out || (nterms && (!d_terms || !d_params))) return MPCASM_ERR_ARG;
  hipError_t err;
  const int rc = launch_goal_distance(d_preview, preview_stride, d_params, n_params, d_terms, nterms,
                                      ngoals, d_out, batch, static_cast<hipStream_t>(stream), &err);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_box_transform(double* d_params, int64_t n_params, int batch, const int32_t* d_facets,
                         int nfacets, int op, const double* d_arg, int64_t arg_stride,
                         void* stream) {
  if (batch < 0 || nfacets < 0 || n_params < 0 || arg_stride < 0 || op < MPCASM_BOX_RECENTER ||
      op > MPCASM_BOX_MARGIN)
    return MPCASM_ERR_ARG;
  if (batch == 0 || nfacets == 0) return MPCASM_OK;
  if (!d_params || !d_facets || !d_arg) return MPCASM_ERR_ARG;
  hipError_t err;
  const int rc = launch_box_transform(d_params, n_params, batch, d_facets, nfacets, op, d_arg,
                                      arg_stride, static_cast<hipStream_t>(stream), &err);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_box_transform_ss(double* d_params, int64_t n_params, int batch,
                            const int32_t* d_facets, int nfacets, int op, const double* d_L,
                            int lrows, int ss_dim, const double* d_arg, int64_t arg_stride,
                            void* stream) {
  if (batch < 0 || nfacets < 0 || n_params < 0 || arg_stride < 0 || lrows < 1 || ss_dim < 1 ||
      (op != MPCASM_BOX_RECENTER && op != MPCASM_BOX_TRANSLATE))
    return MPCASM_ERR_ARG;
  if (batch == 0 || nfacets == 0) return MPCASM_OK;
  if (!d_params || !d_facets || !d_L || !d_arg) return MPCASM_ERR_ARG;
  hipError_t err;
  const int rc = launch_box_transform_ss(d_params, n_params, batch, d_facets, nfacets, op, d_L,
                                         lrows, ss_dim, d_arg, arg_stride,
                                         static_cast<hipStream_t>(stream), &err);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_admm(int no, int nc, const double* d_P, const double* d_q, const double* d_G,
                const double* d_h, double* d_x, double* d_y, double* d_z, double* d_res, double rho,
                double sigma, double alpha, int iters, int warm, int batch, double* d_kinv, int kinv_valid,
                void* stream) {
  if (no < 1 || nc < 0 || batch < 0 || iters < 0 || !(rho > 0.0) || !(sigma > 0.0) || !(alpha > 0.0) ||
      !(alpha < 2.0) || no > (1 << 12) || nc > (1 << 16))
    return MPCASM_ERR_ARG;
  if (batch == 0) return MPCASM_OK;
  if (!d_P || !d_q || !d_x || (nc > 0 && (!d_G || !d_h || !d_y || !d_z))) return MPCASM_ERR_ARG;
  hipError_t err;
  if (kinv_valid != 0 && d_kinv == nullptr) return MPCASM_ERR_ARG;
  const int rc = launch_admm(no, nc, d_P, d_q, d_G, d_h, d_x, d_y, d_z, d_res, rho, sigma, alpha, iters,
                             warm != 0, batch, d_kinv, kinv_valid != 0, static_cast<hipStream_t>(stream), &err);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

int mpcasm_gather(const double* d_src, int64_t src_stride, const int32_t* d_index, int nnz,
                  double* d_dst, int batch, void* stream) {
  if (nnz < 0 || batch < 0 || src_stride < 0) return MPCASM_ERR_ARG;
  if (nnz == 0 || batch == 0) return MPCASM_OK;
  if (!d_src || !d_index || !d_dst) return MPCASM_ERR_ARG;
  hipError_t err;
  const int rc = launch_gather(d_src, src_stride, d_index, nnz, d_dst, batch,
                               static_cast<hipStream_t>(stream), &err);
  if (rc == MPCASM_ERR_HIP) g_last_hip = (int)err;
  return rc;
}

}  // extern "C"

namespace mpcasm {

hipError_t allow_whole_lds(const void* fn) {
  static std::mutex mutex;
  static std::vector<std::pair<const void*, int>> done;
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mutex);
  for (const auto& d : done)
    if (d.first == fn && d.second == device) return hipSuccess;
  // (the runtime of this image refuses the CU's full 160 KiB -- hipErrorInvalidValue -- and grants
  // 156 KiB, what the kernels' own limit RESIDENT_LDS_LIMIT asks for: the larger one first, for a
  // runtime that gives it)
  for (int bytes : {CU_LDS_BYTES, RESIDENT_LDS_LIMIT}) {
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) {
      done.emplace_back(fn, device);
      return e;
    }
    (void)hipGetLastError();
  }
  return e;
}

int g_path = 0;  // test hook (MPCASM_OPT_PATH): 0 best, 1 no resident kernel, 2 staged only
int g_resident_per_cu = 0;  // tuning aid (MPCASM_OPT_RESIDENT_PER_CU): 0 = automatic
int g_resident_grid = 0;    // MPCASM_OPT_RESIDENT_GRID: 0 = every resident workgroup slot
// the values in force for the launch this thread is making (mpcasm_assemble sets them)
thread_local int t_path = 0, t_jit = 0, t_per_cu = 0, t_grid = 0;
thread_local int t_last_kernel = 0;

// dispatch: fused single launch when the problem fits on chip, else staged
int launch_assemble(const PlanDev& plan, const SrcTable& src, const double* params,
                    const double* given, double* P, double* q, double* G, double* h, void* work,
                    int batch, int num_cus, hipStream_t stream, hipError_t* err,
                    const int32_t* h_itab, int device, bool indexed) {
  // a dynamics compiled as ltv: the sweep kernel and nothing else (the other kernels' tables describe
  // the formulation's own horizon matrices, the source slots carry (A_k, B_k))
  if (sweep_eligible(plan)) {
    if (indexed) return MPCASM_ERR_LIMIT;   // (rows of `given` by index: the persistent kernel only)
    t_last_kernel = MPCASM_KERNEL_SWEEP;
    return launch_assemble_sweep(plan, src, params, given, P, q, G, h, batch, stream, err, h_itab);
  }
  PlanDev p = plan;
  p.rs_p_direct = p.rs_ok ? resident_p_direct_for(plan, batch) : 0;  // (the launch's size decides)
  // the persistent kernel may take a whole CU's LDS (one workgroup of 8 wavefronts per
  // CU still beats the staged pipeline by far); the per-instance fused kernel is only
  // worth it while two workgroups fit
  constexpr size_t FUSED_LDS_LIMIT = 80 * 1024;
  const size_t rs = resident_lds_bytes(p);
  if (rs != 0 && rs <= RESIDENT_LDS_LIMIT && (t_path == 0 || p.rs_nlti != 0 || p.csc_pnnz != 0 || p.csc_gnnz != 0) &&
      resident_inputs_aligned(p, src, params, given)) {
    // large batches: the same kernel compiled for this very plan (jit.hip), when available
    if (h_itab != nullptr)
      if (const void* k = jit_kernel_for(p, h_itab, device, batch, rs)) {
        t_last_kernel = MPCASM_KERNEL_RESIDENT_JIT;
        return jit_launch(k, p, src, params, given, P, q, G, h, batch, num_cus, t_per_cu,
                          t_grid, work, stream, err);
      }
    t_last_kernel = MPCASM_KERNEL_RESIDENT;
    return launch_assemble_resident(p, src, params, given, P, q, G, h, work, batch, rs, num_cus,
                                    stream, err);
  }
  if (indexed) return MPCASM_ERR_LIMIT;   // (rows of `given` by index: the persistent kernel only)
  // wide problems: one workgroup per block of P, rows composed straight into the LDS tiles
  if (tiled_eligible(p) && t_path != 2 && p.csc_pnnz == 0 && p.csc_gnnz == 0) {
    t_last_kernel = MPCASM_KERNEL_TILED;
    return launch_assemble_tiled(p, src, params, given, P, q, G, h, work, batch, stream, err, h_itab);
  }
  // elsewhere, horizon matrices generated from (A, B) and the CSC form of the results exist in the
  // persistent kernel only
  if (p.rs_nlti != 0 || p.csc_pnnz != 0 || p.csc_gnnz != 0) return MPCASM_ERR_LIMIT;
  const size_t lds = fused_lds_bytes(p, 4);
  if (lds != 0 && lds <= FUSED_LDS_LIMIT && t_path <= 1) {
    t_last_kernel = MPCASM_KERNEL_FUSED;
    return launch_assemble_fused(p, src, params, given, P, q, G, h, batch, lds, stream, err);
  }
  t_last_kernel = MPCASM_KERNEL_STAGED;
  return launch_assemble_staged(p, src, params, given, P, q, G, h, work, batch, stream, err);
}

}  // namespace mpcasm
