// plan_dev.h -- device-side view of a plan and the source table of a launch: plain structs, no
// HIP runtime types, so that the persistent kernel's source also compiles under hiprtc
// (MPCASM_SPEC, resident.hip).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#include "plan_tables.h"

namespace mpcasm {

// every int field of PlanDev, once: the struct, its upload (capi.hip) and the constants of a
// specialised kernel (jit.hip) are generated from this list
#define MPCASM_PLAN_INT_FIELDS(X)                                                                   \
  X(ng) X(no) X(nc) X(nparams) X(nsrc) X(nbase) X(nseg) X(rtot) X(nent) X(ngterm) X(nlimit)         \
  X(nlax) X(pmrows) X(pm_nent) X(ldv) X(off_seg) X(off_colseg) X(off_rowptr) X(off_entbase)         \
  X(off_entk) X(off_gterm) X(off_limit) X(off_lax) X(off_rowlimit) X(off_pm_rowptr)                 \
  X(off_pm_entbase) X(off_pm_entk) X(doff_entcoef) X(doff_pm_entcoef) X(fused_ok) X(arena_total)    \
  X(off_arena) X(nfd) X(off_fd_idx) X(off_fd_ptr) X(nops) X(off_op) X(ncoef) X(doff_coefpool)       \
  X(max_axes) X(rs_sym_any) X(rs_ok) X(rs_jc) X(rs_sym) X(rs_ntrip) X(off_rs_src) X(off_rs_gidx)    \
  X(off_rs_dst) X(doff_rs_coef) X(off_rs_trip) X(off_rs_wtrip) X(rs_nsplit) X(off_rs_split)         \
  X(off_rs_rr) X(rs_unit) X(rs_nchunk) X(off_rs_inmeta) X(rs_img) X(rs_img_given)                   \
  X(rs_img_params) X(doff_rs_const) X(rs_nlti) X(off_rs_lti) X(rs_img_dma) X(rs_ab)                 \
  X(off_rs_abmeta) X(rr_packed) X(off_rs_dpar) X(doff_rs_dcoef) X(rs_ngdesc) X(off_rs_gdesc)        \
  X(pm_nfd) X(off_pm_map) X(off_pm_fdptr) X(off_pm_op) X(doff_pm_pool) X(doff_diagcoef) X(ndiag)      \
  X(rs_nzblk) X(off_rs_zblk) X(rs_p_direct) X(rs_gsingle)                                           \
  X(csc_pnnz) X(off_csc_p) X(csc_gnnz) X(off_csc_g) X(csc_gsingle)                                  \
  X(t_ci_ok) X(t_nop) X(off_t_cig) X(off_t_cio) X(t_doff_delta) X(t_ok) X(t_nstage)                 \
  X(off_t_stage) X(t_nlti) X(off_t_lti) X(off_t_lti_ids) X(t_work) X(off_t_grow)                    \
  X(off_t_srow) X(t_doff_scoef) X(off_t_pig) X(t_ngrest) X(off_t_grest) X(off_t_brow0) X(t_nbrow) X(t_toeplitz) X(rs_diag_table) X(off_t_bcolptr) X(off_t_bcols) \
  X(rs_ngfix) X(off_rs_gfix) X(rs_compact) X(rs_ldv) X(rs_vd) X(rs_vrow0) X(off_rs_rrwin) X(t_np1) X(off_t_p1ptr) X(off_t_p1ent) X(off_t_p2y) \
  X(t_scan) X(t_scan_nblk) X(off_t_scan_blk) X(off_t_scan_gt) X(t_doff_scan_gc) X(off_t_scan_grow)      \
  X(t_doff_scan_gcoef) X(t_scan_ngrest) X(off_t_scan_grest) X(off_t_scan_colblk) X(t_scan_nother) \
  X(sw_ok) X(sw_n) X(sw_m) X(sw_horizon) X(sw_src_a) X(sw_src_b) X(sw_naxes) X(off_sw_axis) X(sw_nterm)  \
  X(off_sw_term) X(sw_nlim) X(off_sw_lim) X(off_sw_col) X(sw_doff_cvec) X(sw_ncvec) \
  X(off_sw_cptr) X(off_sw_cent) X(sw_ncent) X(off_sw_gptr) X(off_sw_gent) X(sw_ngent) X(off_rs_prog) X(t_scan_fused)

// device-side view of a plan (pointers into the device copies of the tables).
//   rs_p_direct: the persistent kernel sends the blocks of P straight to HBM (set by
//   resident_choose_p_direct: P does not fit in LDS beside the workspace, or MPCASM_OPT_P_DIRECT);
//   rs_sym_any: every Hessian term has A == B;  rs_src16: sources that the 16-byte image loads
//   read (bit per source);  ndiag: number of diagonal gterms, doff_diagcoef: their coefficients
struct PlanDev {
  const int32_t* itab;
  const double* dtab;
#define MPCASM_X(f) int f;
  MPCASM_PLAN_INT_FIELDS(MPCASM_X)
#undef MPCASM_X
  unsigned rs_src16;
};

// LDS the persistent kernel may ask for (a whole CU's 160 KiB less what the runtime keeps)
constexpr int RESIDENT_LDS_LIMIT = 156 * 1024;

// sources of one launch (device pointers + per-instance strides, by value)
struct SrcTable {
  const double* ptr[MAX_SOURCES];
  long long stride[MAX_SOURCES];
};

}  // namespace mpcasm
