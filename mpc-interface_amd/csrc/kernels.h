// kernels.h -- internal launch interface between capi.hip and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mpcasm.h"
#include "plan_tables.h"

namespace mpcasm {

// device-side view of a plan (pointers into the device copies of the tables)
struct PlanDev {
  const int32_t* itab;
  const double* dtab;
  int ng, no, nc, nparams, nsrc, nbase, nseg, rtot, nent, ngterm, nlimit, nlax, pmrows, pm_nent, ldv;
  int off_seg, off_colseg, off_rowptr, off_entbase, off_entk, off_gterm, off_limit, off_lax, off_rowlimit;
  int off_pm_rowptr, off_pm_entbase, off_pm_entk;
  int doff_entcoef, doff_pm_entcoef;
  // fused program
  int fused_ok, arena_total, off_arena, nfd, off_fd_idx, off_fd_ptr, nops, off_op, ncoef,
      doff_coefpool, max_axes, rs_sym_any;  // rs_sym_any: every Hessian term has A == B
  // resident program
  int rs_ok, rs_jc, rs_sym, rs_ntrip, off_rs_src, off_rs_gidx, off_rs_dst, doff_rs_coef,
      off_rs_trip, off_rs_wtrip, rs_nsplit, off_rs_split, off_rs_rr, rs_unit,
      rs_nchunk, off_rs_inmeta, rs_img, rs_img_given, rs_img_params, doff_rs_const, rs_nlti,
      off_rs_lti, rs_img_dma, rs_ab, off_rs_abmeta, rr_packed, off_rs_dpar, doff_rs_dcoef,
      rs_ngdesc, off_rs_gdesc, pm_nfd, off_pm_map, off_pm_fdptr, off_pm_op, doff_pm_pool;
  unsigned rs_src16;  // sources that the 16-byte image loads read (bit per source)
  int doff_diagcoef, ndiag;  // diagonal gterms: coefficient list, number of such terms
};

// sources of one launch (device pointers + per-instance strides, by value)
struct SrcTable {
  const double* ptr[MAX_SOURCES];
  long long stride[MAX_SOURCES];
};

// fill.hip
int launch_fill_su(const double* A, const double* B, double* S, double* U, int batch, int N, int n,
                   int m, int ltv, hipStream_t stream, hipError_t* err);

// assemble.hip
size_t assemble_workspace_bytes(const PlanDev& p, int batch);
int launch_assemble_staged(const PlanDev& p, const SrcTable& src, const double* params,
                           const double* given, double* P, double* q, double* G, double* h,
                           void* work, int batch, hipStream_t stream, hipError_t* err);
// fused.hip
size_t fused_lds_bytes(const PlanDev& p, int nw);
int launch_assemble_fused(const PlanDev& p, const SrcTable& src, const double* params,
                          const double* given, double* P, double* q, double* G, double* h,
                          int batch, size_t lds_bytes, hipStream_t stream, hipError_t* err);
// resident.hip
size_t resident_lds_bytes(const PlanDev& p);
// whether this launch's buffers meet the alignment the plan's input loads assume
bool resident_inputs_aligned(const PlanDev& p, const SrcTable& src, const double* params,
                             const double* given);
int launch_assemble_resident(const PlanDev& p, const SrcTable& src, const double* params,
                             const double* given, double* P, double* q, double* G, double* h,
                             void* work, int batch, size_t lds_bytes, int num_cus,
                             hipStream_t stream, hipError_t* err);
int launch_assemble(const PlanDev& p, const SrcTable& src, const double* params,
                    const double* given, double* P, double* q, double* G, double* h, void* work,
                    int batch, int num_cus, hipStream_t stream, hipError_t* err);
int launch_preview_matrices(const PlanDev& p, const SrcTable& src, double* PM, int batch,
                            hipStream_t stream, hipError_t* err);
int launch_box_transform(double* params, long long nparams, int batch, const int32_t* facets,
                         int nfacets, int op, const double* arg, long long arg_stride,
                         hipStream_t stream, hipError_t* err);
int launch_gather(const double* src, long long src_stride, const int32_t* index, int nnz,
                  double* dst, int batch, hipStream_t stream, hipError_t* err);
int launch_preview(const double* PM, const double* given, const double* optim, double* out,
                   int batch, int rows, int ng, int no, hipStream_t stream, hipError_t* err);

}  // namespace mpcasm
