// kernels.h -- internal launch interface between capi.hip and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mpcasm.h"
#include "plan_dev.h"
#include "plan_tables.h"

namespace mpcasm {

// capi.hip: raise a kernel's dynamic-LDS limit to the CU's whole LDS -- ONCE per (function, device),
// process-wide and under a mutex.  The attribute belongs to the function, not to the calling thread
// or the plan: a limit that followed each launch's size could be lowered by one host thread under
// another thread's larger plan (and a replayed graph node of the larger plan would meet it).
constexpr int CU_LDS_BYTES = 160 * 1024;
hipError_t allow_whole_lds(const void* fn);

// fill.hip
int launch_fill_su(const double* A, const double* B, double* S, double* U, int batch, int N, int n,
                   int m, int ltv, hipStream_t stream, hipError_t* err);

// assemble.hip
size_t assemble_workspace_bytes(const PlanDev& p, int batch);
int launch_assemble_staged(const PlanDev& p, const SrcTable& src, const double* params,
                           const double* given, double* P, double* q, double* G, double* h,
                           void* work, int batch, hipStream_t stream, hipError_t* err);
// tiled.hip
bool tiled_eligible(const PlanDev& p);
size_t tiled_workspace_bytes(const PlanDev& p, int batch);
int launch_assemble_tiled(const PlanDev& p, const SrcTable& src, const double* params,
                          const double* given, double* P, double* q, double* G, double* h,
                          void* work, int batch, hipStream_t stream, hipError_t* err,
                          const int32_t* h_itab);
int launch_lti_tables(const PlanDev& p, const SrcTable& src, double* work, int batch,
                      const int32_t* h_itab, SrcTable* eff, hipStream_t stream);
// sweep.hip: a dynamics compiled as ltv -- per-step, per-instance (A_k, B_k), no horizon matrix
bool sweep_eligible(const PlanDev& p);
int launch_assemble_sweep(const PlanDev& p, const SrcTable& src, const double* params,
                          const double* given, double* P, double* q, double* G, double* h, int batch,
                          hipStream_t stream, hipError_t* err, const int32_t* h_itab);
// preview.hip
int launch_preview_direct(const PlanDev& p, const SrcTable& eff, const double* given,
                          const double* optim, double* out, int batch, int num_cus,
                          hipStream_t stream, hipError_t* err, const int32_t* h_itab = nullptr);
int launch_preview_goals(const PlanDev& p, const SrcTable& eff, const double* given, const double* optim,
                         const double* params, long long nparams, const int32_t* terms, int nterms,
                         int ngoals, double* out, int batch, int num_cus, hipStream_t stream,
                         hipError_t* err, const int32_t* h_itab);
int launch_goal_distance(const double* preview, long long preview_stride, const double* params,
                         long long nparams, const int32_t* terms, int nterms, int ngoals,
                         double* out, int batch, hipStream_t stream, hipError_t* err);
// fused.hip
size_t fused_lds_bytes(const PlanDev& p, int nw);
int launch_assemble_fused(const PlanDev& p, const SrcTable& src, const double* params,
                          const double* given, double* P, double* q, double* G, double* h,
                          int batch, size_t lds_bytes, hipStream_t stream, hipError_t* err);
// resident.hip
size_t resident_lds_bytes(const PlanDev& p);
int resident_choose_p_direct(const PlanDev& p, int option);
int resident_p_direct_for(const PlanDev& p, int batch);
// whether this launch's buffers meet the alignment the plan's input loads assume
bool resident_inputs_aligned(const PlanDev& p, const SrcTable& src, const double* params,
                             const double* given);
int launch_assemble_resident(const PlanDev& p, const SrcTable& src, const double* params,
                             const double* given, double* P, double* q, double* G, double* h,
                             void* work, int batch, size_t lds_bytes, int num_cus,
                             hipStream_t stream, hipError_t* err);
// h_itab / device: the plan's tables on the host and its device, for the per-plan compiled
// kernel (jit.hip); nullptr = ahead-of-time kernels only
// options in force for the launch the calling thread is making (capi.hip: the plan's own where
// mpcasm_plan_set_option gave it some, else the process-wide ones of mpcasm_set_option)
extern thread_local int t_path, t_jit, t_per_cu, t_grid;
extern thread_local int t_last_kernel;  // MPCASM_KERNEL_*: what launch_assemble launched
int launch_assemble(const PlanDev& p, const SrcTable& src, const double* params,
                    const double* given, double* P, double* q, double* G, double* h, void* work,
                    int batch, int num_cus, hipStream_t stream, hipError_t* err,
                    const int32_t* h_itab = nullptr, int device = 0, bool indexed = false);
int launch_preview_matrices(const PlanDev& p, const SrcTable& src, double* PM, int batch,
                            hipStream_t stream, hipError_t* err);
int launch_box_transform(double* params, long long nparams, int batch, const int32_t* facets,
                         int nfacets, int op, const double* arg, long long arg_stride,
                         hipStream_t stream, hipError_t* err);
int launch_box_transform_ss(double* params, long long nparams, int batch, const int32_t* facets,
                            int nfacets, int op, const double* L, int lrows, int ss_dim,
                            const double* arg, long long arg_stride, hipStream_t stream,
                            hipError_t* err);
int launch_gather(const double* src, long long src_stride, const int32_t* index, int nnz,
                  double* dst, int batch, hipStream_t stream, hipError_t* err);
int launch_admm(int no, int nc, const double* P, const double* q, const double* G, const double* h,
                double* x, double* y, double* z, double* res, double rho, double sigma, double alpha,
                int iters, int warm, int batch, double* kinv, int kinv_valid, hipStream_t stream,
                hipError_t* err);
int launch_preview(const double* PM, const double* given, const double* optim, double* out,
                   int batch, int rows, int ng, int no, hipStream_t stream, hipError_t* err);

}  // namespace mpcasm
