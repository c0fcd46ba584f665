/*
 * mpcasm.h -- C ABI of libmpcasm.so, the MI355X (gfx950) batched QP-assembly
 * engine behind the mpc_interface problem-description API.
 *
 * The reference (Gepetto/mpc-interface) has no native boundary on this path:
 * everything is numpy inside one Python process.  Each entry point below
 * replaces the reference function cited next to it (paths relative to the
 * reference checkout); INTEGRATION.md shows the ctypes stubs a maintainer of
 * the reference would add to call them.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer named d_* is a DEVICE pointer
 *     (hipMalloc'ed or a torch-ROCm tensor's data_ptr()), h_* is a HOST pointer;
 *   - the caller owns every buffer; the library allocates only the device copy
 *     of a plan's tables, released by mpcasm_plan_destroy;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as
 *     void*, NULL = default stream); no call synchronises the device;
 *   - every function returns MPCASM_OK (0) or a negative mpcasm_status; no
 *     exception crosses the boundary;
 *   - all floating point data is IEEE binary64, row-major, densely packed.
 */
#ifndef MPCASM_H
#define MPCASM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mpcasm_status {
  MPCASM_OK = 0,
  MPCASM_ERR_ARG = -1,      /* null pointer, non-positive size, bad flag        */
  MPCASM_ERR_PLAN = -2,     /* malformed plan tables (magic/version/bounds)     */
  MPCASM_ERR_HIP = -3,      /* a HIP runtime call failed (see mpcasm_last_hip)  */
  MPCASM_ERR_NODEVICE = -4, /* no HIP device visible                            */
  MPCASM_ERR_LIMIT = -5     /* problem exceeds a kernel limit (LDS, sources)    */
} mpcasm_status;

typedef struct mpcasm_plan mpcasm_plan; /* opaque, immutable after creation */

/* library / device ------------------------------------------------------- */

/* ABI version of this header (major*1000 + minor). */
int mpcasm_abi_version(void);
/* Number of visible HIP devices (0 when none; never fails). */
int mpcasm_device_count(void);
/* Last hipError_t seen by the calling thread inside the library (0 = none). */
int mpcasm_last_hip(void);
/* Static string for a status code. */
const char* mpcasm_status_string(int status);
/* Process-wide options.  MPCASM_OPT_PATH selects the assembly kernels: 0 = best
 * available (persistent fused kernel when one instance fits on chip), 1 = never
 * the persistent kernel (per-instance fused kernel if it fits; the tiled kernel's scan form with its
 * pre-passes instead of the set-up fused into it), 2 = always the
 * staged K2 -> K3 -> K4 pipeline with the workspace in HBM, 3 = as 1, and a wide problem whose
 * rows are windows of generated horizon tables still composes its tiles (the tiled kernel's
 * general form instead of its Toeplitz form), 4 = as 1, and such a problem multiplies its windows
 * on the matrix core even where every Hessian term is the full horizon of one state (the tiled
 * kernel's Toeplitz form instead of its scan form, which sums P along diagonals).  The parity
 * tests use it to exercise every path; all paths give the same results. */
enum { MPCASM_OPT_PATH = 1, MPCASM_OPT_PHASE_MASK = 2, MPCASM_OPT_RESIDENT_PER_CU = 3, MPCASM_OPT_JIT = 4,
       MPCASM_OPT_P_DIRECT = 5, MPCASM_OPT_RESIDENT_GRID = 6 };
/* MPCASM_OPT_PHASE_MASK is a profiling aid (timing-only ablation of the fused
 * kernels: bit 0 compose, 1 Hessian, 2 gradient, 3 constraints, 4 input staging
 * after the first instance, 5 P/q stores, 7 register prefetch of the next instance's
 * inputs; bit 6, off by default, makes the persistent kernel also write per-wavefront
 * cycle sums of its phases into d_work, which mpcasm_workspace_bytes sizes for it;
 * default 0xBF).  Results are WRONG with any of bits 0-5 cleared -- never use it
 * outside a profile.  Bits 8-14 are A/B aids of the persistent kernel that leave the results
 * right: 8 no runs of four consecutive instances per workgroup, 10-12 runs of 2^k instead,
 * 13 the workgroups of one XCD take neighbouring instances, 14 every workgroup starts its G and
 * P streams at their first line (default: at a line of its own); bit 9 computes G without
 * storing it (results WRONG).
 * MPCASM_OPT_RESIDENT_PER_CU (tuning aid): workgroups of the persistent kernel per CU;
 * 0 (default) = chosen from the batch size, never more than are resident at once.
 * MPCASM_OPT_RESIDENT_GRID: at most this many workgroups of the persistent kernel per launch
 * (0, the default: all that are resident at once).  For callers that run launches of several plans
 * side by side on streams of their own (the structure buckets of a walker fleet): every launch gets
 * its share of the chip's workgroup slots and none waits for another to drain.
 * MPCASM_OPT_JIT: the persistent kernel compiled for the very plan by hiprtc (its sizes and
 * matrix-core trip lists become constants; same source, same results as the ahead-of-time
 * kernel): 0 (default) = for batches of at least 512 instances, when libhiprtc.so is there
 * (compiled once per plan structure and device, on the first such launch: that launch blocks
 * for the compilation, a second or two); 1 = for every batch; 2 = never.
 * MPCASM_OPT_P_DIRECT (read by mpcasm_plan_create): how the persistent kernel writes P -- 1: its
 * 4x4 blocks go from the matrix core straight to HBM; 2: collected in LDS and copied out with
 * 16-byte stores whenever P fits there beside the workspace; 0 (default): as 1 when the launch
 * writes less than ~0.55 GB, or when P in LDS would cost a workgroup per CU and an instance's
 * results are small (< 128 KB), else as 2.
 * These options are process-wide test / tuning hooks, not part of a launch's state: set them
 * before other threads start launching. */
int mpcasm_set_option(int option, int value);
/* The same choice for ONE plan (MPCASM_OPT_PATH, MPCASM_OPT_JIT, MPCASM_OPT_RESIDENT_PER_CU,
 * MPCASM_OPT_RESIDENT_GRID; value
 * -1: back to the process-wide value): part of the plan's state, read by every mpcasm_assemble on
 * it -- what a caller with several plans on several threads uses instead of the hooks above. */
int mpcasm_plan_set_option(mpcasm_plan* plan, int option, int value);
/* Needs no device: validates the tables as mpcasm_plan_create does and reports the dynamic LDS
 * bytes one workgroup of the persistent kernel needs for them -- out[0] with P handed over
 * directly, out[1] with P collected in LDS (0: the plan does not run on the persistent kernel).
 * Two workgroups share a CU's 160 KB: a plan compiler uses it to pick the layout of the workspace
 * that fits two (mpcasm.engine.Assembler, workspace="auto"). */
int mpcasm_resident_lds_bytes(const int32_t* h_itab, size_t n_itab, const double* h_dtab, size_t n_dtab,
                              int64_t out[2]);
/* Diagnostic, needs no device: validates the tables as mpcasm_plan_create does, generates the
 * per-plan constants and compiles the persistent kernel for them with hiprtc (gfx950).
 * MPCASM_OK, MPCASM_ERR_LIMIT (no persistent kernel for this plan, or no libhiprtc.so) or
 * MPCASM_ERR_HIP (compilation failed); the compiler's log goes to `log` (may be NULL). */
int mpcasm_jit_check(const int32_t* h_itab, size_t n_itab, const double* h_dtab, size_t n_dtab,
                     char* log, size_t log_capacity);

/* Compile ahead of the first launch what a launch of `batch` instances of this plan would
 * compile (the persistent kernel specialised for the plan, MPCASM_OPT_JIT): a control loop calls
 * it once with the plan's capacity, so that no tick ever blocks for a compilation; launches of
 * fewer instances then run the compiled kernel too.  Other plans keep launching meanwhile (the
 * compilation holds no lock).  Code objects are kept on disk -- $MPCASM_CACHE_DIR, else
 * $XDG_CACHE_HOME/mpcasm, else ~/.cache/mpcasm; MPCASM_NO_DISK_CACHE=1: never -- named by a hash of
 * everything the compiler sees, so every later process (and the other ranks of a node) loads
 * instead of compiling.  MPCASM_OK also when nothing needs compiling or hiprtc is missing (the
 * ahead-of-time kernel runs then); MPCASM_ERR_ARG when the current device is not the plan's. */
int mpcasm_plan_prepare(const mpcasm_plan* plan, int batch);
/* Diagnostic: out[0] = kernels compiled by this process so far, out[1] = code objects loaded from
 * the disk cache, out[2] = code objects written to it. */
int mpcasm_jit_stats(int64_t out[3]);

/* K1  horizon extension ---------------------------------------------------
 * Replaces tools.extend_matrices(N, A, B)      python/mpc_interface/tools.py:14-33
 * (C++ twin gecko::tools::extend_matrices      cpp/src/tools.cc:83-144),
 * batched over `batch` independent systems.
 *
 *   ltv == 0:  d_A [batch][n][n],     d_B [batch][n][m]      x+ = A x + B u
 *   ltv == 1:  d_A [batch][N][n][n],  d_B [batch][N][n][m]   x_{k+1} = A_k x_k + B_k u_k
 *   d_S [batch][N][n][n]      S[b][k][j][i]    = (A^{k+1})[i][j]          (A_k...A_0 when ltv)
 *   d_U [batch][m][N][N][n]   U[b][j][k][l][i] = (A^{k-l} B)[i][j], l<=k  (A_k..A_{l+1} B_l when ltv)
 *                                              = 0,                 l>k
 * i.e. U[b][j] is the reference's list element U[j] (shape N x N x n).
 * Every byte of S and U is written (zeros included).
 */
int mpcasm_fill_su(const double* d_A, const double* d_B, double* d_S, double* d_U,
                   int batch, int N, int n, int m, int ltv, void* stream);

/* plans -------------------------------------------------------------------
 * A plan is the compiled *structure* of one Formulation (index maps, flattened
 * definition graph, cost and constraint tables): what the reference re-derives
 * from dicts on every call of
 *   Formulation.make_preview_matrices          python/mpc_interface/body.py:149-193
 *   Formulation.generate_all_qp_constraints    body.py:304-320
 *   Formulation.generate_all_qp_costs          body.py:322-329
 * The tables are produced by the host plan compiler (mpcasm/plan.py); their
 * layout is documented in csrc/plan_tables.h.  h_itab / h_dtab are copied.
 */
int mpcasm_plan_create(const int32_t* h_itab, size_t n_itab,
                       const double* h_dtab, size_t n_dtab,
                       mpcasm_plan** out_plan);
int mpcasm_plan_destroy(mpcasm_plan* plan);

/* Sizes of a plan: out[0]=given_len ng, out[1]=optim_len no, out[2]=rows of
 * the stacked G (nc), out[3]=params per instance, out[4]=number of sources,
 * out[5]=row-set rows, out[6]=leading dimension of the workspace rows,
 * out[7]=preview rows (sum over definitions). */
int mpcasm_plan_sizes(const mpcasm_plan* plan, int64_t out[8]);

/* f3  sparse hand-off, first form: a plan compiled with csc=... (mpcasm/plan.py) makes
 * mpcasm_assemble write the `data` arrays of
 *   Q = scipy.sparse.csc_matrix(Q); A = scipy.sparse.csc_matrix(A)
 *   python/use_examples/simple_functional_example/biped_mpc_loop.py:57-58
 * directly -- d_P is then [batch][out[0]], d_G [batch][out[1]] (both 0 for a dense plan);
 * indptr / indices are the plan compiler's, one pattern for the whole batch (entries that
 * happen to be 0.0 are stored).  Such a plan exists on the persistent kernel only:
 * mpcasm_plan_create answers MPCASM_ERR_LIMIT when the problem does not fit on chip (assemble
 * dense and convert with mpcasm_gather then). */
int mpcasm_plan_csc_sizes(const mpcasm_plan* plan, int64_t out[2]);

/* Bytes of scratch the assembly of `batch` instances needs (device memory,
 * caller-allocated, 16-byte aligned): a part per instance and, for wide problems, a part per launch
 * behind it (the shared-model form's tables) -- a buffer sized for a larger batch serves every
 * smaller one. */
int mpcasm_workspace_bytes(const mpcasm_plan* plan, int batch, size_t* out_bytes);

/* K2+K3+K4  batched QP assembly --------------------------------------------
 * Replaces, for `batch` instances of one structure,
 *   Formulation.generate_all_qp_matrices(given)  body.py:333-348
 * on top of the preview matrices of make_preview_matrices (body.py:149-193):
 *
 *   h_src        host array of plan.n_sources DEVICE pointers: the horizon
 *                matrices ExtendedSystem.matrices[k] ([N][p][n] each,
 *                dynamics.py:199) the definitions read
 *   h_src_stride host array, elements between consecutive instances of each
 *                source (0 = one copy shared by the whole batch)
 *   d_params     [batch][n_params]  per-instance Cost / Constraint numbers
 *                (weight, aim, cross_aim; arrow, center, extreme)
 *   d_given      [batch][ng]        the 'given' vector (body.py:195-207)
 *   d_P [batch][no][no]  d_q [batch][no]  d_G [batch][nc][no]  d_h [batch][nc]
 *                qpsolvers layout: minimise 1/2 x'Px + q'x  s.t.  Gx <= h
 *                (the reference returns them as A=G, h, Q=P, q).
 *   d_work       scratch of mpcasm_workspace_bytes(plan, batch)
 *
 * Any of d_P/d_q (both) or d_G/d_h (both) may be NULL to skip that half.  The result
 * buffers are 16-byte aligned (MPCASM_ERR_ARG otherwise).
 *
 * K1 fused (plans compiled with lti=[...], mpcasm/plan.py): for a dynamics whose
 * horizon matrices are generated on chip, i.e. what
 *   tools.extend_matrices(N, A, B)             python/mpc_interface/tools.py:14-33
 * would have produced from the system's (A, B), the slot of the group's FIRST
 * horizon matrix (U_0) carries A [n][n] and the slot of its SECOND one (U_1, or S
 * when m = 1) carries B [n][m], each with its own per-instance stride (n*n, n*m or
 * 0); the group's other slots are ignored.  Such a plan runs in the persistent
 * kernel only: MPCASM_ERR_LIMIT when one instance does not fit on chip, and
 * mpcasm_preview_matrices refuses it (there is no S, U to read).
 */
int mpcasm_assemble(const mpcasm_plan* plan, const double* const* h_src,
                    const int64_t* h_src_stride, const double* d_params,
                    const double* d_given, double* d_P, double* d_q, double* d_G,
                    double* d_h, void* d_work, int batch, void* stream);

/* The same with the rows of d_given picked by an index: instance b reads d_given[d_given_index[b]]
 * and row b of everything else (parameters, per-instance sources; results to row b).  What a walker
 * fleet's structure bucket needs (biped_mpc_loop.py:41-56 for many walkers: the bucket's walkers are
 * scattered over the fleet-wide `given`): no gather pass in front of the assembly.  Persistent kernel
 * only: MPCASM_ERR_LIMIT for a plan that runs elsewhere (gather the rows and call mpcasm_assemble). */
int mpcasm_assemble_indexed(const mpcasm_plan* plan, const double* const* h_src,
                            const int64_t* h_src_stride, const double* d_params, const double* d_given,
                            const int32_t* d_given_index, double* d_P, double* d_q, double* d_G,
                            double* d_h, void* d_work, int batch, void* stream);

/* Which kernels the latest successful mpcasm_assemble on this plan launched (a record for
 * benchmarks and tests; 0 before the first launch): the persistent kernel ahead of time / compiled
 * for the plan, the per-instance fused kernel, the staged K2 -> K3 -> K4 pipeline, the tiled
 * kernel for wide problems. */
enum { MPCASM_KERNEL_NONE = 0, MPCASM_KERNEL_RESIDENT = 1, MPCASM_KERNEL_RESIDENT_JIT = 2,
       MPCASM_KERNEL_FUSED = 3, MPCASM_KERNEL_STAGED = 4, MPCASM_KERNEL_TILED = 5,
       MPCASM_KERNEL_TILED_SCAN = 6 /* the tiled kernel's scan form: P summed along diagonals */,
       MPCASM_KERNEL_SWEEP = 7 /* per-step dynamics (a plan compiled with ltv): no horizon matrix */,
       MPCASM_KERNEL_TILED_SHARED = 8 /* wide problem, every source shared by the batch: P and G as
                                         weighted sums of matrices computed once per launch */ };
int mpcasm_plan_last_kernel(const mpcasm_plan* plan);

/* K2 alone  preview matrices ------------------------------------------------
 * Replaces Formulation.make_preview_matrices   body.py:149-193
 * d_PM [batch][preview_rows][ng+no]: for every definition, in definition
 * order, the rows of [Mg | Mo] (Formulation.PM[var] = (Mg, Mo)).
 */
int mpcasm_preview_matrices(const mpcasm_plan* plan, const double* const* h_src,
                            const int64_t* h_src_stride, double* d_PM, int batch,
                            void* stream);

/* f2  batched preview --------------------------------------------------------
 * Replaces Formulation.preview(given, optim, variable)   body.py:209-219
 * for every definition at once: d_out[batch][preview_rows] =
 *   Mg @ given + Mo @ optim  with d_PM as written by mpcasm_preview_matrices.
 */
int mpcasm_preview(const double* d_PM, const double* d_given, const double* d_optim,
                   double* d_out, int batch, int rows, int ng, int no, void* stream);

/* f2, straight from the sources: the same rows without a preview matrix in memory.
 * Replaces Formulation.preview(given, optim, variable)   body.py:209-219
 * for every definition at once: d_out[batch][preview_rows] = Mg @ given + Mo @ optim, every row
 * evaluated as its combination of base rows (rows of the horizon matrices times [given ; optim],
 * kept on chip).  h_src / h_src_stride as for mpcasm_assemble; a plan compiled with lti=[...]
 * takes the groups' (A, B) in the same slots and needs d_work (mpcasm_workspace_bytes) for the
 * horizon tables it generates first. */
int mpcasm_preview_direct(const mpcasm_plan* plan, const double* const* h_src,
                          const int64_t* h_src_stride, const double* d_given, const double* d_optim,
                          double* d_out, void* d_work, int batch, void* stream);

/* f2  goal distances --------------------------------------------------------------
 * Replaces Formulation.goal_distance(given, optim, goal_name)   body.py:221-228
 * (and full_goal_distance, :230-234: the sum over the goals) for a batch, from the rows
 * mpcasm_preview_direct / mpcasm_preview wrote:
 *   d_out[b][g] = sum over the terms t of goal g (one per axis)
 *                 sum_{r < rows_t} (d_preview[b][row0_t + r] - d_params[b][aim_t])^2
 * d_terms: nterms records of 4 int32 -- goal, row0, rows, aim's parameter slot.  As in the
 * reference the whole variable counts, whatever the goal's schedule or L. */
int mpcasm_goal_distance(const double* d_preview, int64_t preview_stride, const double* d_params,
                         int64_t n_params, const int32_t* d_terms, int nterms, int ngoals,
                         double* d_out, int batch, void* stream);

/* ... and the same distances straight from the sources, without the rows ever leaving the chip:
 * what full_goal_distance needs after a solve (body.py:230-234) -- 8 bytes per goal and instance
 * instead of 8 bytes per row of every definition.  Arguments as mpcasm_preview_direct + the goal
 * table of mpcasm_goal_distance; d_out[batch][ngoals].  MPCASM_ERR_LIMIT when the plan does not run
 * on the kernel that does this (sources of an instance's own, more than 16 terms, rows of one
 * instance beyond LDS): take the rows with mpcasm_preview_direct and call mpcasm_goal_distance. */
int mpcasm_preview_goal_distance(const mpcasm_plan* plan, const double* const* h_src,
                                 const int64_t* h_src_stride, const double* d_given,
                                 const double* d_optim, const double* d_params, const int32_t* d_terms,
                                 int nterms, int ngoals, double* d_out, void* d_work, int batch,
                                 void* stream);

/* f3  sparse hand-off -----------------------------------------------------------
 * Replaces the dense -> CSC conversion in front of the solver call of the walking loop
 *   Q = scipy.sparse.csc_matrix(Q); A = scipy.sparse.csc_matrix(A)
 *   python/use_examples/simple_functional_example/biped_mpc_loop.py:57-58
 * for a whole batch with ONE pattern: d_index[k] is the flat offset (row * cols + col)
 * of the k-th stored entry, in CSC order, of a structural sparsity pattern that the plan
 * compiler derives (mpcasm/plan.py: csc_pattern); entries of the pattern that happen to
 * be 0.0 in an instance are stored as explicit zeros, so the pattern -- indptr, indices,
 * shared by the batch, as OSQP's update_values wants it -- never changes.
 *   d_dst[b][k] = d_src[b * src_stride + d_index[k]]      k < nnz, b < batch
 */
int mpcasm_gather(const double* d_src, int64_t src_stride, const int32_t* d_index, int nnz,
                  double* d_dst, int batch, void* stream);

/* f4  batched box transforms ---------------------------------------------------
 * Replaces, for every instance of a batch at once, the per-tick geometry updates of a
 *   Box                                     python/mpc_interface/restrictions.py:380-486
 * on the facets' parameters inside d_params [batch][n_params] (the arrow, center and
 * extreme fields of the box's Constraints, restrictions.py:201-219 incl. normalize()):
 *   MPCASM_BOX_RECENTER   center  = arg[axes]                 recenter_in_TS   :380-388
 *   MPCASM_BOX_TRANSLATE  center += arg[axes]                 translate_in_TS  :411-415
 *   MPCASM_BOX_ROTATE     arrow_r = arrow_r . R^T, R = arg[axes][axes]   rotate_in_TS :436-455
 *   MPCASM_BOX_SCALE      extreme *= arg[0]  (new factor / old factor)   scale_box :474-478
 *   MPCASM_BOX_MARGIN     extreme -= arg[0] * ||arrow||_F     set_safety_margin :480-486
 * After ROTATE, SCALE and MARGIN rows whose extreme turned negative are flipped
 * (extreme, arrow -> -extreme, -arrow), as Constraint.normalize() does (:180-194).
 * d_facets: nfacets records of 7 int32 -- arrow offset, arrow rows, center offset, center
 * rows, extreme offset, extreme rows, axes -- offsets into one instance's parameters.
 * d_arg [batch][arg_stride] (arg_stride 0: one argument for all instances).
 */
enum { MPCASM_BOX_RECENTER = 0, MPCASM_BOX_TRANSLATE, MPCASM_BOX_ROTATE, MPCASM_BOX_SCALE,
       MPCASM_BOX_MARGIN };
int mpcasm_box_transform(double* d_params, int64_t n_params, int batch, const int32_t* d_facets,
                         int nfacets, int op, const double* d_arg, int64_t arg_stride,
                         void* stream);

/* f4, state space: replaces Box.recenter_in_SS / translate_in_SS            restrictions.py:390-404, 417-431
 * (through Constraint.SS_to_TS, :240-250) for the boxes whose facets carry an L -- those of
 * Box.state_space (:342-378: L = the facet's normal in the state space) and task-space boxes
 * built with L.  Per instance a state-space point p [ss_dim][axes] (one column per task-space
 * axis); facet f gets, per centre row r and axis a,
 *   MPCASM_BOX_RECENTER   center[r][a]  = sum_v L[f][a][r][v] p[v][a]
 *   MPCASM_BOX_TRANSLATE  center[r][a] += sum_v L[f][a][r][v] p[v][a]
 * d_L [nfacets][axes][lrows][ss_dim]: the facets' L matrices (the same for every instance:
 * they are part of the plan's structure); a facet's centre field must have lrows rows.
 * d_arg [batch][arg_stride] with arg_stride = ss_dim * axes (0: one point for all). */
int mpcasm_box_transform_ss(double* d_params, int64_t n_params, int batch,
                            const int32_t* d_facets, int nfacets, int op, const double* d_L,
                            int lrows, int ss_dim, const double* d_arg, int64_t arg_stride,
                            void* stream);

/* f3, the "or": the solve itself, batched ---------------------------------------------
 * Replaces, for every instance of a batch at once, the solver call of the walking loop
 *   self.optim = osqp_solve_qp(P=Q, q=q, G=A, h=h)
 *   python/use_examples/simple_functional_example/biped_mpc_loop.py:60
 * on the dense results of mpcasm_assemble where they lie:  min 1/2 x'Px + q'x  s.t.  Gx <= h  (no
 * equalities on this path, body.py:331).  `iters` iterations of OSQP's ADMM (Stellato et al., Math.
 * Prog. Comp. 12 (2020), Algorithm 1) with the steps rho, sigma, alpha (OSQP's defaults: 0.1, 1e-6,
 * 1.6) and without its problem scaling, adaptive rho and polishing:
 *   (P + sigma I + rho G'G) xt = sigma x - q + G'(rho z - y);  zt = G xt;  x+ = alpha xt + (1 - alpha) x
 *   z+ = min(alpha zt + (1 - alpha) z + y / rho, h);  y+ = y + rho (alpha zt + (1 - alpha) z - z+)
 * d_P [batch][no][no] (symmetric), d_q [batch][no], d_G [batch][nc][no], d_h [batch][nc].
 * d_x [batch][no], d_y [batch][nc], d_z [batch][nc]: the iterates -- read when warm != 0 (a walking
 * loop starts a tick from the last one's), else started from x = 0, y = 0, z = min(0, h); always
 * written.  d_res (may be NULL) [batch][2]: OSQP's residuals |Gx - z|_inf and |Px + q + G'y|_inf after
 * the last iteration -- the caller decides whether to iterate on.  An instance whose
 * P + sigma I + rho G'G is not positive definite gets NaNs.
 * d_kinv (may be NULL) [batch][no][no]: the inverse of P + sigma I + rho G'G.  kinv_valid == 0: written by this
 * call; != 0: READ instead of factoring -- for a caller whose P and G (and rho, sigma) did not change since the call
 * that wrote it: the same model and structure with a new `given` changes q and h only (body.py:236-302), and the
 * factorisation is the larger part of a call of a few dozen iterations.  MPCASM_ERR_LIMIT when one instance's
 * matrices do not fit on chip ((no + max(nc, no)) * (no | 1) + 8 no + 4 nc + 4 max(nc, no) doubles in 156 KB of LDS:
 * the biped up to N = 24 and beyond; not C3). */
int mpcasm_admm(int no, int nc, const double* d_P, const double* d_q, const double* d_G,
                const double* d_h, double* d_x, double* d_y, double* d_z, double* d_res, double rho,
                double sigma, double alpha, int iters, int warm, int batch, double* d_kinv, int kinv_valid,
                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MPCASM_H */
