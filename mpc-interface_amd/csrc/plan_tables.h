// plan_tables.h -- layout of the flat plan tables shared by the host plan
// compiler (mpcasm/plan.py, which mirrors these constants) and the kernels.
//
// A plan is two arrays: int32 `itab` (structure) and double `dtab` (numbers
// that are the same for every instance of the batch: definition coefficients
// with L matrices folded in).  itab starts with a header of H_WORDS words; all
// H_OFF_* are word offsets into itab, all H_DOFF_* element offsets into dtab.
//
// Vocabulary (reference python/mpc_interface/body.py):
//   column   one entry of [given | optim]: c < ng is given c, else optim c-ng
//   source   one horizon matrix ExtendedSystem.matrices[k], shape [N][p][n]
//            (dynamics.py:199), shared by the batch or one per instance
//   base     a variable defined directly by a dynamics object (a key of
//            Formulation.of): a domain variable (identity rows) or a state of
//            an ExtendedSystem (rows gathered from sources), body.py:158-177
//   segment  a run of columns of one base variable fed by one source slice
//   row      one row of [Mg | Mo] of some definition, stored as a CSR list of
//            (base, base_row, coefficient): the flattened definition graph of
//            body.py:179-193
//   row-set  consecutive rows consumed by a cost or a constraint (a variable
//            restricted to a schedule, L folded in)
//   gterm    one accumulation  P += w A^T B,  q += w A^T r  of body.py:292-300
//   limit    one Constraint (restrictions.py:15) -> rows of G, h
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

namespace mpcasm {

constexpr int32_t PLAN_MAGIC = 0x4D504341;  // 'MPCA'
constexpr int32_t PLAN_VERSION = 30;

enum HeaderWord : int {
  H_MAGIC = 0,
  H_VERSION,
  H_NG,            // given_len
  H_NO,            // optim_len
  H_NC,            // rows of stacked G
  H_NPARAMS,       // doubles per instance in d_params
  H_NSRC,          // number of sources
  H_NBASE,         // number of base variables
  H_NSEG,          // number of segments
  H_RTOT,          // rows of all row-sets (workspace rows per instance)
  H_NENT,          // CSR entries of the row-set program
  H_NGTERM,
  H_NLIMIT,
  H_NLAX,          // limit-axis records
  H_PMROWS,        // rows of the preview program (all definitions)
  H_PM_NENT,
  H_LDV,           // workspace leading dimension (>= no + 2, even)
  H_OFF_SEG,
  H_OFF_COLSEG,    // [nbase][ng+no] segment id or -1
  H_OFF_ROWPTR,    // [RTOT+1]
  H_OFF_ENTBASE,   // [NENT]
  H_OFF_ENTK,      // [NENT]
  H_OFF_GTERM,
  H_OFF_LIMIT,
  H_OFF_LAX,
  H_OFF_ROWLIMIT,  // [NC] limit index of every row of the stacked G
  H_OFF_PM_ROWPTR, // [PMROWS+1]
  H_OFF_PM_ENTBASE,
  H_OFF_PM_ENTK,
  H_DOFF_ENTCOEF,    // [NENT]
  H_DOFF_PM_ENTCOEF, // [PM_NENT]
  H_NITAB,         // total words of itab (self check)
  H_NDTAB,         // total elements of dtab
  // ---- fused program (small problems: everything of one instance on chip) ----
  H_FUSED_OK,      // 1 when the sections below are present
  H_ARENA_TOTAL,   // doubles of the on-chip source arena (slot 0 holds the constant 1.0)
  H_OFF_ARENA,     // [NSRC][2] arena offset, size of every source
  H_NFD,           // composed elements of the workspace that are structurally non-zero
  H_OFF_FD_IDX,    // [NFD] workspace index r*ldv + c  (c == no: the d column)
  H_OFF_FD_PTR,    // [NFD+1] op range of every element
  H_NOPS,
  H_OFF_OP,        // [NOPS][2]: arena offset of the source value; (given index + 1) << 16 | coef id
  H_NCOEF,
  H_DOFF_COEFPOOL, // [NCOEF] distinct coefficients
  // ---- resident program (persistent fused kernel: tables live in registers / LDS) ----
  H_RS_OK,
  H_RS_JC,          // compose ops per thread (RS_NT threads per instance)
  H_RS_SYM,         // 1: every Hessian term has A == B, only tiles ti <= tj are computed
  H_RS_NTRIP,
  H_OFF_RS_SRC,     // [JC][RS_NT] image offset of the op's source value
  H_OFF_RS_GIDX,    // [JC][RS_NT] image offset of the op's given value (or of a constant 1.0)
  H_OFF_RS_DST,     // [JC][RS_NT] workspace index the running sum goes to (| RS_DST_ACC: added
                    //             to it, the element is shared by two threads), or -1
  H_DOFF_RS_COEF,   // [JC][RS_NT]
  H_OFF_RS_TRIP,    // [NTRIP + 2][8]: up to 16 rows of one gterm into one pack of blocks, see RT_* below;
                    //    two all-zero records behind the last (read ahead)
  H_OFF_RS_WTRIP,   // [RS_WAVES][2] first trip, trip count of every wavefront
  H_RS_NSPLIT,      // workspace elements composed by two threads
  H_OFF_RS_SPLIT,   // [NSPLIT] their workspace indices (zeroed before every compose)
  H_OFF_RS_RR,      // [NC][RS_RR_WORDS] row records of the stacked G (see resident.hip)
  // the input image of one instance in LDS: [1 | sources | given, 1 | params, 0 | 0...],
  // filled by LDS-DMA loads of RS_UNIT bytes per lane, 64 lanes per chunk
  H_RS_UNIT,        // 16 or 4 (bytes per lane)
  H_RS_NCHUNK,
  H_OFF_RS_INMETA,  // [NCHUNK * 64][2] input stream, byte offset inside the instance's slice
  H_RS_IMG,         // doubles of the image (a multiple of 128)
  H_RS_IMG_GIVEN,   // image offset of given[0]  (given[ng] reads 1.0)
  H_RS_IMG_PARAMS,  // image offset of params[0] (params[nparams] reads 0.0)
  H_DOFF_RS_CONST,  // [4] 1, 1, 0, 0: the constant input stream (even offset)
  H_DOFF_DIAGCOEF,  // [NDIAGCOEF] coefficients of the diagonal gterms
  H_NDIAGCOEF,
  // source groups generated on chip (K1 fused into the persistent kernel): the horizon
  // matrices U_0..U_{m-1}, S of an LTI system are not read; its A and B arrive through the
  // streams of the group's first two sources (4-byte loads into a ring of two slots, two
  // instances ahead) and the image gets tables built from them
  H_RS_NLTI,
  H_OFF_RS_LTI,     // [NLTI][RS_LTI_WORDS], see LT_* below
  H_RS_IMG_DMA,     // doubles of the image that the loads fill (a multiple of 128, <= RS_IMG)
  H_RS_AB,          // doubles of one ring slot holding every group's A and B (a multiple of 32)
  H_OFF_RS_ABMETA,  // [RS_AB * 2][2] input stream, byte offset of every 4-byte lane of a slot
  H_RR_PACKED,      // 1: every row record of G carries its RR_PACKED words (no is even, rows
                    //    have at most two axes, offsets and parameter slots fit 16 bits)
  H_OFF_RS_DPAR,    // [NO][4] per column of the unknowns: weight, aim parameter slots of the (at most
                    //    RS_DIAG_MAX = 2) diagonal gterms on it (free: slot NPARAMS, always 0.0)
  H_DOFF_RS_DCOEF,  // [NO][2] their coefficients (free: 0.0)
  H_RS_NGDESC,      // RS_GDESC_PIECES * RS_GDESC_THREADS when the table below exists, else 0
  H_OFF_RS_GDESC,   // [RS_NGDESC][2] small problems: a ready-made descriptor of every 16-byte piece of
                    //    G; piece e = columns 2cp, 2cp+1 of row R = e / (no/2):
                    //    (voff0 + 2cp) | (voff1 + 2cp) << 16, arrow0 | arrow1 << 16 -- the two axes
                    //    in either order (the one that can be non-zero first, see H_RS_GSINGLE)
  // the preview matrices [Mg | Mo] element by element (0 elements: the tables are absent and
  // K2 alone walks the row program): H_OFF_PM_MAP [PMROWS * (NG + NO)] index of the element's
  // op list or -1 (structural zero); element i = sum over ops H_OFF_PM_FDPTR[i] .. [i+1] of
  // pool[cid] * source[sid][offset]; an op is two words: offset inside the source, sid | cid << 8
  // (sid 255: the constant 1)
  H_PM_NFD,
  H_OFF_PM_MAP,
  H_OFF_PM_FDPTR,   // [PM_NFD + 1]
  H_OFF_PM_OP,      // [PM_NOPS][2]
  H_PM_NOPS,
  H_DOFF_PM_POOL,   // [PM_NPOOL] distinct coefficients
  H_PM_NPOOL,
  H_RS_NZBLK,       // 4x4 blocks of P no term reaches (both triangles): exact zeros
  H_OFF_RS_ZBLK,    // [RS_NZBLK] block row << 8 | block column
  H_RS_GSINGLE,     // bit u * 4 + w: every piece of G of round u of stream wave w (descriptor table)
                    //    has at most ONE axis that can be non-zero in its columns -- the first of
                    //    its descriptor: the second workspace row and arrow are not read
  // CSC hand-off (biped_mpc_loop.py:57-58 for a batch): a plan with CSC_PNNZ != 0 makes
  // mpcasm_assemble write, instead of dense P and G, the data arrays of their CSC forms on a
  // fixed pattern -- P [batch][CSC_PNNZ], G [batch][CSC_GNNZ]; q and h as ever.  Only the
  // persistent kernel runs such a plan (P collected in LDS).
  H_CSC_PNNZ,
  H_OFF_CSC_P,      // [CSC_PNNZ] where the k-th stored entry sits in the LDS copy of P:
                    //    row * (NO rounded up to even) + column
  H_CSC_GNNZ,
  H_OFF_CSC_G,      // [CSC_GNNZ][2] per stored entry (R, c) of G, from R's row record:
                    //    (voff0 + c) | (voff1 + c) << 16, arrow0 | arrow1 << 16 (either order)
  H_CSC_GSINGLE,    // 1: at most one axis of every stored entry can be non-zero, the first
  // ---- column tables (every plan): what one column of one base variable reads.  Entry
  // (base b, column c) is two words: element offset inside its stream for base row 0, and
  // rs | stream << 24 (rs: signed 24 bits, elements per base row): the value of base row k in
  // that column is stream[offset + k * rs].  Streams 0 .. NSRC-1 are the launch's sources, stream
  // T_SID_CONST is the plan's dtab: a column no segment covers reads the 0.0 at
  // dtab[T_DOFF_DELTA] with rs = 0; an identity block reads the table of zeros with a single 1.0
  // behind it (rs = -1).  CIG: the given columns [NBASE][NG]; CIO: the unknowns
  // [NBASE][T_NOP], T_NOP = NO rounded up to whole blocks of T_BLOCK columns (the pad reads 0.0).
  H_T_CI_OK,
  H_T_NOP,
  H_OFF_T_CIG,
  H_OFF_T_CIO,      // (16-byte aligned: the tiled kernel reads two columns at once)
  H_T_DOFF_DELTA,   // [T_NDELTA] dtab: 0.0, then 2L+1 doubles all zero but the middle one
  H_T_NDELTA,
  // ---- tiled program (wide problems, no >= T_BLOCK: one workgroup per T_BLOCK x T_BLOCK block of
  // P, the workspace rows composed 16 at a time straight into the matrix core's LDS tiles) ----
  H_T_OK,
  H_T_NSTAGE,
  H_OFF_T_STAGE,    // [T_NSTAGE][T_STAGE_WORDS] see TS_* below
  H_T_NLTI,         // source groups whose horizon tables a pre-pass generates from (A, B)
  H_OFF_T_LTI,      // [T_NLTI][T_LTI_WORDS] see TL_* below
  H_OFF_T_LTI_IDS,  // the groups' source ids: U_0 .. U_{m-1}, S each
  H_T_WORK,         // doubles of scratch per instance: d [RTOT rounded up to even], then the tables
  H_OFF_T_GROW,     // [NC][RS_AXMAX] workspace row of every axis of every row of G (-1: no such axis)
  H_OFF_T_SROW,     // [T_NSTAGE][2][16] base row k of every A row / B row of a simple stage (else 0)
  H_T_DOFF_SCOEF,   // [T_NSTAGE][2][16] dtab: its coefficient (0.0 behind the stage's last row)
  H_OFF_T_PIG,      // [T_NSTAGE][16][T_PIG_MAX][2] rows of G that are arrow * (this A row): row of G,
                    //    arrow's parameter slot (-1: none); every such row of G is listed once
  H_T_NGREST,       // rows of G that ride on no stage (several axes, rows no cost reads ...)
  H_OFF_T_GREST,    // [T_NGREST] their indices, ascending
  H_OFF_T_BROW0,    // [NBASE + 1] first row of every base variable among all base rows; [NBASE] = total
  H_OFF_T_BCOLPTR,  // [NBASE + 1] per base variable: its columns some segment covers, as a range of ...
  H_OFF_T_BCOLS,    // ... this list of column indices in [given | unknowns] (ascending per base)
  H_T_TOEPLITZ,     // 1: every stage is TS_FLAG_TOEPLITZ (one generated group): the kernel keeps the
                    //    group's TB table in LDS and reads the matrix core's operands out of it
  H_RS_NGFIX,       // pieces of G (descriptor table) that need both of their axes while the rest of their
                    //    round needs one, at most one per stream-wave thread: the round adds the second
                    //    axis for that lane alone; 0: the table below is absent
  H_OFF_RS_GFIX,    // [RS_GDESC_THREADS][2] per thread: workspace index of the second axis (as in RS_GDESC)
                    //    | u << 16 -- the piece is its e = t + u RS_GDESC_THREADS --, the second arrow's
                    //    parameter slot; a thread without such a piece: 0 | RS_GFIX_NONE << 16, NPARAMS
  // The persistent kernel's workspace V (plan.py Workspace): row major, leading dimension RS_LDV,
  // RS_VROW0 zero rows in front; a row holds its *window* of the unknowns -- dense (RS_COMPACT 0): all
  // of them, RS_LDV = LDV, RS_VD = NO, RS_VROW0 = 0; compact: the 4-column blocks its row-set can be
  // non-zero in, so that a wide problem of several axes fits two workgroups per CU -- then d in
  // column RS_VD and the one behind it.  Trip offsets (RT_A, RT_B) have the window's first column
  // taken off, so that block bi of the unknowns is found at + 32 bi all the same.
  H_RS_COMPACT,
  H_RS_LDV,
  H_RS_VD,
  H_RS_VROW0,
  H_OFF_RS_RRWIN,   // [NC] (compact) per row of G the windows of its two axes in column pairs:
                    //    first | count << 8, the second axis << 16 (a missing axis: 0)
  // f2 (preview.hip): the column tables unrolled per base row, and the definitions' row entries by
  // base row, ready to be copied into LDS once per workgroup
  H_T_NP1,          // entries of the table below; -1: absent (the column tables do not cover the plan)
  H_OFF_T_P1PTR,    // [total base rows + 1] a base row's entries, as a range of ...
  H_OFF_T_P1ENT,    // ... [T_NP1][2]: element offset in its stream (for this very row),
                    //     column of [given | unknowns] | stream << 24
  H_OFF_T_P2Y,      // [PM_NENT] per entry of a definition's row: its base row among all base rows
  // ---- scan form of the tiled kernel (toeplitz_scan_kernel).  A Toeplitz plan (H_T_TOEPLITZ) whose
  // every Hessian term is w (c M_i)^T (c M_i) over ALL N rows of one state i of the generated group:
  // M_i[k][(j, l)] = T_ij[k - l] (tools.py:27-31), so the block of P on the columns of inputs j, j' is
  //   P[(j,l)][(j',l')] = C[(j,l)][(j',l')] + P[(j,l+1)][(j',l'+1)]      (nothing behind l = N-1),
  //   C[r][c] = sum_g w_g c_g^2 M_g[N-1][r] M_g[N-1][c]                    (the states' LAST rows):
  // a sum along diagonals, O(K) multiply-adds per element instead of the O(K N) of the product.
  H_T_SCAN,           // K = number of such terms (1 .. T_SCAN_KMAX); 0: the plan has no scan form
  H_T_SCAN_NBLK,      // inputs of the group that are unknowns (<= T_SCAN_BLKMAX)
  // The kernel keeps the group's table without the zero halves of TB: Tc[(i m + j) N + d] = T_ij[d].
  H_OFF_T_SCAN_BLK,   // [NBLK][2] first column of the input's N unknowns, j N: what row k of state i
                      //    holds in the input's column l is Tc[i m N + k + j N - l] for l <= k, else 0
  H_OFF_T_SCAN_GT,    // [K][T_SCAN_GT_WORDS] i m N, weight slot, aim slot, first row of d
  H_T_DOFF_SCAN_GC,   // [K] dtab: the term's coefficient c
  H_OFF_T_SCAN_GROW,  // [NC][2] per row of G that is arrow * c * (row k of state i): i m N + k, the
                      //    arrow's parameter slot; else -1, -1
  H_T_DOFF_SCAN_GCOEF,// [NC] dtab: that row's c
  H_T_SCAN_NGREST,    // the other rows of G (composed through the column tables)
  H_OFF_T_SCAN_GREST, // [T_SCAN_NGREST] ascending
  H_OFF_T_SCAN_COLBLK,// [NO] per unknown the index of its block in SCAN_BLK, or -1
  H_T_SCAN_NOTHER,    // unknowns in no block (their rows and columns of P hold diagonal terms only)
  // ---- sweep kernel (sweep.hip): a dynamics compiled as `ltv`, x+ = A_k x + B_k u with its own
  // (A_k, B_k) per step and instance (BASELINE config C5).  No horizon matrix is ever formed: with
  // Phi(k, l) = A_k ... A_l, row k of an output c . x holds c^T Phi(k, l+1) B_l in the columns of step
  // l <= k (tools.py:27-31 with per-step matrices), and
  //   P[(j,l)][(j',l')] = (Psi_l B_l[:,j]) . U[l][l'][:,j'] (l' <= l),  Psi_l = W_l + A_{l+1}^T Psi_{l+1} A_{l+1}
  //   q[(j,l)] = B_l[:,j] . lam_l,   lam_l = rho_l + A_{l+1}^T lam_{l+1}
  // with W_l = sum w c c^T and rho_l = sum w (d - aim) c over the cost rows of step l.
  H_SW_OK,            // 1: the plan runs on the sweep kernel (and on nothing else)
  H_SW_N,             // states (<= SW_NMAX)
  H_SW_M,             // inputs (<= SW_MMAX)
  H_SW_HORIZON,
  H_SW_SRC_A,         // source slots that carry A [N][n][n] and B [N][n][m] of every instance
  H_SW_SRC_B,
  H_SW_NAXES,         // axes sharing the system (<= SW_AXMAX): each has its own initial state and inputs
  H_OFF_SW_AXIS,      // [NAXES][SW_AXIS_WORDS] given column of the initial state, first unknown of input j
  H_SW_NTERM,
  H_OFF_SW_TERM,      // [NTERM][SW_TERM_WORDS] a cost's rows on one axis, see ST_* below
  H_SW_NLIM,
  H_OFF_SW_LIM,       // [NLIM][SW_LIM_WORDS + SW_AXMAX * SW_LAX_WORDS] a limit, see SL_* / SX_* below
  H_OFF_SW_COL,       // [NO] per unknown: axis | input << 8 | step << 16
  H_SW_DOFF_CVEC,     // [NCVEC][SW_NMAX] dtab: the combinations c of the states (0 beyond n)
  H_SW_NCVEC,
  // what happens at every step, as lists the kernel walks: the cost rows of step l are the terms
  // CENT[CPTR[l] .. CPTR[l+1]), the lines of G of step l are GENT[GPTR[l] .. GPTR[l+1]) = (limit, line)
  H_OFF_SW_CPTR,      // [HORIZON + 1]
  H_OFF_SW_CENT,      // [NCENT]
  H_SW_NCENT,
  H_OFF_SW_GPTR,      // [HORIZON + 1]
  H_OFF_SW_GENT,      // [NGENT][2]
  H_SW_NGENT,
  H_OFF_RS_PROG,      // [JC][RS_NT][4] the compose program once more, one 16-byte record per op and thread:
                      //    RS_SRC | RS_GIDX << 16, RS_DST, the two halves of RS_COEF (16-byte aligned)
  H_T_SCAN_FUSED,     // 1: the scan kernel makes its own table and its own d (no pre-pass, no scratch): `given` is the
                      // group's initial state in order, every workspace row a row of one of the K terms, no
                      // row of G through the column tables
  H_WORDS = 160
};
static_assert(H_T_SCAN_FUSED < H_WORDS, "plan header");
constexpr int SW_NMAX = 4, SW_MMAX = 4, SW_AXMAX = 4, SW_AXIS_WORDS = 8, SW_TERM_WORDS = 8, SW_LIM_WORDS = 8,
              SW_LAX_WORDS = 8;
// a cost term on one axis: rows i = 0 .. ST_COUNT-1 are c . x of step ST_K0 + i ST_KSTEP
enum { ST_AXIS = 0, ST_K0, ST_KSTEP, ST_COUNT, ST_WPARAM, ST_AIMPARAM, ST_CVEC, ST_PAD };
// a limit: rows SL_OUT0 .. + SL_COUNT - 1 of G, h; per axis (SL_NAXES records of SW_LAX_WORDS behind
// the head): line i is arrow[i] * (c . x of step SX_K0 + i SX_KSTEP) on axis SX_AXIS; the arrow of
// line i in parameter SX_ARROW + i SX_ARROW_STEP, the centre likewise, the extreme in SL_EXTREME (+ i)
enum { SL_OUT0 = 0, SL_COUNT, SL_NAXES, SL_EXTREME, SL_EXTREME_STEP, SL_PAD0, SL_PAD1, SL_PAD2 };
enum { SX_AXIS = 0, SX_K0, SX_KSTEP, SX_CVEC, SX_ARROW, SX_ARROW_STEP, SX_CENTER, SX_CENTER_STEP };
constexpr int T_SCAN_KMAX = 16, T_SCAN_BLKMAX = 8, T_SCAN_NMAX = 64, T_SCAN_GT_WORDS = 4;
enum { SG_SBOFF = 0, SG_WPARAM, SG_AIMPARAM, SG_DROW };

constexpr int T_BLOCK = 128;       // columns of a block of P (tiled kernel)
constexpr int T_SID_CONST = 32;    // the stream that is the plan's own dtab
// stage record of the tiled kernel: up to 16 consecutive rows of one gterm.  TS_MASKA / TS_MASKB:
// bit t set when 16-column tile t of the unknowns can be non-zero in the A / B rows (64 bits; tiles
// beyond 62 fold onto bit 63).  TS_INFO = rows | flags << 8 | class << 16.  The stages are sorted
// by class: 0 = no Hessian part (gradient, rows of G only); n = 1..4: inside every 64-column
// quarter of the unknowns only the first n 16-column tiles of the A and of the B rows can be
// non-zero, so a wavefront multiplies the first n x n tiles of its 64 x 64 quadrant.  TS_BASE =
// base of the A rows | base of the B rows << 16 when every row of the stage is ONE entry of one
// base (TS_FLAG_SIMPLE_*).  TS_FLAG_G: some A row carries rows of G (H_OFF_T_PIG).
// TS_FLAG_TOEPLITZ (every stage of a plan, or none: H_T_TOEPLITZ): the A rows (and the B rows of a
// Hessian stage) are rows k0 .. k0 + rows - 1 of ONE state of the plan's single generated group
// with one coefficient (H_T_DOFF_SCOEF[stage][side][0]); TS_UA / TS_UB = (state * m) * 2N + k0:
// with the group's TB table (TL_*) in LDS, row r of the stage reads, in a column whose table entry is
// (offset, rs = 1) of one of the group's U_j, the element TB[offset - state * m * 2N + TS_U + r] --
// a window of a Toeplitz table, no tile is ever built.
enum { TS_AROW = 0, TS_BROW, TS_DROW, TS_INFO, TS_WPARAM, TS_AIMPARAM, TS_MASKA_LO, TS_MASKA_HI,
       TS_MASKB_LO, TS_MASKB_HI, TS_BASE, TS_PAD, TS_UA, TS_UB, TS_SBOFFA, TS_SBOFFB,
       T_STAGE_WORDS = 16 };
enum { TS_FLAG_P = 1, TS_FLAG_HALF = 2, TS_FLAG_SIMPLE_A = 4, TS_FLAG_SIMPLE_B = 8, TS_FLAG_SAME = 16,
       TS_FLAG_G = 32, TS_FLAG_TOEPLITZ = 64 };
constexpr int T_PIG_MAX = 2;
// generated group: sizes, where its source ids start in T_LTI_IDS, scratch offsets (doubles, per
// instance) of TA = S itself [N][n][n] and of TB [n][m][2N]: row (i, j) holds N zeros, then
// (A^d B)[i][j], d < N, so that U_j[k][l][i] = TB[i][j][N + k - l] for every k, l (zeros above the
// diagonal included)
enum { TL_N = 0, TL_M, TL_HORIZON, TL_IDS, TL_TA, TL_TB, TL_PAD0, TL_PAD1, T_LTI_WORDS = 8 };

// segment record
enum { SEG_SRC = 0, SEG_OFF0, SEG_ROWSTRIDE, SEG_ELEMSTRIDE, SEG_DST0, SEG_LEN, SEG_KIND, SEG_PAD, SEG_WORDS = 8 };
enum { SEG_KIND_GATHER = 0, SEG_KIND_IDENTITY = 1 };

// gterm record:  P[:, :] += w * A^T B   (when GT_FLAG_P),   q += s * w * A^T (d[D_OFF + k] - aim)
// GT_MASKA / GT_MASKB: bit t set when 16-column tile t of the optim columns holds a
// structural non-zero in the A / B rows (tiles >= 30 fold onto bit 30)
enum { GT_AOFF = 0, GT_BOFF, GT_NROWS, GT_WPARAM, GT_DOFF, GT_AIMPARAM, GT_FLAGS, GT_MASKA, GT_MASKB, GT_PAD, GT_WORDS = 10 };
// GT_FLAG_DIAG: the term's rows are coef_k e_{c0+k} on the unknowns (a cost on a free
// variable itself): no workspace rows; P[c][c] += (w coef) coef, q[c] += w (coef (0 - aim))
// for c = c0 + k, with c0 = GT_AOFF, coef at dtab[H_DOFF_DIAGCOEF + GT_BOFF + k], k < GT_NROWS
enum { GT_FLAG_P = 1, GT_FLAG_HALF = 2, GT_FLAG_DIAG = 4 };

// limit record
enum {
  LM_OUT0 = 0, LM_NROWS, LM_NAXES, LM_LAX0,
  LM_ARROW_P, LM_ARROW_ROWS, LM_CENTER_P, LM_CENTER_ROWS, LM_EXTREME_P, LM_EXTREME_ROWS,
  LM_PAD0, LM_PAD1, LM_WORDS = 12
};
// limit-axis record
enum { LX_ROWOFF = 0, LX_ROWS, LX_WORDS = 2 };

constexpr int MAX_SOURCES = 32;
// RS_NT threads per instance = RS_WAVES wavefronts; the first RS_NW ("matrix waves") fetch
// the inputs, the others stream G; all of them run Hessian tiles; RS_JC_MAX compose ops
// per thread
constexpr int RS_NW = 4, RS_NT = 512, RS_WAVES = RS_NT / 64, RS_JC_MAX = 12, RS_TRIP_WORDS = 8;
constexpr int RS_BLOCKS_MAX = 255;  // 4-column blocks of the unknowns (one byte each): no <= 1020
// generated source group: sizes; offsets of A [n][n] and B [n][m] inside a ring slot; image
// offsets of the tables TA[k][i][j] = (A^{k+1})[i][j] and TB[d][i][j] = (A^d B)[i][j], k, d < N, and of
// the powers A^(2^s) they are built from
enum { LT_N = 0, LT_M, LT_HORIZON, LT_A, LT_B, LT_TA, LT_TB, LT_TP, RS_LTI_WORDS = 8 };
constexpr int RS_LTI_MAX = 4;
// row record of G: voff[4] (index of column 0 of the axis' row in the persistent kernel's
// workspace layout, see RT_* below), arrow param[4], center param[4], naxes, extreme param, the
// first two axes packed once more: voff0 | voff1 << 16, arrow0 | arrow1 << 16
constexpr int RS_AXMAX = 4, RS_RR_WORDS = 16;
constexpr int RS_GDESC_PIECES = 6, RS_GDESC_THREADS = RS_NT - RS_NW * 64;  // pieces per stream-wave thread
constexpr int RS_GFIX_NONE = 7;   // (H_OFF_RS_GFIX: no round of this thread has a two-axis piece)
constexpr int32_t RS_DST_ACC = 1 << 30;
// The Hessian and the gradient are accumulated in 4x4 blocks (v_mfma_f64_4x4x4_4b_f64: four
// independent 4x4 blocks per instruction, 4 rows of the workspace per k-step): block (bi, bj)
// of P is columns 4bi.. of the A rows times columns 4bj.. of the B rows, block bi of q is
// the same A columns times s (d - aim), d = column `no` of the d rows.  Only blocks some term
// reaches structurally exist (plus the diagonal of P and all of q, which the diagonal gterms
// add into); four of them form a *pack*, one accumulator register: lane group j of the
// instruction works on the pack's block j.
// The persistent kernel keeps the workspace in LDS row major, V[r][c] with leading dimension LDV
// (columns: the unknowns, d = Mg.given in column `no`, ones in column no + 1); rows come in groups
// of four: row-sets start on a group boundary and are followed by zero rows up to the next one.
// Trip record (8 words = one s_load_dwordx8): 16 rows (four k-steps; lane row l >> 4 = lk feeds
// k-step u the row base + (0, 8, 4, 12)[lk] + u: the two lane rows a 32-lane LDS read serves lie 8
// rows apart, half the banks with LDV = 2 mod 4) or, *short*, 4 rows (one k-step, lane row lk
// feeds row base + lk) of one gterm into one pack.  RT_A / RT_B: BYTE offset of the first A / B
// row; RT_D: BYTE offset of column `no` of the first d row -- what the lanes of a block of q read
// as their B operand: element 0 of the block row is d, element 1 the ones, so that D[.][0] = sum a d
// and D[.][1] = sum a; RT_W / RT_AIM: BYTE offset of the weight / aim among the parameters;
// RT_WORD: rows (16, 4; 0: a pack nothing adds into -- it is still written) | short << 5 |
// half << 6 (the term is halved) | nop << 7 (no Hessian part: blocks of P get nothing) | first
// trip of its pack << 8 | last << 9 | live lane groups << 10 | lane groups that hold a block of
// q << 14 | last trip of its term in the pack << 18: there the term's sum S enters the pack as
// w S (blocks of P) or (w s) (S[.][0] - aim S[.][1]) (blocks of q).  RT_BI / RT_BJ: block row /
// block column of the four lane groups, a byte each (groups that are not live repeat a live
// one's; the column of a block of q is not used).  A wavefront's trips of one pack are consecutive.
enum { RT_A = 0, RT_B, RT_D, RT_W, RT_AIM, RT_WORD, RT_BI, RT_BJ };
enum { RT_SHORT = 5, RT_HALF = 6, RT_NOP = 7, RT_FIRST = 8, RT_LAST = 9, RT_LIVE = 10, RT_QMASK = 14,
       RT_TERM_END = 18 };
// diagonal gterms the persistent kernel takes on one column of the unknowns
constexpr int RS_DIAG_MAX = 2;
enum { RR_VOFF = 0, RR_ARROW = 4, RR_CENTER = 8, RR_NAXES = 12, RR_EXTREME = 13, RR_PACKED = 14 };

}  // namespace mpcasm
