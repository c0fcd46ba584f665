// resident.hip -- the persistent fused assembly kernel (K2 + K3 + K4, one launch).
//
// Same job as fused.hip, restructured so that a QP instance costs no table traffic,
// no serial chains of dependent on-chip loads and no exposed HBM latency.  The
// workgroups (512 threads = 8 wavefronts) are persistent: a few per CU, each loops
// over instances (b = blockIdx.x, += gridDim.x; large batches in runs of four consecutive
// instances per workgroup, see instance_at).  Everything that is *structure* is
// loaded once per workgroup and then stays on chip for the whole launch:
//   * the compose program (K2): each thread owns a fixed handful of ops
//     `coef * image[src] * image[given]` and keeps them in REGISTERS (JC slots);
//   * the input list: which bytes of the instance's horizon matrices, given vector and
//     parameters land where in the LDS input image (per-lane load addresses);
//   * Hessian (tile, term) item lists, row records of G in LDS.
//
// The wavefronts SPECIALISE.  Waves 0-3 ("matrix waves") are the only ones that read
// HBM: they fetch the next instance's input image with LDS-DMA loads
// (global_load_lds_dwordx4: no registers, no staging pass) into the other half of a
// double buffer.  Waves 4-7 ("stream waves") write the rows of G and h.  All eight run the
// Hessian and gradient on the fp64 matrix core in 4x4 blocks (v_mfma_f64_4x4x4_4b_f64, four
// independent blocks per instruction; plan_tables.h RT_*): the workspace is row major and a lane
// row's four k-steps read rows 8 apart from the next lane row's (eight conflict-free 8-byte LDS
// reads per trip, device_prims.h mfma_trip16), a term's weight is applied once to its block
// sum, the gradient rides along through a column of ones, and the blocks of P go from the
// accumulators straight to HBM (small launches, or when P does not fit) or are collected in
// LDS and written with 16-byte nontemporal stores (launches that stream to HBM); a plan may
// ask for the CSC data arrays of P and G instead of the dense matrices (g_mode 3).
//
// Per instance, three barriers:
//   A | all: diagonal terms, K2 compose the workspace V = [Mo | d = Mg.given | 1] from the
//       image (Mg is never stored; the preview matrices never touch HBM)
//   B | matrix waves: start the next image's loads; everyone: its packs of blocks -> P (HBM),
//       q (LDS); stream waves: G, h -> HBM; the last matrix wave: the next instance's
//       horizon tables when they are generated on chip
//   C | matrix waves: wait for the image, clear the workspace elements that two threads
//       add into; all: q and the zero blocks of P -> HBM
// Barriers order LDS only (lds_barrier), so result stores stay in flight across phases.
// Within a phase all operand loads are issued before the first use (explicit load
// batches), because the compiler will not hoist LDS reads over LDS writes of the same
// array.
//
// Reference semantics: body.py:149-193 (preview rows), :236-264 + restrictions.py:
// 175-199 (constraints), :266-302, :322-329 (costs); identical plan tables and
// numerics contract as assemble.hip / fused.hip.
//
// ONE source, two builds.  Ahead of time (libmpcasm.so) the kernel reads the plan's sizes and
// the trip lists at run time.  Compiled by hiprtc for ONE plan (MPCASM_SPEC, jit.hip) the very
// same body sees them as constants from a generated "plan_spec.h": the plan's sizes fold into
// immediates (no kernel-argument reloads, no scalar-register spills) and every wavefront's
// trip list is unrolled into straight-line code -- no records, no flag tests, no loop.
#ifdef __HIPCC_RTC__
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
typedef unsigned long uintptr_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "device_prims.h"
#include "plan_dev.h"
#ifdef MPCASM_SPEC
#include "plan_spec.h"  // namespace mpcasm::spec: PlanConst, TRIPS, WTRIP (generated per plan)
#endif
#ifndef __HIPCC_RTC__
#include "kernels.h"
#endif

namespace mpcasm {

#ifndef __HIPCC_RTC__
extern int g_phase_mask;  // diagnostic (timing-only ablation), fused.hip
extern int g_resident_per_cu;  // tuning aid, capi.hip
#endif

namespace {

constexpr int NT = RS_NT;         // threads of a workgroup
constexpr int MW = RS_NW;         // wavefronts that run the matrix core
constexpr int WT = NT - MW * 64;  // threads of the worker wavefronts
constexpr int AXMAX = RS_AXMAX;
constexpr int RR_WORDS = RS_RR_WORDS;
// ... of which a row of at most two axes whose fields fit 16 bits (H_RR_PACKED) needs four in LDS:
// voff0 | voff1 << 16, arrow0 | arrow1 << 16, center0 | center1 << 16, extreme (a missing second
// axis points at workspace row 0 with the always-zero parameter: it adds exact zeros)
constexpr int RR_COMPACT = 4;
static_assert(RS_DIAG_MAX == 2, "the per-column diagonal table holds two terms");

__host__ __device__ inline int even_up_i(int x) { return (x + 1) & ~1; }

constexpr int GU = 6;  // 16-byte pieces of G a worker thread may own a descriptor for
// G by 16-byte pieces (columns 2cp, 2cp+1 of a row) from the row's two source rows and
// arrows packed in the RR_PACKED words of its record (rows have at most two axes, workspace
// offsets and parameter indices fit 16 bits): mode 1.  Mode 2: a small problem (every
// stream-wave thread owns at most GU pieces) keeps a ready-made descriptor per piece in
// LDS -- 12 KB that buy back the row/column bookkeeping; 0: the general path.
template <class PlanT>
__host__ __device__ inline int resident_g_mode(const PlanT& p) {
  if (p.csc_gnnz != 0 || p.csc_pnnz != 0) return 3;  // CSC hand-off: entry by entry (plan_tables.h H_CSC_*)
  if (!p.rr_packed) return 0;
  return p.rs_ngdesc != 0 ? 2 : 1;
}
static_assert(GU == RS_GDESC_PIECES && WT == RS_GDESC_THREADS, "descriptor table of G");

// doubles of the workspace: row major (plan_tables.h RT_*), and a few spare doubles behind the
// last row (the lanes of a block of q read two columns past the ones)
template <class PlanT>
__host__ __device__ inline int resident_v_doubles(const PlanT& p) {
  return (p.rs_vrow0 + p.rtot) * p.rs_ldv + 16;
}

struct ResidentLayout {
  // offsets in doubles
  int v, pl, ql, dvec, dcoef, dpar, img, ab, streams, ints, total_doubles;
  int ldp;                                            // leading dimension of P in LDS
  int p_direct;  // 1: P does not fit beside the workspace -- its blocks go straight to HBM
  int i_rr, i_meta, i_wtrip, i_split, i_lti, i_abmeta, i_gdesc, i_gfix, i_rrwin, i_cscp;  // offsets in ints inside the int region
};

template <class PlanT>
__host__ __device__ inline ResidentLayout resident_layout(const PlanT& p) {
  ResidentLayout L;
  L.ldp = even_up_i(p.no);
  int o = 0;
  L.v = o;      o += resident_v_doubles(p);
  L.pl = o;     // (placed last, see below: it is what a wide problem drops)
  L.ql = o;     o += L.ldp;
  // diagonal gterms: addends of P[c][c] and q[c] of this instance, then per column the
  // (weight, aim) parameter slots and coefficients of the (at most RS_DIAG_MAX) terms on it
  L.dvec = o;   o += 2 * L.ldp;
  L.dcoef = o;  o += RS_DIAG_MAX * L.ldp;
  L.dpar = o;   o += RS_DIAG_MAX * L.ldp;
  L.img = o;    o += 2 * p.rs_img;  // two input images: this instance's, the next one's
  L.ab = o;     o += 2 * p.rs_ab;   // ring of two slots: (A, B) of the generated systems
  L.streams = o; o += 2 * (MAX_SOURCES + 3);  // (base pointer, bytes per instance) per stream
  L.ints = o;
  int i = 0;  // the first two start 16-byte aligned
  L.i_rr = i;    i += p.nc * (p.rr_packed ? RR_COMPACT : RR_WORDS);
  L.i_meta = i;  i += p.rs_nchunk * 64 * 2;
  L.i_abmeta = i; i += p.rs_ab * 2 * 2;
  // per-thread piece descriptors of G; CSC hand-off: per-entry descriptors of G, then where the
  // stored entries of P sit in its LDS copy
  L.i_gdesc = i;  i += resident_g_mode(p) == 2 ? GU * WT * 2 : (resident_g_mode(p) == 3 ? 2 * p.csc_gnnz : 0);
  // per stream-wave thread: the second axis of the one piece of its that needs both (H_RS_NGFIX)
  L.i_gfix = i;   i += resident_g_mode(p) == 2 && p.rs_ngfix != 0 ? 2 * WT : 0;
  L.i_rrwin = i;  i += p.rs_compact ? p.nc : 0;  // windows of the rows of G (compact workspace)
  L.i_cscp = i;   i += resident_g_mode(p) == 3 ? p.csc_pnnz : 0;
  L.i_wtrip = i; i += RS_WAVES * 2;
  L.i_split = i; i += p.rs_nsplit;
  L.i_lti = i;   i += p.rs_nlti * RS_LTI_WORDS;
  o += even_up_i(i) / 2;
  // P in LDS (dense, read out with 16-byte stores), or its blocks leave the matrix core for
  // HBM directly (8-byte stores, 32-byte runs): p.rs_p_direct, see resident_choose_p_direct
  L.p_direct = p.rs_p_direct;
  L.pl = o;
  if (!L.p_direct) o += p.no * L.ldp;
  L.total_doubles = o;
  return L;
}

// One LDS-DMA load: every lane's `bytes` (16 or 4) from its own global address to
// lds_base + lane * bytes.  M0 carries the LDS base; it is compiler-reserved, so it is
// saved, set and restored inside the one statement.  The compiler does not count this
// load: the issuing wave waits for it itself (dma_wait).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Row of a 16-row trip that lane row lk feeds to k-step 0 (k-step u: that row + u).  The two
// lane rows one 32-lane LDS read serves (lk = 0, 1 and 2, 3) lie 8 rows apart: with
// ldv = 2 (mod 4) that is 32 banks, so the read touches every bank once.
__device__ __forceinline__ int resident_trip_row(int lk) { return ((lk & 1) << 3) | ((lk & 2) << 1); }

#ifdef MPCASM_SPEC
// ---- K3 of a specialised kernel: a wavefront's trips as straight-line code --------------
struct TripState {
  const char* Vlane;  // workspace + (0, 8, 4, 12)[lk] rows + lx * 8
  const char* prc;    // this instance's parameters
  double *Pl, *ql;  // Pl: P in LDS (leading dimension ldp), or this instance's P in HBM (no)
  int ldpl;
  bool mirror;
  const double* dvec;
  int lx, lg, lk;
  const char *ap, *bp;  // of the current pack
  int bi, bj;
  bool isq, live;
  double acc, sum;
};

template <int I>
__device__ __forceinline__ void spec_trip(TripState& s) {
  using PC = spec::PlanConst;
  constexpr int A = spec::TRIPS[I][RT_A], B = spec::TRIPS[I][RT_B], D = spec::TRIPS[I][RT_D];
  constexpr int W = spec::TRIPS[I][RT_W], AIM = spec::TRIPS[I][RT_AIM], WORD = spec::TRIPS[I][RT_WORD];
  constexpr int BI = spec::TRIPS[I][RT_BI], BJ = spec::TRIPS[I][RT_BJ];
  constexpr int row_bytes = PC::rs_ldv * 8, ldp = (PC::no + 1) & ~1;
  constexpr int qmask = (WORD >> RT_QMASK) & 15, livemask = (WORD >> RT_LIVE) & 15;
  constexpr bool nop = (WORD >> RT_NOP) & 1, half = (WORD >> RT_HALF) & 1;
  if constexpr ((WORD >> RT_FIRST) & 1) {  // a new pack: what this lane reads and owns
    // (from an opaque copy of the lane's group: a handful of instructions per pack and
    // instance -- hoisted out of the instance loop, every pack's constants would sit in
    // registers for the whole launch, and the kernel has none to spare)
    int lg_ = s.lg;
    asm volatile("" : "+v"(lg_));
    s.bi = (BI >> (8 * lg_)) & 255;
    s.bj = (BJ >> (8 * lg_)) & 255;
    s.isq = (qmask >> lg_) & 1;
    s.live = (livemask >> lg_) & 1;
    s.ap = s.Vlane + s.bi * 32;
    s.bp = s.Vlane + (s.isq ? 0 : s.bj * 32);
    s.acc = 0.0;
  }
  if constexpr ((WORD & 31) != 0) {
    const char* bsrc = qmask == 0 ? s.bp + B : (qmask == livemask ? s.bp + D : s.bp + (s.isq ? D : B));
    if constexpr (!((WORD >> RT_SHORT) & 1)) {
      // eight 8-byte reads at immediate offsets (k-step u is row u of the lane's four) and the
      // four products: mfma_trip16.  An offset that does not fit the instruction's 16 bits goes
      // into the address.
      constexpr int span = 3 * row_bytes;
      constexpr int AH = A + span < 65536 ? 0 : A, DH = D + span < 65536 ? 0 : D, BH = B + span < 65536 ? 0 : B;
      const unsigned pa = (unsigned)(uintptr_t)s.ap + AH;
      if constexpr (qmask == 0) {
        s.sum = mfma_trip16<A - AH, B - BH, row_bytes>(pa, (unsigned)(uintptr_t)s.bp + BH, s.sum);
      } else if constexpr (qmask == livemask) {
        s.sum = mfma_trip16<A - AH, D - DH, row_bytes>(pa, (unsigned)(uintptr_t)s.bp + DH, s.sum);
      } else {
        // (mixed pack: the lanes of a block of q read d, the others B -- one offset for both, the
        // difference goes into the address)
        s.sum = mfma_trip16<A - AH, B - BH, row_bytes>(pa, (unsigned)(uintptr_t)s.bp + BH + (s.isq ? D - B : 0), s.sum);
      }
    } else {
      const int short_shift = (s.lk - resident_trip_row(s.lk)) * row_bytes;
      const double a = *reinterpret_cast<const double*>(s.ap + (A + short_shift));
      const double b = *reinterpret_cast<const double*>(bsrc + short_shift);
      s.sum = mfma_f64_4x4x4(a, b, s.sum);
    }
    if constexpr ((WORD >> RT_TERM_END) & 1) {
      const double w = *reinterpret_cast<const double*>(s.prc + W);
      if constexpr (qmask != 0 || nop) {
        const double aim = *reinterpret_cast<const double*>(s.prc + AIM);
        const double ws = half ? 0.5 * w : w;
        const double ones = quad_broadcast<1>(s.sum);
        const double m = s.isq ? ws : (nop ? 0.0 : w);
        s.acc = fma(m, fma(-(s.isq ? aim : 0.0), ones, s.sum), s.acc);
      } else {
        s.acc = fma(w, s.sum, s.acc);
      }
      s.sum = 0.0;
    }
  }
  if constexpr ((WORD >> RT_LAST) & 1) {  // the pack is complete: into P and q in LDS
    // (opaque copies again: a pack whose four blocks share a block row has a constant s.bi, and
    // its row offsets and bounds tests would be kept in registers across instances -- with a
    // width that is no multiple of 4 that is what made the kernel spill)
    int lk_ = s.lk, lx_ = s.lx;
    asm volatile("" : "+v"(lk_), "+v"(lx_));
    const int row = 4 * s.bi + lk_, col = 4 * s.bj + lx_;
    if (s.live && row < PC::no) {
      if (s.isq) {
        if (lx_ == 0) s.ql[row] = s.acc + s.dvec[ldp + row];
      } else if (col < PC::no) {
        const double val = s.acc + (row == col ? s.dvec[col] : 0.0);
        // (to LDS or, block by block, to HBM: ordinary stores -- the 32-byte runs of a block
        // must meet in L2 to leave it as whole lines)
        s.Pl[row * s.ldpl + col] = val;
        if (PC::rs_sym && s.bi != s.bj) s.Pl[col * s.ldpl + row] = val;
      }
    }
  }
}

template <int I, int N>
__device__ __forceinline__ void spec_trips(TripState& s) {
  if constexpr (I < N) {
    spec_trip<I>(s);
    // (one trip's operands at a time: hoisting the reads of later trips over this one's
    // products costs more registers than the 128 a wavefront has at four per SIMD)
    __builtin_amdgcn_sched_barrier(0);
    spec_trips<I + 1, N>(s);
  }
}
template <int WAVE>
__device__ __forceinline__ void spec_wave_trips(TripState& s) {
  spec_trips<spec::WTRIP[WAVE][0], spec::WTRIP[WAVE][0] + spec::WTRIP[WAVE][1]>(s);
}
#endif  // MPCASM_SPEC

// GEN: the plan has source groups generated on chip (K1 fused); without them that code is
// not even compiled in, it costs registers.  PlanT: PlanDev (sizes at run time) or
// spec::PlanConst (the same names as constants); plan_itab / plan_dtab: the plan's tables.
template <int JC, bool STAMPS, bool GEN, class PlanT>
__device__ __forceinline__ void resident_body(
    const PlanT& p, const int32_t* __restrict__ plan_itab, const double* __restrict__ plan_dtab,
    const SrcTable& src, const double* __restrict__ params, const double* __restrict__ given,
    double* __restrict__ P, double* __restrict__ q, double* __restrict__ G,
    double* __restrict__ h, int batch, int phases, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: lives in an SGPR
  // diagnostic build only (MPCASM_OPT_PHASE_MASK bit 6): per-wave cycle sums of the phases,
  // written to the otherwise unused workspace; no result depends on them
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  // ... and, behind those of all workgroups, cycles since the kernel's start at the stations
  // of the set-up (0 stream tables in LDS, 1 first barrier, 2 compose program in registers,
  // 3 first tables generated, 4 structure tables copied, 5 barrier, 6 first image landed)
  const unsigned long long t_begin = STAMPS ? __builtin_amdgcn_s_memtime() : 0;
#define SETUP_STAMP(k)                                                                  \
  if (STAMPS && stamps != nullptr && lane == 0)                                         \
    stamps[((size_t)gridDim.x + blockIdx.x) * (NT / 64) * 8 + wave * 8 + (k)] =         \
        __builtin_amdgcn_s_memtime() - t_begin;
  const bool stamping = STAMPS && stamps != nullptr;
#define MPCASM_STAMP(slot)                                         \
  if (STAMPS && stamping) {                                        \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
    t_acc[slot] += t_now - t_prev;                                 \
    t_prev = t_now;                                                \
  }
  const ResidentLayout L = resident_layout(p);
  const int no = p.no, nc = p.nc, ldv = p.rs_ldv, vd = p.rs_vd, ldp = L.ldp;  // (ldv, vd: plan_tables.h H_RS_LDV)

  double* V = lds + L.v;
  double* Pl = lds + L.pl;
  double* ql = lds + L.ql;
  double* dvec = lds + L.dvec;
  double2* dcoef = reinterpret_cast<double2*>(lds + L.dcoef);
  int4* dpar = reinterpret_cast<int4*>(lds + L.dpar);
  double* strm = lds + L.streams;  // [nsrc + 3] pairs: base pointer, bytes per instance (raw)
  int* itb = reinterpret_cast<int*>(lds + L.ints);
  int* rr = itb + L.i_rr;
  int2* meta = reinterpret_cast<int2*>(itb + L.i_meta);
  int* wtrip = itb + L.i_wtrip;
  int* split = itb + L.i_split;
  int* lti = itb + L.i_lti;
  int2* abmeta = reinterpret_cast<int2*>(itb + L.i_abmeta);
  int2* gdesc = reinterpret_cast<int2*>(itb + L.i_gdesc);  // [GU][WT]
  int2* gfix = reinterpret_cast<int2*>(itb + L.i_gfix);    // [WT]
  int* rrwin = itb + L.i_rrwin;                            // [nc]
  // LDS byte address of the image double buffer (the low half of a flat LDS address)
  const unsigned img_lds = (unsigned)(uintptr_t)(lds + L.img);  // (rs_img doubles each, rs_img_dma of them loaded)
  const int unit = p.rs_unit, nchunk = p.rs_nchunk;

  // ---- once per workgroup, first: what the input fetch needs (stream table, per-lane load
  // tables), so that the first instance's image is on its way while the rest is set up
  {
    const int2* t2 = reinterpret_cast<const int2*>(plan_itab + p.off_rs_abmeta);
    for (int i = tid; i < p.rs_ab * 2; i += NT) abmeta[i] = t2[i];
    t2 = reinterpret_cast<const int2*>(plan_itab + p.off_rs_inmeta);
    for (int i = tid; i < nchunk * 64; i += NT) meta[i] = t2[i];
    for (int i = tid; i < p.rs_nlti * RS_LTI_WORDS; i += NT) lti[i] = (plan_itab + p.off_rs_lti)[i];
    // input streams: the sources, then given, params, the plan's constants
    // (wave-uniform index into the kernel's arguments: scalar loads out of the argument segment, which
    // the scalar cache already holds -- indexed by the lane, the table would be fetched by vector
    // loads from wherever the runtime keeps kernel arguments, and the first barrier would wait for it)
    if (wave == 0)
      for (int t = 0; t < p.nsrc + 3; ++t) {
        const int s = t - p.nsrc;
        const double* base = s < 0 ? src.ptr[t]
                                   : (s == 0 ? given : (s == 1 ? params : plan_dtab + p.doff_rs_const));
        const long long stride =
            s < 0 ? src.stride[t] : (s == 0 ? (long long)p.ng : (s == 1 ? (long long)p.nparams : 0));
        if (lane == 0) {
          reinterpret_cast<const double**>(strm)[2 * t] = base;
          reinterpret_cast<long long*>(strm)[2 * t + 1] = stride * (long long)sizeof(double);
        }
      }
  }
  SETUP_STAMP(0)
  lds_barrier();
  SETUP_STAMP(1)
  // The matrix waves fetch instance `inst`'s image into buffer `buf` (chunks dealt round
  // robin).  With registers to spare (FK > 0) a lane keeps the address of its piece of the
  // first FK chunks of its wave -- pointer into instance 0 and bytes per instance -- so that
  // a load costs one multiply-add; further chunks look their stream up in LDS.
#ifdef MPCASM_SPEC
  constexpr int FK = JC <= 5 ? 3 : 0;
#else
  constexpr int FK = 0;  // (the ahead-of-time kernel has no registers to spare: it looks the streams up)
#endif
  // The `given` rows of a launch may be picked by an index (mpcasm_assemble_indexed: a fleet's bucket takes
  // its walkers' rows out of the fleet-wide array, no gather pass): the table travels in the last,
  // unused slot of the source table, marked by a stride of -1; instance b reads row gix[b] of `given`
  // and row b of everything else.
  const int32_t* gix = src.stride[MAX_SOURCES - 1] == -1
                           ? reinterpret_cast<const int32_t*>(src.ptr[MAX_SOURCES - 1]) : nullptr;
  const char* f_base[FK > 0 ? FK : 1];
  int f_stride[FK > 0 ? FK : 1];
  bool f_isg[FK > 0 ? FK : 1];   // the lane's piece of the chunk comes out of `given`
  // A chunk whose every lane reads a stream shared by the whole batch (stride 0: one model
  // for all instances, the constants) holds the same bytes for every instance: once both
  // halves of the double buffer have it, it is not fetched again.
  bool f_shared[FK > 0 ? FK : 1];
  bool f_fast = FK > 0;
  if (FK > 0 && wave < MW) {
#pragma unroll
    for (int j = 0; j < FK; ++j) {
      const int k = wave + j * MW;
      const int2 m = meta[(k < nchunk ? k : 0) * 64 + lane];
      const long long stride = reinterpret_cast<const long long*>(strm)[2 * m.x + 1];
      f_base[j] = reinterpret_cast<const char* const*>(strm)[2 * m.x] + m.y;
      f_stride[j] = (int)stride;
      f_isg[j] = m.x == p.nsrc;
      f_shared[j] = __all(stride == 0);
      f_fast = f_fast && stride >= 0 && stride < (1ll << 31);
    }
  }
  f_fast = __builtin_amdgcn_readfirstlane(__all(f_fast));
  // `settled`: both image buffers have been filled once by this workgroup
  auto fetch_image = [&](long inst, int buf, bool settled) {
    const unsigned dst0 = img_lds + (unsigned)buf * (unsigned)p.rs_img * 8u;
    const long ginst = gix != nullptr ? (long)gix[inst] : inst;   // (wave-uniform: one scalar load)
    int kfirst = wave;
    if (FK > 0 && f_fast) {
#pragma unroll
      for (int j = 0; j < FK; ++j) {
        const int k = wave + j * MW;
        if (k < nchunk && !(settled && f_shared[j])) {
          const char* a = f_base[j] + (unsigned long long)(f_isg[j] ? ginst : inst) * (unsigned)f_stride[j];
          const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)(k * 64 * unit));
          if (unit == 16)
            dma16(a, dst);
          else
            dma4(a, dst);
        }
      }
      kfirst = wave + FK * MW;
    }
    for (int k = kfirst; k < nchunk; k += MW) {
      const int2 m = meta[k * 64 + lane];
      const char* base = reinterpret_cast<const char* const*>(strm)[2 * m.x];
      const long long stride = reinterpret_cast<const long long*>(strm)[2 * m.x + 1];
      if (settled && __all(stride == 0)) continue;
      const char* a = base + (m.x == p.nsrc ? ginst : inst) * stride + m.y;
      const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)(k * 64 * unit));
      if (unit == 16)
        dma16(a, dst);
      else
        dma4(a, dst);
    }
  };
  const int lx = lane & 3, lg = (lane >> 2) & 3, lk = lane >> 4;
  // K1 on chip: the horizon matrices of an LTI system, as the tables the compose ops read
  // (TA[k][i][j] = (A^{k+1})[i][j], TB[d][i][j] = (A^d B)[i][j]; tools.py:14-33), from the
  // A and B that arrived with the image.  One wavefront per system, doubling the number of
  // finished blocks per pass: X_{h+d} = A^h X_d with A^h = (A^{h/2})^2 -- log2(N) dependent
  // passes instead of the N - 1 of the reference's recurrence (the rounding differs by a few
  // ulp).  LDS operations of one wavefront complete in order, so a pass sees the last one.
  // The (A, B) travel apart from the image, two instances ahead, into a ring of two slots
  // (fetch_ab, wave 0), so that the last stream wave can build the tables of instance i+1
  // beside the assembly of instance i, off the critical path.
  const unsigned ab_lds = (unsigned)(uintptr_t)(lds + L.ab);
  auto fetch_ab = [&](long inst, int slot) {  // wave 0 only
    for (int k = 0; k < p.rs_ab / 32; ++k) {
      const int2 m = abmeta[k * 64 + lane];
      const char* base = reinterpret_cast<const char* const*>(strm)[2 * m.x];
      const long long stride = reinterpret_cast<const long long*>(strm)[2 * m.x + 1];
      dma4(base + inst * stride + m.y,
           __builtin_amdgcn_readfirstlane(ab_lds + (unsigned)((slot * p.rs_ab + k * 32) * 8)));
    }
  };
  auto generate_sources = [&](int buf, int slot) {  // one wavefront
    double* im = lds + L.img + buf * p.rs_img;
    const double* ab = lds + L.ab + slot * p.rs_ab;
    for (int g = 0; g < p.rs_nlti; ++g) {
      // (the record comes out of LDS: the same for every lane, and said so -- sizes and offsets the
      // compiler takes for per-lane values turn every loop below into a masked one)
      const int* rec = lti + g * RS_LTI_WORDS;
      const int n = __builtin_amdgcn_readfirstlane(rec[LT_N]), m = __builtin_amdgcn_readfirstlane(rec[LT_M]);
      const int N = __builtin_amdgcn_readfirstlane(rec[LT_HORIZON]), nn = n * n, nm = n * m;
      const double* Am = ab + __builtin_amdgcn_readfirstlane(rec[LT_A]);
      const double* Bm = ab + __builtin_amdgcn_readfirstlane(rec[LT_B]);
      double* TA = im + __builtin_amdgcn_readfirstlane(rec[LT_TA]);
      double* TB = im + __builtin_amdgcn_readfirstlane(rec[LT_TB]);
      double* TP = im + __builtin_amdgcn_readfirstlane(rec[LT_TP]);
      if (n <= 4 && n + m <= 4) {
        // Small systems: the recurrence X_d = A X_{d-1}, X_0 = [B | A] (tools.py:21-30) in
        // registers.  Lane 4 c + i of a group of 16 holds element i of column c, so the n
        // values a lane needs of its column sit in its own quad: DPP quad broadcasts and
        // n FMAs per step, no LDS round trip, nothing on the matrix core that the Hessian
        // tiles are using.  The chain is cut in four: group 0 makes X_0..X_3 the
        // reference's way (three steps; the A part of X_3 is A^4), then the four groups of
        // 16 lanes advance X_r, r = 0..3, by A^4 per step, side by side -- six dependent
        // steps instead of fifteen for N = 16 (a few ulp from the reference's rounding).
        const int grp = lane >> 4, gi = lane & 3, gc = (lane >> 2) & 3;
        const bool live = gi < n && gc < n + m;
        double ar[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) ar[t] = gi < n && t < n ? Am[gi * n + t] : 0.0;  // A[i][t]
        double x = !live ? 0.0 : (gc < m ? Bm[gi * m + gc] : Am[gi * n + (gc - m)]);
        // element (i, c) of block d: TB[d][i][c] or TA[d][i][c - m]
        double* out = gc < m ? TB + gi * m + gc : TA + gi * n + (gc - m);
        const int step = gc < m ? nm : nn;
        auto advance = [&]() {  // x <- (matrix in ar) . x; x[t] of a column is in lane t of the quad
          double y = ar[0] * quad_broadcast<0>(x);
          y = fma(ar[1], quad_broadcast<1>(x), y);
          y = fma(ar[2], quad_broadcast<2>(x), y);
          y = fma(ar[3], quad_broadcast<3>(x), y);
          x = y;
        };
        const int head = N < 4 ? N : 4;
        for (int d = 0; d < head; ++d) {
          if (live && grp == 0) out[d * step] = x;
          advance();
        }
        if (N > 4) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int t = 0; t < 4; ++t) ar[t] = gi < n && t < n ? TA[3 * nn + gi * n + t] : 0.0;  // A^4
          x = live ? out[grp * step] : 0.0;  // X_grp
          for (int d = grp + 4; d - grp < N; d += 4) {
            advance();
            if (live && d < N) out[d * step] = x;
          }
        }
        continue;
      }
      for (int e = lane; e < nn; e += 64) {
        const double v = Am[e];
        TA[e] = v;
        TP[e] = v;
      }
      for (int e = lane; e < nm; e += 64) TB[e] = Bm[e];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      int have = 1;  // TA[0..have), TB[0..have) done; TP[s] = A^have
      for (int s_ = 0; have < N; ++s_) {
        const double* Pw = TP + s_ * nn;
        const int cnt = have < N - have ? have : N - have;
        const int na = cnt * nn, nb = cnt * nm, total = na + nb + nn;
        for (int e = lane; e < total; e += 64) {
          // element (i, j) of Pw . X with X a finished n x w block; the product lands `have`
          // blocks further (or, for X = Pw itself, in the next power)
          const double* X;
          double* out;
          int w, r;
          if (e < na) {
            const int d = e / nn;
            r = e - d * nn;
            X = TA + d * nn, out = TA + (have + d) * nn, w = n;
          } else if (e < na + nb) {
            const int d = (e - na) / nm;
            r = (e - na) - d * nm;
            X = TB + d * nm, out = TB + (have + d) * nm, w = m;
          } else {
            r = e - na - nb;
            X = Pw, out = TP + (s_ + 1) * nn, w = n;
          }
          const int i = r / w, j = r - i * w;
          double acc = 0.0;
          for (int t = 0; t < n; ++t) acc = fma(Pw[i * n + t], X[t * w + j], acc);
          out[r] = acc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        have += cnt;
      }
    }
  };
  // The n-th instance of this workgroup.  Large batches go round the workgroups in runs of four
  // consecutive instances: an instance's q and h are not whole 128-byte lines, and a line that
  // two workgroups (two XCDs, two L2s) each write a part of goes to HBM twice, as partial
  // writes -- four instances' q and h end on a line boundary for every even problem size, and
  // what one workgroup writes within a few microseconds merges in its L2
  // (tools/microbench/store_rate3.hip: +9 % on the C2 output pattern alone).
  const int run_over = (phases >> 10) & 7;  // (bits 10-12, A/B aid: runs of 2^k instances)
  const int run_shift = run_over ? run_over : ((long)batch >= 16L * gridDim.x && !(phases & 256) ? 2 : 0);  // (bit 8: A/B aid)
  // (bit 13, A/B aid: workgroups of one XCD -- blockIdx.x mod 8 -- take neighbouring instances)
  const unsigned vblock = (phases & 8192) && gridDim.x % 8 == 0
                              ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
  auto instance_at = [&](int n) -> long {
    return ((((long)(n >> run_shift) * gridDim.x + vblock)) << run_shift) + (n & ((1 << run_shift) - 1));
  };
  if (wave < MW) {  // the first instance's inputs start their trip now
    fetch_image(instance_at(0), 0, false);
    if (wave == 0 && (GEN && p.rs_nlti != 0)) {
      fetch_ab(instance_at(0), 0);
      if (instance_at(1) < batch) fetch_ab(instance_at(1), 1);
    }
  }
  // ---- once per workgroup: the compose program into registers ------------------
  int c_sg[JC], c_dst[JC];  // c_sg: image offset of the source | of the given value << 16
  double c_coef[JC];
  {
    // one 16-byte record per op (H_OFF_RS_PROG): a quarter of the load instructions of the four tables
    // it is packed from -- the set-up of a launch is bound by how many loads 512 workgroups issue at once
    const int4* prog = reinterpret_cast<const int4*>(plan_itab + p.off_rs_prog);
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      const bool have = j < p.rs_jc;
      const int4 w = have ? prog[j * NT + tid] : int4{0, -1, 0, 0};
      c_sg[j] = w.x;
      c_dst[j] = w.y;
      c_coef[j] = __hiloint2double(w.w, w.z);
    }
  }
  SETUP_STAMP(2)
  if (wave == 0 && (GEN && p.rs_nlti != 0)) {
    // wave 0 builds the first instance's tables as soon as its (A, B) are there, while the
    // other waves copy the structure tables
    dma_wait();
    SETUP_STAMP(2)   // (wave 0: its station 2 is "the first (A, B) have landed")
    generate_sources(0, 0);
  }
  const bool w0_busy = GEN && p.rs_nlti != 0;  // wave 0 takes no part in the table copies
  const int ct = w0_busy ? tid - 64 : tid, CT = w0_busy ? NT - 64 : NT;
  SETUP_STAMP(3)
  // ---- once per workgroup: structure tables into LDS, workspace zeroed -----------
  {
    // All loads from the plan first (every one a trip to L2), then the LDS stores: table
    // by table the latencies would add up.
    constexpr int RRK = 3;  // row-record words per thread in the first batch
    const int32_t* trr = plan_itab + p.off_rs_rr;  // row records of G, precomputed by the plan compiler
    const int nrr = p.rr_packed ? 0 : nc * RR_WORDS;  // (packed plans keep compact records, below)
    int v_rr[RRK], v_wtrip = 0, v_split = 0, v_rrwin = 0;
    int2 v_gd[GU], v_gfix = int2{0, 0};
    int4 v_dpar = int4{0, 0, 0, 0}, v_rra = int4{0, 0, 0, 0}, v_rrb = int4{0, 0, 0, 0};
    double2 v_dcoef = double2{0.0, 0.0};
    const bool own_gd = resident_g_mode(p) == 2 && tid >= MW * 64;
    const bool own_diag = p.ndiag != 0 && ct >= 0 && ct < no;
    const int wt_ = tid - MW * 64;
    if (ct >= 0) {
#pragma unroll
      for (int k = 0; k < RRK; ++k) v_rr[k] = ct + k * CT < nrr ? trr[ct + k * CT] : 0;
      if (ct < RS_WAVES * 2) v_wtrip = (plan_itab + p.off_rs_wtrip)[ct];
      if (ct < p.rs_nsplit) v_split = (plan_itab + p.off_rs_split)[ct];
      // (the compact row record of row ct of G: words 8 .. 15 of its 64-byte record, two 16-byte
      // loads in this batch -- read word by word behind the LDS stores, they cost a second trip)
      if (p.rr_packed && ct < nc) {
        v_rra = reinterpret_cast<const int4*>(trr + ct * RR_WORDS)[2];
        v_rrb = reinterpret_cast<const int4*>(trr + ct * RR_WORDS)[3];
      }
      if (p.rs_compact && ct < nc) v_rrwin = (plan_itab + p.off_rs_rrwin)[ct];
    }
    if (own_gd) {
      // the descriptors of this thread's pieces of G (made by the plan compiler): piece
      // e = wt + u WT, packed, read back as 8 bytes
      const int2* gd = reinterpret_cast<const int2*>(plan_itab + p.off_rs_gdesc);
#pragma unroll
      for (int u = 0; u < GU; ++u) v_gd[u] = gd[u * WT + wt_];
      if (p.rs_ngfix != 0) v_gfix = reinterpret_cast<const int2*>(plan_itab + p.off_rs_gfix)[wt_];
    }
    if (own_diag) {
      // the diagonal gterms on column ct; free slots read the 0.0 behind the parameters
      v_dpar = reinterpret_cast<const int4*>(plan_itab + p.off_rs_dpar)[ct];
      v_dcoef = reinterpret_cast<const double2*>(plan_dtab + p.doff_rs_dcoef)[ct];
    }
    if (ct >= 0) {
      // blocks of P no term reaches stay zero for the whole launch
      if (!L.p_direct)
        for (int i = ct; i < no * ldp; i += CT) Pl[i] = 0.0;
      double2* V2 = reinterpret_cast<double2*>(V);
      const int n2 = resident_v_doubles(p) / 2;
      for (int i = ct; i < n2; i += CT) V2[i] = double2{0.0, 0.0};
      for (int i = ct; i < 2 * ldp; i += CT) dvec[i] = 0.0;  // stays zero without diagonal gterms
#pragma unroll
      for (int k = 0; k < RRK; ++k)
        if (ct + k * CT < nrr) rr[ct + k * CT] = v_rr[k];
      for (int i = ct + RRK * CT; i < nrr; i += CT) rr[i] = trr[i];
      if (p.rs_compact) {
        if (ct < nc) rrwin[ct] = v_rrwin;
        for (int R = ct + CT; R < nc; R += CT) rrwin[R] = (plan_itab + p.off_rs_rrwin)[R];
      }
      if (p.rr_packed) {
        static_assert(RR_CENTER == 8 && RR_EXTREME == 13 && RR_PACKED == 14, "the row record's second half");
        if (ct < nc)
          reinterpret_cast<int4*>(rr)[ct] = int4{v_rrb.z, v_rrb.w, v_rra.x | (v_rra.y << 16), v_rrb.y};
        for (int R = ct + CT; R < nc; R += CT) {
          const int32_t* g = trr + R * RR_WORDS;
          reinterpret_cast<int4*>(rr)[R] =
              int4{g[RR_PACKED], g[RR_PACKED + 1], g[RR_CENTER] | (g[RR_CENTER + 1] << 16), g[RR_EXTREME]};
        }
      }
      if (resident_g_mode(p) == 3) {  // CSC hand-off tables
        const int2* cg = reinterpret_cast<const int2*>(plan_itab + p.off_csc_g);
        for (int i = ct; i < p.csc_gnnz; i += CT) gdesc[i] = cg[i];
        for (int i = ct; i < p.csc_pnnz; i += CT) (itb + L.i_cscp)[i] = (plan_itab + p.off_csc_p)[i];
      }
      if (ct < RS_WAVES * 2) wtrip[ct] = v_wtrip;
      if (ct < p.rs_nsplit) split[ct] = v_split;
      for (int i = ct + CT; i < p.rs_nsplit; i += CT) split[i] = (plan_itab + p.off_rs_split)[i];
    }
    if (own_gd) {
#pragma unroll
      for (int u = 0; u < GU; ++u) gdesc[u * WT + wt_] = v_gd[u];
      if (p.rs_ngfix != 0) gfix[wt_] = v_gfix;
    }
    if (own_diag) {
      dpar[ct] = v_dpar;
      dcoef[ct] = v_dcoef;
    }
  }
  SETUP_STAMP(4)
  lds_barrier();
  SETUP_STAMP(5)
  // column no + 1 of the workspace: ones, for the whole launch (nothing composes into it)
  for (int r = tid; r < p.rtot; r += NT) V[(p.rs_vrow0 + r) * ldv + vd + 1] = 1.0;

  const bool lookahead = (phases & 128) != 0;  // diagnostic: off = fetch only when needed
  if (wave < MW) {
    dma_wait();  // the first image (requested before the set-up above) has landed
    SETUP_STAMP(6)
  }
  SETUP_STAMP(7)

  // ---- K4 bookkeeping of the worker threads: piece e = wt + u WT of G is the 16 bytes
  // (columns 2cp, 2cp+1) of row R; (R, cp) of the first piece and the step between pieces
  const int wt = tid - MW * 64;  // index among the worker threads (negative on MFMA waves)
  const int npair = no >> 1;
  const int gtotal = nc * npair;
  const int g_mode = resident_g_mode(p);
  // P handed over block by block (L.p_direct): the blocks no term reaches are written as zeros
  // after barrier C, 16 bytes per thread and piece (block, row, half of the row).  A thread's
  // first ZK pieces are the same for every instance: where they go is worked out once, here
  // (element offset in P | 1 << 30 one 16-byte store | 1 << 29 two elements; -1 nothing) --
  // read from the table per instance, the load's trip to L2 and back sat on the stream waves'
  // way to barrier A.
#ifdef MPCASM_SPEC
  constexpr int ZK = 2;
#else
  constexpr int ZK = 0;  // (ahead of time: from the table, no registers held)
#endif
  int zoff[ZK > 0 ? ZK : 1];
  {
    const int t0 = tid >= MW * 64 ? tid - MW * 64 : tid + WT;  // (the order of the P / q phase)
    const int32_t* zb = plan_itab + p.off_rs_zblk;
#pragma unroll
    for (int k = 0; k < ZK; ++k) {
      const int e = t0 + k * NT;
      zoff[k] = -1;
      if (L.p_direct && e < p.rs_nzblk * 8) {
        const int z = zb[e >> 3], row = 4 * (z >> 8) + ((e >> 1) & 3), col = 4 * (z & 255) + 2 * (e & 1);
        if (row < no && col < no)
          zoff[k] = (row * no + col) | (col + 1 < no ? ((no & 1) == 0 ? 1 << 30 : 1 << 29) : 0);
      }
    }
  }
  // (row, column pair) of this thread's first piece, and the step from piece to piece
  // Every workgroup writes its instance's G (row-record path) and P (out of LDS) starting at a place of
  // its own, whole lines on: workgroups run in step, and with results of 147 KB and 72 KB per instance
  // (C3) their streams, all at the same offset, would meet on a few memory channels (the write requests
  // stall for DRAM credits three times as often when the placement is unlucky: profiles/r03_placement.txt)
  const bool skew = !(phases & 16384);  // (bit 14, A/B aid: off)
  const int g_rot = skew && gtotal % 8 == 0 && gtotal > 0 ? (int)((blockIdx.x * 53u) % (unsigned)(gtotal / 8)) * 8 : 0;
  const int g_e0 = wt >= 0 && gtotal > 0 ? (wt + g_rot) % gtotal : 0;
  const int g_first = wt >= 0 && npair > 0 ? ((g_e0 / npair) << 16) | (g_e0 % npair) : 0;
  const int g_dR = npair > 0 ? WT / npair : 0, g_dcp = npair > 0 ? WT % npair : 0;


  // G by 16-byte pieces from the packed row records (g_mode 1): piece e = wt + u WT of G is columns
  // 2cp, 2cp+1 of row R.  Per piece: the packed words of the row record -> arrows and workspace
  // rows -> arithmetic -> one 16-byte store; the reads of three pieces are in flight together
  const auto g_packed = [&](double* Gb, const double* V, const double* prm) __attribute__((always_inline)) {
    double2* G2 = reinterpret_cast<double2*>(Gb);
    int first = g_first;  // opaque copy: nothing derived from it is kept across instances
    asm volatile("" : "+v"(first));
    int R = first >> 16, cp = first & 0xFFFF;
    for (int e0 = wt; e0 < gtotal; e0 += 3 * WT) {
      int2 ds[3];
      int c2[3], wn[3], at[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int Rr = e0 + u * WT < gtotal ? R : 0;
        ds[u] = *reinterpret_cast<const int2*>(rr + Rr * RR_COMPACT);
        wn[u] = p.rs_compact ? rrwin[Rr] : 0;
        c2[u] = 2 * cp;
        at[u] = Rr * npair + cp;  // (the piece's place in G: the workgroup starts at g_rot and wraps)
        cp += g_dcp;
        R += g_dR;
        if (cp >= npair) {
          cp -= npair;
          ++R;
        }
        if (R >= nc) R -= nc;
      }
      double a0[3], a1[3];
      double2 v0[3], v1[3];
      if (p.rs_compact) {
        // a row holds its window of the columns: a piece outside it is zero -- read the
        // window's first piece and the always-zero parameter behind the others instead
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const unsigned win = (unsigned)wn[u], cpair = (unsigned)c2[u] >> 1;
          const unsigned d0 = cpair - (win & 255), d1 = cpair - ((win >> 16) & 255);
          const bool in0 = d0 < ((win >> 8) & 255), in1 = d1 < (win >> 24);
          a0[u] = prm[in0 ? ds[u].y & 0xFFFF : p.nparams];
          a1[u] = prm[in1 ? (unsigned)ds[u].y >> 16 : p.nparams];
          v0[u] = *reinterpret_cast<const double2*>(V + (ds[u].x & 0xFFFF) + (in0 ? 2 * d0 : 0));
          v1[u] = *reinterpret_cast<const double2*>(V + ((unsigned)ds[u].x >> 16) + (in1 ? 2 * d1 : 0));
        }
      } else {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          a0[u] = prm[ds[u].y & 0xFFFF];
          a1[u] = prm[(unsigned)ds[u].y >> 16];
          v0[u] = *reinterpret_cast<const double2*>(V + (ds[u].x & 0xFFFF) + c2[u]);
          v1[u] = *reinterpret_cast<const double2*>(V + ((unsigned)ds[u].x >> 16) + c2[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int e = e0 + u * WT;
        double2 r;
        r.x = fma(a1[u], v1[u].x, a0[u] * v0[u].x);
        r.y = fma(a1[u], v1[u].y, a0[u] * v0[u].y);
        if (e < gtotal) store_result(&G2[at[u]], r);
      }
    }
  };

  if (STAMPS && stamping) t_prev = __builtin_amdgcn_s_memtime();
  int buf = 0, iter = 0;  // iter: instances this workgroup has started
  for (long inst = instance_at(0); inst < batch; inst = instance_at(iter + 1), buf ^= 1, ++iter) {
    const double* img = lds + L.img + buf * p.rs_img;
    const double* prm = img + p.rs_img_params;
    lds_barrier();  // A: this instance's image landed, P and q of the previous one read out
    MPCASM_STAMP(0)

    auto compose = [&]() {
      // diagonal gterms (costs on free variables themselves): their addends of P[c][c], q[c]
      if (p.ndiag != 0) {
        int c = tid;
        asm volatile("" : "+v"(c));  // opaque: nothing derived from it is hoisted out of the loop
        if (c < no) {  // body.py:292-300 for rows coef e_c: P[c][c] += (w coef) coef, q[c] += ...
          const int4 par = dpar[c];
          const double2 co = dcoef[c];
          const double w0 = prm[par.x], a0 = prm[par.y], w1 = prm[par.z], a1 = prm[par.w];
          dvec[c] = (w0 * co.x) * co.x + (w1 * co.y) * co.y;
          dvec[ldp + c] = w0 * (co.x * (0.0 - a0)) + w1 * (co.y * (0.0 - a1));
        }
      }
      // ---- K2: compose the workspace from the register-resident program -------------
      if (phases & 1) {
        double va[JC], vg[JC];
  #pragma unroll
        for (int j = 0; j < JC; ++j) {  // all operand loads first
          va[j] = img[c_sg[j] & 0xFFFF];
          vg[j] = img[(unsigned)c_sg[j] >> 16];
        }
        double acc = 0.0;
  #pragma unroll
        for (int j = 0; j < JC; ++j) {
          acc += c_coef[j] * va[j] * vg[j];
          if (c_dst[j] >= 0) {
            if (c_dst[j] & RS_DST_ACC)  // two threads share the element (cleared after C)
              __hip_atomic_fetch_add(&V[c_dst[j] & ~RS_DST_ACC], acc, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
            else
              V[c_dst[j]] = acc;
          }
          acc = c_dst[j] >= 0 ? 0.0 : acc;
        }
      }
    };
    const long nxt = instance_at(iter + 1), nxt2 = instance_at(iter + 2);
    if (wave < MW) {
      compose();
      lds_barrier();  // B: workspace complete
      MPCASM_STAMP(1)
      // the trips are what the barrier at the end of the phase waits for: they get the issue
      // slots before the other workgroup's G and compose (measured: 49.4 -> 48.4 us on C2) -- and
      // so does the request for the next image, which otherwise queues behind the stream waves'
      // first stores of G for some 600 cycles
      __builtin_amdgcn_s_setprio(2);
      // the next instance's image starts its trip from HBM now
      if (lookahead && nxt < batch && (phases & 16)) fetch_image(nxt, buf ^ 1, iter >= 1);
      // ... and, two instances ahead, the (A, B) of the systems whose matrices are built here
      if (wave == 0 && (GEN && p.rs_nlti != 0) && nxt2 < batch) fetch_ab(nxt2, buf);
      MPCASM_STAMP(7)
      // the last matrix wave builds the next instance's horizon tables (its (A, B) landed and
      // were waited for by wave 0 a whole instance ago) beside the stream waves' G; the plan
      // compiler gives it that many trips fewer
      if (wave == MW - 1 && (GEN && p.rs_nlti != 0) && nxt < batch) generate_sources(buf ^ 1, buf ^ 1);
    } else {
      compose();
      lds_barrier();  // B: workspace complete
      MPCASM_STAMP(1)
      if (G != nullptr && (phases & 8)) {
        // ---- K4: constraint rows straight to HBM ---------------------------------------
        double* Gb = G + (size_t)inst * nc * no;
        if (g_mode == 3) {
          // CSC hand-off: entry k of this instance's G data = arrow . workspace at (R, c) of the
          // k-th stored entry; consecutive lanes write consecutive entries (8 bytes each), the
          // reads of three entries are in flight together
          double* Gd = G + (size_t)inst * p.csc_gnnz;
          int w_ = wt;
          asm volatile("" : "+v"(w_));
          for (int e0 = w_; e0 < p.csc_gnnz; e0 += 3 * WT) {
            int2 ds[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) ds[u] = gdesc[min(e0 + u * WT, p.csc_gnnz - 1)];
            double a0[3], a1[3], v0[3], v1[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              a0[u] = prm[ds[u].y & 0xFFFF];
              v0[u] = V[ds[u].x & 0xFFFF];
              if (!p.csc_gsingle) {
                a1[u] = prm[(unsigned)ds[u].y >> 16];
                v1[u] = V[(unsigned)ds[u].x >> 16];
              }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
              if (e0 + u * WT < p.csc_gnnz)
                Gd[e0 + u * WT] = p.csc_gsingle ? a0[u] * v0[u] : fma(a1[u], v1[u], a0[u] * v0[u]);
          }
        } else if (g_mode == 2) {
          // per piece: its packed descriptor -> arrows and workspace rows -> arithmetic ->
          // one 16-byte store; the reads of three pieces are in flight together
          double2* G2 = reinterpret_cast<double2*>(Gb);
          int wt_ = wt;  // opaque copy: nothing derived from it is kept across instances
          asm volatile("" : "+v"(wt_));
          // h of row wt_ rides with the first batch of pieces: its chain of dependent LDS
          // reads (record -> parameters, d -> arithmetic) costs a wave as much as a whole
          // batch when it runs on its own after G
          // (compiled for the plan this is a constant: the branches below fold away)
          const bool all_single = p.rs_gsingle == (1 << (GU * (WT / 64))) - 1;
          // pieces whose reads are in flight together: three, or -- compiled for a plan whose
          // pieces all have one axis (half the operands) -- all six: two LDS round trips (the
          // descriptors, then arrows and workspace rows) per instance instead of four
#ifdef MPCASM_SPEC
          constexpr int GB = spec::PlanConst::rs_gsingle == (1 << (GU * (WT / 64))) - 1 ? GU : 3;
#else
          constexpr int GB = 3;
#endif
          static_assert(GU % GB == 0, "batches of pieces");
          const bool h_mine = wt_ < nc;
          // one piece of this thread may need its second axis in a round of one-axis pieces
          // (H_RS_NGFIX): requested with the first batch, added in its round for this lane alone
          int fix_u = RS_GFIX_NONE;
          double fa1 = 0.0;
          double2 fv1{0.0, 0.0};
          int2 fd{0, 0};
          if (p.rs_ngfix != 0) fd = gfix[wt_];
          const int4 hc4 = reinterpret_cast<const int4*>(rr)[h_mine ? wt_ : 0];  // compact record
          const int4 hr = int4{0, hc4.w, hc4.x, hc4.y};  // -, extreme, packed rows, packed arrows
          const int2 hc = int2{hc4.z & 0xFFFF, (int)((unsigned)hc4.z >> 16)};
          double ha0 = 0.0, ha1 = 0.0, hc0 = 0.0, hc1 = 0.0, hd0 = 0.0, hd1 = 0.0, hext = 0.0;
#pragma unroll
          for (int u0 = 0; u0 < GU; u0 += GB) {
            int2 ds[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) ds[u] = gdesc[(u0 + u) * WT + wt_];
            if (u0 == 0 && p.rs_ngfix != 0) {
              fix_u = (unsigned)fd.x >> 16;
              fa1 = prm[fd.y];
              fv1 = *reinterpret_cast<const double2*>(V + (fd.x & 0xFFFF));
            }
            if (u0 == 0) {
              ha0 = prm[hr.w & 0xFFFF];
              ha1 = prm[(unsigned)hr.w >> 16];
              hc0 = prm[hc.x];
              hc1 = prm[hc.y];
              hd0 = V[(hr.z & 0xFFFF) + vd];
              hd1 = V[((unsigned)hr.z >> 16) + vd];
              hext = prm[hr.y];
            }
            // rounds whose pieces all have one axis that can be non-zero (p.rs_gsingle: known to
            // the plan compiler, wave-uniform) read one workspace row and one arrow
            double a0[GB], a1[GB];
            double2 v0[GB], v1[GB];
            bool one[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) {
              one[u] = all_single || ((p.rs_gsingle >> ((u0 + u) * (WT / 64) + (wave - MW))) & 1);
              a0[u] = prm[ds[u].y & 0xFFFF];
              // (columns 2cp, 2cp+1 of a row: one 16-byte read; a wavefront's pieces run along
              // the rows, 16 lanes = 256 contiguous bytes where a row is that long)
              v0[u] = *reinterpret_cast<const double2*>(V + (ds[u].x & 0xFFFF));
              if (!one[u]) {
                a1[u] = prm[(unsigned)ds[u].y >> 16];
                v1[u] = *reinterpret_cast<const double2*>(V + ((unsigned)ds[u].x >> 16));
              }
            }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
              const int e = wt_ + (u0 + u) * WT;
              double2 r;
              if (one[u]) {
                r.x = a0[u] * v0[u].x;
                r.y = a0[u] * v0[u].y;
                if (p.rs_ngfix != 0 && u0 + u == fix_u) {
                  r.x = fma(fa1, fv1.x, r.x);
                  r.y = fma(fa1, fv1.y, r.y);
                }
              } else {
                r.x = fma(a1[u], v1[u].x, a0[u] * v0[u].x);
                r.y = fma(a1[u], v1[u].y, a0[u] * v0[u].y);
              }
              if (e < gtotal && !((phases & 512) && r.x == r.x))
                store_result(&G2[e], r);  // (bit 9, A/B aid: G computed, not written)
            }
            if (u0 == 0 && h_mine) {  // h = (extreme + arrow . center) - arrow . d   (body.py:264)
              double ac = ha0 * hc0, ad = ha0 * hd0;
              ac += ha1 * hc1;
              ad = fma(ha1, hd1, ad);
              (h + (size_t)inst * nc)[wt_] = (hext + ac) - ad;
            }
          }
        } else if (g_mode == 1) {
          g_packed(Gb, V, prm);
        } else if ((no & 1) == 0) {
          // (opaque copies of the thread index keep the general paths' per-thread constants
          // from being computed once and then held in registers for the whole launch)
          int w_ = wt;
          asm volatile("" : "+v"(w_));
          const int dR = WT / npair, dcp = WT - dR * npair;
          int e = w_, R = w_ / npair, cp = w_ - (w_ / npair) * npair;
          double2* G2 = reinterpret_cast<double2*>(Gb);
          while (e < gtotal) {
            const int* rec = rr + R * RR_WORDS;
            const int naxes = rec[RR_NAXES];
            double2 accv{0.0, 0.0};
            for (int ax = 0; ax < naxes; ++ax) {
              const double a = prm[rec[RR_ARROW + ax]];
              const double2 v = *reinterpret_cast<const double2*>(V + rec[RR_VOFF + ax] + 2 * cp);
              accv.x = fma(a, v.x, accv.x);
              accv.y = fma(a, v.y, accv.y);
            }
            store_result(&G2[e], accv);
            e += WT;
            cp += dcp;
            R += dR;
            if (cp >= npair) {
              cp -= npair;
              ++R;
            }
          }
        } else {
          int w_ = wt;
          asm volatile("" : "+v"(w_));
          const int total = nc * no;
          const int dR = WT / no, dc = WT - dR * no;
          int e = w_, R = w_ / no, c = w_ - (w_ / no) * no;
          while (e < total) {
            const int* rec = rr + R * RR_WORDS;
            const int naxes = rec[RR_NAXES];
            double accv = 0.0;
            for (int ax = 0; ax < naxes; ++ax)
              accv = fma(prm[rec[RR_ARROW + ax]], V[rec[RR_VOFF + ax] + c], accv);
            store_result(&Gb[e], accv);
            e += WT;
            c += dc;
            R += dR;
            if (c >= no) {
              c -= no;
              ++R;
            }
          }
        }
        double* hb = h + (size_t)inst * nc;
        int Rh = g_mode == 2 ? wt + WT : wt;  // (the descriptor path has done row wt already)
        asm volatile("" : "+v"(Rh));
        for (int R = Rh; R < nc; R += WT) {
          if (p.rr_packed) {
            const int4 c = reinterpret_cast<const int4*>(rr)[R];
            const double a0 = prm[c.y & 0xFFFF], a1 = prm[(unsigned)c.y >> 16];
            double ac = a0 * prm[c.z & 0xFFFF], ad = a0 * V[(c.x & 0xFFFF) + vd];
            ac += a1 * prm[(unsigned)c.z >> 16];
            ad = fma(a1, V[((unsigned)c.x >> 16) + vd], ad);
            hb[R] = (prm[c.w] + ac) - ad;
            continue;
          }
          const int* rec = rr + R * RR_WORDS;
          const int naxes = rec[RR_NAXES];
          double ac = 0.0, ad = 0.0;
          for (int ax = 0; ax < naxes; ++ax) {
            const double a = prm[rec[RR_ARROW + ax]];
            ac += a * prm[rec[RR_CENTER + ax]];
            ad = fma(a, V[rec[RR_VOFF + ax] + vd], ad);
          }
          hb[R] = (prm[rec[RR_EXTREME]] + ac) - ad;
        }
      }
      MPCASM_STAMP(3)
      __builtin_amdgcn_s_setprio(2);  // (as the matrix waves: trips first)
    }
#ifdef MPCASM_SPEC
    if (P != nullptr && (phases & 2)) {
      // ---- K3, specialised: the wavefront's trips unrolled (spec_trip)
      static_assert(RS_WAVES == 8, "one case per wavefront");
      TripState st;
      st.Vlane = reinterpret_cast<const char*>(V) + resident_trip_row(lk) * (ldv * 8) + lx * 8;
      st.prc = reinterpret_cast<const char*>(prm);
      st.Pl = L.p_direct ? P + (size_t)inst * no * no : Pl;
      st.ldpl = L.p_direct ? no : ldp;
      st.ql = ql;
      st.dvec = dvec;
      st.lx = lx, st.lg = lg, st.lk = lk;
      st.ap = st.bp = st.Vlane;
      st.bi = st.bj = 0;
      st.isq = st.live = false;
      st.acc = st.sum = 0.0;
      switch (wave) {
        case 0: spec_wave_trips<0>(st); break;
        case 1: spec_wave_trips<1>(st); break;
        case 2: spec_wave_trips<2>(st); break;
        case 3: spec_wave_trips<3>(st); break;
        case 4: spec_wave_trips<4>(st); break;
        case 5: spec_wave_trips<5>(st); break;
        case 6: spec_wave_trips<6>(st); break;
        default: spec_wave_trips<7>(st); break;
      }
    }
#else
    if (P != nullptr && (phases & 2)) {
      // ---- K3: this wavefront's packs of Hessian and gradient blocks on the matrix core
      // -> P, q in LDS.  v_mfma_f64_4x4x4_4b_f64: lane l feeds element x = l & 3 of block
      // g = (l >> 2) & 3 in k-step row l >> 4 and receives D[l >> 4][l & 3] of block g.
      // A trip is one 32-byte record (wave-uniform, the same for every instance: read from
      // the plan through the scalar cache, the next one on its way), eight 8-byte operand reads
      // (mfma_trip16_at: a lane's four rows are LDV apart, the two lane rows one read serves 8
      // rows apart -- every bank once) and four MFMAs into the TERM's sum; nothing is scaled on
      // the way.  The weight comes in once per term and pack: acc += w S, the lanes of a block of
      // q (B operand: d in element 0, ones in element 1) acc += (w s) (S[.][0] - aim S[.][1]).
      const int t0 = __builtin_amdgcn_readfirstlane(wtrip[2 * wave]);
      const int tn = __builtin_amdgcn_readfirstlane(wtrip[2 * wave + 1]);
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      typedef const __attribute__((address_space(4))) i32x8* const_i32x8_ptr;  // -> s_load
      const_i32x8_ptr tg = (const_i32x8_ptr)(uintptr_t)(plan_itab + p.off_rs_trip) + t0;
      if (tn > 0) {
        const char* prc = reinterpret_cast<const char*>(prm);
        const int row_bytes = ldv * (int)sizeof(double);
        // full trip: lane row lk owns rows resident_trip_row(lk) + u of the trip; short trip:
        // row lk -- `short_shift` moves a lane's address from the one to the other
        const int short_shift = (lk - resident_trip_row(lk)) * row_bytes;
        const char* Vlane = reinterpret_cast<const char*>(V) + resident_trip_row(lk) * row_bytes + lx * 8;
        const char *ap = Vlane, *bp = Vlane;  // of the current pack
        int bi = 0, bj = 0;
        bool isq = false, live = false;
        i32x8 rn = tg[0];
        double acc = 0.0, sum = 0.0;
        for (int t = 0; t < tn; ++t) {
          const i32x8 r = rn;
          rn = tg[t + 1];
          const int word = r[RT_WORD];
          if ((word >> RT_FIRST) & 1) {  // a new pack: what this lane reads and owns
            bi = (r[RT_BI] >> (8 * lg)) & 255;
            bj = (r[RT_BJ] >> (8 * lg)) & 255;
            isq = (word >> (RT_QMASK + lg)) & 1;
            live = (word >> (RT_LIVE + lg)) & 1;
            ap = Vlane + bi * 32;
            bp = Vlane + (isq ? 0 : bj * 32);  // (the d offset of a record points at column `no`)
            acc = 0.0;
          }
          if (word & 31) {
            const int boff = isq ? r[RT_D] : r[RT_B];
            if (!((word >> RT_SHORT) & 1)) {
              sum = mfma_trip16_at((unsigned)(uintptr_t)(ap + r[RT_A]), (unsigned)(uintptr_t)(bp + boff),
                                   (unsigned)row_bytes, sum);
            } else {
              const double a = *reinterpret_cast<const double*>(ap + (r[RT_A] + short_shift));
              const double b = *reinterpret_cast<const double*>(bp + (boff + short_shift));
              sum = mfma_f64_4x4x4(a, b, sum);
            }
            if ((word >> RT_TERM_END) & 1) {
              // a weight of 0 contributes exact zeros (body.py:292)
              const double w = *reinterpret_cast<const double*>(prc + r[RT_W]);
              if (word & ((15 << RT_QMASK) | (1 << RT_NOP))) {
                const double aim = *reinterpret_cast<const double*>(prc + r[RT_AIM]);
                const double ws = (word >> RT_HALF) & 1 ? 0.5 * w : w;
                const double ones = quad_broadcast<1>(sum);  // (blocks of q: sum of the A rows)
                const double m = isq ? ws : ((word >> RT_NOP) & 1 ? 0.0 : w);
                acc = fma(m, fma(-(isq ? aim : 0.0), ones, sum), acc);
              } else {
                acc = fma(w, sum, acc);
              }
              sum = 0.0;
            }
          }
          if ((word >> RT_LAST) & 1) {  // the pack is complete: into P and q in LDS
            const int row = 4 * bi + lk, col = 4 * bj + lx;
            if (live && row < no) {
              if (isq) {
                if (lx == 0) ql[row] = acc + dvec[ldp + row];
              } else if (col < no) {
                // the diagonal gterms' addend where row == col
                const double val = acc + (row == col ? dvec[col] : 0.0);
                double* Pout = L.p_direct ? P + (size_t)inst * no * no : Pl;
                const int ldo = L.p_direct ? no : ldp;
                Pout[row * ldo + col] = val;
                if (p.rs_sym && bi != bj) Pout[col * ldo + row] = val;
              }
            }
          }
        }
      }
    }
#endif  // MPCASM_SPEC
    MPCASM_STAMP(2)
    __builtin_amdgcn_s_setprio(0);
    lds_barrier();  // C: P and q are in LDS, the workspace is dead
    MPCASM_STAMP(4)

    if (wave < MW) {
      if (!lookahead && nxt < batch && (phases & 16)) fetch_image(nxt, buf ^ 1, iter >= 1);
      // The next image is complete before barrier A.  The wait comes before this wave's
      // P stores so that it never waits for a store, only for loads issued a phase ago.
      dma_wait();
      // elements that two threads add into start the next compose from zero
      for (int i = tid; i < p.rs_nsplit; i += MW * 64) V[split[i]] = 0.0;
    }
    MPCASM_STAMP(5)
    if (P != nullptr && (phases & 32)) {
      double* Pb = P + (size_t)inst * no * no;
      // the thread index runs from the stream waves round to the matrix waves, so that what
      // does not divide evenly (and q) falls to the waves that are not busy with the next
      // image: thread t of this phase is thread (t + MW * 64) % NT of the workgroup
      int t_ = tid >= MW * 64 ? tid - MW * 64 : tid + WT;
      asm volatile("" : "+v"(t_));
      if (L.p_direct) {
        // P went out block by block; what no term reaches is written here: zeros (zoff above;
        // beyond ZK pieces per thread, from the table)
#pragma unroll
        for (int k = 0; k < ZK; ++k)
          if (zoff[k] >= 0) {
            double* z = Pb + (zoff[k] & 0xFFFFFF);
            if (zoff[k] & (1 << 30))
              *reinterpret_cast<double2*>(z) = double2{0.0, 0.0};
            else {
              z[0] = 0.0;
              if (zoff[k] & (1 << 29)) z[1] = 0.0;
            }
          }
        const int32_t* zb = plan_itab + p.off_rs_zblk;
        for (int e = t_ + ZK * NT; e < p.rs_nzblk * 8; e += NT) {
          const int z = zb[e >> 3], row = 4 * (z >> 8) + ((e >> 1) & 3), col = 4 * (z & 255) + 2 * (e & 1);
          if (row < no && col < no) {
            if (col + 1 < no && ((no & 1) == 0))
              *reinterpret_cast<double2*>(Pb + (size_t)row * no + col) = double2{0.0, 0.0};
            else {
              Pb[(size_t)row * no + col] = 0.0;
              if (col + 1 < no) Pb[(size_t)row * no + col + 1] = 0.0;
            }
          }
        }
      } else if (g_mode == 3) {
        // CSC hand-off: the stored entries of P, in the pattern's order, out of its LDS copy
        double* Pd = P + (size_t)inst * p.csc_pnnz;
        const int* cp = itb + L.i_cscp;
        for (int e = t_; e < p.csc_pnnz; e += NT) Pd[e] = Pl[cp[e]];
      } else if ((no & 1) == 0) {
        // ldp == no here, so P in LDS is dense: a flat 16-byte copy
        const int total = no * npair;
        double2* P2 = reinterpret_cast<double2*>(Pb);
        const double2* Pl2 = reinterpret_cast<const double2*>(Pl);
        const int rot = skew && total % 8 == 0 ? (int)((blockIdx.x * 37u) % (unsigned)(total / 8)) * 8 : 0;
        for (int e = t_; e < total; e += NT) {
          const int at = e + rot < total ? e + rot : e + rot - total;  // (this workgroup's own starting line)
          store_result(&P2[at], Pl2[at]);
        }
      } else {
        for (int e = t_; e < no * no; e += NT) {
          const int row = e / no;
          store_result(&Pb[e], Pl[row * ldp + (e - row * no)]);
        }
      }
      double* qb = q + (size_t)inst * no;
      for (int c = t_; c < no; c += NT) qb[c] = ql[c];  // (q, h: part lines, they meet their neighbours in L2)
    }
    MPCASM_STAMP(6)
  }
  if (STAMPS && stamping && lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) stamps[((size_t)blockIdx.x * (NT / 64) + wave) * 8 + i] = t_acc[i];
  }
#undef MPCASM_STAMP
#undef SETUP_STAMP
}

#ifdef MPCASM_SPEC
}  // namespace
// the kernel of ONE plan (hiprtc): everything about the plan is a constant
#ifndef MPCASM_SPEC_WAVES_PER_EU
#define MPCASM_SPEC_WAVES_PER_EU 4
#endif
extern "C" __global__ __launch_bounds__(RS_NT, MPCASM_SPEC_WAVES_PER_EU) void resident_spec_kernel(
    const int32_t* __restrict__ plan_itab, const double* __restrict__ plan_dtab, SrcTable src,
    const double* __restrict__ params, const double* __restrict__ given, double* __restrict__ P,
    double* __restrict__ q, double* __restrict__ G, double* __restrict__ h, int batch, int phases,
    unsigned long long* __restrict__ stamps) {
  constexpr spec::PlanConst p{};
#ifdef MPCASM_SPEC_STAMPS  // diagnostic build (MPCASM_OPT_PHASE_MASK bit 6): per-wave cycle sums
  constexpr bool stamped = true;
#else
  constexpr bool stamped = false;
#endif
  // (MPCASM_SPEC_PHASES: the kernel as it ships runs every phase -- with the mask a constant its
  // tests and the code of the profiling variants fold away; a build per mask value otherwise)
#ifdef MPCASM_SPEC_PHASES
  const int mask = MPCASM_SPEC_PHASES;
#else
  const int mask = phases;
#endif
  resident_body<spec::JC, stamped, spec::PlanConst::rs_nlti != 0>(p, plan_itab, plan_dtab, src, params,
                                                               given, P, q, G, h, batch, mask, stamps);
}
namespace {
#else
template <int JC, bool STAMPS, bool GEN>
__global__ __launch_bounds__(NT, 4) void resident_assemble_kernel(
    PlanDev p, SrcTable src, const double* __restrict__ params, const double* __restrict__ given,
    double* __restrict__ P, double* __restrict__ q, double* __restrict__ G,
    double* __restrict__ h, int batch, int phases, unsigned long long* __restrict__ stamps) {
  resident_body<JC, STAMPS, GEN>(p, p.itab, p.dtab, src, params, given, P, q, G, h, batch, phases,
                                 stamps);
}
#endif

#ifndef __HIPCC_RTC__

template <int JC>
int launch_jc(const PlanDev& p, const SrcTable& src, const double* params, const double* given,
              double* P, double* q, double* G, double* h, void* work, int batch, size_t lds_bytes,
              int num_cus, hipStream_t stream, hipError_t* err) {
  // the stamped instantiation exists for the diagnostic option only
  // (the stamped instantiation exists once, with everything compiled in)
  auto kernel = (g_phase_mask & 64) ? resident_assemble_kernel<JC, true, true>
                : p.rs_nlti != 0    ? resident_assemble_kernel<JC, false, true>
                                    : resident_assemble_kernel<JC, false, false>;
  // residency of this instantiation at this LDS size: queried once per (instantiation, device,
  // size) and kept -- plans of different LDS sizes that share the instantiation (the 34- and
  // 36-unknown buckets of a walker fleet) alternate without another runtime call, so the launch
  // path can be graph-captured.  The dynamic-LDS attribute of the function is only ever RAISED, to
  // the largest size seen: a replayed graph node of the larger plan must not meet a lowered limit.
  struct Residency {
    int device;
    size_t lds;
    int per_cu;
  };
  constexpr int KEPT = 8;
  static thread_local Residency kept[3][KEPT];
  static thread_local int nkept[3] = {0, 0, 0};
  const int slot = (g_phase_mask & 64) ? 2 : (p.rs_nlti != 0 ? 1 : 0);
  int device = 0;
  (void)hipGetDevice(&device);  // (the attribute and the occupancy belong to one device)
  int found = -1;
  for (int i = 0; i < nkept[slot]; ++i)
    if (kept[slot][i].device == device && kept[slot][i].lds == lds_bytes) found = kept[slot][i].per_cu;
  if (found < 0) {
    if (lds_bytes > 64 * 1024) {
      *err = allow_whole_lds(reinterpret_cast<const void*>(kernel));
      if (*err != hipSuccess) return MPCASM_ERR_HIP;
    }
    int n = 0;
    *err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, NT, lds_bytes);
    if (*err != hipSuccess) return MPCASM_ERR_HIP;
    found = n;
    const int at = nkept[slot] < KEPT ? nkept[slot]++ : (int)(lds_bytes % KEPT);  // (full: replace one)
    kept[slot][at] = Residency{device, lds_bytes, n};
  }
  // persistent grid: exactly the workgroups that are resident at once, never more
  // than there are instances
  int per_cu = found;
  if (per_cu < 1) return MPCASM_ERR_LIMIT;
  if (t_per_cu > 0 && t_per_cu < per_cu) per_cu = t_per_cu;
  long grid = (long)num_cus * per_cu;
  if (t_grid > 0 && t_grid < grid) grid = t_grid;
  if (grid > batch) grid = batch;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(NT), lds_bytes, stream, p, src, params,
                     given, P, q, G, h, batch, g_phase_mask,
                     (g_phase_mask & 64) ? static_cast<unsigned long long*>(work) : nullptr);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}
#endif  // !__HIPCC_RTC__

}  // namespace

#ifndef __HIPCC_RTC__
// Whether the blocks of P leave the matrix core for HBM directly (8-byte stores, 32-byte runs)
// instead of being collected in LDS and copied out in 16-byte pieces: 1 always -- P does not fit
// beside the workspace, or MPCASM_OPT_P_DIRECT = 1; 0 never (option 2); 2 by the size of the
// launch (resident_p_direct_for).
int resident_choose_p_direct(const PlanDev& p, int option) {
  PlanDev q = p;
  q.rs_p_direct = 0;
  const bool fits = (size_t)resident_layout(q).total_doubles * sizeof(double) <= (size_t)RESIDENT_LDS_LIMIT;
  return (!fits || option == 1) ? 1 : (option == 2 ? 0 : 2);
}

// ... for one launch.  What decides (round 3, tools/ab_n24.py, tools/ab_workspace.py: same box, one
// process, interleaved rounds):
// * the size of the launch: the 32-byte runs of the direct stores merge on their way to HBM while
//   a launch writes less than ~0.55 GB (C2 at B = 16384: 0.73 direct against 0.65 through LDS;
//   round 2 switched at 0.2 GB already), beyond that they reach memory as partial lines and cost a
//   third of the write rate and more (C2 at B = 32768: 0.46 against 0.68; C3 at 16384 with the
//   compact workspace: 0.54 direct against 0.82 through LDS although P is a sixth of its output;
//   tools/microbench/store_rate3.hip shows the same on the bare store pattern);
// * unless P in LDS costs a workgroup per CU AND an instance is small: the biped at N = 24 with the
//   dense workspace needs 85 KB with P beside it and 64 KB without -- one workgroup per CU or two
//   -- and is faster with the direct stores at EVERY batch size (B = 8192: 0.76 against 0.46 of
//   8 TB/s; 65536: 0.58 against 0.48): eight wavefronts on 68 KB of output per instance leave the
//   CU idle between phases.  C3's 226 KB per instance (92 trips, 9408 pieces of G) keep one
//   workgroup's waves busy: there the size of the launch decides as everywhere else.
int resident_p_direct_for(const PlanDev& p, int batch) {
  if (p.rs_p_direct != 2) return p.rs_p_direct;
  PlanDev q = p;
  q.rs_p_direct = 0;
  const size_t with_p = (size_t)resident_layout(q).total_doubles * sizeof(double);
  q.rs_p_direct = 1;
  const size_t without = (size_t)resident_layout(q).total_doubles * sizeof(double);
  constexpr size_t HALF_CU = 80 * 1024;  // two workgroups share a CU's 160 KB
  const double per_instance = 8.0 * ((double)p.no * p.no + p.no + (double)p.nc * p.no + p.nc);
  if (with_p > HALF_CU && without <= HALF_CU && per_instance < 128.0 * 1024) return 1;
  return per_instance * batch < 560e6 ? 1 : 0;
}

// 0 when the resident kernel cannot take this plan, else its dynamic LDS bytes
size_t resident_lds_bytes(const PlanDev& p) {
  if (!p.rs_ok || p.rs_jc > RS_JC_MAX || p.no > WT || p.no < 1 || p.max_axes > AXMAX) return 0;
  if ((long)(p.rs_vrow0 + p.rtot) * p.rs_ldv > (1 << 20)) return 0;
  return (size_t)resident_layout(p).total_doubles * sizeof(double);
}

// the 16-byte input loads need every stream 16-byte aligned in every instance
bool resident_inputs_aligned(const PlanDev& p, const SrcTable& src, const double* params,
                             const double* given) {
  if (p.rs_unit != 16) return true;
  auto ok = [](const void* base, long long stride) {
    return (reinterpret_cast<uintptr_t>(base) & 15) == 0 && (stride & 1) == 0;
  };
  for (int s = 0; s < p.nsrc; ++s)
    if (((p.rs_src16 >> s) & 1) && !ok(src.ptr[s], src.stride[s])) return false;
  return ok(given, p.ng) && ok(params, p.nparams);
}

int launch_assemble_resident(const PlanDev& p, const SrcTable& src, const double* params,
                             const double* given, double* P, double* q, double* G, double* h,
                             void* work, int batch, size_t lds_bytes, int num_cus,
                             hipStream_t stream, hipError_t* err) {
#define MPCASM_RS_ARGS p, src, params, given, P, q, G, h, work, batch, lds_bytes, num_cus, stream, err
  if (p.rs_jc <= 3) return launch_jc<3>(MPCASM_RS_ARGS);
  if (p.rs_jc <= 4) return launch_jc<4>(MPCASM_RS_ARGS);
  if (p.rs_jc <= 5) return launch_jc<5>(MPCASM_RS_ARGS);
  if (p.rs_jc <= 8) return launch_jc<8>(MPCASM_RS_ARGS);
  return launch_jc<RS_JC_MAX>(MPCASM_RS_ARGS);
#undef MPCASM_RS_ARGS
}
#endif  // !__HIPCC_RTC__

}  // namespace mpcasm
