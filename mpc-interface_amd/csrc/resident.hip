// resident.hip -- the persistent fused assembly kernel (K2 + K3 + K4, one launch).
//
// Same job as fused.hip, restructured so that a QP instance costs no table traffic
// and no serial chains of dependent on-chip loads: the workgroups are persistent
// (a few per CU, each loops over instances b = blockIdx.x, += gridDim.x) and
// everything that is *structure* is loaded once per workgroup and stays on chip:
//   * the compose program (K2): each of the 256 threads owns a fixed handful of
//     ops `coef * arena[src] (* given[g])` and keeps them in REGISTERS (JC slots);
//   * the constraint program (K4): each thread owns a fixed handful of 16-byte
//     pieces of G and keeps their operand offsets in registers too;
//   * the gradient records and the wavefront -> Hessian tile map in LDS; the
//     Hessian term descriptors come through scalar loads (wave-uniform).
// Per instance the kernel reads only the instance's horizon matrices, given vector
// and parameters (coalesced, staged in LDS) and writes P, q, G, h.  All operand
// addresses of a phase are known before the phase starts, so every phase issues
// its LDS reads back to back instead of chasing descriptors.  P is assembled in
// LDS from the MFMA accumulators (mirroring symmetric tiles) and streamed out with
// 16-byte stores like G.  The workspace V is zeroed once: its structural zeros are
// never written again.  Barriers order LDS only (lds_barrier), so result stores
// stay in flight across phases.
//
// Reference semantics: body.py:149-193 (preview rows), :236-264 + restrictions.py:
// 175-199 (constraints), :266-302, :322-329 (costs); identical plan tables and
// numerics contract as assemble.hip / fused.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace mpcasm {

extern int g_phase_mask;  // diagnostic (timing-only ablation), fused.hip

namespace {

constexpr int NW = RS_NW, NT = RS_NT, TPW = RS_TPW;
constexpr int AXMAX = 4;
constexpr int RR_WORDS = 2 + 3 * AXMAX;  // naxes, extreme param, voff[], arrow param[], center param[]
constexpr int GU = 6;                    // 16-byte pieces of G a thread may own (fast path)

__host__ __device__ inline int even_up_i(int x) { return (x + 1) & ~1; }

struct ResidentLayout {
  int v, xa, g, prm, qpart, ints, total_doubles;  // offsets in doubles
  int ldp;                                       // leading dimension of P in LDS
  int ns;                                        // row slices of the gradient pass
  int i_tile, i_rr, i_gq, i_item, i_islot;       // offsets in ints inside the int region
};

__host__ __device__ inline ResidentLayout resident_layout(const PlanDev& p) {
  ResidentLayout L;
  L.ldp = even_up_i(p.no);
  L.ns = p.no > 0 ? NT / p.no : 1;
  if (L.ns < 1) L.ns = 1;
  if (L.ns > 16) L.ns = 16;
  int o = 0;
  L.v = o;     o += even_up_i(p.rtot * p.ldv) + 16;
  const int xa = even_up_i(p.arena_total), pl = p.no * L.ldp;
  L.xa = o;    o += xa > pl ? xa : pl;          // source arena, later P
  L.g = o;     o += even_up_i(p.ng + 1);       // + one slot that always holds 1.0
  L.prm = o;   o += even_up_i(p.nparams + 1);   // + one slot that always holds 0.0
  L.qpart = o; o += L.ns * L.ldp;
  L.ints = o;
  int i = 0;
  L.i_gq = i;    i += p.rs_nq * 4;              // first two: 16-byte aligned
  L.i_item = i;  i += (p.rs_nitem + 1) * RS_ITEM_WORDS;
  L.i_islot = i; i += NW * TPW * 2;
  L.i_tile = i;  i += NW * TPW;
  L.i_rr = i;    i += p.nc * RR_WORDS;
  o += even_up_i(i) / 2;
  L.total_doubles = o;
  return L;
}

template <int JC>
__global__ __launch_bounds__(NT, 3) void resident_assemble_kernel(
    PlanDev p, SrcTable src, const double* __restrict__ params, const double* __restrict__ given,
    double* __restrict__ P, double* __restrict__ q, double* __restrict__ G,
    double* __restrict__ h, int batch, int phases) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const ResidentLayout L = resident_layout(p);
  const int no = p.no, ng = p.ng, nc = p.nc, ldv = p.ldv, ldp = L.ldp;

  double* V = lds + L.v;
  double* arena = lds + L.xa;
  double* Pl = lds + L.xa;  // aliases the arena: live only between barriers B and D
  double* gl = lds + L.g;
  double* prm = lds + L.prm;
  double* qpart = lds + L.qpart;
  int* itb = reinterpret_cast<int*>(lds + L.ints);
  int4* gq = reinterpret_cast<int4*>(itb + L.i_gq);
  int4* items = reinterpret_cast<int4*>(itb + L.i_item);
  int* islot = itb + L.i_islot;
  int* tile = itb + L.i_tile;
  int* rr = itb + L.i_rr;

  // ---- once per workgroup: the compose program into registers ------------------
  int c_sg[JC], c_dst[JC];  // c_sg: arena offset | given index << 16 (index ng: the constant 1)
  double c_coef[JC];
  {
    const int32_t* tsrc = p.itab + p.off_rs_src;
    const int32_t* tg = p.itab + p.off_rs_gidx;
    const int32_t* tdst = p.itab + p.off_rs_dst;
    const double* tcoef = p.dtab + p.doff_rs_coef;
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      const bool have = j < p.rs_jc;
      const int gi = have ? tg[j * NT + tid] : -1;
      c_sg[j] = (have ? tsrc[j * NT + tid] : 0) | ((gi < 0 ? ng : gi) << 16);
      c_dst[j] = have ? tdst[j * NT + tid] : -1;
      c_coef[j] = have ? tcoef[j * NT + tid] : 0.0;
    }
  }
  // ---- once per workgroup: structure tables into LDS, workspace zeroed -----------
  {
    const int4* t4 = reinterpret_cast<const int4*>(p.itab + p.off_rs_gq);
    for (int i = tid; i < p.rs_nq; i += NT) gq[i] = t4[i];
    t4 = reinterpret_cast<const int4*>(p.itab + p.off_rs_item);
    for (int i = tid; i <= p.rs_nitem; i += NT)
      items[i] = i < p.rs_nitem ? t4[i] : int4{0, 0, 0, 0};
    const int32_t* t = p.itab + p.off_rs_islot;
    for (int i = tid; i < NW * TPW * 2; i += NT) islot[i] = t[i];
    t = p.itab + p.off_rs_tile;
    for (int i = tid; i < NW * TPW; i += NT) tile[i] = t[i];
    const int32_t* rowlimit = p.itab + p.off_rowlimit;
    const int32_t* limits = p.itab + p.off_limit;
    const int32_t* lax = p.itab + p.off_lax;
    for (int R = tid; R < nc; R += NT) {
      const int32_t* lm = limits + rowlimit[R] * LM_WORDS;
      const int r = R - lm[LM_OUT0];
      const int naxes = lm[LM_NAXES];
      const int32_t* lx = lax + lm[LM_LAX0] * LX_WORDS;
      int* rec = rr + R * RR_WORDS;
      rec[0] = naxes;
      rec[1] = lm[LM_EXTREME_P] + (lm[LM_EXTREME_ROWS] == 1 ? 0 : r);
      for (int ax = 0; ax < AXMAX; ++ax) {
        if (ax < naxes) {
          const int vr = lx[ax * LX_WORDS + LX_ROWS] == 1 ? 0 : r;
          rec[2 + ax] = (lx[ax * LX_WORDS + LX_ROWOFF] + vr) * ldv;
          rec[2 + AXMAX + ax] = lm[LM_ARROW_P] + (lm[LM_ARROW_ROWS] == 1 ? 0 : r) * naxes + ax;
          rec[2 + 2 * AXMAX + ax] =
              lm[LM_CENTER_P] + (lm[LM_CENTER_ROWS] == 1 ? 0 : r) * naxes + ax;
        } else {  // missing axis: weight 0 (the extra parameter slot) on row 0
          rec[2 + ax] = 0;
          rec[2 + AXMAX + ax] = p.nparams;
          rec[2 + 2 * AXMAX + ax] = p.nparams;
        }
      }
    }
    double2* V2 = reinterpret_cast<double2*>(V);
    const int n2 = (even_up_i(p.rtot * ldv) + 16) / 2;
    for (int i = tid; i < n2; i += NT) V2[i] = double2{0.0, 0.0};
    if (tid == 0) {
      prm[p.nparams] = 0.0;
      gl[ng] = 1.0;
    }
  }
  lds_barrier();

  // ---- once per workgroup: the constraint program into registers -------------------
  // piece e = tid + u NT of G (16 bytes = columns 2cp, 2cp+1 of row R)
  const bool g_fast = (no & 1) == 0 && p.max_axes <= 2 && p.nparams < 65535 &&
                      (long)nc * (no >> 1) <= (long)GU * NT;
  int g_v0[GU], g_v1[GU], g_a[GU];  // g_a: arrow param of axis 0 | axis 1 << 16
  const int npair = no >> 1;
  const int gtotal = nc * npair;
  if (g_fast) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int e = tid + u * NT;
      g_v0[u] = g_v1[u] = 0;
      g_a[u] = p.nparams | (p.nparams << 16);
      if (e < gtotal) {
        const int R = e / npair, cp = e - R * npair;
        const int* rec = rr + R * RR_WORDS;
        g_v0[u] = rec[2] + 2 * cp;
        g_v1[u] = rec[3] + 2 * cp;
        g_a[u] = rec[2 + AXMAX] | (rec[3 + AXMAX] << 16);
      }
    }
  }
  // gradient pass: thread = (column qc, row slice qs)
  const int qs = no > 0 ? tid / no : 0, qc = tid - qs * (no > 0 ? no : 1);
  const int NS = L.ns;

  const int32_t* arec = p.itab + p.off_arena;
  const int nt = (no + 15) >> 4;
  const int li = lane & 15, lk = lane >> 4;
  bool first = true;

  for (long inst = blockIdx.x; inst < batch; inst += gridDim.x) {
    // ---- stage this instance's inputs (the arena region is free: barrier D) -----
    if (tid == 0) arena[0] = 1.0;
    if ((phases & 16) || first) {
      for (int s = 0; s < p.nsrc; ++s) {
        const long stride = src.stride[s];
        // a shared source survives in the arena only when P does not reuse the region
        if (stride == 0 && !first && P == nullptr) continue;
        const int off = arec[2 * s], size = arec[2 * s + 1];
        const double* sp = src.ptr[s] + inst * stride;
        for (int i = tid; i < size; i += NT) arena[off + i] = sp[i];
      }
      const double* gb = given + inst * ng;
      for (int i = tid; i < ng; i += NT) gl[i] = gb[i];
      const double* pb = params + inst * p.nparams;
      for (int i = tid; i < p.nparams; i += NT) prm[i] = pb[i];
    }
    first = false;
    lds_barrier();  // A: inputs staged

    // ---- K2: compose the workspace from the register-resident program -------------
    if (phases & 1) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < JC; ++j) {
        acc += c_coef[j] * arena[c_sg[j] & 0xFFFF] * gl[(unsigned)c_sg[j] >> 16];
        if (c_dst[j] >= 0) {
          V[c_dst[j]] = acc;
          acc = 0.0;
        }
      }
    }
    lds_barrier();  // B: workspace complete, arena dead

    if (P != nullptr && (phases & 2)) {
      // ---- K3: Hessian tiles on the matrix core -> P in LDS --------------------------
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int t = __builtin_amdgcn_readfirstlane(tile[wave * TPW + s]);
        if (t < 0) continue;
        const int ti = t / nt, tj = t - ti * nt;
        f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
        const int i0 = __builtin_amdgcn_readfirstlane(islot[(wave * TPW + s) * 2]);
        const int cnt = __builtin_amdgcn_readfirstlane(islot[(wave * TPW + s) * 2 + 1]);
        int4 nxt = items[i0];
        for (int it = 0; it < cnt; ++it) {
          const int4 cur = nxt;
          nxt = items[i0 + it + 1];  // prefetch the next pair (the table has a spare record)
          const int nrows = __builtin_amdgcn_readfirstlane(cur.z);
          // a weight of 0 contributes exact zeros through the products (body.py:292)
          const double w = prm[__builtin_amdgcn_readfirstlane(cur.w)];
          const double* ap = V + __builtin_amdgcn_readfirstlane(cur.x) + li;
          const double* bp = V + __builtin_amdgcn_readfirstlane(cur.y) + li;
          for (int k0 = 0; k0 < nrows; k0 += 16) {  // four MFMA k-steps per trip
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // all eight loads in flight together
              const int k = k0 + 4 * u + lk;
              const int kc = k < nrows ? k : nrows - 1;
              a[u] = ap[kc * ldv];
              b[u] = bp[kc * ldv];
              a[u] = k < nrows ? w * a[u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (k0 + 4 * u < nrows) acc = mfma_f64_16x16x4(a[u], b[u], acc);
          }
        }
        const bool mirror = p.rs_sym && ti != tj;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = ti * 16 + lk + 4 * reg, col = tj * 16 + li;
          if (row < no && col < no) {
            Pl[row * ldp + col] = acc[reg];
            if (mirror) Pl[col * ldp + row] = acc[reg];
          }
        }
      }
    }
    if (P != nullptr && (phases & 4)) {
      // ---- gradient: q[c] = sum_records w s V[a][c] (V[d] - aim), rows sliced NS ways --
      double qa = 0.0;
      if (qs < NS) {
        const int nq = p.rs_nq;
        for (int i0 = qs; i0 < nq; i0 += 4 * NS) {  // four records per trip, loads in flight
          int4 e[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NS;
            e[u] = gq[i < nq ? i : nq - 1];
          }
          double w[4], d[4], aim[4], a[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            w[u] = prm[e[u].w & 0x3FFFFFFF];
            d[u] = V[e[u].y];
            aim[u] = prm[e[u].z];
            a[u] = V[e[u].x + qc];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double r = ((e[u].w >> 30) & 1 ? 0.5 : 1.0) * (d[u] - aim[u]);
            const double t = fma(w[u] * a[u], r, qa);
            qa = i0 + u * NS < nq ? t : qa;
          }
        }
        qpart[qs * ldp + qc] = qa;
      }
    }

    if (G != nullptr && (phases & 8)) {
      // ---- K4: constraint rows straight to HBM -------------------------------------------
      double* Gb = G + (size_t)inst * nc * no;
      if (g_fast) {
        double2* G2 = reinterpret_cast<double2*>(Gb);
#pragma unroll
        for (int u = 0; u < GU; ++u) {
          const int e = tid + u * NT;
          const double a0 = prm[g_a[u] & 0xFFFF], a1 = prm[(unsigned)g_a[u] >> 16];
          const double2 v0 = *reinterpret_cast<const double2*>(V + g_v0[u]);
          const double2 v1 = *reinterpret_cast<const double2*>(V + g_v1[u]);
          double2 r;
          r.x = fma(a1, v1.x, a0 * v0.x);
          r.y = fma(a1, v1.y, a0 * v0.y);
          if (e < gtotal) G2[e] = r;
        }
      } else if ((no & 1) == 0) {
        const int dR = NT / npair, dcp = NT - dR * npair;
        int e = tid, R = tid / npair, cp = tid - (tid / npair) * npair;
        double2* G2 = reinterpret_cast<double2*>(Gb);
        while (e < gtotal) {
          const int* rec = rr + R * RR_WORDS;
          const int naxes = rec[0];
          double2 accv{0.0, 0.0};
          for (int ax = 0; ax < naxes; ++ax) {
            const double a = prm[rec[2 + AXMAX + ax]];
            const double2 v = *reinterpret_cast<const double2*>(V + rec[2 + ax] + 2 * cp);
            accv.x = fma(a, v.x, accv.x);
            accv.y = fma(a, v.y, accv.y);
          }
          G2[e] = accv;
          e += NT;
          cp += dcp;
          R += dR;
          if (cp >= npair) {
            cp -= npair;
            ++R;
          }
        }
      } else {
        const int total = nc * no;
        const int dR = NT / no, dc = NT - dR * no;
        int e = tid, R = tid / no, c = tid - (tid / no) * no;
        while (e < total) {
          const int* rec = rr + R * RR_WORDS;
          const int naxes = rec[0];
          double accv = 0.0;
          for (int ax = 0; ax < naxes; ++ax)
            accv = fma(prm[rec[2 + AXMAX + ax]], V[rec[2 + ax] + c], accv);
          Gb[e] = accv;
          e += NT;
          c += dc;
          R += dR;
          if (c >= no) {
            c -= no;
            ++R;
          }
        }
      }
      double* hb = h + (size_t)inst * nc;
      for (int R = tid; R < nc; R += NT) {
        const int* rec = rr + R * RR_WORDS;
        const int naxes = rec[0];
        double ac = 0.0, ad = 0.0;
        for (int ax = 0; ax < naxes; ++ax) {
          const double a = prm[rec[2 + AXMAX + ax]];
          ac += a * prm[rec[2 + 2 * AXMAX + ax]];
          ad = fma(a, V[rec[2 + ax] + no], ad);
        }
        hb[R] = (prm[rec[1]] + ac) - ad;
      }
    }
    lds_barrier();  // C: P and the q partials are in LDS

    if (P != nullptr && (phases & 32)) {
      double* Pb = P + (size_t)inst * no * no;
      if ((no & 1) == 0) {
        // ldp == no here, so P in LDS is dense: a flat 16-byte copy
        const int total = no * npair;
        double2* P2 = reinterpret_cast<double2*>(Pb);
        const double2* Pl2 = reinterpret_cast<const double2*>(Pl);
        for (int e = tid; e < total; e += NT) P2[e] = Pl2[e];
      } else {
        for (int e = tid; e < no * no; e += NT) {
          const int row = e / no;
          Pb[e] = Pl[row * ldp + (e - row * no)];
        }
      }
      double* qb = q + (size_t)inst * no;
      for (int c = tid; c < no; c += NT) {
        double s = 0.0;
        for (int w = 0; w < NS; ++w) s += qpart[w * ldp + c];
        qb[c] = s;
      }
    }
    lds_barrier();  // D: P read out, the region may take the next instance's sources
  }
}

template <int JC>
int launch_jc(const PlanDev& p, const SrcTable& src, const double* params, const double* given,
              double* P, double* q, double* G, double* h, int batch, size_t lds_bytes, int num_cus,
              hipStream_t stream, hipError_t* err) {
  auto kernel = resident_assemble_kernel<JC>;
  if (lds_bytes > 64 * 1024) {
    *err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (*err != hipSuccess) return MPCASM_ERR_HIP;
  }
  // persistent grid: exactly the workgroups that are resident at once, never more
  // than there are instances
  int per_cu = 0;
  *err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, NT, lds_bytes);
  if (*err != hipSuccess) return MPCASM_ERR_HIP;
  if (per_cu < 1) return MPCASM_ERR_LIMIT;
  long grid = (long)num_cus * per_cu;
  if (grid > batch) grid = batch;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(NT), lds_bytes, stream, p, src, params,
                     given, P, q, G, h, batch, g_phase_mask);
  *err = hipGetLastError();
  return *err == hipSuccess ? MPCASM_OK : MPCASM_ERR_HIP;
}

}  // namespace

// 0 when the resident kernel cannot take this plan, else its dynamic LDS bytes
size_t resident_lds_bytes(const PlanDev& p) {
  if (!p.rs_ok || p.rs_jc > RS_JC_MAX || p.no > NT || p.no < 1 || p.max_axes > AXMAX) return 0;
  if ((long)p.rtot * p.ldv > (1 << 20)) return 0;
  return (size_t)resident_layout(p).total_doubles * sizeof(double);
}

int launch_assemble_resident(const PlanDev& p, const SrcTable& src, const double* params,
                             const double* given, double* P, double* q, double* G, double* h,
                             int batch, size_t lds_bytes, int num_cus, hipStream_t stream,
                             hipError_t* err) {
#define MPCASM_RS_ARGS p, src, params, given, P, q, G, h, batch, lds_bytes, num_cus, stream, err
  if (p.rs_jc <= 8) return launch_jc<8>(MPCASM_RS_ARGS);
  if (p.rs_jc <= 12) return launch_jc<12>(MPCASM_RS_ARGS);
  if (p.rs_jc <= 16) return launch_jc<16>(MPCASM_RS_ARGS);
  return launch_jc<RS_JC_MAX>(MPCASM_RS_ARGS);
#undef MPCASM_RS_ARGS
}

}  // namespace mpcasm
