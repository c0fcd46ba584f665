// device_prims.h -- device primitives shared by the kernels (gfx950): matrix-core wrappers, the
// LDS-only barrier, DPP / wave reductions.  No dependency beyond the HIP device built-ins, so the
// persistent kernel's source can carry it to hiprtc.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace mpcasm {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16) + C on the fp64 matrix core.
// Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; D register r of lane l
// is element (row = (l >> 4) + 4 r, col = l & 15).
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// four independent 4x4x4 products: lane l feeds A[x][k] / B[k][x] and receives D[k][x] of
// block g, where x = l & 3, g = (l >> 2) & 3, k = l >> 4 (tools/microbench/mfma4x4_layout.hip)
__device__ __forceinline__ double mfma_f64_4x4x4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// A 16-row trip of the persistent kernel: four k-steps A[.][u] B[u][.] into `sum`, the operands
// of lane row lk in four rows RB bytes apart, as ONE statement the compiler does not look into.
// Left to itself it pairs the eight 8-byte reads into ds_read2_b64, which moves half the bytes
// per LDS cycle (MI355X_MICROARCH.md, LDS: 8 cycles per 16 bytes against 2 x 2): here they stay
// eight ds_read_b64 at immediate offsets, the products start as their operands arrive (LDS reads
// return in order), and the waits between dependent products are what the ISA asks for.
// pa / pb: LDS byte addresses; OA / OB: offsets of the lane's first row (both below 64 KiB with
// 3 RB added).
template <int OA, int OB, int RB>
__device__ __forceinline__ double mfma_trip16(unsigned pa, unsigned pb, double sum) {
  static_assert(OA >= 0 && OB >= 0 && OA + 3 * RB < 65536 && OB + 3 * RB < 65536, "ds_read offset field");
  double a0, a1, a2, a3, b0, b1, b2, b3;
  asm volatile(
      "ds_read_b64 %1, %9 offset:%11\n\t"
      "ds_read_b64 %5, %10 offset:%15\n\t"
      "ds_read_b64 %2, %9 offset:%12\n\t"
      "ds_read_b64 %6, %10 offset:%16\n\t"
      "ds_read_b64 %3, %9 offset:%13\n\t"
      "ds_read_b64 %7, %10 offset:%17\n\t"
      "ds_read_b64 %4, %9 offset:%14\n\t"
      "ds_read_b64 %8, %10 offset:%18\n\t"
      "s_waitcnt lgkmcnt(6)\n\t"
      "v_mfma_f64_4x4x4_4b_f64 %0, %1, %5, %0\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "s_nop 2\n\t"
      "v_mfma_f64_4x4x4_4b_f64 %0, %2, %6, %0\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "s_nop 2\n\t"
      "v_mfma_f64_4x4x4_4b_f64 %0, %3, %7, %0\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_nop 2\n\t"
      "v_mfma_f64_4x4x4_4b_f64 %0, %4, %8, %0\n\t"
      "s_nop 5"
      : "+v"(sum), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
      : "v"(pa), "v"(pb), "n"(OA), "n"(OA + RB), "n"(OA + 2 * RB), "n"(OA + 3 * RB), "n"(OB),
        "n"(OB + RB), "n"(OB + 2 * RB), "n"(OB + 3 * RB));
  return sum;
}

// ... with the rows' distance known only at run time (the ahead-of-time kernel): the same
// statement on eight addresses.
__device__ __forceinline__ double mfma_trip16_at(unsigned pa, unsigned pb, unsigned row_bytes, double sum) {
  double a0, a1, a2, a3, b0, b1, b2, b3;
  const unsigned pa1 = pa + row_bytes, pa2 = pa1 + row_bytes, pa3 = pa2 + row_bytes;
  const unsigned pb1 = pb + row_bytes, pb2 = pb1 + row_bytes, pb3 = pb2 + row_bytes;
  asm volatile(
      "ds_read_b64 %1, %9\n\t"
      "ds_read_b64 %5, %13\n\t"
      "ds_read_b64 %2, %10\n\t"
      "ds_read_b64 %6, %14\n\t"
      "ds_read_b64 %3, %11\n\t"
      "ds_read_b64 %7, %15\n\t"
      "ds_read_b64 %4, %12\n\t"
      "ds_read_b64 %8, %16\n\t"
      "s_waitcnt lgkmcnt(6)\n\t"
      "v_mfma_f64_4x4x4_4b_f64 %0, %1, %5, %0\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "s_nop 2\n\t"
      "v_mfma_f64_4x4x4_4b_f64 %0, %2, %6, %0\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "s_nop 2\n\t"
      "v_mfma_f64_4x4x4_4b_f64 %0, %3, %7, %0\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_nop 2\n\t"
      "v_mfma_f64_4x4x4_4b_f64 %0, %4, %8, %0\n\t"
      "s_nop 5"
      : "+v"(sum), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
      : "v"(pa), "v"(pa1), "v"(pa2), "v"(pa3), "v"(pb), "v"(pb1), "v"(pb2), "v"(pb3));
  return sum;
}

// Results that a wavefront writes as whole cache lines (16 bytes per lane, consecutive lanes)
// leave for HBM with nontemporal stores: written once, not read again by this launch, they need
// not stay in L2 -- measured on the persistent assembly kernel at B = 65536 (2.2 GB of outputs):
// 491 -> 419 us on the same box, nothing lost at B = 4096.  NOT for short runs that only make a
// whole line together with a neighbour's store (the 32-byte runs of a 4x4 block of P, q, h): those
// must meet in L2 (same kernel, B = 4096, blocks of P nontemporal: 34 -> 51 us).
#ifndef MPCASM_PLAIN_STORES
__device__ __forceinline__ void store_result(double2* dst, double2 v) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d w = {v.x, v.y};
  __builtin_nontemporal_store(w, reinterpret_cast<v2d*>(dst));
}
__device__ __forceinline__ void store_result(double* dst, double v) { __builtin_nontemporal_store(v, dst); }
#else
__device__ __forceinline__ void store_result(double2* dst, double2 v) { *dst = v; }
__device__ __forceinline__ void store_result(double* dst, double v) { *dst = v; }
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// the vector-memory counter (s_waitcnt vmcnt(0)), i.e. it waits for every global
// store in flight to be acknowledged by HBM -- a full memory round trip per
// barrier in kernels that stream results out between LDS phases.  LDS operations
// of a wavefront complete in order, so lgkmcnt(0) + s_barrier is sufficient for
// data handed over through LDS; loads whose results feed LDS writes are waited
// for by the compiler through the data dependency.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// the value of lane T of every quad, in all four lanes of the quad (DPP quad_perm)
template <int T>
__device__ __forceinline__ double quad_broadcast(double v) {
  constexpr int ctl = T | (T << 2) | (T << 4) | (T << 6);
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), ctl, 0xF, 0xF, true),
                          __builtin_amdgcn_mov_dpp(__double2loint(v), ctl, 0xF, 0xF, true));
}

// sum over the 64 lanes of a wavefront, result in every lane
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace mpcasm
