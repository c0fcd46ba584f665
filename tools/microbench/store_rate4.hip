// store_rate4.hip -- the K1 fill's OUTPUT pattern alone for the C2 shape (per system S 1152 B and
// U 6144 B into two arrays), 65536 and 524288 systems: which split of systems over wavefronts
// and workgroups lets pure stores run fastest.
//   wave4  : 64-thread workgroups, a wavefront writes 4 consecutive systems (the kernel as it is)
//   wave1  : 64-thread workgroups, one system each
//   wg4    : 256-thread workgroups, 4 systems, one per wavefront
//   wg1    : 256-thread workgroups, ONE system written by all four wavefronts
//   wg2    : 256-thread workgroups, two systems
// Build: hipcc -O3 --offload-arch=gfx950 store_rate4.hip -o store_rate4 ; run: ./store_rate4
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int S2 = 72, U2 = 384;  // 16-byte words per system

template <int NT, int SYS>
__global__ __launch_bounds__(NT) void k_coop(double2* S, double2* U, long nsys) {
  // the workgroup's SYS systems are contiguous in both arrays: flat cooperative copy
  const double2 v = {1.0, 2.0};
  const long s0 = (long)blockIdx.x * SYS;
  if (s0 >= nsys) return;
  for (int e = threadIdx.x; e < SYS * U2; e += NT) U[s0 * U2 + e] = v;
  for (int e = threadIdx.x; e < SYS * S2; e += NT) S[s0 * S2 + e] = v;
}
template <int NT, int SPW>
__global__ __launch_bounds__(NT) void k_wave(double2* S, double2* U, long nsys) {
  // every wavefront writes its own SPW consecutive systems
  const double2 v = {1.0, 2.0};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long s0 = ((long)blockIdx.x * (NT / 64) + wave) * SPW;
  if (s0 >= nsys) return;
  for (int e = lane; e < SPW * U2; e += 64) U[s0 * U2 + e] = v;
  for (int e = lane; e < SPW * S2; e += 64) S[s0 * S2 + e] = v;
}

int main() {
  double2 *S, *U;
  const long maxsys = 524288;
  (void)hipMalloc(&S, maxsys * S2 * 16);
  (void)hipMalloc(&U, maxsys * U2 * 16);
  for (long nsys : {8192L, 65536L, 524288L}) {
    const double bytes = (double)nsys * (S2 + U2) * 16;
    auto run = [&](const char* name, auto launch) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) launch();
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const float us = ms / 10 * 1e3f;
      printf("%7ld systems %7.1f MB  %-44s %9.2f us %7.0f GB/s  %.3f\n", nsys, bytes / 1e6, name, us,
             bytes / us / 1e3, bytes / us / 8e6);
    };
    run("wave4: 64 thr, 4 systems per wavefront", [&] { k_wave<64, 4><<<(nsys + 3) / 4, 64>>>(S, U, nsys); });
    run("wave1: 64 thr, 1 system", [&] { k_wave<64, 1><<<nsys, 64>>>(S, U, nsys); });
    run("wg4  : 256 thr, 4 systems, one per wavefront", [&] { k_wave<256, 1><<<(nsys + 3) / 4, 256>>>(S, U, nsys); });
    run("wg1  : 256 thr, 1 system by all", [&] { k_coop<256, 1><<<nsys, 256>>>(S, U, nsys); });
    run("wg2  : 256 thr, 2 systems by all", [&] { k_coop<256, 2><<<(nsys + 1) / 2, 256>>>(S, U, nsys); });
    run("wg4c : 256 thr, 4 systems by all", [&] { k_coop<256, 4><<<(nsys + 3) / 4, 256>>>(S, U, nsys); });
    run("wg1/128: 128 thr, 1 system by all", [&] { k_coop<128, 1><<<nsys, 128>>>(S, U, nsys); });
  }
  return 0;
}
