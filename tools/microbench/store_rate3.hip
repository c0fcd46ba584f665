// store_rate3.hip -- the persistent assembly kernel's OUTPUT pattern alone (C2: per instance P
// 10368 B, q 288 B, G 21888 B, h 608 B into four arrays), B = 65536 instances = 2.17 GB: what
// pure stores reach in that shape, and what the order of the instances in time is worth.
//   static : workgroup w takes instances w, w + grid, ... (the kernel as it is)
//   ticket : a workgroup takes the next instance from an atomic counter (order follows time)
//   launch : one workgroup per instance, the dispatcher's order
// Build: hipcc -O3 --offload-arch=gfx950 store_rate3.hip -o store_rate3 ; run: ./store_rate3
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int NO = 36, NC = 76;
constexpr int P2 = NO * NO / 2, Q2 = NO / 2, G2 = NC * NO / 2, H2 = NC / 2;  // 16-byte words

__device__ __forceinline__ void write_instance(double2* P, double2* q, double2* G, double2* h, long i,
                                               int tid, int nt, int mode) {
  const double2 v = {1.0, 2.0};
  const bool p8 = mode & 1, noqh = mode & 2, qh4 = mode & 4;
  // G: threads nt/2 .. nt-1 (the stream waves); P, q, h: the others
  const int half = nt / 2;
  if (tid >= half) {
    for (int e = tid - half; e < G2; e += half) G[i * G2 + e] = v;
    if (qh4) {
      if ((i & 3) == 3)
        for (int e = tid - half; e < 4 * H2; e += half) h[(i - 3) * H2 + e] = v;
    } else if (!noqh)
      for (int e = tid - half; e < H2; e += half) h[i * H2 + e] = v;
  } else {
    if (p8) {
      // as the matrix core hands P over: 4x4 blocks, a lane one double (row stride 288 B)
      double* Pd = reinterpret_cast<double*>(P + i * P2);
      for (int e = tid; e < NO * NO; e += half) {
        const int blk = e >> 4, r = (e >> 2) & 3, c = e & 3;
        const int bi = blk / (NO / 4), bj = blk % (NO / 4);
        Pd[(4 * bi + r) * NO + 4 * bj + c] = 1.0;
      }
    } else {
      for (int e = tid; e < P2; e += half) P[i * P2 + e] = v;
    }
    if (qh4) {
      if ((i & 3) == 3)
        for (int e = tid; e < 4 * Q2; e += half) q[(i - 3) * Q2 + e] = v;
    } else if (!noqh)
      for (int e = tid; e < Q2; e += half) q[i * Q2 + e] = v;
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_static(double2* P, double2* q, double2* G, double2* h, long B, int p8, int*) {
  if (p8 & 4) {
    for (long g = blockIdx.x; 4 * g < B; g += gridDim.x)
      for (int k = 0; k < 4; ++k) write_instance(P, q, G, h, 4 * g + k, threadIdx.x, NT, p8);
  } else if (p8 & 8) {  // groups of 4 consecutive instances, nothing else changed
    for (long g = blockIdx.x; 4 * g < B; g += gridDim.x)
      for (int k = 0; k < 4; ++k) write_instance(P, q, G, h, 4 * g + k, threadIdx.x, NT, p8 & 3);
  } else
    for (long i = blockIdx.x; i < B; i += gridDim.x) write_instance(P, q, G, h, i, threadIdx.x, NT, p8);
}
template <int NT>
__global__ __launch_bounds__(NT) void k_ticket(double2* P, double2* q, double2* G, double2* h, long B, int p8, int* ctr) {
  __shared__ int next;
  for (;;) {
    if (threadIdx.x == 0) next = atomicAdd(ctr, 1);
    __syncthreads();
    const long i = next;
    __syncthreads();
    if (i >= B) break;
    write_instance(P, q, G, h, i, threadIdx.x, NT, p8);
  }
}
template <int NT>
__global__ __launch_bounds__(NT) void k_launch(double2* P, double2* q, double2* G, double2* h, long B, int p8, int*) {
  write_instance(P, q, G, h, blockIdx.x, threadIdx.x, NT, p8);
}

typedef void (*kern_t)(double2*, double2*, double2*, double2*, long, int, int*);

int main() {
  const long B = 65536;
  double2 *P, *q, *G, *h;
  int* ctr;
  (void)hipMalloc(&P, B * P2 * 16);
  (void)hipMalloc(&q, B * Q2 * 16);
  (void)hipMalloc(&G, B * G2 * 16);
  (void)hipMalloc(&h, B * H2 * 16);
  (void)hipMalloc(&ctr, 4);
  const double bytes = (double)B * (P2 + Q2 + G2 + H2) * 16;
  auto run = [&](const char* name, kern_t k, unsigned grid, unsigned block, int p8) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float total = 0;
    for (int it = 0; it < 7; ++it) {
      (void)hipMemsetAsync(ctr, 0, 4, 0);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, grid, block, 0, 0, P, q, G, h, B, p8, ctr);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (it >= 2) total += ms;
    }
    const float us = total / 5 * 1e3f;
    printf("%-52s %9.1f us %7.0f GB/s  %.3f\n", name, us, bytes / us / 1e3, bytes / us / 8e6);
  };
  printf("%.1f MB per launch\n", bytes / 1e6);
  const char* names[] = {"P 16-byte contiguous", "P 8-byte blocks", "no q, h; P 16-byte", "no q, h; P 8-byte",
                         "q, h per 4 instances (aligned); P 16-byte", "q, h per 4 instances; P 8-byte"};
  for (int mode = 0; mode < 6; ++mode) {
    printf("-- %s\n", names[mode]);
    run("static, 512 workgroups x 512", k_static<512>, 512, 512, mode);
    run("static, 1024 workgroups x 256", k_static<256>, 1024, 256, mode);
    if (mode < 4) run("launch order, 65536 workgroups x 512", k_launch<512>, 65536, 512, mode);
  }
  printf("-- groups of 4 consecutive instances per workgroup, q and h per instance\n");
  run("static, 512 workgroups x 512, P 16-byte", k_static<512>, 512, 512, 8);
  run("static, 512 workgroups x 512, P 8-byte", k_static<512>, 512, 512, 9);
  return 0;
}
