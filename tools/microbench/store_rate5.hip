// store_rate5.hip -- the C3 assembly's OUTPUT pattern alone (3-D LIPM N=32: per instance P 73 728 B,
// G 150 528 B; q, h left out), B = 16 384 instances = 3.67 GB, by persistent workgroups of 512 threads
// that run in step: does the time depend on WHERE the arrays were allocated (profiles/r03_placement.txt:
// the real kernel takes 0.55 or 0.75 ms by placement), and does the distance between consecutive
// instances matter?  Both strides are multiples of 1 KiB (P: 2^13 x 9, G: 2^10 x 147): workgroups that
// write the same offset of consecutive instances at the same time meet in the low address bits.
//   pad   : bytes added to both instance strides (0 = dense, the C-ABI's layout)
//   rot   : workgroup w starts its instance's G and P `rot * w` 16-byte pieces in (and wraps)
// Every variant on `sets` fresh allocations, all kept alive.
// Build: hipcc -O3 --offload-arch=gfx950 store_rate5.hip -o store_rate5 ; run: ./store_rate5
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

constexpr int NO = 96, NC = 196, NT = 512;
constexpr long P2 = NO * NO / 2, G2 = NC * NO / 2;  // 16-byte pieces per instance

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(NT) void k_c3(double2* P, double2* G, long B, long sp, long sg, int rot) {
  const v2d v = {1.0, 2.0};
  const int tid = threadIdx.x;
  const long shift = (long)rot * blockIdx.x;
  for (long i = blockIdx.x; i < B; i += gridDim.x) {
    double2* g = G + i * sg;
    double2* p = P + i * sp;
    // the stream waves (threads 256..511) write G, the matrix waves P -- as the kernel does
    if (tid >= NT / 2) {
      for (long e = tid - NT / 2; e < G2; e += NT / 2) {
        long x = e + shift;
        x -= (x >= G2) ? G2 * (x / G2) : 0;
        __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(g + x));
      }
    } else {
      for (long e = tid; e < P2; e += NT / 2) {
        long x = e + shift;
        x -= (x >= P2) ? P2 * (x / P2) : 0;
        __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(p + x));
      }
    }
  }
}

int main() {
  const long B = 16384;
  const int sets = 6;
  const long pads[] = {0, 128, 256, 2048 + 128, 4096 + 128};
  const int rots[] = {0, 8, 24, 257};
  const double bytes = (double)B * (P2 + G2) * 16;
  printf("C3 output pattern, %.2f GB per launch, 256 workgroups x 512; us per launch over %d placements: min / median / max\n",
         bytes / 1e9, sets);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  std::vector<void*> keep;
  for (long pad : pads)
    for (int rot : rots) {
      if (pad != 0 && rot != 0) continue;
      const long sp = P2 + pad / 16, sg = G2 + pad / 16;
      std::vector<float> t;
      for (int s = 0; s < sets; ++s) {
        double2 *P, *G;
        if (hipMalloc(&P, B * sp * 16) != hipSuccess || hipMalloc(&G, B * sg * 16) != hipSuccess) {
          printf("out of memory\n");
          return 1;
        }
        keep.push_back(P);
        keep.push_back(G);
        float total = 0;
        for (int it = 0; it < 6; ++it) {
          (void)hipEventRecord(e0);
          hipLaunchKernelGGL(k_c3, 256, NT, 0, 0, P, G, B, sp, sg, rot);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (it >= 2) total += ms;
        }
        t.push_back(total / 4 * 1e3f);
      }
      std::sort(t.begin(), t.end());
      printf("pad %5ld B  rot %3d : %8.1f / %8.1f / %8.1f us   (%.3f / %.3f / %.3f of 8 TB/s)\n", pad, rot, t.front(),
             t[t.size() / 2], t.back(), bytes / t.front() / 8e6, bytes / t[t.size() / 2] / 8e6, bytes / t.back() / 8e6);
      if (keep.size() > 24) {  // (keep at most ~12 sets alive: 44 GB)
        for (size_t i = 0; i < 12; ++i) (void)hipFree(keep[i]);
        keep.erase(keep.begin(), keep.begin() + 12);
      }
    }
  return 0;
}
