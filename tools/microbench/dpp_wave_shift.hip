// dpp_wave_shift.hip -- what v_mov_b32_dpp wave_shl:1 / wave_shr:1 move on gfx950: every lane
// holds its own index, the shifted value is printed per lane (bound_ctrl: lanes without a source
// read 0).  The scan form of the tiled kernel (csrc/tiled.hip) carries P[r+1][c+1] into column c:
// lane i <- lane i + 1.
// Build: hipcc -O3 --offload-arch=gfx950 dpp_wave_shift.hip -o dpp_wave_shift
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(0, lane + 100, 0x130, 0xF, 0xF, true);       // wave_shl:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(0, lane + 100, 0x138, 0xF, 0xF, true);  // wave_shr:1
  out[128 + lane] = __shfl_down(lane + 100, 1, 64);
}

int main() {
  int* d;
  if (hipMalloc(&d, 192 * 4) != hipSuccess) return 1;
  probe<<<1, 64>>>(d);
  static int h[192];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const char* name[3] = {"wave_shl:1", "wave_shr:1", "__shfl_down 1"};
  for (int v = 0; v < 3; ++v) {
    printf("%-14s", name[v]);
    for (int l = 0; l < 64; ++l) printf(" %d", h[v * 64 + l] ? h[v * 64 + l] - 100 : -1);
    printf("\n");
  }
  int ok = 1;
  for (int l = 0; l < 64; ++l) ok &= h[l] == (l < 63 ? l + 101 : 0);
  printf("wave_shl:1 is lane i <- lane i + 1 with 0 into lane 63: %s\n", ok ? "yes" : "NO");
  return 0;
}
