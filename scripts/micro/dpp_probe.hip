// Which lane does a DPP row shift / rotate read from?  (gfx950)   hipcc --offload-arch=gfx950 -O3 dpp_probe.hip -o dpp_probe && ./dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x101, 0xf, 0xf, false);        // row_shl:1, bound_ctrl off, old = -1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x12f, 0xf, 0xf, false);   // row_ror:15
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x102, 0xf, 0xf, false);  // row_shl:2
  out[192 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x12e, 0xf, 0xf, false);  // row_ror:14
  out[256 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xf, 0xf, false);  // row_shr:1
}
int main() {
  int* d; hipMalloc(&d, 320 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[320]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[5] = {"row_shl:1", "row_ror:15", "row_shl:2", "row_ror:14", "row_shr:1"};
  for (int t = 0; t < 5; ++t) { printf("%-10s:", names[t]); for (int i = 0; i < 20; ++i) printf(" %d", h[t * 64 + i]); printf("\n"); }
  return 0;
}
