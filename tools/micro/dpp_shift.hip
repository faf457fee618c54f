// prints what DPP wave_shr:1 / wave_shl:1 deliver per lane on this part (gfx9 DPP controls 0x138 / 0x130)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  const int v = 100 + lane;
  out[lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, true);
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, true);
}
int main() {
  int* d; hipMalloc(&d, 128 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("wave_shr:1 (0x138): lane0=%d lane1=%d lane31=%d lane32=%d lane63=%d\n", h[0], h[1], h[31], h[32], h[63]);
  printf("wave_shl:1 (0x130): lane0=%d lane1=%d lane31=%d lane32=%d lane62=%d lane63=%d\n", h[64], h[65], h[95], h[96], h[126], h[127]);
  return 0;
}
