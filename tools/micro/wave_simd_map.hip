// Which SIMD does wave w of a 512-thread workgroup run on (gfx950)?  HW_REG_HW_ID (id 4): wave_id [3:0], simd_id [5:4], cu_id [11:8].
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_ID, offset 0, width 32
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned* d; hipMalloc(&d, 16 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(16), dim3(512), 0, 0, d); hipDeviceSynchronize();
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) {
    printf("workgroup %d:", b);
    for (int w = 0; w < 8; ++w) printf("  w%d->simd%u(slot%u,cu%u)", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
    printf("\n");
  }
  return 0;
}
