// ds_read_b64_tr_b16 semantics probe (gfx950): every lane supplies its own 8-byte-aligned LDS address; what lands where?
// Image: 16-bit value = its own element index.  Lane l (group q = l>>4, i = l&15) supplies the address of element
// q*256 + (i>>2)*ROW + 4*(i&3)  (a [4][16] block with row stride ROW elements); expectation from the guide:
// out[lane c][j] = block[j][c].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint16_t* out, int row) {
  __shared__ __attribute__((aligned(16))) uint16_t img[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) img[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x, q = l >> 4, i = l & 15;
  const uint32_t addr = (uint32_t)(uintptr_t)img + 2u * (q * 1024 + (i >> 2) * row + 4 * (i & 3));
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[l * 4 + 0] = r.x & 0xffff; out[l * 4 + 1] = r.x >> 16; out[l * 4 + 2] = r.y & 0xffff; out[l * 4 + 3] = r.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int row : {16, 128}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, row); hipDeviceSynchronize();
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      const int want = (l >> 4) * 1024 + j * row + (l & 15);
      if (h[l * 4 + j] != want) ++bad;
    }
    printf("row stride %d: %d mismatches vs out[lane c][j] = block[j][c]\n", row, bad);
    for (int l = 0; l < 20; ++l) printf("  lane %2d: %4u %4u %4u %4u\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
