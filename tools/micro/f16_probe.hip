// gfx950 probes behind csrc/fused_bwd6.hip: (1) the rounding of v_cvt_pk_f16_f32, (2) whether the f16 MFMA honours fp16 denormal
// inputs, (3) whether v_cvt_pk_f16_f32 produces denormals.   hipcc --offload-arch=gfx950 -o f16_probe f16_probe.hip && ./f16_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
using f16x8 = __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  uint32_t p;
  const float a = 1.f + 0x1p-11f + 0x1p-20f, b = 1.f + 0x1p-11f;     // RN: 1 + 2^-10, (tie -> even) 1;  RTZ: 1, 1
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
  const float d0 = 0x1p-20f, d1 = 0x1p-24f;                          // fp16 denormals
  uint32_t q;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(q) : "v"(d0), "v"(d1));
  // MFMA: A[row][k] = 2^-20 for k = 0 only (row = lane & 15, kg = lane >> 4 holds k = 8 kg .. + 7), B[k][col] = 2^10 for k = 0
  union { f16x8 v; uint32_t u[4]; } A, B;
  for (int i = 0; i < 4; ++i) { A.u[i] = 0; B.u[i] = 0; }
  if ((lane >> 4) == 0) { A.u[0] = 0x0010u; /* 2^-20 as an fp16 denormal: mantissa 2^-24 * 16 */ B.u[0] = 0x6400u; /* 1024 */ }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.v, B.v, acc, 0, 0, 0);
  if (lane == 0) {
    out[0] = __uint_as_float(p);
    out[1] = __uint_as_float(q);
    out[2] = acc[0];
  }
}
int main() {
  float* d;
  hipMalloc(&d, 16);
  probe<<<1, 64>>>(d);
  float h[4];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  uint32_t p, q;
  memcpy(&p, &h[0], 4); memcpy(&q, &h[1], 4);
  printf("cvt_pk_f16(1+2^-11+2^-20, 1+2^-11) = %04x %04x  (RN: 3c01 3c00; RTZ: 3c00 3c00)\n", p & 0xffff, p >> 16);
  printf("cvt_pk_f16(2^-20, 2^-24) = %04x %04x  (denormals kept: 0010 0001; flushed: 0000 0000)\n", q & 0xffff, q >> 16);
  printf("mfma f16: 2^-20 (denormal) x 2^10 = %g  (honoured: %g; flushed: 0)\n", h[2], 0x1p-10);
  return 0;
}
