// Does a VALU-only wave run concurrently with an MFMA-only wave on the SAME SIMD?  512-thread workgroups, one per CU: waves 0-3
// (one per SIMD) issue MFMAs only, waves 4-7 (their SIMD-mates) issue v_fma only.  Times: MFMA waves alone, VALU waves alone,
// both together -- for v_mfma_f32_16x16x32_bf16 (4 passes) and v_mfma_f32_32x32x16_bf16 (8 passes).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_corun.hip -o /tmp/corun
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MF>   // 1: 16x16x32, 2: 32x32x16
__global__ __launch_bounds__(512) void k(int mfma_iters, int valu_iters, float* out) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  if (wave < 4) {
    union { uint4 u; bf16x8 v; } a, b;
    a.u = make_uint4(threadIdx.x, 1, 2, 3); b.u = make_uint4(4, 5, 6, threadIdx.x);
    f32x4 c[4]; f32x16 C[4];
    for (int i = 0; i < 4; ++i) { c[i] = f32x4{0, 0, 0, 0}; for (int j = 0; j < 16; ++j) C[i][j] = 0.f; }
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (MF == 1) c[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c[m & 3], 0, 0, 0);
        else C[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, C[m & 3], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) s += c[i][0] + C[i][3];
  } else {
    float f[8]; for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
    float k1 = 1.0001f + out[0] * 0.f, k2 = 0.5f + out[1] * 0.f;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j & 7]) : "v"(k1), "v"(k2));
    }
    for (int i = 0; i < 8; ++i) s += f[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MF>
float run(int mi, int vi, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MF>), dim3(256), dim3(512), 0, 0, mi / 10 + 1, vi / 10 + 1, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL((k<MF>), dim3(256), dim3(512), 0, 0, mi, vi, out); (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}
template <int MF> void sweep(const char* name, float* out) {
  // MFMA: 16 per iteration; VALU: 64 per iteration.  Pick counts so that each side alone takes ~the same time.
  const int mi = MF == 1 ? 8000 : 4000;           // 128k x 16 cyc  or  64k x 32 cyc  ~ 2M cycles
  for (int vi : {0, 2000, 4000, 6000, 8000}) {
    const float both = run<MF>(mi, vi, out), m_only = run<MF>(mi, 0, out), v_only = run<MF>(0, vi, out);
    printf("%s: %6d MFMA + %7d v_fma per SIMD: MFMA wave alone %7.1f us, VALU wave alone %7.1f us, together %7.1f us (sum %7.1f)\n",
           name, mi * 16, vi * 64, m_only, v_only, both, m_only + v_only);
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMemset(out, 0, 256 * 512 * 4);
  sweep<1>("16x16x32", out);
  sweep<2>("32x32x16", out);
  return 0;
}
