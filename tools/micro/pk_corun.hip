// Packed fp32 VALU (v_pk_fma_f32: two FMAs per lane) in a VALU-only wave next to an MFMA-only wave on the same SIMD: does the packed
// instruction issue at the scalar one's rate (2x the arithmetic), and does it disturb the matrix wave?  (MI355X_MICROARCH.md prices
// packed f32 as an anti-lever INSIDE an MFMA wave; the split-role kernels keep vector and matrix work in different waves.)
// 768-thread workgroups, one per CU: waves 0-3 MFMA only (f16 16x16x32), waves 4-11 VALU only (two per SIMD, as in fused_bwd6.hip).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/pk_corun.hip -o /tmp/pk_corun
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8 = __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(768) void k(int mfma_iters, int valu_iters, float* out) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  if (wave < 4) {
    union { uint4 u; f16x8 v; } a, b;
    a.u = make_uint4(threadIdx.x, 1, 2, 3); b.u = make_uint4(4, 5, 6, threadIdx.x);
    f32x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) c[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c[m & 3], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += c[i][0];
  } else {
    f32x2 f[8]; for (int i = 0; i < 8; ++i) f[i] = f32x2{float(threadIdx.x + i), float(i)};
    f32x2 k1 = f32x2{1.0001f + out[0] * 0.f, 1.0002f}, k2 = f32x2{0.5f + out[1] * 0.f, 0.25f};
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(f[j & 7]) : "v"(k1), "v"(k2));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j & 7].x) : "v"(k1.x), "v"(k2.x));
      }
    }
    for (int i = 0; i < 8; ++i) s += f[i].x + f[i].y;
  }
  out[blockIdx.x * 768 + threadIdx.x] = s;
}

template <int PK>
float run(int mi, int vi, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PK>), dim3(256), dim3(768), 0, 0, mi / 10 + 1, vi / 10 + 1, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL((k<PK>), dim3(256), dim3(768), 0, 0, mi, vi, out); (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}
template <int PK> void sweep(const char* name, float* out) {
  const int mi = 8000;
  for (int vi : {2000, 4000}) {
    const float both = run<PK>(mi, vi, out), m_only = run<PK>(mi, 0, out), v_only = run<PK>(0, vi, out);
    printf("%s: %6d MFMA + 2 x %7d instructions per SIMD: MFMA wave alone %7.1f us, VALU waves alone %7.1f us, together %7.1f us (sum %7.1f)\n",
           name, mi * 16, vi * 64, m_only, v_only, both, m_only + v_only);
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 768 * 4); (void)hipMemset(out, 0, 256 * 768 * 4);
  sweep<0>("v_fma_f32   ", out);
  sweep<1>("v_pk_fma_f32", out);
  return 0;
}
