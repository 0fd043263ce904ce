// Two waves per SIMD, each running the SAME program: M MFMAs and V v_fma per iteration, either phase-separated
// (all MFMAs, then all FMAs -- what a "compute, then stage" kernel looks like) or interleaved 1 MFMA : V/M FMAs.
// Answers: does the hardware overlap wave A's MFMA phase with wave B's VALU phase by itself, or must the instruction
// stream of each wave be interleaved?   build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_phases.hip -o /tmp/mvp
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>   // 0: MFMA only, 1: FMA only, 2: phase-separated, 3: interleaved
__global__ __launch_bounds__(512) void k(int iters, float* out) {
  union { uint4 u; bf16x8 v; } a, b;
  a.u = make_uint4(threadIdx.x, 1, 2, 3); b.u = make_uint4(4, 5, 6, threadIdx.x);
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f;
#define M4() c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c1, 0, 0, 0); \
             c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c3, 0, 0, 0)
#define F4() f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f)
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0 || MODE == 2) { M4(); M4(); M4(); M4(); M4(); M4(); }                     // 24 MFMAs
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < 24; ++j) { F4(); }                                                // 96 FMAs
    }
    if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0); F4();
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c1, 0, 0, 0); F4();
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c2, 0, 0, 0); F4();
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c3, 0, 0, 0); F4();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3;
}

template <int MODE>
void run(const char* what, int threads, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, 100, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s %d wave(s)/SIMD: %7.1f ns per iteration (24 MFMA + 96 FMA per wave)\n", what, threads / 256, ms * 1e6 / iters);
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<0>("MFMA only", threads, out);
    run<1>("FMA only", threads, out);
    run<2>("phase-separated (24 MFMA, then 96 FMA)", threads, out);
    run<3>("interleaved (1 MFMA : 4 FMA)", threads, out);
  }
  return 0;
}
