// How many VALU instructions hide in the shadow of one MFMA on gfx950?  One or two waves per SIMD; per loop trip a wave
// issues 8 MFMAs (ring of 4 independent accumulators), each followed by NF filler instructions (register-only v_fma_f32 on
// 8 independent chains, or v_cvt_pk_bf16_f32 + v_sub_f32 pairs = the bf16 split's mix).  Prints ns and shader cycles per
// MFMA slot (s_memtime).    build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_filler.hip -o /tmp/mfma_filler
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MF, int NF, int FT>   // MF 0: no MFMA, 1: 16x16x32, 2: 32x32x16;  FT 0: v_fma, 1: split mix
__global__ __launch_bounds__(512) void k(int iters, float* out, unsigned long long* cyc) {
  union { uint4 u; bf16x8 v; } a, b;
  a.u = make_uint4(threadIdx.x, 1, 2, 3); b.u = make_uint4(4, 5, 6, threadIdx.x);
  f32x4 c[4]; f32x16 C[4];
  for (int i = 0; i < 4; ++i) { c[i] = f32x4{0, 0, 0, 0}; for (int j = 0; j < 16; ++j) C[i][j] = 0.f; }
  float f[8]; for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
  float k1 = 1.0001f + out[0] * 0.f, k2 = 0.5f + out[1] * 0.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MF == 1) c[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c[m & 3], 0, 0, 0);
      if (MF == 2) C[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, C[m & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int q = (m * NF + j) & 7;
        if (FT == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q]) : "v"(k1), "v"(k2));
        else if (j & 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[q]) : "v"(k2));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[q]) : "v"(k1));
      }
      if (MF) __builtin_amdgcn_sched_group_barrier(8, 1, 0);
      if (NF) __builtin_amdgcn_sched_group_barrier(2, NF, 0);   // asm volatile counts as... keep order anyway
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += c[i][0] + C[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MF, int NF, int FT>
void run(int threads, float* out, unsigned long long* cyc) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MF, NF, FT>), dim3(256), dim3(threads), 0, 0, 50, out, cyc); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL((k<MF, NF, FT>), dim3(256), dim3(threads), 0, 0, iters, out, cyc); (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("mfma=%s fill=%s NF=%d waves/SIMD=%d : %6.2f ns, %6.1f cyc(memtime) per slot\n", MF == 0 ? "none " : MF == 1 ? "16x32" : "32x16",
         FT ? "split" : "fma  ", NF, threads / 256, ms * 1e6 / iters / 8, (double)h / iters / 8);
}

template <int MF, int FT> void sweep(float* out, unsigned long long* cyc) {
  for (int threads : {256, 512}) {
    run<MF, 0, FT>(threads, out, cyc); run<MF, 1, FT>(threads, out, cyc); run<MF, 2, FT>(threads, out, cyc); run<MF, 3, FT>(threads, out, cyc);
    run<MF, 4, FT>(threads, out, cyc); run<MF, 5, FT>(threads, out, cyc); run<MF, 6, FT>(threads, out, cyc); run<MF, 8, FT>(threads, out, cyc);
    run<MF, 12, FT>(threads, out, cyc); run<MF, 16, FT>(threads, out, cyc);
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMemset(out, 0, 256 * 512 * 4);
  unsigned long long* cyc; (void)hipMalloc(&cyc, 8);
  sweep<0, 0>(out, cyc); sweep<1, 0>(out, cyc); sweep<2, 0>(out, cyc); sweep<1, 1>(out, cyc); sweep<2, 1>(out, cyc);
  return 0;
}
