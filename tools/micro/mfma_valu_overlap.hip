// Do MFMA and VALU / LDS-store streams from DIFFERENT waves of one SIMD overlap on gfx950?
// 8 waves per workgroup, one workgroup per CU.  mode bit 0: waves 0-3 issue a long chain-free MFMA stream;
// bit 1: waves 4-7 issue a VALU stream (v_fma); bit 2: waves 4-7 issue ds_write_b128; bit 3: the VALU / DS work is done by
// the SAME waves as the MFMAs, interleaved in one instruction stream (waves 4-7 idle).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  __shared__ uint4 lds[512 * 4];
  const int wave = threadIdx.x >> 6;
  union { uint4 u; bf16x8 v; } a, b;
  a.u = make_uint4(threadIdx.x, 1, 2, 3); b.u = make_uint4(4, 5, 6, threadIdx.x);
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f;
  const bool same = mode & 8;
  const bool do_mfma = (mode & 1) && wave < 4;
  const bool do_valu = (mode & 2) && (same ? wave < 4 : wave >= 4);
  const bool do_ds = (mode & 4) && (same ? wave < 4 : wave >= 4);
  for (int i = 0; i < iters; ++i) {
    if (do_mfma && !same) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c3, 0, 0, 0);
      }
    }
    if (do_valu && !same) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
        f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
      }
    }
    if (do_ds && !same) {
#pragma unroll
      for (int j = 0; j < 2; ++j) lds[threadIdx.x + j * 512] = make_uint4(__float_as_uint(f0), i, j, 3);
    }
    if (same && do_mfma) {
      // one stream: 16 MFMAs, after each one 2 fma (32 VALU total) and every 8th a ds_write_b128 (2 total)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
        if (do_valu) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); }
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c1, 0, 0, 0);
        if (do_valu) { f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f); }
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c2, 0, 0, 0);
        if (do_valu) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); }
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c3, 0, 0, 0);
        if (do_valu) { f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f); }
        if (do_ds && (j & 1)) lds[threadIdx.x + (j >> 1) * 512] = make_uint4(__float_as_uint(f0), i, j, 3);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + lds[threadIdx.x ^ 1].x;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int modes[] = {1, 2, 4, 6, 3, 5, 7, 9, 11, 13, 15};
  for (int m : modes) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, m, 100, out); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, m, iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %2d (%s%s%s%s): %8.3f ms   per iteration %.0f ns\n", m, (m & 1) ? "MFMAx16 " : "", (m & 2) ? "VALUx32 " : "",
           (m & 4) ? "DSWx2 " : "", (m & 8) ? "[same waves, interleaved]" : "[MFMA waves 0-3, others waves 4-7]", ms, ms * 1e6 / iters);
  }
  return 0;
}
