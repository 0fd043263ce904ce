// The one-pass backward of the fused Linear (math and operand images: fused_bwd.hip; role split: fused_bwd4.hip) with THREE waves
// per SIMD: 768 threads, waves 0-7 vector, waves 8-11 matrix; wave w runs on SIMD w % 4, so every SIMD holds two vector waves and
// one matrix wave (168 registers each).
//
// Why: in fused_bwd4.hip a SIMD's single vector wave ran at ~7.6 cycles per instruction -- not issue-bound (removing 10 % of its
// issue slots changed nothing) but latency-bound: one wave has two independent rows of dependent chains (DPP row sums, hashes,
// three-plane splits) to interleave.  Two vector waves per SIMD hide each other's latency, as they do in the forward
// (fused_fwd2.hip).  What made that impossible there was the matrix wave: W slice (96 registers) + its 64 x 64 tile of gW (64) do
// not fit 168.  Here the matrix waves keep ONLY the W slice and do backward-data; the weight gradient moves to the vector waves --
// the workgroup's one gW is eight 64 x 32 tiles of 32 accumulator registers each, and a vector wave's weight-gradient step
// (24 x v_mfma_f32_32x32x16_bf16 per stage, operands by ds_read_b64_tr_b16 from the shared images) sits in the tick where its
// vector work is light.  Per SIMD and stage the matrix pipe carries the same 3072 cycles as before (1536 backward-data + 2 x 768).
//
// Stage = 32 rows, two workgroup barriers per stage:
//     tick 2k    vector: S0(k+1): gy(k+1) under the mask -> ga[(k+1) % 3];  S2a(k): x(k) -> xhat, keep factors, u -> u[k % 2]
//                matrix: S1(k): gu = ga[k % 3] @ W
//     tick 2k+1  vector: S2b(k): gu -> LayerNorm backward -> gx;  S3(k-1): gW tile += ga[(k-1) % 3]^T u[(k-1) % 2]
//                matrix: idle (half of its issue slots are the vector waves' anyway)
// Vector wave v owns row 4 v + rg (rg = lane >> 4) of a stage -- one row per lane, 8 elements, the row inside one DPP row of 16
// lanes -- and the gW tile o in [64 (v >> 2), +64), i in [32 (v & 3), +32).  gx and the gW partial are bit-identical to
// fused_bwd4.hip's (same fragment layouts, same accumulation order); the column sums fold eight waves' rows in another order.
// LDS: 3 x 24 KB ga + 2 x 24 KB u + 16.5 KB gu + gamma / beta = 138 KB.
#include <stdlib.h>

#include "common.h"

namespace allset {

using bf16x8t = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4t = __attribute__((ext_vector_type(4))) float;
using f32x16t = __attribute__((ext_vector_type(16))) float;
typedef short v4st_t __attribute__((ext_vector_type(4)));
union FragT { uint4 u; bf16x8t v; struct { v4st_t lo, hi; } t; };
constexpr int kTBlock = 768;
constexpr int kTRows = 32;                     // rows per stage
constexpr int kTVWaves = 8;

template <int CTRL>
__device__ __forceinline__ float dpp_ft(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum_t(float v) {     // sum over the 16 lanes of a DPP row, result in every lane of it
  v += dpp_ft<0xB1>(v);
  v += dpp_ft<0x4E>(v);
  v += dpp_ft<0x141>(v);
  v += dpp_ft<0x140>(v);
  return v;
}
__device__ __forceinline__ bf16x8t tr_frag2_t(const uint8_t* lo, const uint8_t* hi) {
  FragT f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4st_t*)(lo));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4st_t*)(hi));
  return f.v;
}
// byte offset of (row, column byte) in a [rows][256 B] bf16 plane (fused_bwd4.hip img_off_r)
__device__ __forceinline__ int img_off_t(int row, int colbyte) {
  return row * 256 + ((((colbyte >> 6) ^ row) & 3) << 6) + (((((colbyte >> 4) & 3) ^ (row >> 2)) & 3) << 4) + (colbyte & 15);
}
__device__ __forceinline__ uint32_t hash_mix_t(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; return x; }
#define ALLSET_T3_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define ALLSET_FRESH_LANE_T(name) \
  int name = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); __asm__ volatile("" : "+v"(name))

template <bool HAS_LN, bool DROP_IN, bool RELU_IN, bool HAS_MASK, bool HAS_ACC>
__global__ __launch_bounds__(kTBlock) void fused_linear_bwd_roles3_kernel(
    const float* __restrict__ gy, int64_t ldg, const uint32_t* __restrict__ mask, float p_out, const float* __restrict__ W,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float p_in, uint64_t seed_in, float* gx, int64_t ldgx,
    float* __restrict__ part_ln, float* __restrict__ part_w, float* __restrict__ part_b, int64_t n,
    const uint64_t* __restrict__ seed_base, int64_t pstride_w, int64_t pstride_b, int64_t pstride_ln, const float* acc_in,
    int64_t ldacc) {
  constexpr int OD = 128, ID = 128;
  constexpr int R = kTRows;
  constexpr int PLANE = R * 256;                 // bytes per bf16 plane of an image
  constexpr int IMG = 3 * PLANE;                 // one image: planes h, m, l
  constexpr int SPG = 132;                       // pitch (floats) of the gu tile
  __shared__ __attribute__((aligned(16))) uint8_t sGA[3 * IMG];
  __shared__ __attribute__((aligned(16))) uint8_t sU[2 * IMG];
  __shared__ __attribute__((aligned(16))) float sGU[R * SPG];
  __shared__ __attribute__((aligned(16))) float sG[ID];
  __shared__ __attribute__((aligned(16))) float sB[ID];
  seed_in = resolve_seed(seed_base, seed_in);
  const int tid = threadIdx.x;
  if (tid < ID) { sG[tid] = HAS_LN ? gamma[tid] : 1.f; sB[tid] = HAS_LN ? beta[tid] : 0.f; }
  const int lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t n_stages = (n + R - 1) / R;
  // this workgroup's stages: blockIdx.x + k * gridDim.x, k = 0 .. T - 1 (T >= 1: the grid never exceeds the stage count)
  const int64_t T = (n_stages - blockIdx.x + gridDim.x - 1) / gridDim.x;
  auto stage_of = [&](int64_t k) -> int64_t { return blockIdx.x + k * static_cast<int64_t>(gridDim.x); };
  auto rows_left = [&](int64_t stage) -> int {
    const int64_t left = n - stage * R;
    return left >= R ? R : (left > 0 ? static_cast<int>(left) : 0);
  };
  __syncthreads();

  if (wave < kTVWaves) {
    // =================================================== vector waves ===================================================
    const float inv_i = 1.f / static_cast<float>(ID);
    const float keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
    const float keep_in = DROP_IN ? 1.f / (1.f - p_in) : 1.f;
    const uint32_t thr_in = drop_threshold(p_in);
    const uint32_t seed_lo = static_cast<uint32_t>(seed_in);
    const int c = lane0 & 15, rg = lane0 >> 4;
    const int lr = 4 * wave + rg;                // this lane's row of a stage; columns 64 hb + 4 c .. + 3, hb = 0, 1
    const int oh = wave >> 2, iq = wave & 3;     // weight-gradient tile: o in [64 oh, +64), i in [32 iq, +32)
    float4 dg[2], db[2], gbv[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      dg[hb] = make_float4(0.f, 0.f, 0.f, 0.f); db[hb] = make_float4(0.f, 0.f, 0.f, 0.f); gbv[hb] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x16t gw[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) gw[a][q] = 0.f;
    // ONE register set per operand (168 registers: the wave also holds a gW tile), requested as soon as the previous stage's copy is
    // consumed and used two ticks later -- a whole stage time in flight; 8 vector waves x 4 KB = 32 KB per CU, what fused_bwd4.hip's
    // four waves x two sets had.
    float4 agS[2]; uint32_t amS[2];              // [hb]: gy row / mask words of the next stage
    float4 xrS[2]; float2 stS;                   // [hb]: x row; statistics
    auto request_gy = [&](int64_t k, float4 (&ag)[2], uint32_t (&am)[2]) {
      const int64_t s0 = k < T ? stage_of(k) : stage_of(T - 1);           // past the end: re-read the last stage (never consumed)
      const int nrc = max(rows_left(s0), 1);
      const int lrc = min(lr, nrc - 1);
      const char* base = reinterpret_cast<const char*>(gy + s0 * R * ldg);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        ag[hb] = *reinterpret_cast<const float4*>(base + static_cast<uint32_t>(lrc) * static_cast<uint32_t>(ldg) * 4u + 256 * hb + 16 * c);
        if constexpr (HAS_MASK)     // "mask layout" (include/allset_hip.h): block (row / 16, column / 64), dword (row % 16, 32-column group)
          am[hb] = (mask + ((s0 * (R / 16) + (lrc >> 4)) * (OD / 64) + hb) * 32)[((lrc & 15) >> 2) * 8 + (lrc & 3) * 2 + (c >> 3)];
      }
    };
    auto request_x = [&](int64_t k, float4 (&xr)[2], float2& st) {
      const int64_t s0 = k < T ? stage_of(k) : stage_of(T - 1);
      const int nrc = max(rows_left(s0), 1);
      const int lrc = min(lr, nrc - 1);
      const char* xb = reinterpret_cast<const char*>(x + s0 * R * ldx);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
        xr[hb] = *reinterpret_cast<const float4*>(xb + static_cast<uint32_t>(lrc) * static_cast<uint32_t>(ldx) * 4u + 256 * hb + 16 * c);
      if constexpr (HAS_LN) st = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(stats + s0 * R * 2) + lrc * 8);
    };
    // ---- S0(k): ga = gy under the forward's epilogue mask, three bf16 planes into ga[k % 3]; then the request for gy(k + 2)
    auto S0 = [&](int64_t k, float4 (&ag)[2], uint32_t (&am)[2]) {
      const bool valid = lr < rows_left(stage_of(k));
      uint8_t* img = sGA + (k % 3) * IMG;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        float4 v = ag[hb];
        if constexpr (HAS_MASK) {
          const uint32_t bits = valid ? (am[hb] >> (c & 7)) : 0u;         // bit 8 q + (c % 8) for column 64 hb + 4 c + q
          v.x = (bits & 0x1u) ? v.x * keep_out : 0.f; v.y = (bits & 0x100u) ? v.y * keep_out : 0.f;
          v.z = (bits & 0x10000u) ? v.z * keep_out : 0.f; v.w = (bits & 0x1000000u) ? v.w * keep_out : 0.f;
        } else {                                          // (selects, not a branch: dead rows exist in the last stage only)
          v.x = valid ? v.x : 0.f; v.y = valid ? v.y : 0.f; v.z = valid ? v.z : 0.f; v.w = valid ? v.w : 0.f;
        }
        gbv[hb].x += v.x; gbv[hb].y += v.y; gbv[hb].z += v.z; gbv[hb].w += v.w;        // bias gradient: column sums of ga
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_bf16(v.x, v.y, h0, m0, l0);
        split3_bf16(v.z, v.w, h1, m1, l1);
        const int wo = img_off_t(lr, 128 * hb + 8 * c);
        *reinterpret_cast<uint2*>(img + 0 * PLANE + wo) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(img + 1 * PLANE + wo) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(img + 2 * PLANE + wo) = make_uint2(l0, l1);
      }
      __builtin_amdgcn_sched_barrier(0);
      request_gy(k + 1, ag, am);                // into the registers just consumed: the next stage
      __builtin_amdgcn_sched_barrier(0);
    };
    // ---- S2a(k) needs only x: u = dropout_in(LN(relu_in(x))) -> three bf16 planes into u[k % 2]; xhat, the keep factors and the
    // relu signs stay in registers for S2b(k), which is what needs gu
    float4 xhK[2];             // xhat (LayerNorm) or relu_in(x) of stage k, [hb]
    float4 kpK[2];             // dropout-in keep factors (keep_in or 0)
    uint32_t xbK = 0;          // "raw x > 0" flags, bit 4 hb + q
    float rstdK = 1.f;
    auto S2a = [&](int64_t k, float4 (&xr)[2], float2& st) {
      const int64_t stage = stage_of(k);
      const bool live = lr < rows_left(stage);
      uint8_t* img = sU + (k % 2) * IMG;
      const uint64_t stage_pair = static_cast<uint64_t>(stage) * (R * ID / 2);
      const uint32_t stage_pair_lo = static_cast<uint32_t>(stage_pair);
      const uint32_t hi_term = __umul24(static_cast<uint32_t>(stage_pair >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed_in >> 32);
      xbK = 0;
      rstdK = HAS_LN ? st.y : 1.f;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        float4 kp = make_float4(1.f, 1.f, 1.f, 1.f);
        if constexpr (DROP_IN) {
          // pair index of (row, column) = stage * 2048 + (lr * 128 + column) / 2: the lane's part is < 2048 -> an OR (common.h pair_hash)
          const uint32_t lo = stage_pair_lo | static_cast<uint32_t>((lr * ID + 64 * hb + 4 * c) >> 1);
          const uint32_t h0 = hash_mix_t((lo ^ seed_lo) * 0x9E3779B1U + hi_term);
          const uint32_t h1 = hash_mix_t(((lo + 1u) ^ seed_lo) * 0x9E3779B1U + hi_term);
          kp.x = (h0 & 0xffffu) >= thr_in ? keep_in : 0.f; kp.y = (h0 >> 16) >= thr_in ? keep_in : 0.f;
          kp.z = (h1 & 0xffffu) >= thr_in ? keep_in : 0.f; kp.w = (h1 >> 16) >= thr_in ? keep_in : 0.f;
        }
        kpK[hb] = kp;
        float4 t = xr[hb];
        if (RELU_IN) {
          xbK |= ((t.x > 0.f ? 1u : 0u) | (t.y > 0.f ? 2u : 0u) | (t.z > 0.f ? 4u : 0u) | (t.w > 0.f ? 8u : 0u)) << (4 * hb);
          t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
        }
        float4 u = t;
        if constexpr (HAS_LN) {
          const float mean = st.x, rstd = st.y;
          float4 xh = make_float4((t.x - mean) * rstd, (t.y - mean) * rstd, (t.z - mean) * rstd, (t.w - mean) * rstd);
          xh.x = live ? xh.x : 0.f; xh.y = live ? xh.y : 0.f; xh.z = live ? xh.z : 0.f; xh.w = live ? xh.w : 0.f;
          t = xh;
          const float4 gam = *reinterpret_cast<const float4*>(&sG[64 * hb + 4 * c]);
          const float4 bet = *reinterpret_cast<const float4*>(&sB[64 * hb + 4 * c]);
          u = make_float4(fmaf(xh.x, gam.x, bet.x), fmaf(xh.y, gam.y, bet.y), fmaf(xh.z, gam.z, bet.z), fmaf(xh.w, gam.w, bet.w));
        }
        xhK[hb] = t;
        if constexpr (DROP_IN) { u.x *= kp.x; u.y *= kp.y; u.z *= kp.z; u.w *= kp.w; }
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_bf16(u.x, u.y, h0, m0, l0);
        split3_bf16(u.z, u.w, h1, m1, l1);
        const int wo = img_off_t(lr, 128 * hb + 8 * c);
        *reinterpret_cast<uint2*>(img + 0 * PLANE + wo) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(img + 1 * PLANE + wo) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(img + 2 * PLANE + wo) = make_uint2(l0, l1);
      }
      if constexpr (RELU_IN) __asm__ volatile("" : "+v"(xbK));     // (packed here, not at its use)
      __builtin_amdgcn_sched_barrier(0);
      request_x(k + 1, xr, st);                   // x is consumed: the request for the next stage
      __builtin_amdgcn_sched_barrier(0);
    };
    auto S2b = [&](int64_t k) {
      const int64_t stage = stage_of(k);
      const int nrows = rows_left(stage);
      const bool live = lr < nrows;
      float4 gam[2];
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) gam[hb] = *reinterpret_cast<const float4*>(&sG[64 * hb + 4 * c]);
      // gx = acc_in + ...: a second gradient branch of the same tensor, summed here (may alias gx: each element is read and
      // written by the same lane).  Requested first, consumed last.
      float4 acc[2];
      if constexpr (HAS_ACC) {
        const int lrc = min(lr, max(nrows, 1) - 1);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          acc[hb] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(acc_in + stage * R * ldacc) +
                                                     static_cast<uint32_t>(lrc) * static_cast<uint32_t>(ldacc) * 4u + 256 * hb + 16 * c);
      }
      float4 v[2];
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        v[hb] = *reinterpret_cast<const float4*>(&sGU[lr * SPG + 64 * hb + 4 * c]);
        if constexpr (DROP_IN) { v[hb].x *= kpK[hb].x; v[hb].y *= kpK[hb].y; v[hb].z *= kpK[hb].z; v[hb].w *= kpK[hb].w; }
        if constexpr (HAS_LN) {
          const float4 xh = xhK[hb];
          dg[hb].x = fmaf(v[hb].x, xh.x, dg[hb].x); dg[hb].y = fmaf(v[hb].y, xh.y, dg[hb].y);
          dg[hb].z = fmaf(v[hb].z, xh.z, dg[hb].z); dg[hb].w = fmaf(v[hb].w, xh.w, dg[hb].w);
          db[hb].x += v[hb].x; db[hb].y += v[hb].y; db[hb].z += v[hb].z; db[hb].w += v[hb].w;
          v[hb].x *= gam[hb].x; v[hb].y *= gam[hb].y; v[hb].z *= gam[hb].z; v[hb].w *= gam[hb].w;
          a1 += (v[hb].x + v[hb].y) + (v[hb].z + v[hb].w);
          a2 = fmaf(v[hb].x, xh.x, fmaf(v[hb].y, xh.y, fmaf(v[hb].z, xh.z, fmaf(v[hb].w, xh.w, a2))));
        }
      }
      float s1 = 0.f, s2 = 0.f;
      if constexpr (HAS_LN) { s1 = row16_sum_t(a1) * inv_i; s2 = row16_sum_t(a2) * inv_i; }
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        float4 o = v[hb];
        if constexpr (HAS_LN) {
          const float rstd = rstdK;
          const float4 xh = xhK[hb];
          o = make_float4(rstd * (v[hb].x - s1 - xh.x * s2), rstd * (v[hb].y - s1 - xh.y * s2),
                          rstd * (v[hb].z - s1 - xh.z * s2), rstd * (v[hb].w - s1 - xh.w * s2));
        }
        if (RELU_IN) {
          const uint32_t xb = xbK >> (4 * hb);
          o.x = (xb & 1u) ? o.x : 0.f; o.y = (xb & 2u) ? o.y : 0.f; o.z = (xb & 4u) ? o.z : 0.f; o.w = (xb & 8u) ? o.w : 0.f;
        }
        if constexpr (HAS_ACC) { o.x += acc[hb].x; o.y += acc[hb].y; o.z += acc[hb].z; o.w += acc[hb].w; }
        if (live)
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(gx + stage * R * ldgx) +
                                     static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldgx) * 4u + 256 * hb + 16 * c) = o;
      }
    };
    // ---- S3(k): weight gradient, this wave's 64 x 32 tile of gW; K = the stage's 32 rows in two steps of 16; A = ga^T, B = u
    auto S3 = [&](int64_t k) {
      ALLSET_FRESH_LANE_T(lane_w);
      const uint8_t* ia = sGA + (k % 3) * IMG;
      const uint8_t* iu = sU + (k % 2) * IMG;
      const int q4 = lane_w >> 4, tr_r = (lane_w & 15) >> 2, tr_row = 8 * (q4 >> 1) + tr_r, tr_in = 32 * (q4 & 1) + 8 * (lane_w & 3);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16x8t wa[3], wb[3];                 // one o-tile's fragments at a time (registers): 24 instead of 36 live
        const int b_lo = img_off_t(16 * kb + tr_row, 64 * iq + tr_in), b_hi = img_off_t(16 * kb + tr_row + 4, 64 * iq + tr_in);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wb[pl] = tr_frag2_t(iu + pl * PLANE + b_lo, iu + pl * PLANE + b_hi);
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};     // l.h, h.l, m.m, m.h, h.m, h.h
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          const int a_lo = img_off_t(16 * kb + tr_row, 64 * (2 * oh + tl) + tr_in), a_hi = img_off_t(16 * kb + tr_row + 4, 64 * (2 * oh + tl) + tr_in);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) wa[pl] = tr_frag2_t(ia + pl * PLANE + a_lo, ia + pl * PLANE + a_hi);
#pragma unroll
          for (int pr = 0; pr < 6; ++pr) gw[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[PA_[pr]], wb[PB_[pr]], gw[tl], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // Tick 2k+1.  (Tried: the two vector waves of a SIMD, v and v + 4, taking S2b / S3 in OPPOSITE order so that one is in its
    // matrix phase while the other is in its vector phase -- the duplicated bodies push six instantiations over 168 registers,
    // 11-30 spills, 0.61 ms.)
#define ALLSET_T3_TICKB(kk) do { S2b(kk); S3((kk) - 1); } while (0)
    request_gy(0, agS, amS);
    request_x(0, xrS, stS);
    S0(0, agS, amS);
    ALLSET_T3_TICK();
    // Per stage: S2a(k) first (consumes x(k), requests x(k+1)), then S0(k+1) (consumes gy(k+1), requests gy(k+2)): at every use the
    // loads still outstanding behind the needed one were issued in THIS tick by unconditional code, and everything older is a
    // stage time old -- no s_waitcnt of the compiler's ever waits for a request that has just gone out.  First and last stage
    // peeled: the trip itself has no conditional part.
    S2a(0, xrS, stS);
    if (1 < T) S0(1, agS, amS);
    ALLSET_T3_TICK();
    S2b(0);
    ALLSET_T3_TICK();
    int64_t k = 1;
    for (; k + 1 < T; ++k) {
      S2a(k, xrS, stS);
      S0(k + 1, agS, amS);
      ALLSET_T3_TICK();
      ALLSET_T3_TICKB(k);
      ALLSET_T3_TICK();
    }
    if (k < T) {                            // the last stage
      S2a(k, xrS, stS);
      ALLSET_T3_TICK();
      ALLSET_T3_TICKB(k);
      ALLSET_T3_TICK();
    }
    S3(T - 1);
    // ---- this wave's 64 x 32 tile of the workgroup's gW partial
    {
      float* pw = part_w + static_cast<int64_t>(blockIdx.x) * pstride_w;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = (2 * oh + a) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane0 >> 5);
          pw[o * ID + iq * 32 + (lane0 & 31)] = gw[a][q];
        }
    }
    ALLSET_T3_TICK();                       // every wave is done with the images: ga's space takes the column sums
    // ---- column sums held by the vector waves (dgamma, dbeta, bias gradient): the four row groups of a lane column fold first,
    // then the eight waves through LDS in a fixed order
    {
      float* red = reinterpret_cast<float*>(sGA);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        float4 a = dg[hb], b = db[hb], g3 = gbv[hb];
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
          a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
          b.x += __shfl_xor(b.x, off); b.y += __shfl_xor(b.y, off); b.z += __shfl_xor(b.z, off); b.w += __shfl_xor(b.w, off);
          g3.x += __shfl_xor(g3.x, off); g3.y += __shfl_xor(g3.y, off); g3.z += __shfl_xor(g3.z, off); g3.w += __shfl_xor(g3.w, off);
        }
        if (lane0 < 16) {
          *reinterpret_cast<float4*>(&red[wave * 3 * ID + 64 * hb + 4 * lane0]) = a;
          *reinterpret_cast<float4*>(&red[wave * 3 * ID + ID + 64 * hb + 4 * lane0]) = b;
          *reinterpret_cast<float4*>(&red[wave * 3 * ID + 2 * ID + 64 * hb + 4 * lane0]) = g3;
        }
      }
    }
  } else {
    // =================================================== matrix waves ===================================================
    const int m = wave - kTVWaves;
    // this wave's slice of W as MFMA B fragments: column tile ct (16 columns 32 m + 16 ct + n), k-step t, plane pl; lane
    // (n = lane & 15, kg = lane >> 4) holds W[o = 32 kg + 8 t + j][column], j = 0..7 (k-order of fused_mlp.hip / fused_bwd.hip)
    FragT wq[2][4][3];
    {
      const int nn = lane0 & 15, kg = lane0 >> 4;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          uint32_t ph[4], pm[4], pl[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int o = 32 * kg + 8 * t + 2 * j, i = 32 * m + 16 * ct + nn;
            split3_bf16(W[o * ID + i], W[(o + 1) * ID + i], ph[j], pm[j], pl[j]);
          }
          wq[ct][t][0].u = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          wq[ct][t][1].u = make_uint4(pm[0], pm[1], pm[2], pm[3]);
          wq[ct][t][2].u = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
    }
    // ---- S1(k): backward-data for this wave's 32 output columns of the stage's 32 rows: 2 row tiles x 2 column tiles = four
    // independent accumulator chains; the A fragments of step t + 1 are requested before step t's MFMAs
    auto S1 = [&](int64_t k) {
      ALLSET_FRESH_LANE_T(lane);
      const int ri = lane & 15, kg = lane >> 4;
      const uint8_t* img = sGA + (k % 3) * IMG;
      auto load_a = [&](FragT (&f0)[3], FragT (&f1)[3], int t) {
        const int o0 = img_off_t(ri, 64 * kg + 16 * t), o1 = img_off_t(16 + ri, 64 * kg + 16 * t);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          f0[pl].u = *reinterpret_cast<const uint4*>(img + pl * PLANE + o0);
          f1[pl].u = *reinterpret_cast<const uint4*>(img + pl * PLANE + o1);
        }
      };
      FragT fa0[2][3], fa1[2][3];
      f32x4t acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4t{0.f, 0.f, 0.f, 0.f};
      load_a(fa0[0], fa1[0], 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) load_a(fa0[(t + 1) & 1], fa1[(t + 1) & 1], t + 1);
        const FragT (&a0)[3] = fa0[t & 1];
        const FragT (&a1)[3] = fa1[t & 1];
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};     // l.h, h.l, m.m, m.h, h.m, h.h
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[PA_[pr]].v, wq[0][t][PB_[pr]].v, acc[0][0], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[PA_[pr]].v, wq[0][t][PB_[pr]].v, acc[1][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[PA_[pr]].v, wq[1][t][PB_[pr]].v, acc[0][1], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[PA_[pr]].v, wq[1][t][PB_[pr]].v, acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // acc[rt][ct][r] = gu[row 16 rt + 4 kg + r][column 32 m + 16 ct + ri]
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) sGU[(16 * rt + 4 * kg + r) * SPG + 32 * m + 16 * ct + ri] = acc[rt][ct][r];
    };
    ALLSET_T3_TICK();
    for (int64_t k = 0; k < T; ++k) {
      S1(k);
      ALLSET_T3_TICK();
      ALLSET_T3_TICK();
    }
    ALLSET_T3_TICK();                       // (the vector waves' last weight-gradient step)
  }
  __syncthreads();
  if (tid < 3 * ID) {
    const float* red = reinterpret_cast<const float*>(sGA);
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < kTVWaves; ++v) s += red[v * 3 * ID + tid];
    const int64_t slice = blockIdx.x;
    if (tid < 2 * ID) { if constexpr (HAS_LN) part_ln[slice * pstride_ln + tid] = s; }
    else if (part_b != nullptr) part_b[slice * pstride_b + (tid - 2 * ID)] = s;
  }
}

}  // namespace allset

using namespace allset;

// OFF by default: measured on the bench (same box) 0.427 ms per [1M,128] Linear against 0.411 for fused_bwd4.hip -- the third wave per
// SIMD does not pay here (has the numbers).  ALLSET_BWD_ROLES3=1 selects it (O = I = 128; bf16x6 mode); gx and gW are
// bit-identical to fused_bwd4.hip's (tests/test_gpu_dense.py::test_one_pass_backward_three_waves_per_simd_variant).
int fused_linear_bwd_roles3_supported(int64_t O, int64_t I) {
  const char* e = getenv("ALLSET_BWD_ROLES3");
  if (!(e && e[0] == '1')) return 0;
  return (dense_mfma_x6() && O == 128 && I == 128) ? 1 : 0;
}

// Same grid and slice count as fused_bwd4.hip (one persistent workgroup per CU, ONE partial slice per workgroup).
int launch_fused_linear_bwd_roles3(unsigned grid, hipStream_t st, bool ln, bool drop, bool relu, bool hm, const float* gy,
                                   int64_t ldg, const uint32_t* mask, float p_out, const float* W, const float* x, int64_t ldx,
                                   const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                                   float* gx, int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n,
                                   const uint64_t* seed_base, int64_t psw, int64_t psb, int64_t psl, const float* acc_in,
                                   int64_t ldacc) {
#define ALLSET_R3_K(LN, DI, RI, HM, HA)                                                                                        \
  fused_linear_bwd_roles3_kernel<LN, DI, RI, HM, HA><<<grid, kTBlock, 0, st>>>(gy, ldg, mask, p_out, W, x, ldx, stats, gamma,    \
                                                                              beta, p_in, seed_in, gx, ldgx, part_ln, part_w,   \
                                                                              part_b, n, seed_base, psw, psb, psl, acc_in, ldacc)
  if (acc_in != nullptr) { ALLSET_R3_K(false, false, false, false, true); return 0; }     // (bwd_all_combo: plain Linear only)
#define ALLSET_R3_M(LN, DI, RI) do { if (hm) ALLSET_R3_K(LN, DI, RI, true, false); else ALLSET_R3_K(LN, DI, RI, false, false); } while (0)
  if (!relu) { if (ln) ALLSET_R3_M(true, false, false); else ALLSET_R3_M(false, false, false); }
  else if (ln) { if (drop) ALLSET_R3_M(true, true, true); else ALLSET_R3_M(true, false, true); }
  else { if (drop) ALLSET_R3_M(false, true, true); else ALLSET_R3_M(false, false, true); }
#undef ALLSET_R3_M
#undef ALLSET_R3_K
  return 0;
}
