// The one-pass backward of the fused Linear (math and operand images: fused_bwd.hip) with the CU's eight waves SPLIT BY ROLE:
// waves 0-3 do all the vector work, waves 4-7 all the matrix work, and wave w shares its SIMD with wave w + 4 -- one of each.
//
// Why: tools/micro/mfma_valu_corun.hip -- an MFMA-only wave and a VALU-only wave on the same SIMD run concurrently (1000 us of
// MFMAs + 570 us of v_fma finish in 1040 us), while two waves that are both in a matrix phase share the pipe and two waves
// that are both in a vector phase share the issue port.  Every symmetric organisation of this kernel (one wave per SIMD,
// fused_bwd.hip; pairs, fused_bwd2.hip; eight cooperating waves in lock-step phases, fused_bwd3.hip) measured
// vector time + matrix time (+ barrier skew): 0.47-0.64 ms per [1M,128] x [128,128] Linear.  With the roles split the two
// kinds of work overlap BY CONSTRUCTION and the kernel costs max(vector, matrix) per stage.
//
// Stage = 32 rows.  Dependencies of a stage: S0 (vector: ga = masked gy -> three bf16 planes) -> S1 (matrix: gu = ga @ W)
// -> S2 (vector: LayerNorm backward -> gx; u recomputed -> three bf16 planes) -> S3 (matrix: gW += ga^T u).  Software pipeline,
// one workgroup barrier per tick, two ticks per stage:
//     tick 2k    vector waves: S0(k+1) -> ga[(k+1) % 3]; S2a(k): x -> u[k % 2]  | matrix waves: S1(k): ga[k % 3] -> gu
//     tick 2k+1  vector waves: S2b(k): gu -> gx                             | matrix waves: S3(k-1): ga[(k-1) % 3], u[(k-1) % 2] -> gW
// Three ga buffers, two u buffers and one gu buffer make every hand-off a tick boundary (checked case by case in DESIGN.md 6a).
//   vector wave v: rows 8 v .. 8 v + 7 of the stage, complete rows, each row in the 16 lanes of one DPP row (the LayerNorm row
//     sums are four DPP adds, no LDS trip; every global access moves 256-byte row segments); it has no accumulators, so the next stage's gy / mask words / x / statistics sit in registers a
//     stage ahead;
//   matrix wave m: backward-data for output columns 32 m .. 32 m + 31 -- its slice of W, three bf16 planes of W[:, 32 columns],
//     is 96 registers and never leaves them (no W in LDS) -- and the 64 x 64 tile (m >> 1, m & 1) of the workgroup's ONE gW
//     (64 accumulator registers); operands from the row-major LDS images: 16-byte fragments (backward-data), ds_read_b64_tr_b16
//     transposes (weight gradient).
// LDS: 3 x 24 KB ga + 2 x 24 KB u + 16.5 KB gu + gamma / beta = 138 KB.
#include <stdlib.h>

#include "common.h"

namespace allset {

using bf16x8r = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4r = __attribute__((ext_vector_type(4))) float;
using f32x16r = __attribute__((ext_vector_type(16))) float;
typedef short v4sr_t __attribute__((ext_vector_type(4)));
union FragR { uint4 u; bf16x8r v; struct { v4sr_t lo, hi; } t; };
constexpr int kRBlock = 512;
constexpr int kRRows = 32;                     // rows per stage

template <int CTRL>
__device__ __forceinline__ float dpp_fr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum_r(float v) {     // sum over the 16 lanes of a DPP row, result in every lane of it
  v += dpp_fr<0xB1>(v);
  v += dpp_fr<0x4E>(v);
  v += dpp_fr<0x141>(v);
  v += dpp_fr<0x140>(v);
  return v;
}
__device__ __forceinline__ bf16x8r tr_frag2_r(const uint8_t* lo, const uint8_t* hi) {
  FragR f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4sr_t*)(lo));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4sr_t*)(hi));
  return f.v;
}
// byte offset of (row, column byte) in a [rows][256 B] bf16 plane: 64-byte chunk XOR row & 3, 16-byte piece XOR (row >> 2) & 3
// (fused_bwd3.hip: both the row-wise 16-byte fragment reads and the transpose reads are then conflict-free)
__device__ __forceinline__ int img_off_r(int row, int colbyte) {
  return row * 256 + ((((colbyte >> 6) ^ row) & 3) << 6) + (((((colbyte >> 4) & 3) ^ (row >> 2)) & 3) << 4) + (colbyte & 15);
}
__device__ __forceinline__ uint32_t hash_mix_r(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; return x; }
#ifdef ALLSET_ABL4_NOBAR            // ablation builds only: timing without the barriers, results wrong
#define ALLSET_ROLE_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define ALLSET_ROLE_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#define ALLSET_FRESH_LANE_R(name) \
  int name = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); __asm__ volatile("" : "+v"(name))

template <bool HAS_LN, bool DROP_IN, bool RELU_IN, bool HAS_MASK, bool HAS_ACC, bool HAS_AUX>
__global__ __launch_bounds__(kRBlock, 2) void fused_linear_bwd_roles_kernel(
    const float* __restrict__ gy, int64_t ldg, const uint32_t* __restrict__ mask, float p_out, const float* __restrict__ W,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float p_in, uint64_t seed_in, float* gx, int64_t ldgx,
    float* __restrict__ part_ln, float* __restrict__ part_w, float* __restrict__ part_b, int64_t n,
    const uint64_t* __restrict__ seed_base, int64_t pstride_w, int64_t pstride_b, int64_t pstride_ln, const float* acc_in,
    int64_t ldacc, const float* __restrict__ aux_g, const float* __restrict__ aux_w, int64_t gcb, int64_t xcb, int64_t gxcb,
    float ln_inv) {
  // ln_inv: 1 / 128 for the LayerNorm backward; 0 drops its two row-mean terms -- with the {0, 1} row statistics the forward wrote
  // in that mode, gx = gu * gamma: the backward of the per-column affine prologue (ALLSET_NORM_COLUMN_AFFINE, fused_fwd2.hip);
  // part_ln then holds sum_r gu * x and sum_r gu per column.
  // gcb / xcb / gxcb: 0 = row-major with the operand's leading dimension; cb > 0 = COLUMN-BLOCKED [128 / cb][n][cb] (ld == cb) for gy /
  // x / gx: the layout of the column-sharded layer's exchange buffers (fused_fwd2.hip has the forward side).
  // HAS_AUX (plain Linear only): four auxiliary output columns rode along in the forward (PMA's folded logits, fused_mlp.hip
  // aux_out = x aux_w^T + aux_b); here gx += aux_g[n,4] @ aux_w[4,I] as four FMAs per element in S2b, and the columns' own
  // weight / bias gradient (aux_g^T x [4,I], column sums of aux_g) accumulate in the vector waves' registers next to the bias
  // gradient -- partials go to part_ln[slice][4 I + 4], which a LayerNorm-free Linear does not use.
  static_assert(!HAS_AUX || (!HAS_LN && !DROP_IN && !RELU_IN && !HAS_MASK && !HAS_ACC), "aux columns: plain Linear only");
  constexpr int OD = 128, ID = 128;
  constexpr int R = kRRows;
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
#ifdef ALLSET_ABL4_TIMING          // diagnostic builds only: cycles per segment of waves 0 (vector) and 4 (matrix) of workgroup 0
  uint64_t tph[4] = {0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define ALLSET_RMARK(k) do { const uint64_t tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } while (0)
#else
#define ALLSET_RMARK(k) do {} while (0)
#endif

  if (wave < 4) {
    // =================================================== vector waves ===================================================
    const float inv_i = ln_inv;
    const float keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
    const float keep_in = DROP_IN ? 1.f / (1.f - p_in) : 1.f;
    const uint32_t thr_in = drop_threshold(p_in);
    const uint32_t seed_lo = static_cast<uint32_t>(seed_in);
    float4 dg[2], db[2], gbv[2];
    // A lane's columns 64 hb + 4 c: byte offset of columns 4 c (one 32-bit VGPR per operand) + hb x a wave-uniform step --
    // 256 bytes in a row-major row, 64 columns' worth of blocks (256 n bytes) in the blocked layout (cb <= 64 divides 64)
    const int c40 = 4 * (lane0 & 15);
    const uint32_t cog0 = gcb ? static_cast<uint32_t>(((c40 / gcb) * n * gcb + c40 % gcb) * 4) : 4u * c40;
    const uint32_t cox0 = xcb ? static_cast<uint32_t>(((c40 / xcb) * n * xcb + c40 % xcb) * 4) : 4u * c40;
    const uint32_t cogx0 = gxcb ? static_cast<uint32_t>(((c40 / gxcb) * n * gxcb + c40 % gxcb) * 4) : 4u * c40;
    const int64_t dhg = gcb ? 256 * n : 256, dhx = xcb ? 256 * n : 256, dhgx = gxcb ? 256 * n : 256;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      dg[hb] = make_float4(0.f, 0.f, 0.f, 0.f); db[hb] = make_float4(0.f, 0.f, 0.f, 0.f); gbv[hb] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // lane (c = lane & 15, rg = lane >> 4) owns rows 8 wave + rg + 4 j (j = 0, 1) of a stage, columns 64 hb + 4 c .. + 3 (hb = 0, 1):
    // a row lives in the 16 lanes of ONE DPP row, so the LayerNorm row sums are four DPP adds and never take an LDS round trip
    // (a vector wave is alone on its SIMD as far as vector work goes: every such trip is exposed latency); a load / store
    // instruction moves 256-byte segments of four rows.
    // Two register sets each: the operands of the next TWO stages are in flight.  The vector waves are the only ones that touch
    // HBM, four per CU; with one stage ahead that is 32 KB in flight per CU = 8 MB chip-wide, and at ~2 us of loaded latency
    // 4 TB/s -- exactly where every earlier variant of this kernel saturated with its matrix work compiled out.  Two stages
    // ahead doubles the bytes in flight (the waves have the registers: no accumulators here).
    float4 agS[2][2][2]; uint32_t amS[2][2][2];      // [set][j][hb]: gy rows / mask words
    float4 xrS[2][2][2]; float2 stS[2][2];           // [set][j][hb]: x rows; [set][j]: statistics
    float4 g4S[2][2];                                // [set][j]: the row's four auxiliary-column gradients (HAS_AUX)
    float4 w4r[4][2], ga4[4][2], g4K[2];             // aux_w slice of this lane's columns; aux weight-gradient accumulators; stage k's g4
    float4 gb4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (HAS_AUX) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          w4r[q][hb] = *reinterpret_cast<const float4*>(aux_w + q * ID + 64 * hb + 4 * (lane0 & 15));
          ga4[q][hb] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    auto request_gy = [&](int64_t k, int lane, float4 (&ag)[2][2], uint32_t (&am)[2][2]) {
      const int c = lane & 15, rg = lane >> 4;
      const int64_t s0 = k < T ? stage_of(k) : stage_of(T - 1);           // past the end: re-read the last stage (never consumed)
      const int nrc = max(rows_left(s0), 1);
      const char* base = reinterpret_cast<const char*>(gy + s0 * R * ldg);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = min(8 * wave + rg + 4 * j, nrc - 1);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          ag[j][hb] = *reinterpret_cast<const float4*>(base + hb * dhg + (static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldg) * 4u + cog0));
          if constexpr (HAS_MASK)   // "mask layout" (include/allset_hip.h): block (row / 16, column / 64), dword (row % 16, 32-column group)
            am[j][hb] = (mask + ((s0 * (R / 16) + (lr >> 4)) * (OD / 64) + hb) * 32)[((lr & 15) >> 2) * 8 + (lr & 3) * 2 + (c >> 3)];
        }
      }
    };
    auto request_x = [&](int64_t k, int lane, float4 (&xr)[2][2], float2 (&st)[2], float4 (&g4)[2]) {
      const int c = lane & 15, rg = lane >> 4;
      const int64_t s0 = k < T ? stage_of(k) : stage_of(T - 1);
      const int nrc = max(rows_left(s0), 1);
      const char* xb = reinterpret_cast<const char*>(x + s0 * R * ldx);
      const char* sb = reinterpret_cast<const char*>(stats + s0 * R * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = min(8 * wave + rg + 4 * j, nrc - 1);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          xr[j][hb] = *reinterpret_cast<const float4*>(xb + hb * dhx + (static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldx) * 4u + cox0));
        if constexpr (HAS_LN) st[j] = *reinterpret_cast<const float2*>(sb + lr * 8);
        if constexpr (HAS_AUX) g4[j] = *reinterpret_cast<const float4*>(aux_g + (s0 * R + lr) * 4);
      }
    };
    // ---- S0(k): ga = gy under the forward's epilogue mask, three bf16 planes into ga[k % 3]; then the request for gy(k + 1)
    auto S0 = [&](int64_t k, float4 (&ag)[2][2], uint32_t (&am)[2][2]) {
      const int lane = lane0;                   // (no re-derivation here: the vector waves have registers for the hoisted offsets)
      const int c = lane & 15, rg = lane >> 4;
      const int nrows = rows_left(stage_of(k));
      uint8_t* img = sGA + (k % 3) * IMG;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = 8 * wave + rg + 4 * j;
        const bool valid = lr < nrows;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          float4 v = ag[j][hb];
          if constexpr (HAS_MASK) {
            const uint32_t bits = valid ? (am[j][hb] >> (c & 7)) : 0u;    // bit 8 q + (c % 8) for column 64 hb + 4 c + q
            v.x = (bits & 0x1u) ? v.x * keep_out : 0.f; v.y = (bits & 0x100u) ? v.y * keep_out : 0.f;
            v.z = (bits & 0x10000u) ? v.z * keep_out : 0.f; v.w = (bits & 0x1000000u) ? v.w * keep_out : 0.f;
          } else {                                        // (selects, not a branch: dead rows exist in the last stage only)
            v.x = valid ? v.x : 0.f; v.y = valid ? v.y : 0.f; v.z = valid ? v.z : 0.f; v.w = valid ? v.w : 0.f;
          }
          gbv[hb].x += v.x; gbv[hb].y += v.y; gbv[hb].z += v.z; gbv[hb].w += v.w;      // bias gradient: column sums of ga
          uint32_t h0, m0, l0, h1, m1, l1;
          split3_bf16(v.x, v.y, h0, m0, l0);
          split3_bf16(v.z, v.w, h1, m1, l1);
          const int wo = img_off_r(lr, 128 * hb + 8 * c);
          *reinterpret_cast<uint2*>(img + 0 * PLANE + wo) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(img + 1 * PLANE + wo) = make_uint2(m0, m1);
          *reinterpret_cast<uint2*>(img + 2 * PLANE + wo) = make_uint2(l0, l1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      request_gy(k + 2, lane, ag, am);          // into the set just consumed: two stages ahead
      __builtin_amdgcn_sched_barrier(0);
    };
    // ---- S2 in two halves.  S2a(k) needs only x: u = dropout_in(LN(relu_in(x))) -> three bf16 planes into u[k % 2]; it runs in the
    // SAME tick as S0(k + 1), where the matrix waves' backward-data step left the vector waves idle for ~900 cycles, and leaves
    // xhat, the keep factors and the relu signs in registers.  S2b(k) is what needs gu: the LayerNorm backward -> gx.
    float4 xhK[2][2];          // xhat (LayerNorm) or relu_in(x) of stage k, [j][hb]
    float4 kpK[2][2];          // dropout-in keep factors (keep_in or 0)
    uint32_t xbK = 0;          // "raw x > 0" flags, bit 8 j + 4 hb + q
    float rstdK[2];
    auto S2a = [&](int64_t k, float4 (&xr)[2][2], float2 (&st)[2], float4 (&g4)[2]) {
      const int lane = lane0;
      const int c = lane & 15, rg = lane >> 4;
      const int64_t stage = stage_of(k);
      const int nrows = rows_left(stage);
      uint8_t* img = sU + (k % 2) * IMG;
      float4 gam[2], bet[2];
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        gam[hb] = *reinterpret_cast<const float4*>(&sG[64 * hb + 4 * c]);
        bet[hb] = *reinterpret_cast<const float4*>(&sB[64 * hb + 4 * c]);
      }
      const uint64_t stage_pair = static_cast<uint64_t>(stage) * (R * ID / 2);
      const uint32_t stage_pair_lo = static_cast<uint32_t>(stage_pair);
      const uint32_t hi_term = __umul24(static_cast<uint32_t>(stage_pair >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed_in >> 32);
      const uint64_t stage_quad = static_cast<uint64_t>(stage) * (R * ID / 4);
      const uint32_t stage_quad_lo = static_cast<uint32_t>(stage_quad);
      const uint32_t hi_term_q = __umul24(static_cast<uint32_t>(stage_quad >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed_in >> 32);
      xbK = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = 8 * wave + rg + 4 * j;
        const bool live = lr < nrows;
        rstdK[j] = HAS_LN ? st[j].y : 1.f;
        if constexpr (HAS_AUX) {          // (a dead row's clamped re-read is a live row's gradient: zero it)
          float4 g = g4[j];
          g.x = live ? g.x : 0.f; g.y = live ? g.y : 0.f; g.z = live ? g.z : 0.f; g.w = live ? g.w : 0.f;
          g4K[j] = g;
          gb4.x += g.x; gb4.y += g.y; gb4.z += g.z; gb4.w += g.w;
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          float4 kp = make_float4(1.f, 1.f, 1.f, 1.f);
          if constexpr (DROP_IN) {
            // pair index of (row, column) = stage * 2048 + (lr * 128 + column) / 2: the lane's part is < 2048 -> an OR (common.h pair_hash)
            if (thr_in & kDrop8) {     // 8 bits per element: ONE hash for the lane's float4 (quad index = stage * 1024 + lane part)
              const uint32_t lo = stage_quad_lo | static_cast<uint32_t>((lr * ID + 64 * hb + 4 * c) >> 2);
              const uint32_t h = hash_mix_r((lo ^ seed_lo) * 0x9E3779B1U + hi_term_q), t8 = thr_in & 0xffu;
              kp.x = (h & 0xffu) >= t8 ? keep_in : 0.f; kp.y = ((h >> 8) & 0xffu) >= t8 ? keep_in : 0.f;
              kp.z = ((h >> 16) & 0xffu) >= t8 ? keep_in : 0.f; kp.w = (h >> 24) >= t8 ? keep_in : 0.f;
            } else {
              const uint32_t lo = stage_pair_lo | static_cast<uint32_t>((lr * ID + 64 * hb + 4 * c) >> 1);
              const uint32_t h0 = hash_mix_r((lo ^ seed_lo) * 0x9E3779B1U + hi_term);
              const uint32_t h1 = hash_mix_r(((lo + 1u) ^ seed_lo) * 0x9E3779B1U + hi_term);
              kp.x = (h0 & 0xffffu) >= thr_in ? keep_in : 0.f; kp.y = (h0 >> 16) >= thr_in ? keep_in : 0.f;
              kp.z = (h1 & 0xffffu) >= thr_in ? keep_in : 0.f; kp.w = (h1 >> 16) >= thr_in ? keep_in : 0.f;
            }
          }
          kpK[j][hb] = kp;
          float4 t = xr[j][hb];
          if (RELU_IN) {
            xbK |= ((t.x > 0.f ? 1u : 0u) | (t.y > 0.f ? 2u : 0u) | (t.z > 0.f ? 4u : 0u) | (t.w > 0.f ? 8u : 0u)) << (8 * j + 4 * hb);
            t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
          }
          float4 u = t;
          if constexpr (HAS_LN) {
            const float mean = st[j].x, rstd = st[j].y;
            float4 xh = make_float4((t.x - mean) * rstd, (t.y - mean) * rstd, (t.z - mean) * rstd, (t.w - mean) * rstd);
            xh.x = live ? xh.x : 0.f; xh.y = live ? xh.y : 0.f; xh.z = live ? xh.z : 0.f; xh.w = live ? xh.w : 0.f;
            t = xh;
            u = make_float4(fmaf(xh.x, gam[hb].x, bet[hb].x), fmaf(xh.y, gam[hb].y, bet[hb].y), fmaf(xh.z, gam[hb].z, bet[hb].z),
                            fmaf(xh.w, gam[hb].w, bet[hb].w));
          }
          xhK[j][hb] = t;
          if constexpr (HAS_AUX) {        // aux_g^T x: this lane's four columns of the four auxiliary rows
            const float4 g = g4K[j];
            ga4[0][hb].x = fmaf(g.x, t.x, ga4[0][hb].x); ga4[0][hb].y = fmaf(g.x, t.y, ga4[0][hb].y); ga4[0][hb].z = fmaf(g.x, t.z, ga4[0][hb].z); ga4[0][hb].w = fmaf(g.x, t.w, ga4[0][hb].w);
            ga4[1][hb].x = fmaf(g.y, t.x, ga4[1][hb].x); ga4[1][hb].y = fmaf(g.y, t.y, ga4[1][hb].y); ga4[1][hb].z = fmaf(g.y, t.z, ga4[1][hb].z); ga4[1][hb].w = fmaf(g.y, t.w, ga4[1][hb].w);
            ga4[2][hb].x = fmaf(g.z, t.x, ga4[2][hb].x); ga4[2][hb].y = fmaf(g.z, t.y, ga4[2][hb].y); ga4[2][hb].z = fmaf(g.z, t.z, ga4[2][hb].z); ga4[2][hb].w = fmaf(g.z, t.w, ga4[2][hb].w);
            ga4[3][hb].x = fmaf(g.w, t.x, ga4[3][hb].x); ga4[3][hb].y = fmaf(g.w, t.y, ga4[3][hb].y); ga4[3][hb].z = fmaf(g.w, t.z, ga4[3][hb].z); ga4[3][hb].w = fmaf(g.w, t.w, ga4[3][hb].w);
          }
          if (j == 0) {         // row 0's planes here, row 1's in S2b: the split that balances the two ticks against the matrix waves
            if constexpr (DROP_IN) { u.x *= kp.x; u.y *= kp.y; u.z *= kp.z; u.w *= kp.w; }
            uint32_t h0, m0, l0, h1, m1, l1;
            split3_bf16(u.x, u.y, h0, m0, l0);
            split3_bf16(u.z, u.w, h1, m1, l1);
            const int wo = img_off_r(lr, 128 * hb + 8 * c);
            *reinterpret_cast<uint2*>(img + 0 * PLANE + wo) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(img + 1 * PLANE + wo) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(img + 2 * PLANE + wo) = make_uint2(l0, l1);
          }
        }
      }
      if constexpr (RELU_IN) __asm__ volatile("" : "+v"(xbK));     // (packed here, not at its use)
      __builtin_amdgcn_sched_barrier(0);
      request_x(k + 2, lane, xr, st, g4);         // x is consumed: the request for two stages ahead goes out a tick earlier
      __builtin_amdgcn_sched_barrier(0);
    };
    auto S2b = [&](int64_t k) {
      const int lane = lane0;
      const int c = lane & 15, rg = lane >> 4;
      const int64_t stage = stage_of(k);
      const int nrows = rows_left(stage);
      float4 gam[2];
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) gam[hb] = *reinterpret_cast<const float4*>(&sG[64 * hb + 4 * c]);
      // gx = acc_in + ...: a second gradient branch of the same tensor, summed here (may alias gx: each element is read and
      // written by the same lane).  Requested first, consumed last.
      float4 acc[2][2];
      if constexpr (HAS_ACC) {
        const int nrc = max(nrows, 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int lrc = min(8 * wave + rg + 4 * j, nrc - 1);
#pragma unroll
          for (int hb = 0; hb < 2; ++hb)
            acc[j][hb] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(acc_in + stage * R * ldacc) +
                                                          static_cast<uint32_t>(lrc) * static_cast<uint32_t>(ldacc) * 4u + 256 * hb + 16 * c);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {            // (the two rows are independent chains: left to the scheduler to interleave)
        const int lr = 8 * wave + rg + 4 * j;
        const bool live = lr < nrows;
        float4 v[2];
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          v[hb] = *reinterpret_cast<const float4*>(&sGU[lr * SPG + 64 * hb + 4 * c]);
          if constexpr (DROP_IN) {
            v[hb].x *= kpK[j][hb].x; v[hb].y *= kpK[j][hb].y; v[hb].z *= kpK[j][hb].z; v[hb].w *= kpK[j][hb].w;
          }
          if constexpr (HAS_LN) {
            const float4 xh = xhK[j][hb];
            dg[hb].x = fmaf(v[hb].x, xh.x, dg[hb].x); dg[hb].y = fmaf(v[hb].y, xh.y, dg[hb].y);
            dg[hb].z = fmaf(v[hb].z, xh.z, dg[hb].z); dg[hb].w = fmaf(v[hb].w, xh.w, dg[hb].w);
            db[hb].x += v[hb].x; db[hb].y += v[hb].y; db[hb].z += v[hb].z; db[hb].w += v[hb].w;
            v[hb].x *= gam[hb].x; v[hb].y *= gam[hb].y; v[hb].z *= gam[hb].z; v[hb].w *= gam[hb].w;
            a1 += (v[hb].x + v[hb].y) + (v[hb].z + v[hb].w);
            a2 = fmaf(v[hb].x, xh.x, fmaf(v[hb].y, xh.y, fmaf(v[hb].z, xh.z, fmaf(v[hb].w, xh.w, a2))));
          }
        }
        float s1 = 0.f, s2 = 0.f;
        if constexpr (HAS_LN) { s1 = row16_sum_r(a1) * inv_i; s2 = row16_sum_r(a2) * inv_i; }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          float4 o = v[hb];
          if constexpr (HAS_LN) {
            const float rstd = rstdK[j];
            const float4 xh = xhK[j][hb];
            o = make_float4(rstd * (v[hb].x - s1 - xh.x * s2), rstd * (v[hb].y - s1 - xh.y * s2),
                            rstd * (v[hb].z - s1 - xh.z * s2), rstd * (v[hb].w - s1 - xh.w * s2));
          }
          if (RELU_IN) {
            const uint32_t xb = xbK >> (8 * j + 4 * hb);
            o.x = (xb & 1u) ? o.x : 0.f; o.y = (xb & 2u) ? o.y : 0.f; o.z = (xb & 4u) ? o.z : 0.f; o.w = (xb & 8u) ? o.w : 0.f;
          }
          if constexpr (HAS_ACC) { o.x += acc[j][hb].x; o.y += acc[j][hb].y; o.z += acc[j][hb].z; o.w += acc[j][hb].w; }
          if constexpr (HAS_AUX) {        // + aux_g @ aux_w, added in the order q = 0..3 (the two-kernel path's order)
            const float4 g = g4K[j];
            o.x = fmaf(g.w, w4r[3][hb].x, fmaf(g.z, w4r[2][hb].x, fmaf(g.y, w4r[1][hb].x, fmaf(g.x, w4r[0][hb].x, o.x))));
            o.y = fmaf(g.w, w4r[3][hb].y, fmaf(g.z, w4r[2][hb].y, fmaf(g.y, w4r[1][hb].y, fmaf(g.x, w4r[0][hb].y, o.y))));
            o.z = fmaf(g.w, w4r[3][hb].z, fmaf(g.z, w4r[2][hb].z, fmaf(g.y, w4r[1][hb].z, fmaf(g.x, w4r[0][hb].z, o.z))));
            o.w = fmaf(g.w, w4r[3][hb].w, fmaf(g.z, w4r[2][hb].w, fmaf(g.y, w4r[1][hb].w, fmaf(g.x, w4r[0][hb].w, o.w))));
          }
#ifdef ALLSET_ABL4_NOSTORE
          if (live && o.x == 123.456f)
#else
          if (live)
#endif
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(gx + stage * R * ldgx) +
                                       hb * dhgx + (static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldgx) * 4u + cogx0)) = o;
          if (j == 1) {         // the u planes of row 1 (from the kept xhat: see S2a)
            float4 u = xhK[j][hb];
            if constexpr (HAS_LN) {
              const float4 bet = *reinterpret_cast<const float4*>(&sB[64 * hb + 4 * c]);
              u = make_float4(fmaf(u.x, gam[hb].x, bet.x), fmaf(u.y, gam[hb].y, bet.y), fmaf(u.z, gam[hb].z, bet.z), fmaf(u.w, gam[hb].w, bet.w));
            }
            if constexpr (DROP_IN) { u.x *= kpK[j][hb].x; u.y *= kpK[j][hb].y; u.z *= kpK[j][hb].z; u.w *= kpK[j][hb].w; }
            uint32_t h0, m0, l0, h1, m1, l1;
            split3_bf16(u.x, u.y, h0, m0, l0);
            split3_bf16(u.z, u.w, h1, m1, l1);
            uint8_t* img = sU + (k % 2) * IMG;
            const int wo = img_off_r(lr, 128 * hb + 8 * c);
            *reinterpret_cast<uint2*>(img + 0 * PLANE + wo) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(img + 1 * PLANE + wo) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(img + 2 * PLANE + wo) = make_uint2(l0, l1);
          }
        }
      }
    };

    request_gy(0, lane0, agS[0], amS[0]);
    request_x(0, lane0, xrS[0], stS[0], g4S[0]);
    request_gy(1, lane0, agS[1], amS[1]);
    request_x(1, lane0, xrS[1], stS[1], g4S[1]);
    S0(0, agS[0], amS[0]);
    ALLSET_ROLE_TICK();
    ALLSET_RMARK(3);
    // Two stages per trip: stage k lives in register set 0, stage k + 1 in set 1.  The trip has NO conditional half: the compiler's
    // s_waitcnt insertion takes the shortest path between a load and its use, and with "if (k + 1 < T) { second half }" inside the
    // loop that path skipped the half's 14 memory operations -- the first S2 of every trip then waited with vmcnt(3) / vmcnt(0),
    // i.e. for the operand prefetch it had just issued (an HBM round trip exposed per stage).  An odd last stage is peeled off.
    int64_t k = 0;
    for (; k + 1 < T; k += 2) {
      S0(k + 1, agS[1], amS[1]);
      S2a(k, xrS[0], stS[0], g4S[0]);
      ALLSET_RMARK(0);
      ALLSET_ROLE_TICK();
      ALLSET_RMARK(1);
      S2b(k);
      ALLSET_RMARK(2);
      ALLSET_ROLE_TICK();
      ALLSET_RMARK(3);
      if (k + 2 < T) S0(k + 2, agS[0], amS[0]);
      S2a(k + 1, xrS[1], stS[1], g4S[1]);
      ALLSET_RMARK(0);
      ALLSET_ROLE_TICK();
      ALLSET_RMARK(1);
      S2b(k + 1);
      ALLSET_RMARK(2);
      ALLSET_ROLE_TICK();
      ALLSET_RMARK(3);
    }
    if (k < T) {                            // odd stage count: the last stage, in set 0
      S2a(k, xrS[0], stS[0], g4S[0]);
      ALLSET_ROLE_TICK();
      S2b(k);
      ALLSET_ROLE_TICK();
    }
    ALLSET_ROLE_TICK();                     // (the matrix waves' last weight-gradient step)
    // ---- column sums held by the vector waves (dgamma, dbeta, bias gradient): the four row groups of a lane column fold first,
    // then the four waves through LDS in a fixed order
    {
      const int lane = lane0;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        float4 a = dg[hb], b = db[hb], g3 = gbv[hb];
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
          a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
          b.x += __shfl_xor(b.x, off); b.y += __shfl_xor(b.y, off); b.z += __shfl_xor(b.z, off); b.w += __shfl_xor(b.w, off);
          g3.x += __shfl_xor(g3.x, off); g3.y += __shfl_xor(g3.y, off); g3.z += __shfl_xor(g3.z, off); g3.w += __shfl_xor(g3.w, off);
        }
        if (lane < 16) {                        // (gu is free: the last S2 read it two ticks ago)
          *reinterpret_cast<float4*>(&sGU[wave * 3 * ID + 64 * hb + 4 * lane]) = a;
          *reinterpret_cast<float4*>(&sGU[wave * 3 * ID + ID + 64 * hb + 4 * lane]) = b;
          *reinterpret_cast<float4*>(&sGU[wave * 3 * ID + 2 * ID + 64 * hb + 4 * lane]) = g3;
        }
      }
      if constexpr (HAS_AUX) {                  // behind the four waves' 3 I column sums: per wave [4][I] + 4
        float* sa = sGU + 4 * 3 * ID + wave * (4 * ID + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            float4 a = ga4[q][hb];
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
              a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
            }
            if (lane < 16) *reinterpret_cast<float4*>(&sa[q * ID + 64 * hb + 4 * lane]) = a;
          }
        float4 b = gb4;                         // (the 16 lanes of a row group hold the same sums: lane 0 of each group counts)
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
          b.x += __shfl_xor(b.x, off); b.y += __shfl_xor(b.y, off); b.z += __shfl_xor(b.z, off); b.w += __shfl_xor(b.w, off);
        }
        if (lane == 0) *reinterpret_cast<float4*>(&sa[4 * ID]) = b;
      }
    }
  } else {
    // =================================================== matrix waves ===================================================
    const int m = wave - 4;
    const int oh = m >> 1, ih = m & 1;           // weight-gradient tile: o in [64 oh, +64), i in [64 ih, +64)
    // ---- this wave's slice of W as MFMA B fragments: column tile ct (16 columns 32 m + 16 ct + n), k-step t, plane pl;
    // lane (n = lane & 15, kg = lane >> 4) holds W[o = 32 kg + 8 t + j][column], j = 0..7 (k-order of fused_mlp.hip / fused_bwd.hip:
    // the input gradient is bit-identical to theirs)
    FragR wq[2][4][3];
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
    f32x16r gw[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k = 0; k < 16; ++k) gw[a][b][k] = 0.f;

    // ---- S1(k): backward-data for this wave's 32 output columns of the stage's 32 rows: 2 row tiles x 2 column tiles = four
    // independent accumulator chains; the A fragments of step t + 1 are requested before step t's MFMAs
    auto S1 = [&](int64_t k) {
      ALLSET_FRESH_LANE_R(lane);
      const int ri = lane & 15, kg = lane >> 4;
      const uint8_t* img = sGA + (k % 3) * IMG;
      auto load_a = [&](FragR (&f0)[3], FragR (&f1)[3], int t) {
        const int o0 = img_off_r(ri, 64 * kg + 16 * t), o1 = img_off_r(16 + ri, 64 * kg + 16 * t);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          f0[pl].u = *reinterpret_cast<const uint4*>(img + pl * PLANE + o0);
          f1[pl].u = *reinterpret_cast<const uint4*>(img + pl * PLANE + o1);
        }
      };
      FragR fa0[2][3], fa1[2][3];
      f32x4r acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4r{0.f, 0.f, 0.f, 0.f};
      load_a(fa0[0], fa1[0], 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) load_a(fa0[(t + 1) & 1], fa1[(t + 1) & 1], t + 1);
        const FragR (&a0)[3] = fa0[t & 1];
        const FragR (&a1)[3] = fa1[t & 1];
#ifndef ALLSET_ABL4_NOMFMA
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};     // l.h, h.l, m.m, m.h, h.m, h.h
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[PA_[pr]].v, wq[0][t][PB_[pr]].v, acc[0][0], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[PA_[pr]].v, wq[0][t][PB_[pr]].v, acc[1][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[PA_[pr]].v, wq[1][t][PB_[pr]].v, acc[0][1], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[PA_[pr]].v, wq[1][t][PB_[pr]].v, acc[1][1], 0, 0, 0);
        }
#else
        acc[0][0][0] += __builtin_bit_cast(float, a0[0].u.x ^ a0[1].u.y ^ a0[2].u.z); acc[1][0][0] += __builtin_bit_cast(float, a1[0].u.x ^ a1[1].u.y ^ a1[2].u.z);
#endif
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
    // ---- S3(k): weight gradient, this wave's 64 x 64 tile of gW; K = the stage's 32 rows in two steps of 16; A = ga^T, B = u
    auto S3 = [&](int64_t k) {
      ALLSET_FRESH_LANE_R(lane_w);
      const uint8_t* ia = sGA + (k % 3) * IMG;
      const uint8_t* iu = sU + (k % 2) * IMG;
      const int q4 = lane_w >> 4, tr_r = (lane_w & 15) >> 2, tr_row = 8 * (q4 >> 1) + tr_r, tr_in = 32 * (q4 & 1) + 8 * (lane_w & 3);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16x8r wa[2][3], wb[2][3];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          const int a_lo = img_off_r(16 * kb + tr_row, 64 * (2 * oh + tl) + tr_in), a_hi = img_off_r(16 * kb + tr_row + 4, 64 * (2 * oh + tl) + tr_in);
          const int b_lo = img_off_r(16 * kb + tr_row, 64 * (2 * ih + tl) + tr_in), b_hi = img_off_r(16 * kb + tr_row + 4, 64 * (2 * ih + tl) + tr_in);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            wa[tl][pl] = tr_frag2_r(ia + pl * PLANE + a_lo, ia + pl * PLANE + a_hi);
            wb[tl][pl] = tr_frag2_r(iu + pl * PLANE + b_lo, iu + pl * PLANE + b_hi);
          }
        }
#ifndef ALLSET_ABL4_NOMFMA
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          gw[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][PA_[pr]], wb[0][PB_[pr]], gw[0][0], 0, 0, 0);
          gw[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][PA_[pr]], wb[0][PB_[pr]], gw[1][0], 0, 0, 0);
          gw[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][PA_[pr]], wb[1][PB_[pr]], gw[0][1], 0, 0, 0);
          gw[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][PA_[pr]], wb[1][PB_[pr]], gw[1][1], 0, 0, 0);
        }
#else
        { FragR f; f.v = wa[0][0]; FragR g2; g2.v = wb[1][1]; FragR g3; g3.v = wa[1][2]; FragR g4; g4.v = wb[0][2];
          gw[0][0][0] += __builtin_bit_cast(float, f.u.x ^ g2.u.y ^ g3.u.z ^ g4.u.w); }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    ALLSET_ROLE_TICK();
    ALLSET_RMARK(3);
    for (int64_t k = 0; k < T; ++k) {
      S1(k);
      ALLSET_RMARK(0);
      ALLSET_ROLE_TICK();
      ALLSET_RMARK(1);
      if (k >= 1) S3(k - 1);
      ALLSET_RMARK(2);
      ALLSET_ROLE_TICK();
      ALLSET_RMARK(3);
    }
    S3(T - 1);
    ALLSET_ROLE_TICK();
    // ---- the workgroup's gW partial: each matrix wave its 64 x 64 tile
    {
      const int lane = lane0;
      float* pw = part_w + static_cast<int64_t>(blockIdx.x) * pstride_w;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int o = (2 * oh + a) * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
            pw[o * ID + (2 * ih + b) * 32 + (lane & 31)] = gw[a][b][k];
          }
    }
  }
  __syncthreads();
#ifdef ALLSET_ABL4_TIMING
  // vector wave 0: [0] S0, [1] wait, [2] S2, [3] wait; matrix wave 4: [4] S1, [5] wait, [6] S3, [7] wait  (cycles, all stages)
  if (blockIdx.x == 0 && (tid == 0 || tid == 256)) {
    float* dbg = part_w + (tid == 0 ? 0 : 4);         // over this workgroup's own gW entries (stored before the barrier above)
    for (int k = 0; k < 4; ++k) dbg[k] = static_cast<float>(tph[k]);
  }
#endif
  if (tid < 3 * ID) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) s += sGU[v * 3 * ID + tid];
    const int64_t slice = blockIdx.x;
    if (tid < 2 * ID) { if constexpr (HAS_LN) part_ln[slice * pstride_ln + tid] = s; }
    else if (part_b != nullptr) part_b[slice * pstride_b + (tid - 2 * ID)] = s;
  }
  if constexpr (HAS_AUX) {
    for (int idx = tid; idx < 4 * ID + 4; idx += kRBlock) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) s += sGU[4 * 3 * ID + v * (4 * ID + 4) + idx];
      part_ln[static_cast<int64_t>(blockIdx.x) * pstride_ln + idx] = s;
    }
  }
}

}  // namespace allset

using namespace allset;

// 1 = the split-role kernel takes this call (O = I = 128): a pure function of the widths
int fused_linear_bwd_roles_supported(int64_t O, int64_t I, int has_acc) {
  (void)has_acc;                         // (acc_in is built for the plain Linear, the one combination the one-wave kernel has it for too)
  return (O == 128 && I == 128) ? 1 : 0;
}

unsigned fused_linear_bwd_roles_grid(int64_t n) {
  const int64_t blocks = (n + kRRows - 1) / kRRows;
  return static_cast<unsigned>(blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks));      // one persistent workgroup per CU
}

// Called by allset_fused_linear_bwd_all (fused_bwd.hip) after its argument checks; ONE partial slice per workgroup.
int launch_fused_linear_bwd_roles(unsigned grid, hipStream_t st, bool ln, bool drop, bool relu, bool hm, const float* gy,
                                  int64_t ldg, const uint32_t* mask, float p_out, const float* W, const float* x, int64_t ldx,
                                  const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                                  float* gx, int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n,
                                  const uint64_t* seed_base, int64_t psw, int64_t psb, int64_t psl, const float* acc_in,
                                  int64_t ldacc, const float* aux_g, const float* aux_w, int64_t gcb, int64_t xcb, int64_t gxcb,
                                  float ln_inv) {
#define ALLSET_ROLES_KA(LN, DI, RI, HM, HA, AX)                                                                                \
  fused_linear_bwd_roles_kernel<LN, DI, RI, HM, HA, AX><<<grid, kRBlock, 0, st>>>(gy, ldg, mask, p_out, W, x, ldx, stats, gamma, \
                                                                             beta, p_in, seed_in, gx, ldgx, part_ln, part_w,    \
                                                                             part_b, n, seed_base, psw, psb, psl, acc_in, ldacc, \
                                                                             aux_g, aux_w, gcb, xcb, gxcb, ln_inv)
#define ALLSET_ROLES_K(LN, DI, RI, HM, HA) ALLSET_ROLES_KA(LN, DI, RI, HM, HA, false)
  if (aux_g != nullptr) { ALLSET_ROLES_KA(false, false, false, false, false, true); return 0; }   // (plain Linear + four aux columns)
  if (acc_in != nullptr) { ALLSET_ROLES_K(false, false, false, false, true); return 0; }     // (bwd_all_combo: plain Linear only)
#define ALLSET_ROLES_M(LN, DI, RI) do { if (hm) ALLSET_ROLES_K(LN, DI, RI, true, false); else ALLSET_ROLES_K(LN, DI, RI, false, false); } while (0)
  if (!relu) { if (ln) ALLSET_ROLES_M(true, false, false); else ALLSET_ROLES_M(false, false, false); }
  else if (ln) { if (drop) ALLSET_ROLES_M(true, true, true); else ALLSET_ROLES_M(true, false, true); }
  else { if (drop) ALLSET_ROLES_M(false, true, true); else ALLSET_ROLES_M(false, false, true); }
#undef ALLSET_ROLES_M
#undef ALLSET_ROLES_K
#undef ALLSET_ROLES_KA
  return 0;
}
