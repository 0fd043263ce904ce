// The one-pass backward of the fused Linear (math and operand images: fused_bwd.hip) with ONE weight-gradient accumulator per
// workgroup, W in registers and two waves per SIMD -- the organisation DESIGN.md section 8 item 0 sketched in round 2.
//
// fused_bwd.hip: every wave owns a full gW (256 registers) -> one wave per SIMD, nothing covers its stalls.
// fused_bwd2.hip: two waves share a 16-row chunk, 128 accumulator registers each -> two waves per SIMD, but the other 128
//   registers cannot hold an operand prefetch AND the transposed ga fragments, and the W planes fill LDS: measured no faster.
// Here all 8 waves of the CU work on the same STAGE of 64 rows:
//   * backward-data is cut by OUTPUT COLUMNS: wave w computes gu[64 rows, 16 w .. 16 w + 16).  Its slice of W -- three bf16 planes
//     of W[:, 16 columns] -- is 48 registers and stays there for the whole kernel: no W in LDS at all;
//   * the weight gradient is cut into eight 64 x 32 tiles of ONE gW [128 x 128] per workgroup: 32 accumulator registers per wave
//     instead of 128 / 256, one partial gW per workgroup instead of four (a quarter of the partial-sum traffic);
//   * the vector work (mask, bf16 splits, LayerNorm backward, dropout hash, recomputation of u) is cut by ROWS: wave w owns rows
//     8 w .. 8 w + 7 of the stage, complete rows, so the LayerNorm row sums never leave the wave and every global load / store
//     is a whole 512-byte row;
//   * what the waves exchange goes through three LDS images: ga planes [3][64][256 B] and u planes (same shape), both row-major
//     and read back as 16-byte fragments (backward-data) or by ds_read_b64_tr_b16 (weight gradient), and gu [64][128] fp32;
//   * ~140 registers per wave are left for the NEXT stage's gy rows, mask words, x rows and statistics (requested a whole stage
//     ahead: ~5 us of latency cover) and for double-buffered operand fragments.
// Four workgroup barriers per 64 rows (LDS-only waits in front of them: the global prefetch stays in flight):
//     B0 the previous stage's weight gradient is done with the ga image   -> ga planes of this stage
//     B1 ga complete -> backward-data MFMAs -> gu tile to LDS              B2 gu complete -> LayerNorm backward, gx, u planes
//     B3 u complete  -> weight-gradient MFMAs
// LDS: 48 + 48 + 33 KB + gamma / beta.
#include <stdlib.h>

#include "common.h"

namespace allset {

using bf16x8s = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4s = __attribute__((ext_vector_type(4))) float;
using f32x16s = __attribute__((ext_vector_type(16))) float;
typedef short v4ss_t __attribute__((ext_vector_type(4)));
typedef __bf16 v2bfs_t __attribute__((ext_vector_type(2)));
union FragS { uint4 u; bf16x8s v; struct { v4ss_t lo, hi; } t; };
constexpr int kSBlock = 512;
constexpr int kSRows = 64;                     // rows per stage

template <int CTRL>
__device__ __forceinline__ float dpp_fs(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row32_sum(float v) {       // sum over the 32 lanes of a half-wave, result in every lane of it
  v += dpp_fs<0xB1>(v);
  v += dpp_fs<0x4E>(v);
  v += dpp_fs<0x141>(v);
  v += dpp_fs<0x140>(v);
  v += __shfl_xor(v, 16);
  return v;
}
__device__ __forceinline__ bf16x8s tr_frag2_s(const uint8_t* lo, const uint8_t* hi) {
  FragS f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4ss_t*)(lo));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4ss_t*)(hi));
  return f.v;
}
__device__ __forceinline__ bf16x8s tr_frag_s(const uint8_t* p, int half_stride) {
  FragS f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4ss_t*)(p));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4ss_t*)(p + half_stride));
  return f.v;
}
// byte offset of (row, column byte) in a [rows][256 B] bf16 plane.  Rows are 256 B = a whole number of bank rounds apart, so the
// row must be folded into the column: the 64-byte chunk index is XORed with row & 3 (the four rows a transpose-read touches land
// in four bank quarters, as in fused_bwd.hip) AND the 16-byte piece inside the chunk with (row >> 2) & 3 -- sixteen different
// rows reading the same (chunk, piece) then hit sixteen different bank quads: the backward-data A fragments (ds_read_b128, lane =
// row) are conflict-free; with the chunk swizzle alone they were 4-way conflicted and made that phase LDS-bound (6.2 k cycles per
// stage where its MFMAs need 3.1 k; tools/bwd_stage_ablation.py phase timing).
__device__ __forceinline__ int img_off_s(int row, int colbyte) {
  return row * 256 + ((((colbyte >> 6) ^ row) & 3) << 6) + (((((colbyte >> 4) & 3) ^ (row >> 2)) & 3) << 4) + (colbyte & 15);
}
__device__ __forceinline__ uint32_t hash_mix_s(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; return x; }
#ifdef ALLSET_ABL3_NOBAR            // ablation builds only: timing without the barriers, results wrong
#define ALLSET_STAGE_BARRIER() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define ALLSET_STAGE_BARRIER() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#define ALLSET_FRESH_LANE_S(name) \
  int name = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); __asm__ volatile("" : "+v"(name))

template <bool HAS_LN, bool DROP_IN, bool RELU_IN, bool HAS_MASK>
__global__ __launch_bounds__(kSBlock, 2) void fused_linear_bwd_stage_kernel(
    const float* __restrict__ gy, int64_t ldg, const uint32_t* __restrict__ mask, float p_out, const float* __restrict__ W,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float p_in, uint64_t seed_in, float* gx, int64_t ldgx,
    float* __restrict__ part_ln, float* __restrict__ part_w, float* __restrict__ part_b, int64_t n,
    const uint64_t* __restrict__ seed_base, int64_t pstride_w, int64_t pstride_b, int64_t pstride_ln) {
  constexpr int OD = 128, ID = 128;
  constexpr int R = kSRows;
  constexpr int PLANE = R * 256;                 // bytes per bf16 plane of an image
  constexpr int SPG = 132;                       // pitch (floats) of the gu tile: 16-byte rows, 2-way conflicts at worst
  __shared__ __attribute__((aligned(16))) uint8_t sGA[3 * PLANE];
  __shared__ __attribute__((aligned(16))) uint8_t sU[3 * PLANE];
  __shared__ __attribute__((aligned(16))) float sGU[R * SPG];
  __shared__ __attribute__((aligned(16))) float sG[ID];
  __shared__ __attribute__((aligned(16))) float sB[ID];
  seed_in = resolve_seed(seed_base, seed_in);
  const int tid = threadIdx.x;
  if (tid < ID) { sG[tid] = HAS_LN ? gamma[tid] : 1.f; sB[tid] = HAS_LN ? beta[tid] : 0.f; }
  const int lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this wave's slice of W as MFMA B fragments, three bf16 planes: lane (n = lane & 15, kg = lane >> 4), k-step t holds
  // W[o = 32 kg + 8 t + j][i = 16 wave + n], j = 0..7 (the A fragment of that step is ga[row][32 kg + 8 t .. +7]: the k-order of
  // fused_mlp.hip / fused_bwd.hip, so the input gradient is bit-identical to theirs)
  FragS wq[4][3];
  {
    const int nn = lane0 & 15, kg = lane0 >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t ph[4], pm[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = 32 * kg + 8 * t + 2 * j;
        split3_bf16(W[o * ID + 16 * wave + nn], W[(o + 1) * ID + 16 * wave + nn], ph[j], pm[j], pl[j]);
      }
      wq[t][0].u = make_uint4(ph[0], ph[1], ph[2], ph[3]);
      wq[t][1].u = make_uint4(pm[0], pm[1], pm[2], pm[3]);
      wq[t][2].u = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }
  }
  __syncthreads();

  const float inv_i = 1.f / static_cast<float>(ID);
  const float keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  const float keep_in = DROP_IN ? 1.f / (1.f - p_in) : 1.f;
  const uint32_t thr_in = drop_threshold(p_in);
  const uint32_t seed_lo = static_cast<uint32_t>(seed_in);
  const int64_t n_stages = (n + R - 1) / R;
  const int oh = wave >> 2, iq = wave & 3;       // this wave's weight-gradient tile: o in [64 oh, +64), i in [32 iq, +32)

  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f);     // bias gradient of columns 4 c .. 4 c + 3 over this wave's rows
  f32x16s gw[2];
#pragma unroll
  for (int k = 0; k < 16; ++k) { gw[0][k] = 0.f; gw[1][k] = 0.f; }

  // Row-major layout of the vector phases: lane (c = lane & 31, rr = lane >> 5) owns rows 8 wave + rr + 2 j (j = 0..3) of the
  // stage, columns 4 c .. 4 c + 3: a load / store instruction moves two complete 512-byte rows.
  float4 ag[4];            // gy rows of the NEXT stage to process (requested a stage ahead)
  uint32_t am[4];          // their activation-mask words
  float4 xr[4];            // x rows, same schedule
  float2 st[4];
  auto rows_left = [&](int64_t stage) -> int {         // valid rows of the stage, clamped to 0..64
    const int64_t left = n - stage * R;
    return left >= R ? R : (left > 0 ? static_cast<int>(left) : 0);
  };
  auto request_gy = [&](int64_t stage, int lane) {
    const int c = lane & 31, rr = lane >> 5;
    const int nr = rows_left(stage);
    const int64_t s0 = nr > 0 ? stage : n_stages - 1;   // past the end: re-read the last stage (never consumed)
    const int nrc = max(rows_left(s0), 1);
    const char* base = reinterpret_cast<const char*>(gy + s0 * R * ldg);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lr = min(8 * wave + rr + 2 * j, nrc - 1);
      ag[j] = *reinterpret_cast<const float4*>(base + static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldg) * 4u + 16 * c);
      if constexpr (HAS_MASK)     // "mask layout" (include/allset_hip.h): block (row / 16, column / 64), dword (row % 16, 32-column group)
        am[j] = (mask + ((s0 * (R / 16) + (lr >> 4)) * (OD / 64) + (c >> 4)) * 32)[((lr & 15) >> 2) * 8 + (lr & 3) * 2 + ((c >> 3) & 1)];
    }
  };
  auto request_x = [&](int64_t stage, int lane) {
    const int c = lane & 31, rr = lane >> 5;
    const int nr = rows_left(stage);
    const int64_t s0 = nr > 0 ? stage : n_stages - 1;
    const int nrc = max(rows_left(s0), 1);
    const char* xb = reinterpret_cast<const char*>(x + s0 * R * ldx);
    const char* sb = reinterpret_cast<const char*>(stats + s0 * R * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lr = min(8 * wave + rr + 2 * j, nrc - 1);
      xr[j] = *reinterpret_cast<const float4*>(xb + static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldx) * 4u + 16 * c);
      if constexpr (HAS_LN) st[j] = *reinterpret_cast<const float2*>(sb + lr * 8);
    }
  };

#ifdef ALLSET_ABL3_TIMING          // diagnostic builds only: cycles per phase of wave 0 of workgroup 0 (tools/bwd_stage_ablation.py)
  uint64_t tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define ALLSET_TMARK(k) do { const uint64_t tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } while (0)
#else
#define ALLSET_TMARK(k) do {} while (0)
#endif
  int64_t stage = blockIdx.x;
  request_gy(stage, lane0);
  request_x(stage, lane0);
  for (; stage < n_stages; stage += gridDim.x) {
    ALLSET_FRESH_LANE_S(lane);
    const int c = lane & 31, rr = lane >> 5;
    const int nrows = rows_left(stage);
    ALLSET_TMARK(7);
    ALLSET_STAGE_BARRIER();                                                                  // B0
    ALLSET_TMARK(0);
    // ---- S0: ga = gy under the forward's epilogue mask, three bf16 planes into the image (this wave's 8 rows)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lr = 8 * wave + rr + 2 * j;
      float4 v = ag[j];
      const bool valid = lr < nrows;
      if constexpr (HAS_MASK) {
        const uint32_t bits = valid ? (am[j] >> (c & 7)) : 0u;          // bit 8 q + (c % 8) for column 4 c + q
        v.x = (bits & 0x1u) ? v.x * keep_out : 0.f; v.y = (bits & 0x100u) ? v.y * keep_out : 0.f;
        v.z = (bits & 0x10000u) ? v.z * keep_out : 0.f; v.w = (bits & 0x1000000u) ? v.w * keep_out : 0.f;
      } else if (!valid) {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      gbv.x += v.x; gbv.y += v.y; gbv.z += v.z; gbv.w += v.w;          // bias gradient: column sums of ga
      uint32_t h0, m0, l0, h1, m1, l1;
      split3_bf16(v.x, v.y, h0, m0, l0);
      split3_bf16(v.z, v.w, h1, m1, l1);
      const int wo = img_off_s(lr, 8 * c);
      *reinterpret_cast<uint2*>(sGA + 0 * PLANE + wo) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(sGA + 1 * PLANE + wo) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(sGA + 2 * PLANE + wo) = make_uint2(l0, l1);
    }
    __builtin_amdgcn_sched_barrier(0);
    request_gy(stage + gridDim.x, lane);                 // the next stage's gy: a whole stage of latency cover
    __builtin_amdgcn_sched_barrier(0);
    ALLSET_TMARK(1);
    ALLSET_STAGE_BARRIER();                                                                  // B1
    ALLSET_TMARK(2);
    // ---- S1: backward-data, this wave's 16 output columns of all 64 rows (bf16x6: six of nine plane products); two row tiles
    // at a time = two independent accumulator chains
    {
      const int ri = lane & 15, kg = lane >> 4;
      // fragment addresses: (row tile pair rp, k-step t) -> rows 32 rp + ri and + 16, column byte 64 kg + 16 t; the fragments of
      // step t + 1 (or of the next row-tile pair) are requested before step t's MFMAs: with both waves of the SIMD in this phase
      // nobody else covers the LDS latency
      auto load_a = [&](FragS (&f0)[3], FragS (&f1)[3], int rp, int t) {
        const int row0 = 32 * rp + ri, row1 = row0 + 16;
        const int o0 = img_off_s(row0, 64 * kg + 16 * t), o1 = img_off_s(row1, 64 * kg + 16 * t);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          f0[pl].u = *reinterpret_cast<const uint4*>(sGA + pl * PLANE + o0);
          f1[pl].u = *reinterpret_cast<const uint4*>(sGA + pl * PLANE + o1);
        }
      };
      FragS fa0[2][3], fa1[2][3];
      load_a(fa0[0], fa1[0], 0, 0);
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        f32x4s acc0 = f32x4s{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int cur = (rp * 4 + t) & 1;
          if (rp * 4 + t + 1 < 8) load_a(fa0[cur ^ 1], fa1[cur ^ 1], (rp * 4 + t + 1) >> 2, (rp * 4 + t + 1) & 3);
          const FragS (&a0)[3] = fa0[cur];
          const FragS (&a1)[3] = fa1[cur];
#ifndef ALLSET_ABL3_NOMFMA
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[2].v, wq[t][0].v, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[2].v, wq[t][0].v, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[0].v, wq[t][2].v, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0].v, wq[t][2].v, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[1].v, wq[t][1].v, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[1].v, wq[t][1].v, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[1].v, wq[t][0].v, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[1].v, wq[t][0].v, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[0].v, wq[t][1].v, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0].v, wq[t][1].v, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[0].v, wq[t][0].v, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0].v, wq[t][0].v, acc1, 0, 0, 0);
#else
          acc0[0] += __builtin_bit_cast(float, a0[0].u.x ^ a0[1].u.y ^ a0[2].u.z); acc1[0] += __builtin_bit_cast(float, a1[0].u.x ^ a1[1].u.y ^ a1[2].u.z);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        // acc[r] = gu[row 16 (2 rp + k) + 4 kg + r][column 16 wave + ri]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sGU[(32 * rp + 4 * kg + r) * SPG + 16 * wave + ri] = acc0[r];
          sGU[(32 * rp + 16 + 4 * kg + r) * SPG + 16 * wave + ri] = acc1[r];
        }
      }
    }
    ALLSET_TMARK(3);
    ALLSET_STAGE_BARRIER();                                                                  // B2
    ALLSET_TMARK(4);
    // ---- S2: dropout-in mask, LayerNorm backward, relu-in mask -> gx;  u = dropout_in(LN(relu_in(x))) -> bf16 planes
    {
      const float4 gam = *reinterpret_cast<const float4*>(&sG[4 * c]);
      const float4 bet = *reinterpret_cast<const float4*>(&sB[4 * c]);
      const uint64_t stage_pair = static_cast<uint64_t>(stage) * (R * ID / 2);
      const uint32_t stage_pair_lo = static_cast<uint32_t>(stage_pair);
      const uint32_t hi_term = __umul24(static_cast<uint32_t>(stage_pair >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed_in >> 32);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lr = 8 * wave + rr + 2 * j;
        const bool live = lr < nrows;
        float4 v = *reinterpret_cast<const float4*>(&sGU[lr * SPG + 4 * c]);
        float4 kp = make_float4(1.f, 1.f, 1.f, 1.f);
        if constexpr (DROP_IN) {
          // pair index of (row, column 4 c) = stage * 4096 + (lr * 128 + 4 c) / 2: the lane's part is < 4096 -> an OR (common.h pair_hash)
          const uint32_t lo = stage_pair_lo | static_cast<uint32_t>((lr * ID + 4 * c) >> 1);
          const uint32_t h0 = hash_mix_s((lo ^ seed_lo) * 0x9E3779B1U + hi_term);
          const uint32_t h1 = hash_mix_s(((lo + 1u) ^ seed_lo) * 0x9E3779B1U + hi_term);
          kp.x = (h0 & 0xffffu) >= thr_in ? keep_in : 0.f; kp.y = (h0 >> 16) >= thr_in ? keep_in : 0.f;
          kp.z = (h1 & 0xffffu) >= thr_in ? keep_in : 0.f; kp.w = (h1 >> 16) >= thr_in ? keep_in : 0.f;
          v.x *= kp.x; v.y *= kp.y; v.z *= kp.z; v.w *= kp.w;
        }
        const float4 xraw = xr[j];
        float4 t = xraw;
        if (RELU_IN) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
        float4 o = v;
        if constexpr (HAS_LN) {
          const float mean = st[j].x, rstd = st[j].y;
          float4 xh = make_float4((t.x - mean) * rstd, (t.y - mean) * rstd, (t.z - mean) * rstd, (t.w - mean) * rstd);
          if (!live) xh = make_float4(0.f, 0.f, 0.f, 0.f);
          dg.x = fmaf(v.x, xh.x, dg.x); dg.y = fmaf(v.y, xh.y, dg.y); dg.z = fmaf(v.z, xh.z, dg.z); dg.w = fmaf(v.w, xh.w, dg.w);
          db.x += v.x; db.y += v.y; db.z += v.z; db.w += v.w;
          v.x *= gam.x; v.y *= gam.y; v.z *= gam.z; v.w *= gam.w;
          const float s1 = row32_sum((v.x + v.y) + (v.z + v.w)) * inv_i;
          const float s2 = row32_sum(fmaf(v.x, xh.x, fmaf(v.y, xh.y, fmaf(v.z, xh.z, v.w * xh.w)))) * inv_i;
          o = make_float4(rstd * (v.x - s1 - xh.x * s2), rstd * (v.y - s1 - xh.y * s2), rstd * (v.z - s1 - xh.z * s2),
                          rstd * (v.w - s1 - xh.w * s2));
          t = make_float4(fmaf(xh.x, gam.x, bet.x), fmaf(xh.y, gam.y, bet.y), fmaf(xh.z, gam.z, bet.z), fmaf(xh.w, gam.w, bet.w));
        }
        if (RELU_IN) {
          o.x = xraw.x > 0.f ? o.x : 0.f; o.y = xraw.y > 0.f ? o.y : 0.f; o.z = xraw.z > 0.f ? o.z : 0.f; o.w = xraw.w > 0.f ? o.w : 0.f;
        }
#ifdef ALLSET_ABL3_NOSTORE
        if (live && o.x == 123.456f)
#else
        if (live)
#endif
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(gx + stage * R * ldgx) +
                                     static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldgx) * 4u + 16 * c) = o;
        if constexpr (DROP_IN) { t.x *= kp.x; t.y *= kp.y; t.z *= kp.z; t.w *= kp.w; }
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_bf16(t.x, t.y, h0, m0, l0);
        split3_bf16(t.z, t.w, h1, m1, l1);
        const int wo = img_off_s(lr, 8 * c);
        *reinterpret_cast<uint2*>(sU + 0 * PLANE + wo) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(sU + 1 * PLANE + wo) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(sU + 2 * PLANE + wo) = make_uint2(l0, l1);
        __builtin_amdgcn_sched_barrier(0);        // one row at a time
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    request_x(stage + gridDim.x, lane);                  // the next stage's x rows and statistics
    __builtin_amdgcn_sched_barrier(0);
    ALLSET_TMARK(5);
    ALLSET_STAGE_BARRIER();                                                                  // B3
    ALLSET_TMARK(6);
    // ---- S3: weight gradient, this wave's 64 x 32 tile of gW: K = the stage's 64 rows in four steps of 16; A = ga^T, B = u,
    // both by transpose-reads of the row-major images
    {
      ALLSET_FRESH_LANE_S(lane_w);
      const int q4 = lane_w >> 4, tr_r = (lane_w & 15) >> 2, tr_row = 8 * (q4 >> 1) + tr_r, tr_in = 32 * (q4 & 1) + 8 * (lane_w & 3);
      // (rows 16 kb + tr_row and + 4: (row >> 2) & 3 = 2 (q4 >> 1) and + 1 -- the piece swizzle differs between the two halves
      // of a fragment, so both addresses are spelled out)
      const int a_lo0 = img_off_s(tr_row, 64 * (2 * oh) + tr_in), a_hi0 = img_off_s(tr_row + 4, 64 * (2 * oh) + tr_in);
      const int a_lo1 = img_off_s(tr_row, 64 * (2 * oh + 1) + tr_in), a_hi1 = img_off_s(tr_row + 4, 64 * (2 * oh + 1) + tr_in);
      const int b_lo = img_off_s(tr_row, 64 * iq + tr_in), b_hi = img_off_s(tr_row + 4, 64 * iq + tr_in);
      auto load_w = [&](bf16x8s (&w0)[3], bf16x8s (&w1)[3], bf16x8s (&wbb)[3], int kb) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          w0[pl] = tr_frag2_s(sGA + pl * PLANE + kb * 16 * 256 + a_lo0, sGA + pl * PLANE + kb * 16 * 256 + a_hi0);
          w1[pl] = tr_frag2_s(sGA + pl * PLANE + kb * 16 * 256 + a_lo1, sGA + pl * PLANE + kb * 16 * 256 + a_hi1);
          wbb[pl] = tr_frag2_s(sU + pl * PLANE + kb * 16 * 256 + b_lo, sU + pl * PLANE + kb * 16 * 256 + b_hi);
        }
      };
      bf16x8s wa0[2][3], wa1[2][3], wb[2][3];
      load_w(wa0[0], wa1[0], wb[0], 0);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (kb + 1 < 4) load_w(wa0[(kb + 1) & 1], wa1[(kb + 1) & 1], wb[(kb + 1) & 1], kb + 1);   // the next step's fragments first
#ifndef ALLSET_ABL3_NOMFMA
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          gw[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa0[kb & 1][PA_[pr]], wb[kb & 1][PB_[pr]], gw[0], 0, 0, 0);
          gw[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1[kb & 1][PA_[pr]], wb[kb & 1][PB_[pr]], gw[1], 0, 0, 0);
        }
#else
        { FragS f; f.v = wa0[kb & 1][0]; FragS g2; g2.v = wb[kb & 1][1]; FragS g3; g3.v = wa1[kb & 1][2]; gw[0][0] += __builtin_bit_cast(float, f.u.x ^ g2.u.y ^ g3.u.z); }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- the workgroup's partials: gW [O][I] (each wave its 64 x 32 tile), gb [O], LayerNorm (dgamma, dbeta) [2][I]
  const int64_t slice = blockIdx.x;
  const int lane = lane0;
  float* pw = part_w + slice * pstride_w;
#pragma unroll
  for (int ot2 = 0; ot2 < 2; ++ot2)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int o = (2 * oh + ot2) * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
      pw[o * ID + iq * 32 + (lane & 31)] = gw[ot2][k];
    }
  {     // column sums held per wave (dgamma, dbeta, bias gradient): rows rr = 0 / 1 of a lane column fold first, then the 8 waves
        // through LDS in a fixed order
    float4 a = dg, b = db, g3 = gbv;
    a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
    b.x += __shfl_xor(b.x, 32); b.y += __shfl_xor(b.y, 32); b.z += __shfl_xor(b.z, 32); b.w += __shfl_xor(b.w, 32);
    g3.x += __shfl_xor(g3.x, 32); g3.y += __shfl_xor(g3.y, 32); g3.z += __shfl_xor(g3.z, 32); g3.w += __shfl_xor(g3.w, 32);
    __syncthreads();                                  // (sGU is free: every wave left the stage loop)
    if (lane < 32) {
      *reinterpret_cast<float4*>(&sGU[wave * 3 * ID + 4 * lane]) = a;
      *reinterpret_cast<float4*>(&sGU[wave * 3 * ID + ID + 4 * lane]) = b;
      *reinterpret_cast<float4*>(&sGU[wave * 3 * ID + 2 * ID + 4 * lane]) = g3;
    }
    __syncthreads();
    if (tid < 3 * ID) {
      float s = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) s += sGU[w8 * 3 * ID + tid];
      if (tid < 2 * ID) { if constexpr (HAS_LN) part_ln[slice * pstride_ln + tid] = s; }
      else if (part_b != nullptr) part_b[slice * pstride_b + (tid - 2 * ID)] = s;
    }
  }
#ifdef ALLSET_ABL3_TIMING
  // [0] wait B0, [1] S0, [2] wait B1, [3] S1, [4] wait B2, [5] S2, [6] wait B3, [7] S3 (+ loop overhead)
  if (blockIdx.x == 0 && tid == 0) {          // (over wave 0's own gW entries o = 0, i = 0..7: same wave, later stores)
    for (int k = 0; k < 8; ++k) part_w[k] = static_cast<float>(tph[k]);
  }
#endif
}

}  // namespace allset

using namespace allset;

// 1 = the stage kernel takes this call (O = I = 128, no acc_in; bf16x6 mode); ALLSET_BWD_STAGE=0 keeps the one-wave kernel
int fused_linear_bwd_stage_supported(int64_t O, int64_t I, int has_acc) {
  const char* e = getenv("ALLSET_BWD_STAGE");
  if (e && e[0] == '0') return 0;
  return (dense_mfma_x6() && O == 128 && I == 128 && !has_acc) ? 1 : 0;
}

unsigned fused_linear_bwd_stage_grid(int64_t n) {
  const int64_t blocks = (n + kSRows - 1) / kSRows;
  return static_cast<unsigned>(blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks));      // one persistent workgroup per CU
}

// Called by allset_fused_linear_bwd_all (fused_bwd.hip) after its argument checks; ONE partial slice per workgroup.
int launch_fused_linear_bwd_stage(unsigned grid, hipStream_t st, bool ln, bool drop, bool relu, bool hm, const float* gy,
                                  int64_t ldg, const uint32_t* mask, float p_out, const float* W, const float* x, int64_t ldx,
                                  const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                                  float* gx, int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n,
                                  const uint64_t* seed_base, int64_t psw, int64_t psb, int64_t psl) {
#define ALLSET_STAGE_K(LN, DI, RI, HM)                                                                                         \
  fused_linear_bwd_stage_kernel<LN, DI, RI, HM><<<grid, kSBlock, 0, st>>>(gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta,   \
                                                                         p_in, seed_in, gx, ldgx, part_ln, part_w, part_b, n,   \
                                                                         seed_base, psw, psb, psl)
#define ALLSET_STAGE_M(LN, DI, RI) do { if (hm) ALLSET_STAGE_K(LN, DI, RI, true); else ALLSET_STAGE_K(LN, DI, RI, false); } while (0)
  if (!relu) { if (ln) ALLSET_STAGE_M(true, false, false); else ALLSET_STAGE_M(false, false, false); }
  else if (ln) { if (drop) ALLSET_STAGE_M(true, true, true); else ALLSET_STAGE_M(true, false, true); }
  else { if (drop) ALLSET_STAGE_M(false, true, true); else ALLSET_STAGE_M(false, false, true); }
#undef ALLSET_STAGE_M
#undef ALLSET_STAGE_K
  return 0;
}
