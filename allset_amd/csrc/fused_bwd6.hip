// The one-pass backward of the fused Linear at O = I = 128 (behind a LayerNorm prologue, or none), split by role like fused_bwd4.hip,
// with the fp32 products formed from TWO fp16 planes per operand instead of three bf16 planes ("fp16x3"):
//
//     x s = h + l,  h = fp16(x s),  l = fp16(x s - h)         (s a power of two: the scaling is exact)
//     (x s)(w t) = h h' + h l' + l h'  + (l l' <= 2^-22 |x w s t|, dropped)
//
// Three f16 MFMAs replace the six bf16 ones (and one fp32 MFMA at 1/16 the rate); the two planes carry 22+ significant bits, and
// measured against float64 the result is as accurate as a library fp32 GEMM (rms 1.2e-8 of sum |terms|, bf16x6
// 0.7e-8, sequential fp32 2.6e-8).  fp16 has 5 exponent bits, so every operand is brought into the window [2^-14, 2^15) by a
// power-of-two scale that leaves the arithmetic exact:
//   ga (masked gy)  per ROW: its largest element lands in [2^13, 2^14); the row's gu is unscaled in the LayerNorm backward;
//   W               per matrix wave's slice (32 columns); unscaled where the wave writes its gu tile;
//   u (the Linear's input, recomputed)  the weight gradient sums over rows, so a row scale cannot be taken out of the sum: the
//       scales of ga and u must multiply to ONE constant.  With e_r = the exponent of ga's row, eu_r = an exponent bounding u's row
//       and q_r = e_r + eu_r, row r of u is written as u 2^(140 + e_r - Q), Q = the largest q_r the workgroup has met so far: rows
//       whose product of magnitudes is far below the largest one lose low bits of a contribution that is that much smaller.  When a
//       stage raises Q the matrix waves rescale their gW accumulators by the (exact) power of two before adding it -- the
//       online-softmax trick.  Behind a LayerNorm eu_r needs no data pass: |u| <= (sqrt(127) max|gamma| + max|beta|) keep; without
//       one it is the exponent of the row's largest |u| (a DPP row maximum of the x row already in registers).
// An element below 2^-14 of its window is an fp16 denormal (produced by v_cvt_pk_f16_f32 and honoured by the f16 MFMA,
// tools/micro/f16_probe.hip): an absolute error <= 2^-38 of the row's largest element, below fp32 rounding of the sum.
//
// What the lighter matrix side buys is OCCUPANCY for the vector role: a matrix wave holds 32 registers of W fragments (the h plane;
// the l plane, used once per k-step, is read fragment by fragment from LDS) where fused_bwd4.hip held 96, so the kernel fits 168
// registers per wave and the CU twelve waves -- EIGHT vector waves, two per SIMD, each with one row per 16-lane group instead of
// two -- and the vector role's dependent chains (LDS round trips, DPP row sums, the split itself) are hidden by the twin on the
// same SIMD; fused_bwd4's single vector wave per SIMD issued one instruction per ~8 cycles.
//
// Stage = 32 rows, two ticks per stage, one barrier per tick (the pipeline of fused_bwd4.hip):
//     tick 2k    vector: S0(k+1): gy -> mask -> row scale -> ga[(k+1) % 3];  S2a(k): x -> xhat, keep factors, q_r   | matrix: S1(k): gu = ga W
//     tick 2k+1  vector: S2b(k): gu -> LayerNorm backward -> gx;  u -> u[k % 2]                                     | matrix: S3(k-1): gW += ga^T u
//   vector wave v: rows 4 v .. 4 v + 3 of the stage, one row per DPP row of 16 lanes, 8 elements per lane;
//   matrix wave m: S1 for output columns 32 m .. 32 m + 31 (48 MFMAs 16x16x32 per stage); S3 for the 64 x 64 tile (m >> 1, m & 1)
//     of the workgroup's ONE gW (24 MFMAs 32x32x16 per stage).
// LDS: 3 x 16 KB ga + 2 x 16 KB u + 16.5 KB gu + 32 KB W l-plane + gamma / beta + the q_r slots = 130 KB.
#include <stdlib.h>

#include "common.h"

namespace allset {

using f16x8s = __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16;
using f32x4s = __attribute__((ext_vector_type(4))) float;
using f32x16s = __attribute__((ext_vector_type(16))) float;
typedef short v4ss_t __attribute__((ext_vector_type(4)));
union FragS { uint4 u; f16x8s v; struct { v4ss_t lo, hi; } t; };
constexpr int kSBlock = 768;
constexpr int kSRows = 32;                     // rows per stage
constexpr int kSVWaves = 8;
constexpr int kSTop = 13;                      // a scaled row's largest element lies in [2^13, 2^14)
constexpr int kSEMin = 20;                     // floor of the biased row exponent (rows below 2^-107 are treated as that small)

template <int CTRL>
__device__ __forceinline__ float dpp_fs(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum_s(float v) {     // sum over the 16 lanes of a DPP row, result in every lane of it
  v += dpp_fs<0xB1>(v);
  v += dpp_fs<0x4E>(v);
  v += dpp_fs<0x141>(v);
  v += dpp_fs<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max_s(float v) {     // max over the 16 lanes of a DPP row (v >= 0)
  v = fmaxf(v, dpp_fs<0xB1>(v));
  v = fmaxf(v, dpp_fs<0x4E>(v));
  v = fmaxf(v, dpp_fs<0x141>(v));
  v = fmaxf(v, dpp_fs<0x140>(v));
  return v;
}
__device__ __forceinline__ float amax4_s(float4 a, float m) {
  return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), m);
}
// 2^(field - 127) for a biased exponent field in [0, 254] (0 -> 0.0)
__device__ __forceinline__ float pow2_field_s(int field) { return __uint_as_float(static_cast<uint32_t>(field) << 23); }
// x0, x1 -> packed fp16 planes {hi half: x1, lo half: x0}: h = RN16(x), l = RN16(x - h) (x - h is exact in fp32)
__device__ __forceinline__ void split2_f16(float x0, float x1, uint32_t& ph, uint32_t& pl) {
  float r0, r1;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(x0), "v"(x1));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(ph), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(ph), "v"(x1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pl) : "v"(r0), "v"(r1));
}
__device__ __forceinline__ f16x8s tr_frag2_s(const uint8_t* lo, const uint8_t* hi) {
  FragS f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4ss_t*)(lo));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4ss_t*)(hi));
  return f.v;
}
// LDS byte offsets as 32-bit integers (address space 3 kept: a round trip through a generic pointer turns the reads into flat loads)
using lds_u8_s = __attribute__((address_space(3))) uint8_t;
__device__ __forceinline__ uint32_t lds_off_s(const void* p) {
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8_s*)(p)));
}
typedef uint32_t u32x4_s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 lds_read16_s(uint32_t off) {
  const u32x4_s v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_s*>(static_cast<uintptr_t>(off));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ f16x8s tr_frag2_off_s(uint32_t lo, uint32_t hi) {
  FragS f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) v4ss_t*>(static_cast<uintptr_t>(lo)));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) v4ss_t*>(static_cast<uintptr_t>(hi)));
  return f.v;
}
// byte offset of (row, column byte) in a [rows][256 B] 16-bit plane (fused_bwd4.hip img_off_r: conflict-free for both the row-wise
// 16-byte fragment reads and the transpose reads)
__device__ __forceinline__ int img_off_s(int row, int colbyte) {
  return row * 256 + ((((colbyte >> 6) ^ row) & 3) << 6) + (((((colbyte >> 4) & 3) ^ (row >> 2)) & 3) << 4) + (colbyte & 15);
}
__device__ __forceinline__ uint32_t hash_mix_s(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; return x; }
#ifdef ALLSET_ABL6_NOBAR            // ablation builds only (tools/bwd_f16x3_ablation.py): timing without the barriers, results wrong
#define ALLSET_S_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define ALLSET_S_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#define ALLSET_FRESH_LANE_S(name) \
  int name = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); __asm__ volatile("" : "+v"(name))

// PT (round 6, the PMA tail's first rFF Linear; reference layers.py:153-157): this Linear's input is out = ln0(pooled + att_r) and out
// ALSO feeds the residual add in front of ln1, so the gradient of out is (this Linear's input gradient) + (the residual branch's gs).
// With PT the kernel takes x = pooled, colb = att_r, ln0's statistics / gamma / beta and `gres` = gs: it recomputes out for the weight
// gradient (the saved copy is not read), adds gs to gu BEFORE the LayerNorm backward, and writes what allset_ln_res_bwd_pma wrote in
// a pass of its own -- the gradient of pooled, dcolb (its column sums), and per (row, head) the pooling's backward statistics
// {m + log(l + 1e-16), delta = <pooled_head, gpooled_head>} for allset_pma_bwd_src.  One launch and one [n, 128] round trip less per conv.
// PT2 (round 6, the PMA tail's SECOND rFF Linear): its output z feeds s = out + relu(z), y = dropout_p(relu_post?(ln1(s))).  With PT2
// the kernel takes the gradient of y, the saved sum s, ln1's statistics / gamma / beta and the conv's dropout (p2, seed2): stage S0
// runs ln1's backward on the row it is about to stage -- the dropout's keep factors from the counter hash, the relu mask from the
// recomputed LayerNorm output, two DPP row sums -- writes gs (the residual branch's gradient of out, which the first Linear's backward
// adds in front of ln0's: PT) and stages gs under the relu mask of z as this Linear's gy.  dgamma1 / dbeta1 leave through part_ln.
// What allset_ln_res_bwd computed in a pass of its own.
template <bool HAS_LN, bool DROP_IN, bool RELU_IN, bool HAS_MASK, bool HAS_ACC, bool PT = false, bool PT2 = false>
__global__ __launch_bounds__(kSBlock) void fused_linear_bwd_f16x3_kernel(
    const float* __restrict__ gy, int64_t ldg, const uint32_t* __restrict__ mask, float p_out, const float* __restrict__ W,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float p_in, uint64_t seed_in, float* gx, int64_t ldgx,
    float* __restrict__ part_ln, float* __restrict__ part_w, float* __restrict__ part_b, int64_t n,
    const uint64_t* __restrict__ seed_base, int64_t pstride_w, int64_t pstride_b, int64_t pstride_ln, int64_t gcb, int64_t xcb,
    int64_t gxcb, const float* acc_in, int64_t ldacc, float ln_inv,
    const float* __restrict__ colb = nullptr, const float* __restrict__ gres = nullptr, int64_t ldgres = 0,
    float* __restrict__ part_c = nullptr, int64_t pstride_c = 0, const float* __restrict__ pma_m = nullptr,
    const float* __restrict__ pma_l = nullptr, float* __restrict__ pma_stats = nullptr, int pma_heads = 1,
    const float* __restrict__ s2 = nullptr, int64_t lds2 = 0, const float* __restrict__ stats2 = nullptr,
    const float* __restrict__ gamma2 = nullptr, const float* __restrict__ beta2 = nullptr, int relu_post = 0, float p2 = 0.f,
    uint64_t seed2 = 0, float* __restrict__ gs_out = nullptr, int64_t ldgs = 0) {
  static_assert(!PT || (HAS_LN && !DROP_IN && !RELU_IN && !HAS_MASK && !HAS_ACC), "PT: the LayerNorm prologue, nothing else");
  static_assert(!PT2 || (!HAS_LN && !DROP_IN && HAS_MASK && !HAS_ACC && !PT), "PT2: the masked gy of a Linear without a norm");
  // ln_inv: 1 / 128 for the LayerNorm backward; 0 = the per-column affine prologue (ALLSET_NORM_COLUMN_AFFINE, fused_bwd4.hip: the
  // forward wrote {0, 1} row statistics, gx = gu * gamma, part_ln = sum_r gu * x and sum_r gu) -- then u = x gamma + beta has no bound
  // known in advance and its window comes from the row's own largest element, as without a norm.
  // HAS_LN: a true LayerNorm prologue (stats / gamma / beta); without it u = dropout(relu(x)) and the window of a row of u comes from
  // the row's own largest element.  HAS_ACC (plain Linear only): gx = acc_in + this Linear's input gradient (may alias gx).
  static_assert(!HAS_ACC || (!HAS_LN && !DROP_IN && !RELU_IN && !HAS_MASK), "acc_in: plain Linear only");
  // gcb / xcb / gxcb: 0 = row-major with the operand's leading dimension; cb > 0 = COLUMN-BLOCKED [128 / cb][n][cb] (ld == cb) for
  // gy / x / gx (fused_bwd4.hip)
  constexpr int OD = 128, ID = 128;
  constexpr int R = kSRows;
  constexpr int PLANE = R * 256;                 // bytes per fp16 plane of an image
  constexpr int IMG = 2 * PLANE;                 // one image: planes h, l
  constexpr int SPG = 132;                       // pitch (floats) of the gu tile
  constexpr int kSCols = PT ? 4 : 3;             // column-sum arrays the vector waves hand over at the end (dgamma, dbeta, gb [, dcolb])
  static_assert(kSVWaves * 4 * ID <= R * SPG, "the final column sums reuse the gu tile");
  __shared__ __attribute__((aligned(1024))) uint8_t sGA[3 * IMG];       // (1 KB-aligned: fragment addresses are formed by XOR on the low bits)
  __shared__ __attribute__((aligned(1024))) uint8_t sU[2 * IMG];
  __shared__ __attribute__((aligned(16))) float sGU[R * SPG];
  __shared__ __attribute__((aligned(16))) float sG[ID];
  __shared__ __attribute__((aligned(16))) float sB[ID];
  __shared__ __attribute__((aligned(16))) uint8_t sWL[4 * 2 * 4 * 64 * 16];     // W's l plane: [matrix wave][ct][k-step][lane] fragments
  __shared__ __attribute__((aligned(16))) int sE[4 * kSVWaves];       // [stage % 4][vector wave]: largest q_r = e_r + eu_r of its rows
  seed_in = resolve_seed(seed_base, seed_in);
  const int tid = threadIdx.x;
  if (tid < ID) { sG[tid] = HAS_LN ? gamma[tid] : 1.f; sB[tid] = HAS_LN ? beta[tid] : 0.f; }
  __shared__ __attribute__((aligned(16))) float sCb[PT ? ID : 4];
  if constexpr (PT) { if (tid < ID) sCb[tid] = colb ? colb[tid] : 0.f; }
  __shared__ __attribute__((aligned(16))) float sG2[PT2 ? 2 * OD : 4];      // PT2: ln1's gamma | beta
  if constexpr (PT2) { if (tid < OD) { sG2[tid] = gamma2[tid]; sG2[OD + tid] = beta2[tid]; } }
  seed2 = resolve_seed(seed_base, seed2);
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
#ifdef ALLSET_ABL6_TIMING          // diagnostic builds only: cycles per segment of waves 0 (vector) and 8 (matrix) of workgroup 0
  uint64_t tph[4] = {0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define ALLSET_SMARK(k) do { const uint64_t tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } while (0)
#else
#define ALLSET_SMARK(k) do {} while (0)
#endif
  // The window of u.  Row r of u is written as u 2^(140 + e_r - Q): e_r = the biased exponent of ga's row, eu_r = a biased exponent
  // with |u[r, :]| < 2^(eu_r - 126), q_r = e_r + eu_r, Q = the largest q_r the workgroup has met -- so u' < 2^14, and ga' u' = ga u
  // 2^(kSTop + 267 - Q) whatever the row.  Behind a LayerNorm eu_r is one number for the whole launch, from
  // |u| <= (sqrt(127) max|gamma| + max|beta|) keep -- no data pass; without one it is the exponent of the row's largest |u|.
  const float keep_in_all = DROP_IN ? 1.f / (1.f - p_in) : 1.f;
  int eUc = kSEMin;
  if constexpr (HAS_LN) {
    float g = fmaxf(fabsf(sG[lane0]), fabsf(sG[lane0 + 64])), b = fmaxf(fabsf(sB[lane0]), fabsf(sB[lane0 + 64]));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { g = fmaxf(g, __shfl_xor(g, off)); b = fmaxf(b, __shfl_xor(b, off)); }
    const float U = (11.27f * g + b) * keep_in_all;
    eUc = __builtin_amdgcn_readfirstlane(min(max(static_cast<int>(__float_as_uint(U) >> 23), kSEMin), 254));          // U < 2^(eUc - 126)
  }
  // the largest q_r of stage k, from the slots the vector waves filled in S2a
  auto stage_emax = [&](int64_t k) -> int {
    const int4 a = *reinterpret_cast<const int4*>(&sE[(k & 3) * kSVWaves]), b = *reinterpret_cast<const int4*>(&sE[(k & 3) * kSVWaves + 4]);
    return __builtin_amdgcn_readfirstlane(max(max(max(a.x, a.y), max(a.z, a.w)), max(max(b.x, b.y), max(b.z, b.w))));
  };

  if (wave < kSVWaves) {
    // =================================================== vector waves ===================================================
    const float inv_i = ln_inv;
    const bool affine = HAS_LN && ln_inv == 0.f;
    const float keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
    const float keep_in = keep_in_all;
    const uint32_t thr_in = drop_threshold(p_in);
    const uint32_t seed_lo = static_cast<uint32_t>(seed_in);
    const int c = lane0 & 15, rg = lane0 >> 4;
    const int lr = 4 * wave + rg;                // this lane's row of a stage; columns 64 hb + 4 c .. + 3, hb = 0, 1
    float4 dg[2], db[2], gbv[2], dcv[2];
    const int c40 = 4 * c;
    const uint32_t cog0 = gcb ? static_cast<uint32_t>(((c40 / gcb) * n * gcb + c40 % gcb) * 4) : 4u * c40;
    const uint32_t cox0 = xcb ? static_cast<uint32_t>(((c40 / xcb) * n * xcb + c40 % xcb) * 4) : 4u * c40;
    const uint32_t cogx0 = gxcb ? static_cast<uint32_t>(((c40 / gxcb) * n * gxcb + c40 % gxcb) * 4) : 4u * c40;
    const int64_t dhg = gcb ? 256 * n : 256, dhx = xcb ? 256 * n : 256, dhgx = gxcb ? 256 * n : 256;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      dg[hb] = make_float4(0.f, 0.f, 0.f, 0.f); db[hb] = make_float4(0.f, 0.f, 0.f, 0.f); gbv[hb] = make_float4(0.f, 0.f, 0.f, 0.f);
      dcv[hb] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // PT: lanes per head g (a power of two; 32: one head over both column halves), this lane's head per column half; the residual
    // branch's rows and the pooling's softmax statistics of the lane's heads are requested in S2a for S2b of the same stage, one
    // tick later (two register sets of them, two stages ahead like x, spilled: 58 dwords of scratch per lane)
    const int pt_g = PT ? (ID / pma_heads) / 4 : 1;
    const int pt_h0 = PT ? (pt_g >= 32 ? 0 : c / pt_g) : 0, pt_h1 = PT ? (pt_g >= 32 ? 0 : (16 + c) / pt_g) : 0;

    // Two register sets each: the operands of the next TWO stages are in flight (16 waves x 2 KB x 2 = 64 KB per CU)
    float4 agS[2][2]; uint32_t amS[2][2];            // [set][hb]: gy row / mask words
    float4 sgS[2][2]; float2 s2S[2];                 // PT2: [set][hb]: the saved sum's row; [set]: ln1's statistics of the row
    float4 xrS[2][2]; float2 stS[2];                 // [set][hb]: x row; [set]: statistics
    auto request_s2 = [&](int64_t k, float4 (&sg)[2], float2& st2) {
      if constexpr (PT2) {
        const int64_t s0 = k < T ? stage_of(k) : stage_of(T - 1);
        const int nrc = max(rows_left(s0), 1);
        const int lrc = min(lr, nrc - 1);
        const char* sb = reinterpret_cast<const char*>(s2 + s0 * R * lds2);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          sg[hb] = *reinterpret_cast<const float4*>(sb + hb * 256 + (static_cast<uint32_t>(lrc) * static_cast<uint32_t>(lds2) * 4u + 16u * c));
        st2 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(stats2 + s0 * R * 2) + lrc * 8);
      }
    };
    auto request_gy = [&](int64_t k, float4 (&ag)[2], uint32_t (&am)[2]) {
      const int64_t s0 = k < T ? stage_of(k) : stage_of(T - 1);           // past the end: re-read the last stage (never consumed)
      const int nrc = max(rows_left(s0), 1);
      const int lrc = min(lr, nrc - 1);
      const char* base = reinterpret_cast<const char*>(gy + s0 * R * ldg);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        ag[hb] = *reinterpret_cast<const float4*>(base + hb * dhg + (static_cast<uint32_t>(lrc) * static_cast<uint32_t>(ldg) * 4u + cog0));
        if constexpr (HAS_MASK)   // "mask layout" (include/allset_hip.h): block (row / 16, column / 64), dword (row % 16, 32-column group)
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
        xr[hb] = *reinterpret_cast<const float4*>(xb + hb * dhx + (static_cast<uint32_t>(lrc) * static_cast<uint32_t>(ldx) * 4u + cox0));
      if constexpr (HAS_LN) st = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(stats + s0 * R * 2) + lrc * 8);
    };
    // PT: the residual branch's rows / the softmax statistics of stage k, requested in S2a(k) for S2b(k), one tick ahead.  Measured
    // on one box at [1M, 128] (tools/pma_tail_bwd_bench.py, 4 heads; the two passes this replaces: 0.76 ms): this form 0.61 ms; the
    // requests a whole stage ahead (issued at the end of S2b(k - 1): eight more registers live through S0 / S2a) 0.66; two register
    // sets two stages ahead, like x: 58 dwords of scratch per lane, not timed
    auto request_gres = [&](int64_t k, float4 (&gr)[2]) {
      if constexpr (PT) {
        const int64_t s0 = k < T ? stage_of(k) : stage_of(T - 1);
        const int nrc = max(rows_left(s0), 1);
        const int lrc = min(lr, nrc - 1);
        const char* gb_ = reinterpret_cast<const char*>(gres + s0 * R * ldgres);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          gr[hb] = *reinterpret_cast<const float4*>(gb_ + hb * 256 + (static_cast<uint32_t>(lrc) * static_cast<uint32_t>(ldgres) * 4u + 16u * c));
      }
    };
    auto request_ml = [&](int64_t k, float2 (&ml)[2]) {
      if constexpr (PT) {
        const int64_t s0 = k < T ? stage_of(k) : stage_of(T - 1);
        const int nrc = max(rows_left(s0), 1);
        const int64_t row = s0 * R + min(lr, nrc - 1);
        ml[0] = make_float2(pma_m[row * pma_heads + pt_h0], pma_l[row * pma_heads + pt_h0]);
        ml[1] = make_float2(pma_m[row * pma_heads + pt_h1], pma_l[row * pma_heads + pt_h1]);
      }
    };
    int eNext = kSEMin, eCur = kSEMin;            // biased row exponent of ga: stage k + 1 (written by S0), stage k (read by S2b)
    int Erun = 2 * kSEMin;                        // the largest q_r over the stages up to the one S2b is working on
    int quK = kSEMin;                             // eu_r of stage k (S2a -> S2b)
    // ---- S0(k): ga = gy under the forward's epilogue mask, scaled to the row's window, two fp16 planes into ga[k % 3]
    const float keep2 = (PT2 && p2 > 0.f) ? 1.f / (1.f - p2) : 1.f;
    const uint32_t thr2 = drop_threshold(p2);
    auto S0 = [&](int64_t k, float4 (&ag)[2], uint32_t (&am)[2], float4 (&sg)[2], float2& st2) {
      const int nrows = rows_left(stage_of(k));
      const bool valid = lr < nrows;
      uint8_t* img = sGA + (k % 3) * IMG;
      float4 v[2];
      float amax = 0.f;
      if constexpr (PT2) {
        // ---- ln1's backward on this row: ag = the gradient of y = dropout_p2(relu_post?(LayerNorm(s))), sg = s, st2 = {mean, rstd}
        const int64_t stage = stage_of(k);
        const float mean = st2.x, rstd = st2.y;
        float4 xh[2], gh1[2];
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const float4 g4 = *reinterpret_cast<const float4*>(&sG2[64 * hb + 4 * c]), b4 = *reinterpret_cast<const float4*>(&sG2[OD + 64 * hb + 4 * c]);
          xh[hb] = make_float4((sg[hb].x - mean) * rstd, (sg[hb].y - mean) * rstd, (sg[hb].z - mean) * rstd, (sg[hb].w - mean) * rstd);
          float4 g = ag[hb];
          if (p2 > 0.f) {                           // the conv's dropout on y: the forward's hash (fused_fwd2.hip keep4), element (row, column) of [n, 128]
            float4 kp;
            if (thr2 & kDrop8) {
              const uint64_t stage_quad = static_cast<uint64_t>(stage) * (R * OD / 4);
              const uint32_t hi_term = __umul24(static_cast<uint32_t>(stage_quad >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed2 >> 32);
              const uint32_t lo = static_cast<uint32_t>(stage_quad) | static_cast<uint32_t>((lr * OD + 64 * hb + 4 * c) >> 2);
              const uint32_t h = hash_mix_s((lo ^ static_cast<uint32_t>(seed2)) * 0x9E3779B1U + hi_term), t8 = thr2 & 0xffu;
              kp = make_float4((h & 0xffu) >= t8 ? keep2 : 0.f, ((h >> 8) & 0xffu) >= t8 ? keep2 : 0.f,
                               ((h >> 16) & 0xffu) >= t8 ? keep2 : 0.f, (h >> 24) >= t8 ? keep2 : 0.f);
            } else {
              const uint64_t stage_pair = static_cast<uint64_t>(stage) * (R * OD / 2);
              const uint32_t hi_term = __umul24(static_cast<uint32_t>(stage_pair >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed2 >> 32);
              const uint32_t lo = static_cast<uint32_t>(stage_pair) | static_cast<uint32_t>((lr * OD + 64 * hb + 4 * c) >> 1);
              const uint32_t h0 = hash_mix_s((lo ^ static_cast<uint32_t>(seed2)) * 0x9E3779B1U + hi_term);
              const uint32_t h1 = hash_mix_s(((lo + 1u) ^ static_cast<uint32_t>(seed2)) * 0x9E3779B1U + hi_term);
              kp = make_float4((h0 & 0xffffu) >= thr2 ? keep2 : 0.f, (h0 >> 16) >= thr2 ? keep2 : 0.f,
                               (h1 & 0xffffu) >= thr2 ? keep2 : 0.f, (h1 >> 16) >= thr2 ? keep2 : 0.f);
            }
            g.x *= kp.x; g.y *= kp.y; g.z *= kp.z; g.w *= kp.w;
          }
          if (relu_post) {                          // relu mask from the recomputed LayerNorm output (the forward's own expression)
            if (!(fmaf(xh[hb].x, g4.x, b4.x) > 0.f)) g.x = 0.f;
            if (!(fmaf(xh[hb].y, g4.y, b4.y) > 0.f)) g.y = 0.f;
            if (!(fmaf(xh[hb].z, g4.z, b4.z) > 0.f)) g.z = 0.f;
            if (!(fmaf(xh[hb].w, g4.w, b4.w) > 0.f)) g.w = 0.f;
          }
          if (!valid) g = make_float4(0.f, 0.f, 0.f, 0.f);          // (a dead row re-read the last row: it contributes nothing)
          dg[hb].x = fmaf(g.x, xh[hb].x, dg[hb].x); dg[hb].y = fmaf(g.y, xh[hb].y, dg[hb].y);
          dg[hb].z = fmaf(g.z, xh[hb].z, dg[hb].z); dg[hb].w = fmaf(g.w, xh[hb].w, dg[hb].w);
          db[hb].x += g.x; db[hb].y += g.y; db[hb].z += g.z; db[hb].w += g.w;
          gh1[hb] = make_float4(g.x * g4.x, g.y * g4.y, g.z * g4.z, g.w * g4.w);
          a1 += (gh1[hb].x + gh1[hb].y) + (gh1[hb].z + gh1[hb].w);
          a2 = fmaf(gh1[hb].x, xh[hb].x, fmaf(gh1[hb].y, xh[hb].y, fmaf(gh1[hb].z, xh[hb].z, fmaf(gh1[hb].w, xh[hb].w, a2))));
        }
        const float s1 = row16_sum_s(a1) * (1.f / 128.f), s2r = row16_sum_s(a2) * (1.f / 128.f);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const float4 o = make_float4(rstd * (gh1[hb].x - s1 - xh[hb].x * s2r), rstd * (gh1[hb].y - s1 - xh[hb].y * s2r),
                                       rstd * (gh1[hb].z - s1 - xh[hb].z * s2r), rstd * (gh1[hb].w - s1 - xh[hb].w * s2r));
          if (valid)
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(gs_out + stage * R * ldgs) + hb * 256 +
                                       (static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldgs) * 4u + 16u * c)) = o;
          ag[hb] = o;                               // this Linear's gy (under the relu mask of z, below)
        }
      }
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        v[hb] = ag[hb];
        if constexpr (HAS_MASK) {
          // bit 8 q + (c % 8) of the word for column 64 hb + 4 c + q -> all-ones / zero -> AND with the float's bits (two instructions
          // per element; a test, a select and a multiply were four): 1 / keep rides on the bias sums' FMA and on the row scale below
          const int w = valid ? static_cast<int>(am[hb]) : 0;
          v[hb].x = __int_as_float(__float_as_int(v[hb].x) & __builtin_amdgcn_sbfe(w, (c & 7), 1));
          v[hb].y = __int_as_float(__float_as_int(v[hb].y) & __builtin_amdgcn_sbfe(w, (c & 7) + 8, 1));
          v[hb].z = __int_as_float(__float_as_int(v[hb].z) & __builtin_amdgcn_sbfe(w, (c & 7) + 16, 1));
          v[hb].w = __int_as_float(__float_as_int(v[hb].w) & __builtin_amdgcn_sbfe(w, (c & 7) + 24, 1));
          gbv[hb].x = fmaf(v[hb].x, keep_out, gbv[hb].x); gbv[hb].y = fmaf(v[hb].y, keep_out, gbv[hb].y);      // bias gradient: column sums of ga
          gbv[hb].z = fmaf(v[hb].z, keep_out, gbv[hb].z); gbv[hb].w = fmaf(v[hb].w, keep_out, gbv[hb].w);
        } else {
          v[hb].x = valid ? v[hb].x : 0.f; v[hb].y = valid ? v[hb].y : 0.f; v[hb].z = valid ? v[hb].z : 0.f; v[hb].w = valid ? v[hb].w : 0.f;
          gbv[hb].x += v[hb].x; gbv[hb].y += v[hb].y; gbv[hb].z += v[hb].z; gbv[hb].w += v[hb].w;
        }
        amax = amax4_s(v[hb], amax);
      }
      amax = row16_max_s(amax) * (HAS_MASK ? keep_out : 1.f);
      const int e = min(max(static_cast<int>(__float_as_uint(amax) >> 23), kSEMin), 254);
      eNext = e;
      const float sa = pow2_field_s(254 + kSTop - e) * (HAS_MASK ? keep_out : 1.f);        // row max -> [2^13, 2^14); ga = gy keep_out under the mask
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        uint32_t h0, l0, h1, l1;
        split2_f16(v[hb].x * sa, v[hb].y * sa, h0, l0);
        split2_f16(v[hb].z * sa, v[hb].w * sa, h1, l1);
        const int wo = img_off_s(lr, 128 * hb + 8 * c);
        *reinterpret_cast<uint2*>(img + 0 * PLANE + wo) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(img + 1 * PLANE + wo) = make_uint2(l0, l1);
      }
      __builtin_amdgcn_sched_barrier(0);
      request_gy(k + 2, ag, am);                  // into the set just consumed: two stages ahead
      request_s2(k + 2, sg, st2);
      __builtin_amdgcn_sched_barrier(0);
    };
    // ---- S2a(k): what needs only x: xhat, the keep factors, the relu signs -- kept in registers for S2b(k)
    float4 xhK[2];             // xhat of stage k, [hb]
    int4 kmK[2];               // dropout-in keep MASKS (all-ones / zero; 1 / keep rides on the scales the values meet anyway)
    uint32_t xbK = 0;          // "raw x > 0" flags, bit 4 hb + q
    float rstdK = 1.f, meanK = 0.f;
    float4 grK[2]; float2 mlK[2];                    // PT: the residual branch's row / the softmax statistics of stage k (S2a -> S2b)
    auto S2a = [&](int64_t k, float4 (&xr)[2], float2& st) {
      request_gres(k, grK); request_ml(k, mlK);       // (PT; no-ops otherwise)
      const int64_t stage = stage_of(k);
      const uint64_t stage_pair = static_cast<uint64_t>(stage) * (R * ID / 2);
      const uint32_t stage_pair_lo = static_cast<uint32_t>(stage_pair);
      const uint32_t hi_term = __umul24(static_cast<uint32_t>(stage_pair >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed_in >> 32);
      const uint64_t stage_quad = static_cast<uint64_t>(stage) * (R * ID / 4);
      const uint32_t stage_quad_lo = static_cast<uint32_t>(stage_quad);
      const uint32_t hi_term_q = __umul24(static_cast<uint32_t>(stage_quad >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed_in >> 32);
      xbK = 0;
      const float mean = HAS_LN ? st.x : 0.f, rstd = HAS_LN ? st.y : 1.f;
      rstdK = rstd; meanK = mean;
      float uamax = 0.f;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        int4 km = make_int4(-1, -1, -1, -1);
        if constexpr (DROP_IN) {
          // pair index of (row, column) = stage * 2048 + (lr * 128 + column) / 2: the lane's part is < 2048 -> an OR (common.h pair_hash)
          if (thr_in & kDrop8) {     // 8 bits per element: ONE hash for the lane's float4 (quad index = stage * 1024 + lane part)
            const uint32_t lo = stage_quad_lo | static_cast<uint32_t>((lr * ID + 64 * hb + 4 * c) >> 2);
            const int h = static_cast<int>(hash_mix_s((lo ^ seed_lo) * 0x9E3779B1U + hi_term_q));
            const uint32_t t8 = thr_in & 0xffu;
            if (t8 == 128u) {        // (uniform) p = 0.5, the reference's default: keep iff the byte's top bit is set -- one instruction per element
              km = make_int4(__builtin_amdgcn_sbfe(h, 7, 1), __builtin_amdgcn_sbfe(h, 15, 1), __builtin_amdgcn_sbfe(h, 23, 1), h >> 31);
            } else {
              const uint32_t hu = static_cast<uint32_t>(h);
              km = make_int4((hu & 0xffu) >= t8 ? -1 : 0, ((hu >> 8) & 0xffu) >= t8 ? -1 : 0, ((hu >> 16) & 0xffu) >= t8 ? -1 : 0, (hu >> 24) >= t8 ? -1 : 0);
            }
          } else {
            const uint32_t lo = stage_pair_lo | static_cast<uint32_t>((lr * ID + 64 * hb + 4 * c) >> 1);
            const uint32_t h0 = hash_mix_s((lo ^ seed_lo) * 0x9E3779B1U + hi_term);
            const uint32_t h1 = hash_mix_s(((lo + 1u) ^ seed_lo) * 0x9E3779B1U + hi_term);
            km = make_int4((h0 & 0xffffu) >= thr_in ? -1 : 0, (h0 >> 16) >= thr_in ? -1 : 0, (h1 & 0xffffu) >= thr_in ? -1 : 0, (h1 >> 16) >= thr_in ? -1 : 0);
          }
        }
        kmK[hb] = km;
        float4 t = xr[hb];
        if constexpr (PT) {
          const float4 cb4 = *reinterpret_cast<const float4*>(&sCb[64 * hb + 4 * c]);
          t.x += cb4.x; t.y += cb4.y; t.z += cb4.z; t.w += cb4.w;
        }
        if (RELU_IN) {
          xbK |= ((t.x > 0.f ? 1u : 0u) | (t.y > 0.f ? 2u : 0u) | (t.z > 0.f ? 4u : 0u) | (t.w > 0.f ? 8u : 0u)) << (4 * hb);
          t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
        }
        float4 xh = t;
        if constexpr (HAS_LN) xh = make_float4((t.x - mean) * rstd, (t.y - mean) * rstd, (t.z - mean) * rstd, (t.w - mean) * rstd);
        // (a row past the end re-reads the last row: its xhat is finite and meets ga = 0, gu = 0 and fu = 0 wherever it is used)
        xhK[hb] = xh;
        if constexpr (!HAS_LN) uamax = amax4_s(xh, uamax);
        else if (affine) {
          const float4 gm = *reinterpret_cast<const float4*>(&sG[64 * hb + 4 * c]), bt = *reinterpret_cast<const float4*>(&sB[64 * hb + 4 * c]);
          uamax = amax4_s(make_float4(fmaf(xh.x, gm.x, bt.x), fmaf(xh.y, gm.y, bt.y), fmaf(xh.z, gm.z, bt.z), fmaf(xh.w, gm.w, bt.w)), uamax);
        }
      }
      if constexpr (RELU_IN) __asm__ volatile("" : "+v"(xbK));     // (packed here, not at its use)
      // eu_r, q_r = e_r + eu_r, and the wave's largest q_r -> its slot of the stage (all lanes store the same word)
      if (HAS_LN && !affine) quK = eUc;
      else quK = min(max(static_cast<int>(__float_as_uint(row16_max_s(uamax) * keep_in) >> 23) + 1, kSEMin), 254);     // |u| < 2^(eu - 126)
      {
        const int q = eCur + quK;
        const int qw = max(max(__builtin_amdgcn_readlane(q, 0), __builtin_amdgcn_readlane(q, 16)),
                           max(__builtin_amdgcn_readlane(q, 32), __builtin_amdgcn_readlane(q, 48)));
        sE[(k & 3) * kSVWaves + wave] = qw;
      }
      __builtin_amdgcn_sched_barrier(0);
      request_x(k + 2, xr, st);                   // x is consumed: the request for two stages ahead goes out a tick earlier
      __builtin_amdgcn_sched_barrier(0);
    };
    // ---- S2b(k): gu -> LayerNorm backward -> gx; then u, scaled against the workgroup's largest row exponent, into u[k % 2]
    auto S2b = [&](int64_t k) {
      const int64_t stage = stage_of(k);
      const int nrows = rows_left(stage);
      const bool live = lr < nrows;
      Erun = max(Erun, stage_emax(k));
      const float inv_sa = pow2_field_s(eCur - kSTop) * keep_in;  // undoes the row scale of ga (the wave's W scale is undone by S1); x the dropout's 1 / keep
      const int ffield = 267 + eCur - Erun;
      const float fu = pow2_field_s(live ? max(ffield, 0) : 0) * keep_in;     // (a dead row's u is 0: its xhat is, but beta is not)
      float4 gam[2], v[2];
      float a1 = 0.f, a2 = 0.f;
      // gx = acc_in + ...: a second gradient branch of the same tensor, summed here (may alias gx: each element is read and written
      // by the same lane).  Requested first, consumed last.
      float4 acc[2];
      if constexpr (HAS_ACC) {
        const int lrc = min(lr, max(nrows, 1) - 1);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          acc[hb] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(acc_in + stage * R * ldacc) +
                                                     static_cast<uint32_t>(lrc) * static_cast<uint32_t>(ldacc) * 4u + 256 * hb + 16 * c);
      }
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        gam[hb] = *reinterpret_cast<const float4*>(&sG[64 * hb + 4 * c]);
        v[hb] = *reinterpret_cast<const float4*>(&sGU[lr * SPG + 64 * hb + 4 * c]);
        v[hb].x *= inv_sa; v[hb].y *= inv_sa; v[hb].z *= inv_sa; v[hb].w *= inv_sa;          // (inv_sa carries the dropout's 1 / keep)
        if constexpr (DROP_IN) {
          v[hb].x = __int_as_float(__float_as_int(v[hb].x) & kmK[hb].x); v[hb].y = __int_as_float(__float_as_int(v[hb].y) & kmK[hb].y);
          v[hb].z = __int_as_float(__float_as_int(v[hb].z) & kmK[hb].z); v[hb].w = __int_as_float(__float_as_int(v[hb].w) & kmK[hb].w);
        }
        if constexpr (PT) {                         // d out = this Linear's + the residual branch's (a dead row re-read the last row: it adds nothing)
          const float kzr = live ? 1.f : 0.f;
          v[hb].x = fmaf(grK[hb].x, kzr, v[hb].x); v[hb].y = fmaf(grK[hb].y, kzr, v[hb].y); v[hb].z = fmaf(grK[hb].z, kzr, v[hb].z); v[hb].w = fmaf(grK[hb].w, kzr, v[hb].w);
        }
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
      if constexpr (HAS_LN) { s1 = row16_sum_s(a1) * inv_i; s2 = row16_sum_s(a2) * inv_i; }
      const float rstd = rstdK;
      const float ptSd = PT ? 1.f / rstd : 0.f, ptMean = PT ? meanK : 0.f;
      float ptDot[2] = {0.f, 0.f};
      uint8_t* img = sU + (k % 2) * IMG;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const float4 xh = xhK[hb];
        float4 o = v[hb];
        if constexpr (HAS_LN)
          o = make_float4(rstd * (v[hb].x - s1 - xh.x * s2), rstd * (v[hb].y - s1 - xh.y * s2),
                          rstd * (v[hb].z - s1 - xh.z * s2), rstd * (v[hb].w - s1 - xh.w * s2));
        if (RELU_IN) {
          const uint32_t xb = xbK >> (4 * hb);
          o.x = (xb & 1u) ? o.x : 0.f; o.y = (xb & 2u) ? o.y : 0.f; o.z = (xb & 4u) ? o.z : 0.f; o.w = (xb & 8u) ? o.w : 0.f;
        }
        if constexpr (HAS_ACC) { o.x += acc[hb].x; o.y += acc[hb].y; o.z += acc[hb].z; o.w += acc[hb].w; }
#ifdef ALLSET_ABL6_NOSTORE
        if (live && o.x == 123.456f)
#else
        if (live)
#endif
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(gx + stage * R * ldgx) +
                                     hb * dhgx + (static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldgx) * 4u + cogx0)) = o;
        if constexpr (PT) {
          const float kz = live ? 1.f : 0.f;
          dcv[hb].x = fmaf(o.x, kz, dcv[hb].x); dcv[hb].y = fmaf(o.y, kz, dcv[hb].y); dcv[hb].z = fmaf(o.z, kz, dcv[hb].z); dcv[hb].w = fmaf(o.w, kz, dcv[hb].w);
          // delta = <pooled_head, gpooled_head>: pooled = xhat / rstd + mean - colb, recomputed (eight registers of the raw row saved)
          const float4 cb4 = *reinterpret_cast<const float4*>(&sCb[64 * hb + 4 * c]);
          const float sd = ptSd, mu = ptMean;
          ptDot[hb] = fmaf(fmaf(xh.x, sd, mu) - cb4.x, o.x, fmaf(fmaf(xh.y, sd, mu) - cb4.y, o.y,
                      fmaf(fmaf(xh.z, sd, mu) - cb4.z, o.z, (fmaf(xh.w, sd, mu) - cb4.w) * o.w)));
        }
        float4 u = xh;
        if constexpr (HAS_LN) {
          const float4 bet = *reinterpret_cast<const float4*>(&sB[64 * hb + 4 * c]);
          u = make_float4(fmaf(xh.x, gam[hb].x, bet.x), fmaf(xh.y, gam[hb].y, bet.y), fmaf(xh.z, gam[hb].z, bet.z), fmaf(xh.w, gam[hb].w, bet.w));
        }
        if constexpr (DROP_IN) {                    // (fu carries the dropout's 1 / keep)
          u.x = __int_as_float(__float_as_int(u.x) & kmK[hb].x); u.y = __int_as_float(__float_as_int(u.y) & kmK[hb].y);
          u.z = __int_as_float(__float_as_int(u.z) & kmK[hb].z); u.w = __int_as_float(__float_as_int(u.w) & kmK[hb].w);
        }
        uint32_t h0, l0, h1, l1;
        split2_f16(u.x * fu, u.y * fu, h0, l0);
        split2_f16(u.z * fu, u.w * fu, h1, l1);
        const int wo = img_off_s(lr, 128 * hb + 8 * c);
        *reinterpret_cast<uint2*>(img + 0 * PLANE + wo) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(img + 1 * PLANE + wo) = make_uint2(l0, l1);
      }
      if constexpr (PT) {
        // per head: the lanes of a head are pt_g consecutive lanes of the DPP row (both column halves together when there is one head)
        float d0 = ptDot[0], d1 = ptDot[1];
        if (pt_g >= 32) { d0 += d1; d1 = d0; }
        if (pt_g >= 2) { d0 += dpp_fs<0xB1>(d0); d1 += dpp_fs<0xB1>(d1); }
        if (pt_g >= 4) { d0 += dpp_fs<0x4E>(d0); d1 += dpp_fs<0x4E>(d1); }
        if (pt_g >= 8) { d0 += dpp_fs<0x141>(d0); d1 += dpp_fs<0x141>(d1); }
        if (pt_g >= 16) { d0 += dpp_fs<0x140>(d0); d1 += dpp_fs<0x140>(d1); }
        const int64_t row = stage * R + lr;
        const bool w0 = live && (pt_g >= 32 ? c == 0 : (c % pt_g) == 0), w1 = live && pt_g < 32 && (c % pt_g) == 0;
        // empty target: never gathered; exp(a - FLT_MAX) = 0 (the convention of pma_bwd_stats_kernel, csrc/pma.hip)
        if (w0) *reinterpret_cast<float2*>(pma_stats + (row * pma_heads + pt_h0) * 2) =
            make_float2(mlK[0].y > 0.f ? mlK[0].x + __logf(mlK[0].y + 1e-16f) : 3.402823466e+38f, d0);
        if (w1) *reinterpret_cast<float2*>(pma_stats + (row * pma_heads + pt_h1) * 2) =
            make_float2(mlK[1].y > 0.f ? mlK[1].x + __logf(mlK[1].y + 1e-16f) : 3.402823466e+38f, d1);
      }
      eCur = eNext;                               // S0(k + 1) ran in the previous tick
    };

    request_gy(0, agS[0], amS[0]);
    request_s2(0, sgS[0], s2S[0]);
    request_x(0, xrS[0], stS[0]);
    request_gy(1, agS[1], amS[1]);
    request_s2(1, sgS[1], s2S[1]);
    request_x(1, xrS[1], stS[1]);
    S0(0, agS[0], amS[0], sgS[0], s2S[0]);
    eCur = eNext;
    ALLSET_S_TICK();
    // Two stages per trip: stage k lives in register set 0, stage k + 1 in set 1; no conditional half inside the trip (fused_bwd4.hip:
    // the compiler's s_waitcnt insertion); an odd last stage is peeled off.
    int64_t k = 0;
    ALLSET_SMARK(3);
    for (; k + 1 < T; k += 2) {
      S0(k + 1, agS[1], amS[1], sgS[1], s2S[1]);
      S2a(k, xrS[0], stS[0]);
      ALLSET_SMARK(0);
      ALLSET_S_TICK();
      ALLSET_SMARK(1);
      S2b(k);
      ALLSET_SMARK(2);
      ALLSET_S_TICK();
      ALLSET_SMARK(3);
      if (k + 2 < T) S0(k + 2, agS[0], amS[0], sgS[0], s2S[0]);
      S2a(k + 1, xrS[1], stS[1]);
      ALLSET_SMARK(0);
      ALLSET_S_TICK();
      ALLSET_SMARK(1);
      S2b(k + 1);
      ALLSET_SMARK(2);
      ALLSET_S_TICK();
      ALLSET_SMARK(3);
    }
    if (k < T) {                            // odd stage count: the last stage, in set 0
      S2a(k, xrS[0], stS[0]);
      ALLSET_S_TICK();
      S2b(k);
      ALLSET_S_TICK();
    }
    ALLSET_S_TICK();                        // (the matrix waves' last weight-gradient step)
    // ---- column sums held by the vector waves (dgamma, dbeta, bias gradient): the four row groups of a lane column fold first,
    // then the eight waves through LDS in a fixed order
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      float4 a = dg[hb], b = db[hb], g3 = gbv[hb], c4 = dcv[hb];
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
        b.x += __shfl_xor(b.x, off); b.y += __shfl_xor(b.y, off); b.z += __shfl_xor(b.z, off); b.w += __shfl_xor(b.w, off);
        g3.x += __shfl_xor(g3.x, off); g3.y += __shfl_xor(g3.y, off); g3.z += __shfl_xor(g3.z, off); g3.w += __shfl_xor(g3.w, off);
        if constexpr (PT) { c4.x += __shfl_xor(c4.x, off); c4.y += __shfl_xor(c4.y, off); c4.z += __shfl_xor(c4.z, off); c4.w += __shfl_xor(c4.w, off); }
      }
      if (lane0 < 16) {                         // (gu is free: the last S2b read it two ticks ago)
        *reinterpret_cast<float4*>(&sGU[wave * kSCols * ID + 64 * hb + 4 * lane0]) = a;
        *reinterpret_cast<float4*>(&sGU[wave * kSCols * ID + ID + 64 * hb + 4 * lane0]) = b;
        *reinterpret_cast<float4*>(&sGU[wave * kSCols * ID + 2 * ID + 64 * hb + 4 * lane0]) = g3;
        if constexpr (PT) *reinterpret_cast<float4*>(&sGU[wave * kSCols * ID + 3 * ID + 64 * hb + 4 * lane0]) = c4;
      }
    }
  } else {
    // =================================================== matrix waves ===================================================
    const int m = wave - kSVWaves;
    const int oh = m >> 1, ih = m & 1;           // weight-gradient tile: o in [64 oh, +64), i in [64 ih, +64)
    // ---- this wave's slice of W (32 columns 32 m + 16 ct + nn) as MFMA B fragments: column tile ct, k-step t; lane (nn = lane & 15,
    // kg = lane >> 4) holds W[o = 32 kg + 8 t + j][column], j = 0..7, scaled so that the slice's largest element lies in
    // [2^13, 2^14).  The h plane stays in 32 registers; the l plane is used once per k-step and lives in LDS, fragment by fragment.
    FragS wq[2][4];
    float inv_sw;
    uint4* wl = reinterpret_cast<uint4*>(sWL) + m * (2 * 4 * 64) + lane0;        // [m][ct][t][lane]
    {
      const int nn = lane0 & 15, kg = lane0 >> 4;
      float wv[2][4][8];                          // (read once: 64 registers that the accumulators take over afterwards)
      float amax = 0.f;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            wv[ct][t][j] = W[(32 * kg + 8 * t + j) * ID + 32 * m + 16 * ct + nn];
            amax = fmaxf(amax, fabsf(wv[ct][t][j]));
          }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
      const int ew = __builtin_amdgcn_readfirstlane(min(max(static_cast<int>(__float_as_uint(amax) >> 23), kSEMin), 254));
      const float sw = pow2_field_s(254 + kSTop - ew);
      inv_sw = pow2_field_s(ew - kSTop);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          uint32_t ph[4], pl[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) split2_f16(wv[ct][t][2 * j] * sw, wv[ct][t][2 * j + 1] * sw, ph[j], pl[j]);
          wq[ct][t].u = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          wl[(ct * 4 + t) * 64] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
    }
    f32x16s gw[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) gw[a][b][q] = 0.f;
    int Erun = 2 * kSEMin;

    // ---- S1(k): backward-data for this wave's 32 output columns of the stage's 32 rows: 2 row tiles x 2 column tiles = four
    // independent accumulator chains; the A fragments of step t + 1 are requested before step t's MFMAs
    // LDS addresses: every fragment address of a stage is ONE per-lane base (kept in a register) XOR a constant in the swizzle's
    // chunk / piece bits, plus an immediate (row tile, plane, 16-row step) and the image buffer's offset -- a handful of vector
    // instructions per stage where the generic img_off_s arithmetic cost ~130 (the matrix waves are the critical role, and each of
    // their vector instructions queues behind two vector waves on the same SIMD).
    //   S1: row ri (+16), column byte 64 kg + 16 t:  base1 ^ (t << 4);   rows +16: + 4096 (same swizzle class)
    //   S3: row 16 kb + tr_row + 4 hi, column byte 64 (2 o + tl) + tr_in:  base3 ^ (tl << 6) ^ (hi << 4), + 4096 kb + 1024 hi
    const int base1 = img_off_s(lane0 & 15, 64 * (lane0 >> 4));
    int base3a, base3b, base_gu;
    {
      const int q4 = lane0 >> 4, tr_r = (lane0 & 15) >> 2, tr_row = 8 * (q4 >> 1) + tr_r, tr_in = 32 * (q4 & 1) + 8 * (lane0 & 3);
      base3a = img_off_s(tr_row, 64 * (2 * oh) + tr_in);
      base3b = img_off_s(tr_row, 64 * (2 * ih) + tr_in);
      base_gu = (4 * (lane0 >> 4) * SPG + 32 * m + (lane0 & 15)) * 4;
    }
    auto S1 = [&](int64_t k, int b3) {
      const uint32_t img = lds_off_s(sGA) + static_cast<uint32_t>(b3 * IMG + base1);
      auto load_a = [&](FragS (&f0)[2], FragS (&f1)[2], FragS (&fl)[2], int t) {
        const uint32_t pa = img ^ static_cast<uint32_t>(t << 4);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          f0[pl].u = lds_read16_s(pa + pl * PLANE);
          f1[pl].u = lds_read16_s(pa + pl * PLANE + 16 * 256);
        }
        fl[0].u = wl[(0 * 4 + t) * 64];
        fl[1].u = wl[(1 * 4 + t) * 64];
      };
      FragS fa0[2][2], fa1[2][2], fwl[2][2];
      f32x4s acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4s{0.f, 0.f, 0.f, 0.f};
      load_a(fa0[0], fa1[0], fwl[0], 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) load_a(fa0[(t + 1) & 1], fa1[(t + 1) & 1], fwl[(t + 1) & 1], t + 1);
        const FragS (&a0)[2] = fa0[t & 1];
        const FragS (&a1)[2] = fa1[t & 1];
        const FragS (&bl)[2] = fwl[t & 1];
        // l.h, h.l, h.h
#ifdef ALLSET_ABL6_NOMFMA
        acc[0][0][0] += __builtin_bit_cast(float, a0[0].u.x ^ a0[1].u.y ^ bl[0].u.z ^ bl[1].u.w); acc[1][0][0] += __builtin_bit_cast(float, a1[0].u.x ^ a1[1].u.y);
#else
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[1].v, wq[0][t].v, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[1].v, wq[0][t].v, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[1].v, wq[1][t].v, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[1].v, wq[1][t].v, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[0].v, bl[0].v, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[0].v, bl[0].v, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[0].v, bl[1].v, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[0].v, bl[1].v, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[0].v, wq[0][t].v, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[0].v, wq[0][t].v, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[0].v, wq[1][t].v, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[0].v, wq[1][t].v, acc[1][1], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      // acc[rt][ct][r] = gu[row 16 rt + 4 kg + r][column 32 m + 16 ct + ri] * (row scale) * (slice scale): the slice scale goes here
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(sGU) + base_gu + ((16 * rt + r) * SPG + 16 * ct) * 4) = acc[rt][ct][r] * inv_sw;
    };
    // ---- S3(k): weight gradient, this wave's 64 x 64 tile of gW; K = the stage's 32 rows in two steps of 16; A = ga^T, B = u
    auto S3 = [&](int64_t k, int b3, int b2) {
      const int Ek = stage_emax(k);
      if (Ek > Erun) {                            // a larger row exponent: bring the accumulated sum to the new scale (exact)
        const float f = pow2_field_s(max(127 - (Ek - Erun), 0));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) gw[a][b][q] *= f;
        Erun = Ek;
      }
      const uint32_t ia = lds_off_s(sGA) + static_cast<uint32_t>(b3 * IMG + base3a), iu = lds_off_s(sU) + static_cast<uint32_t>(b2 * IMG + base3b);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f16x8s wa[2][2], wb[2][2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          const uint32_t a_lo = (ia ^ static_cast<uint32_t>(tl << 6)) + kb * 4096, a_hi = (ia ^ static_cast<uint32_t>((tl << 6) | 16)) + kb * 4096 + 1024;
          const uint32_t b_lo = (iu ^ static_cast<uint32_t>(tl << 6)) + kb * 4096, b_hi = (iu ^ static_cast<uint32_t>((tl << 6) | 16)) + kb * 4096 + 1024;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            wa[tl][pl] = tr_frag2_off_s(a_lo + pl * PLANE, a_hi + pl * PLANE);
            wb[tl][pl] = tr_frag2_off_s(b_lo + pl * PLANE, b_hi + pl * PLANE);
          }
        }
#ifdef ALLSET_ABL6_NOMFMA
        { FragS f; f.v = wa[0][0]; FragS g2; g2.v = wb[1][1]; FragS g3; g3.v = wa[1][1]; FragS g4; g4.v = wb[0][0];
          gw[0][0][0] += __builtin_bit_cast(float, f.u.x ^ g2.u.y ^ g3.u.z ^ g4.u.w); }
#else
        constexpr int PA_[3] = {1, 0, 0}, PB_[3] = {0, 1, 0};     // l.h, h.l, h.h
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          gw[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[0][PA_[pr]], wb[0][PB_[pr]], gw[0][0], 0, 0, 0);
          gw[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[1][PA_[pr]], wb[0][PB_[pr]], gw[1][0], 0, 0, 0);
          gw[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[0][PA_[pr]], wb[1][PB_[pr]], gw[0][1], 0, 0, 0);
          gw[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[1][PA_[pr]], wb[1][PB_[pr]], gw[1][1], 0, 0, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    ALLSET_S_TICK();
    ALLSET_SMARK(3);
    int c3 = 0, p3 = 2, c2 = 0;                   // k % 3, (k - 1) % 3, k % 2 without the divisions
    for (int64_t k = 0; k < T; ++k) {
      S1(k, c3);
      ALLSET_SMARK(0);
      ALLSET_S_TICK();
      ALLSET_SMARK(1);
      if (k >= 1) S3(k - 1, p3, c2 ^ 1);
      ALLSET_SMARK(2);
      ALLSET_S_TICK();
      ALLSET_SMARK(3);
      p3 = c3; c3 = c3 == 2 ? 0 : c3 + 1; c2 ^= 1;
    }
    S3(T - 1, p3, c2 ^ 1);
    ALLSET_S_TICK();
    // ---- the workgroup's gW partial: each matrix wave its 64 x 64 tile; the accumulators hold gW 2^(kSTop + 267 - Erun)
    {
      const int X = Erun - 267 - kSTop;           // in [-240, 228]: applied as two factors
      const int X1 = X >> 1, X2 = X - X1;
      const float f1 = pow2_field_s(127 + X1), f2 = pow2_field_s(127 + X2);
      float* pw = part_w + static_cast<int64_t>(blockIdx.x) * pstride_w;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int o = (2 * oh + a) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane0 >> 5);
            pw[o * ID + (2 * ih + b) * 32 + (lane0 & 31)] = (gw[a][b][q] * f1) * f2;
          }
    }
  }
  __syncthreads();
#ifdef ALLSET_ABL6_TIMING
  // vector wave 0: [0] S0 + S2a, [1] wait, [2] S2b, [3] wait; matrix wave 8: [4] S1, [5] wait, [6] S3, [7] wait  (cycles, all stages)
  if (blockIdx.x == 0 && (tid == 0 || tid == 512)) {
    float* dbg = part_w + (tid == 0 ? 0 : 4);         // over this workgroup's own gW entries (stored before the barrier above)
    for (int q = 0; q < 4; ++q) dbg[q] = static_cast<float>(tph[q]);
  }
#endif
  if (tid < kSCols * ID) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < kSVWaves; ++v) s += sGU[v * kSCols * ID + tid];
    const int64_t slice = blockIdx.x;
    if (tid < 2 * ID) { if constexpr (HAS_LN || PT2) part_ln[slice * pstride_ln + tid] = s; }
    else if (tid < 3 * ID) { if (part_b != nullptr) part_b[slice * pstride_b + (tid - 2 * ID)] = s; }
    else if (part_c != nullptr) part_c[slice * pstride_c + (tid - 3 * ID)] = s;
  }
}

}  // namespace allset

using namespace allset;

// 1 = the fp16x3 kernel takes this call: O = I = 128 without auxiliary columns (LayerNorm, column-affine or no prologue; acc_in)
int fused_linear_bwd_f16x3_supported(int64_t O, int64_t I, int has_ln, int norm_mode, int has_acc, int has_aux) {
  (void)has_ln; (void)norm_mode; (void)has_acc;
  return (O == 128 && I == 128 && !has_aux) ? 1 : 0;
}

// Called by allset_fused_linear_bwd_all (fused_bwd.hip) after its argument checks (bwd_all_combo: dropout_in only behind relu_in,
// acc_in only on the plain Linear); ONE partial slice per workgroup, the grid of fused_linear_bwd_roles_grid.
int launch_fused_linear_bwd_f16x3(unsigned grid, hipStream_t st, bool ln, bool drop, bool relu, bool hm, const float* gy, int64_t ldg,
                                  const uint32_t* mask, float p_out, const float* W, const float* x, int64_t ldx,
                                  const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                                  float* gx, int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n,
                                  const uint64_t* seed_base, int64_t psw, int64_t psb, int64_t psl, int64_t gcb, int64_t xcb,
                                  int64_t gxcb, const float* acc_in, int64_t ldacc, float ln_inv) {
#define ALLSET_S_K(LN, DI, RI, HM, HA)                                                                                            \
  fused_linear_bwd_f16x3_kernel<LN, DI, RI, HM, HA><<<grid, kSBlock, 0, st>>>(gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta,  \
                                                                              p_in, seed_in, gx, ldgx, part_ln, part_w, part_b, n, \
                                                                              seed_base, psw, psb, psl, gcb, xcb, gxcb, acc_in, ldacc, ln_inv)
  if (acc_in != nullptr) { ALLSET_S_K(false, false, false, false, true); return 0; }
#define ALLSET_S_M(LN, DI, RI) do { if (hm) ALLSET_S_K(LN, DI, RI, true, false); else ALLSET_S_K(LN, DI, RI, false, false); } while (0)
  if (!relu) { if (ln) ALLSET_S_M(true, false, false); else ALLSET_S_M(false, false, false); }
  else if (ln) { if (drop) ALLSET_S_M(true, true, true); else ALLSET_S_M(true, false, true); }
  else { if (drop) ALLSET_S_M(false, true, true); else ALLSET_S_M(false, false, true); }
#undef ALLSET_S_M
#undef ALLSET_S_K
  return 0;
}

// ---- the PMA tail's first rFF Linear, backward, with ln0's backward and the pooling's backward statistics in the same pass (PT) ----
// gu = gy W + gres;  gx = d(pooled) through ln0's backward (x = pooled, colb = att_r, stats / gamma / beta of ln0);  gW = gy^T out,
// gb = colsum(gy) with out = ln0(pooled + att_r) recomputed;  part: [slices][stride] = gW [128*128] | gb [128] | dgamma, dbeta [256] |
// dcolb [128];  pma_stats[r, h] = {m + log(l + 1e-16), <pooled_h, gx_h>}
extern "C" int allset_fused_linear_bwd_pma_tail_supported(int64_t O, int64_t I, int64_t heads) {
#ifdef ALLSET_NO_F16X3
  (void)O; (void)I; (void)heads;
  return 0;
#else
  if (O != 128 || I != 128 || heads < 1 || 128 % heads != 0 || (128 / heads) % 4 != 0) return 0;
  const int64_t g = (128 / heads) / 4;
  return (g & (g - 1)) == 0 ? 1 : 0;
#endif
}

unsigned fused_linear_bwd_roles_grid(int64_t n);

extern "C" int allset_fused_linear_bwd_pma_tail(const float* gy, int64_t ldg, const float* W, const float* pooled, int64_t ldx,
                                                const float* colb, const float* stats, const float* gamma, const float* beta,
                                                const float* gres, int64_t ldgres, float* gx, int64_t ldgx, float* part,
                                                int64_t part_stride, int64_t n_slices, const float* pma_m, const float* pma_l,
                                                float* pma_stats, int64_t heads, int64_t n, int64_t O, int64_t I, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "fused_linear_bwd_pma_tail: negative size");
  if (!allset_fused_linear_bwd_pma_tail_supported(O, I, heads)) {
    set_error("fused_linear_bwd_pma_tail: out=%lld in=%lld heads=%lld is not built (128 x 128, channels per head 4 x a power of two; "
              "allset_fused_linear_bwd_pma_tail_supported)", static_cast<long long>(O), static_cast<long long>(I), static_cast<long long>(heads));
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(part != nullptr && part_stride >= O * I + O + 3 * I, "fused_linear_bwd_pma_tail: part_stride smaller than O*I + O + 3*I");
  const unsigned grid = fused_linear_bwd_roles_grid(n);
  ALLSET_REQUIRE(n_slices == static_cast<int64_t>(grid), "fused_linear_bwd_pma_tail: part must hold allset_fused_linear_bwd_all_slices_for() slices");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(part, 0, static_cast<size_t>(n_slices) * part_stride * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && W && pooled && stats && gamma && beta && gres && gx && pma_m && pma_l && pma_stats, "fused_linear_bwd_pma_tail: null pointer");
  ALLSET_REQUIRE(aligned16(gy) && aligned16(W) && aligned16(pooled) && aligned16(gres) && aligned16(gx) && (colb == nullptr || aligned16(colb)),
                 "fused_linear_bwd_pma_tail: gy / W / pooled / gres / gx / colb must be 16-byte aligned");
  ALLSET_REQUIRE(ldg >= O && ldg % 4 == 0 && ldx >= I && ldx % 4 == 0 && ldgres >= I && ldgres % 4 == 0 && ldgx >= I && ldgx % 4 == 0 &&
                 ldg < (1 << 24) && ldx < (1 << 24) && ldgres < (1 << 24) && ldgx < (1 << 24),
                 "fused_linear_bwd_pma_tail: rows must be 16-byte aligned, leading dimensions below 2^24");
  ALLSET_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 7u) == 0 && (reinterpret_cast<uintptr_t>(pma_stats) & 7u) == 0,
                 "fused_linear_bwd_pma_tail: stats / pma_stats must be 8-byte aligned");
#ifndef ALLSET_NO_F16X3
  fused_linear_bwd_f16x3_kernel<true, false, false, false, false, true><<<grid, kSBlock, 0, st>>>(
      gy, ldg, nullptr, 0.f, W, pooled, ldx, stats, gamma, beta, 0.f, 0, gx, ldgx, part + O * I + O, part, part + O * I, n, nullptr, part_stride,
      part_stride, part_stride, 0, 0, 0, nullptr, 0, 1.f / 128.f, colb, gres, ldgres, part + O * I + O + 2 * I, part_stride, pma_m, pma_l,
      pma_stats, static_cast<int>(heads));
  ALLSET_LAUNCH_CHECK();
#endif
  return ALLSET_OK;
}

// ---- the PMA tail's second rFF Linear, backward, with ln1's backward as its gy prologue (PT2) ----------------------------------------
// gs = d s through y = dropout_p(relu_post?(LayerNorm_{gamma2,beta2}(s)))  (s, stats2 saved by allset_fused_linear_fwd_res_ln; written to gs_out);
// gx = (gs under the 1-bit mask of z > 0) W through relu_in?(x);  gW, gb as allset_fused_linear_bwd_all;
// part: [slices][stride] = gW [128*128] | gb [128] | dgamma2 [128] | dbeta2 [128]
extern "C" int allset_fused_linear_bwd_ln_pro_supported(int64_t O, int64_t I) {
#ifdef ALLSET_NO_F16X3
  (void)O; (void)I;
  return 0;
#else
  return (O == 128 && I == 128) ? 1 : 0;
#endif
}

extern "C" int allset_fused_linear_bwd_ln_pro(const float* gy, int64_t ldg, const float* s, int64_t lds, const float* stats2,
                                              const float* gamma2, const float* beta2, int relu_post, float p, uint64_t seed,
                                              const uint64_t* seed_base, const uint32_t* mask, const float* W, const float* x, int64_t ldx,
                                              int relu_in, float* gs_out, int64_t ldgs, float* gx, int64_t ldgx, float* part,
                                              int64_t part_stride, int64_t n_slices, int64_t n, int64_t O, int64_t I, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "fused_linear_bwd_ln_pro: negative size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "fused_linear_bwd_ln_pro: dropout p must be in [0,1)");
  if (!allset_fused_linear_bwd_ln_pro_supported(O, I)) {
    set_error("fused_linear_bwd_ln_pro: out=%lld in=%lld is not built (128 x 128; allset_fused_linear_bwd_ln_pro_supported)",
              static_cast<long long>(O), static_cast<long long>(I));
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(part != nullptr && part_stride >= O * I + 3 * O, "fused_linear_bwd_ln_pro: part_stride smaller than O*I + 3*O");
  const unsigned grid = fused_linear_bwd_roles_grid(n);
  ALLSET_REQUIRE(n_slices == static_cast<int64_t>(grid), "fused_linear_bwd_ln_pro: part must hold allset_fused_linear_bwd_all_slices_for() slices");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(part, 0, static_cast<size_t>(n_slices) * part_stride * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && s && stats2 && gamma2 && beta2 && mask && W && x && gs_out && gx, "fused_linear_bwd_ln_pro: null pointer");
  ALLSET_REQUIRE(aligned16(gy) && aligned16(s) && aligned16(W) && aligned16(x) && aligned16(gs_out) && aligned16(gx) && aligned16(gamma2) && aligned16(beta2),
                 "fused_linear_bwd_ln_pro: gy / s / W / x / gs_out / gx / gamma2 / beta2 must be 16-byte aligned");
  ALLSET_REQUIRE(ldg >= O && ldg % 4 == 0 && lds >= O && lds % 4 == 0 && ldgs >= O && ldgs % 4 == 0 && ldx >= I && ldx % 4 == 0 && ldgx >= I &&
                 ldgx % 4 == 0 && ldg < (1 << 24) && lds < (1 << 24) && ldgs < (1 << 24) && ldx < (1 << 24) && ldgx < (1 << 24),
                 "fused_linear_bwd_ln_pro: rows must be 16-byte aligned, leading dimensions below 2^24");
  ALLSET_REQUIRE((reinterpret_cast<uintptr_t>(stats2) & 7u) == 0, "fused_linear_bwd_ln_pro: stats2 must be 8-byte aligned");
#ifndef ALLSET_NO_F16X3
#define ALLSET_S_PT2(RI)                                                                                                              \
  fused_linear_bwd_f16x3_kernel<false, false, RI, true, false, false, true><<<grid, kSBlock, 0, st>>>(                                 \
      gy, ldg, mask, 0.f, W, x, ldx, nullptr, nullptr, nullptr, 0.f, 0, gx, ldgx, part + O * I + O, part, part + O * I, n, seed_base,     \
      part_stride, part_stride, part_stride, 0, 0, 0, nullptr, 0, 1.f / 128.f, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 1, \
      s, lds, stats2, gamma2, beta2, relu_post, p, seed, gs_out, ldgs)
  if (relu_in) ALLSET_S_PT2(true); else ALLSET_S_PT2(false);
#undef ALLSET_S_PT2
  ALLSET_LAUNCH_CHECK();
#endif
  return ALLSET_OK;
}
