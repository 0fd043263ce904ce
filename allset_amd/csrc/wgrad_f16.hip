// The weight gradient of a wide fused Linear behind a LayerNorm prologue (widths 256 / 512) on TWO fp16 planes per operand ("fp16x3",
// fused_bwd6.hip's arithmetic):
//
//     gW[o][i] = sum_r ga[r][o] u[r][i],   gb[o] = sum_r ga[r][o]
//     ga = gy * (forward output > 0 ? keep_out : 0)                  (the forward's 1-bit activation mask, or no epilogue at all)
//     u  = dropout_in(LN_{stats,gamma,beta}(relu_in(x)))             (recomputed from x and the saved row statistics)
//
// dense.hip's wgrad_x6_kernel does the same from three bf16 planes (six products) in 128 x 128 tiles: at 256 x 256 it reads both
// operands twice and runs at a quarter of the HBM rate, the matrix pipe and the staging VALU work adding up on every SIMD
// (DESIGN.md 6.4).  Here a workgroup owns a 256 (o) x 128 (i) tile -- ga is staged in whole rows, u once per i tile; the i tiles of a
// slice sit on one XCD (the launch-order remap below), so the second read of ga is an L2 hit -- and a stage's 32 rows cost three
// f16 MFMAs per 16 x 16 x 32 product instead of six bf16 ones.
//
// fp16 has five exponent bits; every operand is brought into its window by a power of two (exact):
//   ga   per STAGE (32 rows x 256 columns): the largest |gy| keep_out of the stage lands in [2^13, 2^14) (scale 2^(140 - e_k), e_k the
//        biased exponent of that maximum).  The maximum must be known before the stage is split: every thread posts the maximum of the
//        registers it will stage, one stage ahead, into one of three LDS slots (ds_max_u32 on the float's bits) -- no extra barrier.
//   u    |u| <= (sqrt(I) max|gamma| + max|beta|) keep_in = U: one power of two 2^Su for the launch, TIMES 2^(e_k - Q), Q = the largest e_k
//        this workgroup has met so far: the product of the two scales is then 2^(140 + Su - Q) for every stage and comes out of the sum.
//        A stage whose gradients are far below the largest one loses low bits of a contribution that is that much smaller.  When a stage
//        raises Q the waves rescale their accumulators by the (exact) power of two before adding it (fused_bwd6.hip's online window).
// What the windows cost: an element more than 2^14 below its stage's largest one is an fp16 denormal (absolute error <= 2^-38 of the
// stage's largest element); a gradient COLUMN that far below the others therefore loses low bits -- the contract of the 128-wide
// one-pass backward (include/allset_hip_ext.h, "arithmetic").  ALLSET_ARITH_BF16X6 callers keep wgrad_x6_kernel.
#include "common.h"

namespace allset {

using f16x8w_t = __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16;
using f32x4w_t = __attribute__((ext_vector_type(4))) float;
union WfFrag { uint4 u; f16x8w_t v; };
constexpr int kWfBlock = 512;
constexpr int kWfRows = 32;                          // rows per stage (the k extent of one MFMA)
constexpr int kWfTO = 256, kWfTI = 128;              // the workgroup's tile of gW
constexpr int kWfPlaneA = kWfTO * 16, kWfPlaneB = kWfTI * 16;       // dwords per plane: [feature][16 row pairs]
constexpr int kWfBuf = 2 * kWfPlaneA + 2 * kWfPlaneB;               // dwords per buffer (48 KB)
constexpr int kWfEMin = 20;                          // floor of a stage's biased exponent (stages below 2^-107 are treated as that small)

struct WfArgs {
  const float* gy; int64_t ldg;
  const uint32_t* mask; int mask_nh; float keep_out;
  const float* x; int64_t ldx; const float* stats; const float* gamma; const float* beta;
  int relu_in; float p_in; uint64_t seed_in; const uint64_t* seed_base;
  float* part_w; float* part_b; int64_t pw_stride, pb_stride;
  int64_t n; int O, I, tiles_i; int64_t rows_per_slice; int n_slices;
};

template <int CTRL>
__device__ __forceinline__ float wf_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// max over the 16 lanes of a DPP row (v >= 0, no NaN handling wanted: one v_max_f32_dpp per step)
__device__ __forceinline__ float wf_row16_max(float v) {
  asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(v));
  return v;
}
// max(|a.x|, |a.y|, |a.z|, |a.w|, m): two v_max3_f32 with |.| source modifiers (fmaxf / fabsf cost a canonicalising max each)
__device__ __forceinline__ float wf_amax4(float4 a, float m) {
  asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(a.x), "v"(a.y));
  asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(a.z), "v"(a.w));
  return m;
}
// ds_max_u32 without the compiler's wave-reduction loop around an atomic on a uniform address (four lanes post per wave)
__device__ __forceinline__ void wf_lds_max(uint32_t* p, uint32_t v) {
  const uint32_t off = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)(p)));
  asm volatile("ds_max_u32 %0, %1" :: "v"(off), "v"(v) : "memory");
}
// 2^(field - 127) for a biased exponent field; fields <= 0 give 0.0
__device__ __forceinline__ float wf_pow2(int field) { return field > 0 ? __uint_as_float(static_cast<uint32_t>(field) << 23) : 0.f; }
__device__ __forceinline__ int wf_swz(int feat) { return (((feat >> 2) & 3) >> 1) * 3; }      // dense.hip wx6_swz: the same LDS image

// MK: the source of the forward's "output > 0" test on gy: 0 none (the Linear has no relu / dropout epilogue), 2 its 1-bit activation mask.
// DROP: the input dropout's resolution (common.h drop_threshold): 0 none, 1 eight bits per element (one hash per four), 2 sixteen.
// (Template parameters, not flags: a branch inside the staging path splits it into basic blocks the scheduler cannot mix, and on a
// SIMD every vector instruction of that path is paid at the price of a quarter MFMA -- DESIGN.md 6.4.)
#ifdef ALLSET_ABL_WF_NOBAR            // (ablation builds: timing only, results wrong)
#define WF_SYNC() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define WF_SYNC() __syncthreads()
#endif
template <int MK, int DROP>
__global__ __launch_bounds__(kWfBlock) void wgrad_f16_kernel(WfArgs g) {
  __shared__ __attribute__((aligned(16))) uint32_t sP[2][kWfBuf];    // [buffer][A h | A l | B h | B l][feature * 16 + ..]
  __shared__ __attribute__((aligned(16))) float sGB[2 * kWfTI];      // gamma | beta of this tile's input columns
  __shared__ uint32_t sMax[3];                                       // stage maxima (float bits), three stages in rotation
  __shared__ uint32_t sGBm[2];                                       // max |gamma|, max |beta| over the LayerNorm row
  // Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest): remapped, XCD k owns ALL tiles of slices k, k + 8, ... --
  // the tiles of a slice walk the same rows at the same time and the re-read of ga (and of u across o tiles) hits that XCD's L2.
  const int T = gridDim.x;
  const int64_t L = static_cast<int64_t>(blockIdx.y) * T + blockIdx.x;
  const int rem8 = static_cast<int>(L % (8 * T));
  const int slice = static_cast<int>(L / (8 * T)) * 8 + (rem8 & 7);
  const int tile = rem8 >> 3;
  if (slice >= g.n_slices) return;
  const int tile_o = tile / g.tiles_i, tile_i = tile % g.tiles_i;
  const int o_base = tile_o * kWfTO, i_base = tile_i * kWfTI;
  const int64_t r_begin = static_cast<int64_t>(slice) * g.rows_per_slice;      // a multiple of 32
  const int64_t r_end = min(g.n, r_begin + g.rows_per_slice);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int I = g.I;
  float* pw = g.part_w + static_cast<int64_t>(slice) * g.pw_stride;
  // staging map: row pair rp of the 32-row stage; ga columns a_col..+3 and a_col + 128..+3, u columns u_col..+3
  const int rp = lane & 15, cq = wave * 4 + (lane >> 4);
  const int a_col = o_base + cq * 4, u_col = i_base + cq * 4;
  const int w_off = cq * 64 + 4 * ((rp >> 2) ^ wf_swz(cq * 4)) + (rp & 3);          // + c * 16 for the quad's column c
  if (r_begin >= r_end) {                                             // an empty slice still owns a (zero) partial
    for (int e = tid; e < kWfTO * kWfTI; e += kWfBlock) pw[static_cast<int64_t>(o_base + e / kWfTI) * I + i_base + e % kWfTI] = 0.f;
    if (tile_i == 0 && g.part_b != nullptr && tid < kWfTO) g.part_b[static_cast<int64_t>(slice) * g.pb_stride + o_base + tid] = 0.f;
    return;
  }
  const float keep_in = DROP ? 1.f / (1.f - g.p_in) : 1.f;
  const uint32_t thr_in = drop_threshold(g.p_in);
  const uint64_t seed_in = resolve_seed(g.seed_base, g.seed_in);
  const float relu_floor = g.relu_in ? 0.f : -INFINITY;               // fmaxf(t, floor): the relu without a branch
  const int m_col = (a_col / 64) * 32 + (a_col % 64) / 32, m_bit = (a_col % 32) / 4;   // "mask layout" (include/allset_hip_ext.h)
  float4 bsum0 = make_float4(0.f, 0.f, 0.f, 0.f), bsum1 = bsum0;

  // Addresses = a UNIFORM base per stage (scalar registers, scalar arithmetic) + a 32-bit per-thread offset that is a constant of the
  // thread in every full stage: the loads take the SGPR-base form and the staging path spends no vector instruction on 64-bit
  // address arithmetic (the first version of this kernel: 100 of its 410 vector instructions per stage).  Only a slice's last,
  // partial stage clamps its rows (a uniform branch).
  const uint32_t ldg = static_cast<uint32_t>(g.ldg), ldx = static_cast<uint32_t>(g.ldx);
  auto mask_off = [&](int rl) -> uint32_t {
    return static_cast<uint32_t>((rl >> 4) * (g.mask_nh * 32) + m_col + ((rl & 15) >> 2) * 8 + (rl & 3) * 2);
  };
  uint32_t oa[2], ox[2], os[2], om[2];                               // of local rows 2 rp, 2 rp + 1
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    oa[h] = (2 * rp + h) * ldg + a_col; ox[h] = (2 * rp + h) * ldx + u_col; os[h] = (2 * rp + h) * 2; om[h] = mask_off(2 * rp + h);
  }
  const int64_t r_last = (r_end - 1) & ~static_cast<int64_t>(31);    // first row of the slice's last stage
  struct Stage { float4 a0[2], a1[2], u[2]; float2 st[2]; uint32_t m0[2], m1[2]; };
  auto load_stage = [&](Stage& sg, int64_t r0) {                      // issue only; stages past the end re-read the last one (never used)
    const int64_t rc = r0 < r_last ? r0 : r_last;
    const int rows_here = static_cast<int>(min(r_end - rc, static_cast<int64_t>(kWfRows)));
    const float* gp = g.gy + rc * g.ldg;
    const float* xp = g.x + rc * g.ldx;
    const float* sp = g.stats + rc * 2;
    const uint32_t* mp = MK == 2 ? g.mask + (rc >> 4) * (g.mask_nh * 32) : nullptr;
    uint32_t la[2] = {oa[0], oa[1]}, lx[2] = {ox[0], ox[1]}, ls[2] = {os[0], os[1]}, lm[2] = {om[0], om[1]};
    if (rows_here < kWfRows) {                                        // (uniform) the partial stage: rows clamped to the last one
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rl = min(2 * rp + h, rows_here - 1);
        la[h] = rl * ldg + a_col; lx[h] = rl * ldx + u_col; ls[h] = rl * 2; lm[h] = mask_off(rl);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sg.a0[h] = *reinterpret_cast<const float4*>(gp + la[h]);
      sg.a1[h] = *reinterpret_cast<const float4*>(gp + la[h] + 128);
      sg.u[h] = *reinterpret_cast<const float4*>(xp + lx[h]);
      if constexpr (MK == 2) { sg.m0[h] = mp[lm[h]]; sg.m1[h] = mp[lm[h] + 64]; }     // columns + 128: two 64-column blocks further
      sg.st[h] = *reinterpret_cast<const float2*>(sp + ls[h]);
    }
  };
  // One stage ahead of the split: the forward's mask onto the registers (bit -> all-ones / zero -> AND with the float's bits), the
  // stage's largest |ga| keep_out into its slot, and -- in the workgroups that own the bias partial -- the column sums of ga.
  const bool want_b = tile_i == 0 && g.part_b != nullptr;
  auto post_max = [&](Stage& sg, int slot, int64_t r0) {
    if constexpr (MK == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int w0 = static_cast<int>(sg.m0[h]), w1 = static_cast<int>(sg.m1[h]);
#define WF_MSK(v, w, sh) v = __int_as_float(__float_as_int(v) & __builtin_amdgcn_sbfe(w, m_bit + (sh), 1))
        WF_MSK(sg.a0[h].x, w0, 0); WF_MSK(sg.a0[h].y, w0, 8); WF_MSK(sg.a0[h].z, w0, 16); WF_MSK(sg.a0[h].w, w0, 24);
        WF_MSK(sg.a1[h].x, w1, 0); WF_MSK(sg.a1[h].y, w1, 8); WF_MSK(sg.a1[h].z, w1, 16); WF_MSK(sg.a1[h].w, w1, 24);
#undef WF_MSK
      }
    }
    float m = wf_amax4(sg.a0[0], 0.f);
    m = wf_amax4(sg.a0[1], m); m = wf_amax4(sg.a1[0], m); m = wf_amax4(sg.a1[1], m);
    m = wf_row16_max(m) * g.keep_out;
    if (rp == 0) wf_lds_max(&sMax[slot], __float_as_uint(m));
    if (want_b) {                                                     // (uniform)
      const int rows_here = static_cast<int>(min(max(r_end - r0, static_cast<int64_t>(0)), static_cast<int64_t>(kWfRows)));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float kk = (2 * rp + h) < rows_here ? g.keep_out : 0.f;
        const float4 a = sg.a0[h], b = sg.a1[h];
        bsum0.x = fmaf(a.x, kk, bsum0.x); bsum0.y = fmaf(a.y, kk, bsum0.y); bsum0.z = fmaf(a.z, kk, bsum0.z); bsum0.w = fmaf(a.w, kk, bsum0.w);
        bsum1.x = fmaf(b.x, kk, bsum1.x); bsum1.y = fmaf(b.y, kk, bsum1.y); bsum1.z = fmaf(b.z, kk, bsum1.z); bsum1.w = fmaf(b.w, kk, bsum1.w);
      }
    }
  };

  // ---- prologue: Su from the LayerNorm parameters, this tile's gamma / beta into LDS, stages 0 and 1 on their way
  if (tid < 3) sMax[tid] = 0u;
  if (tid < 2) sGBm[tid] = 0u;
  __syncthreads();
  {
    float gm = 0.f, bm = 0.f;
    for (int i = tid; i < I; i += kWfBlock) { gm = fmaxf(gm, fabsf(g.gamma[i])); bm = fmaxf(bm, fabsf(g.beta[i])); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { gm = fmaxf(gm, __shfl_xor(gm, off)); bm = fmaxf(bm, __shfl_xor(bm, off)); }
    if (lane == 0) { atomicMax(&sGBm[0], __float_as_uint(gm)); atomicMax(&sGBm[1], __float_as_uint(bm)); }
    if (tid < kWfTI) { sGB[tid] = g.gamma[i_base + tid]; sGB[kWfTI + tid] = g.beta[i_base + tid]; }
  }
  Stage s0, s1;
  load_stage(s0, r_begin);
  load_stage(s1, r_begin + kWfRows);
  post_max(s0, 0, r_begin);
  post_max(s1, 1, r_begin + kWfRows);
  __syncthreads();
  // |u| <= (sqrt(I) max|gamma| + max|beta|) keep_in < 2^(eU - 126): 2^Su brings it below 2^14
  int S;                                                              // 127 + Su
  {
    const float U = (sqrtf(static_cast<float>(I)) * __uint_as_float(sGBm[0]) + __uint_as_float(sGBm[1])) * keep_in;
    const int eU = static_cast<int>(__float_as_uint(U) >> 23);
    S = 127 + min(max(140 - eU, -100), 100);
  }
  const float4 g4 = *reinterpret_cast<const float4*>(sGB + cq * 4), be4 = *reinterpret_cast<const float4*>(sGB + kWfTI + cq * 4);
  int Q = kWfEMin;                                                    // the largest stage exponent met so far (uniform over the workgroup)
  const uint32_t hp_lane = static_cast<uint32_t>(i_base / 4 + cq);   // dropout counter (element index / 4) = row * (I / 4) + this

  auto store_stage = [&](Stage& sg, int buf, int slot, int64_t r0) -> int {   // returns the Q this stage was scaled against
    const int e = min(max(static_cast<int>(__builtin_amdgcn_readfirstlane(static_cast<int>(sMax[slot])) >> 23), kWfEMin), 254);
    Q = max(Q, e);
    const float sa = wf_pow2(267 - e);                                // 2^(140 - e)
    const float su = wf_pow2(S + e - Q);                              // 2^(Su + e - Q)
    const int rows_here = static_cast<int>(min(r_end - r0, static_cast<int64_t>(kWfRows)));
    // gamma / beta carry the launch scale, the stage's distance to the window and the dropout's 1 / keep
    const float sk = su * keep_in;
    const float4 gs = make_float4(g4.x * sk, g4.y * sk, g4.z * sk, g4.w * sk), bs = make_float4(be4.x * sk, be4.y * sk, be4.z * sk, be4.w * sk);
    float4 va0[2], va1[2], vu[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // rows past the slice's end were loaded from its last row: their ga is zeroed, so whatever u holds there is multiplied by zero
      const float ks = (2 * rp + h) < rows_here ? g.keep_out * sa : 0.f;
      const float4 a = sg.a0[h], b = sg.a1[h];
      float4 t = sg.u[h];
      va0[h] = make_float4(a.x * ks, a.y * ks, a.z * ks, a.w * ks);
      va1[h] = make_float4(b.x * ks, b.y * ks, b.z * ks, b.w * ks);
      t.x = fmaxf(t.x, relu_floor); t.y = fmaxf(t.y, relu_floor); t.z = fmaxf(t.z, relu_floor); t.w = fmaxf(t.w, relu_floor);
      const float2 st = sg.st[h];
      t.x = fmaf((t.x - st.x) * st.y, gs.x, bs.x); t.y = fmaf((t.y - st.x) * st.y, gs.y, bs.y);
      t.z = fmaf((t.z - st.x) * st.y, gs.z, bs.z); t.w = fmaf((t.w - st.x) * st.y, gs.w, bs.w);
      if constexpr (DROP != 0) {
        // element index / 4 = (r0 + 2 rp + h) (I / 4) + i_base / 4 + cq: a uniform 64-bit part plus a 32-bit one
        const int64_t quad = (r0 + h) * (I / 4) + static_cast<int64_t>(static_cast<uint32_t>(2 * rp * (I / 4)) + hp_lane);
        if constexpr (DROP == 1) {
          const uint32_t hsh = pair_hash(seed_in, quad), t8 = thr_in & 0xffu;
          t.x = (hsh & 0xffu) >= t8 ? t.x : 0.f; t.y = ((hsh >> 8) & 0xffu) >= t8 ? t.y : 0.f;
          t.z = ((hsh >> 16) & 0xffu) >= t8 ? t.z : 0.f; t.w = (hsh >> 24) >= t8 ? t.w : 0.f;
        } else {
          const float4 k = keep_scale4(seed_in, quad * 4, thr_in, 1.f);
          t.x *= k.x; t.y *= k.y; t.z *= k.z; t.w *= k.w;
        }
      }
      vu[h] = t;
    }
    uint32_t* pa = &sP[buf][w_off];
    uint32_t* pb = &sP[buf][2 * kWfPlaneA + w_off];
    uint32_t hh, ll;
    split2_f16c(va0[0].x, va0[1].x, hh, ll); pa[0] = hh; pa[kWfPlaneA] = ll;
    split2_f16c(va0[0].y, va0[1].y, hh, ll); pa[16] = hh; pa[kWfPlaneA + 16] = ll;
    split2_f16c(va0[0].z, va0[1].z, hh, ll); pa[32] = hh; pa[kWfPlaneA + 32] = ll;
    split2_f16c(va0[0].w, va0[1].w, hh, ll); pa[48] = hh; pa[kWfPlaneA + 48] = ll;
    split2_f16c(va1[0].x, va1[1].x, hh, ll); pa[128 * 16] = hh; pa[kWfPlaneA + 128 * 16] = ll;
    split2_f16c(va1[0].y, va1[1].y, hh, ll); pa[128 * 16 + 16] = hh; pa[kWfPlaneA + 128 * 16 + 16] = ll;
    split2_f16c(va1[0].z, va1[1].z, hh, ll); pa[128 * 16 + 32] = hh; pa[kWfPlaneA + 128 * 16 + 32] = ll;
    split2_f16c(va1[0].w, va1[1].w, hh, ll); pa[128 * 16 + 48] = hh; pa[kWfPlaneA + 128 * 16 + 48] = ll;
    split2_f16c(vu[0].x, vu[1].x, hh, ll); pb[0] = hh; pb[kWfPlaneB] = ll;
    split2_f16c(vu[0].y, vu[1].y, hh, ll); pb[16] = hh; pb[kWfPlaneB + 16] = ll;
    split2_f16c(vu[0].z, vu[1].z, hh, ll); pb[32] = hh; pb[kWfPlaneB + 32] = ll;
    split2_f16c(vu[0].w, vu[1].w, hh, ll); pb[48] = hh; pb[kWfPlaneB + 48] = ll;
    return Q;
  };

  // ---- MFMA side: 4 (o) x 2 (i) waves, wave tile 64 x 64 = 4 x 4 accumulators of 16 x 16; lane (fj, fg) reads piece fg of feature row fj
  const int fj = lane & 15, fg = lane >> 4;
  const int ob = (wave >> 1) * 64, ib = (wave & 1) * 64;
  f32x4w_t acc[4][4];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot)
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[ot][it] = f32x4w_t{0.f, 0.f, 0.f, 0.f};
  int a_off[4], b_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int fa = ob + t * 16 + fj, fb = ib + t * 16 + fj;
    a_off[t] = fa * 16 + 4 * (fg ^ wf_swz(fa));
    b_off[t] = 2 * kWfPlaneA + fb * 16 + 4 * (fg ^ wf_swz(fb));
  }
  int Qacc = kWfEMin;                                                 // the accumulators are in units of 2^(140 + Su - Qacc)
  // The two waves that share a SIMD (w and w + 4: waves are dealt to SIMDs cyclically) run a half trip's two phases in opposite order
  // -- one multiplies while the other splits and stores (the phases touch different LDS buffers) -- so that the fragment reads and
  // load waits of one are covered by the other's arithmetic instead of both stalling together (wide_mlp.hip's mfma_first).
#ifdef ALLSET_ABL_WF_SAMEPHASE
  const bool mfma_first = true;
#else
  const bool mfma_first = wave < 4;
#endif
  auto mfma_stage = [&](int buf, int qb) {
    if (qb != Qacc) {                                                 // (uniform) the window moved up: bring the sums along
      const float s = wf_pow2(127 + Qacc - qb);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[ot][it] *= s;
      Qacc = qb;
    }
#ifdef ALLSET_ABL_WF_NOMFMA          // (ablation builds: timing only)
    return;
#endif
    WfFrag b[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) b[it][pl].u = *reinterpret_cast<const uint4*>(&sP[buf][pl * kWfPlaneB + b_off[it]]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
      WfFrag a[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) a[pl].u = *reinterpret_cast<const uint4*>(&sP[buf][pl * kWfPlaneA + a_off[ot]]);
      // smallest products first; four independent accumulators between two MFMAs into the same one
#pragma unroll
      for (int it = 0; it < 4; ++it) acc[ot][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1].v, b[it][0].v, acc[ot][it], 0, 0, 0);
#pragma unroll
      for (int it = 0; it < 4; ++it) acc[ot][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].v, b[it][1].v, acc[ot][it], 0, 0, 0);
#pragma unroll
      for (int it = 0; it < 4; ++it) acc[ot][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].v, b[it][0].v, acc[ot][it], 0, 0, 0);
    }
  };

  // stage k lives in LDS buffer k & 1 and was scaled from slot k % 3; register set s1 holds stage k + 1, s0 stage k + 2 (roles swap
  // every half trip).  Half trip k: multiply stage k | split stage k + 1 (its slot was posted a half trip ago) | request stage k + 3 |
  // post stage k + 2's maximum | clear stage k's slot -- three distinct slots, one barrier.
  int q0 = store_stage(s0, 0, 0, r_begin), q1 = Q;
  load_stage(s0, r_begin + 2 * kWfRows);
  __syncthreads();
  int sl0 = 0, sl1 = 1, sl2 = 2;                                      // slots of stages k, k + 1, k + 2
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * kWfRows) {
    const int qa = q0;
    if (mfma_first) mfma_stage(0, qa);
    if (r0 + kWfRows < r_end) q1 = store_stage(s1, 1, sl1, r0 + kWfRows);
    if (!mfma_first) mfma_stage(0, qa);
    load_stage(s1, r0 + 3 * kWfRows);
#ifdef ALLSET_ABL_WF_POST          // (ablation builds: timing only, results wrong) post from the set that has had two half trips to land
    post_max(s1, sl2, r0 + 2 * kWfRows);
#else
    post_max(s0, sl2, r0 + 2 * kWfRows);
#endif
    if (tid == 0) sMax[sl0] = 0u;
    WF_SYNC();
    if (r0 + kWfRows < r_end) {
      const int qb = q1;
      if (mfma_first) mfma_stage(1, qb);
      if (r0 + 2 * kWfRows < r_end) q0 = store_stage(s0, 0, sl2, r0 + 2 * kWfRows);
      if (!mfma_first) mfma_stage(1, qb);
      load_stage(s0, r0 + 4 * kWfRows);
#ifdef ALLSET_ABL_WF_POST
      post_max(s0, sl0, r0 + 3 * kWfRows);
#else
      post_max(s1, sl0, r0 + 3 * kWfRows);
#endif
      if (tid == 0) sMax[sl1] = 0u;
      WF_SYNC();
    }
    const int t0 = sl0; sl0 = sl2; sl2 = sl1; sl1 = t0;               // two stages on: (k, k+1, k+2) -> (k+2, k+3, k+4) = slots (sl2, sl0, sl1)
  }

  // ---- epilogue: partial tile -> part_w[slice][O][I], unscaled; acc[ot][it][r] is (o = .. + 4 fg + r, i = .. + fj)
  const float f1 = wf_pow2(Qacc - 13), f2 = wf_pow2(254 - S);         // 2^(Qacc - 140) 2^(-Su)
#pragma unroll
  for (int ot = 0; ot < 4; ++ot)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = i_base + ib + it * 16 + fj;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o_base + ob + ot * 16 + 4 * fg + r;
        pw[static_cast<int64_t>(o) * I + i] = acc[ot][it][r] * f1 * f2;
      }
    }
  // bias partial: the 16 row-pair lanes of a column quad fold by shuffles (only i-tile 0 writes)
  if (tile_i == 0 && g.part_b != nullptr) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      bsum0.x += __shfl_xor(bsum0.x, off); bsum0.y += __shfl_xor(bsum0.y, off);
      bsum0.z += __shfl_xor(bsum0.z, off); bsum0.w += __shfl_xor(bsum0.w, off);
      bsum1.x += __shfl_xor(bsum1.x, off); bsum1.y += __shfl_xor(bsum1.y, off);
      bsum1.z += __shfl_xor(bsum1.z, off); bsum1.w += __shfl_xor(bsum1.w, off);
    }
    if (rp == 0) {
      float* pbv = g.part_b + static_cast<int64_t>(slice) * g.pb_stride + a_col;
      *reinterpret_cast<float4*>(pbv) = bsum0;
      *reinterpret_cast<float4*>(pbv + 128) = bsum1;
    }
  }
}

}  // namespace allset

using namespace allset;

// 1 = allset_wgrad_f16x3 takes a [O, I] weight: O a multiple of 256 (a tile stages whole 256-column blocks of gy), I of 128, I <= 512
// (the LayerNorm row)
extern "C" int allset_wgrad_f16x3_supported(int64_t O, int64_t I) {
  return (O >= kWfTO && O % kWfTO == 0 && I >= kWfTI && I % kWfTI == 0 && I <= 512 && O <= 4096) ? 1 : 0;
}

extern "C" int allset_wgrad_f16x3_slices(int64_t n, int64_t O, int64_t I, int64_t* n_slices) {
  clear_error();
  ALLSET_REQUIRE(n_slices != nullptr && n >= 0, "wgrad_f16x3_slices: bad argument");
  if (!allset_wgrad_f16x3_supported(O, I)) { set_error("wgrad_f16x3_slices: [%lld, %lld] not supported", (long long)O, (long long)I); return ALLSET_ERR_UNSUPPORTED; }
  const int64_t tiles = (O / kWfTO) * (I / kWfTI);
  // one 96-KiB-LDS workgroup per CU: 256 workgroups in all, slices in multiples of 8 (the XCD remap), at least 256 rows per slice
  int64_t s = 256 / tiles / 8 * 8;
  if (s < 8) s = 8;
  const int64_t max_by_rows = (n + 255) / 256;
  if (s > max_by_rows) s = max_by_rows;
  if (s < 1) s = 1;
  *n_slices = s;
  return ALLSET_OK;
}

// The weight / bias gradient partials of out = epilogue(LN(relu_in?(x)) dropped @ W^T + b) from gy, x, the saved row statistics and the
// forward's activation mask (NULL: the Linear has no relu / dropout epilogue); part: [n_slices] rows of part_stride floats, gW [O, I]
// then (want_bias) gb [O]; the caller sums the rows (allset_reduce_partials).  Same contract as allset_wgrad_fused_ex otherwise.
extern "C" int allset_wgrad_f16x3(const float* gy, int64_t ldg, const uint32_t* mask, float p_out, const float* x, int64_t ldx,
                                  const float* stats, const float* gamma, const float* beta, int relu_in, float p_in, uint64_t seed_in,
                                  float* part, int64_t part_stride, int want_bias, int64_t n_slices, int64_t n, int64_t O, int64_t I,
                                  const uint64_t* seed_base, void* stream) {
  clear_error();
  if (!allset_wgrad_f16x3_supported(O, I)) { set_error("wgrad_f16x3: [%lld, %lld] not supported (O %% 256 == 0, I %% 128 == 0, I <= 512)", (long long)O, (long long)I); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(n >= 0 && n_slices >= 1 && n_slices < 65536, "wgrad_f16x3: bad row / slice count");
  ALLSET_REQUIRE(part != nullptr && part_stride >= O * I + (want_bias ? O : 0) && part_stride % 4 == 0 && aligned16(part),
                 "wgrad_f16x3: part must be 16-byte aligned rows of at least O*I (+O) floats, stride a multiple of 4");
  ALLSET_REQUIRE(p_in >= 0.f && p_in < 1.f && p_out >= 0.f && p_out < 1.f, "wgrad_f16x3: dropout p must be in [0,1)");
  ALLSET_REQUIRE(mask != nullptr || p_out == 0.f, "wgrad_f16x3: an output dropout needs the activation mask");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(part, 0, static_cast<size_t>(n_slices) * part_stride * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && x && stats && gamma && beta, "wgrad_f16x3: null input (the LayerNorm prologue is not optional here)");
  ALLSET_REQUIRE(ldg >= O && ldx >= I && ldg % 4 == 0 && ldx % 4 == 0 && aligned16(gy) && aligned16(x),
                 "wgrad_f16x3: rows must be 16-byte aligned and at least as long as the feature width");
  ALLSET_REQUIRE(ldg < (int64_t{1} << 24) && ldx < (int64_t{1} << 24), "wgrad_f16x3: leading dimensions are 32-bit offsets within a 32-row stage");
  WfArgs a;
  a.gy = gy; a.ldg = ldg; a.mask = mask; a.mask_nh = static_cast<int>(O / 64); a.keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  a.x = x; a.ldx = ldx; a.stats = stats; a.gamma = gamma; a.beta = beta; a.relu_in = relu_in; a.p_in = p_in; a.seed_in = seed_in;
  a.seed_base = seed_base;
  a.part_w = part; a.part_b = want_bias ? part + O * I : nullptr; a.pw_stride = part_stride; a.pb_stride = part_stride;
  a.n = n; a.O = static_cast<int>(O); a.I = static_cast<int>(I); a.tiles_i = static_cast<int>(I / kWfTI);
  int64_t rps = (n + n_slices - 1) / n_slices;
  rps = (rps + kWfRows - 1) / kWfRows * kWfRows;
  a.rows_per_slice = rps; a.n_slices = static_cast<int>(n_slices);
  const dim3 grid(static_cast<unsigned>((O / kWfTO) * (I / kWfTI)), static_cast<unsigned>((n_slices + 7) / 8 * 8));
  // the dropout's resolution as common.h drop_threshold() chooses it: 8 bits per element iff p * 256 is an integer
  const float t8 = p_in * 256.0f;
#ifdef ALLSET_ABL_DROP16          // (ablation builds: the 16-bit form for every p, common.h)
  const int drop = p_in > 0.f ? 2 : 0;
  (void)t8;
#else
  const int drop = p_in > 0.f ? (t8 == floorf(t8) ? 1 : 2) : 0;
#endif
#define ALLSET_WF16(MKV) do { if (drop == 0) wgrad_f16_kernel<MKV, 0><<<grid, kWfBlock, 0, st>>>(a); \
    else if (drop == 1) wgrad_f16_kernel<MKV, 1><<<grid, kWfBlock, 0, st>>>(a); else wgrad_f16_kernel<MKV, 2><<<grid, kWfBlock, 0, st>>>(a); } while (0)
  if (mask != nullptr) ALLSET_WF16(2); else ALLSET_WF16(0);
#undef ALLSET_WF16
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
