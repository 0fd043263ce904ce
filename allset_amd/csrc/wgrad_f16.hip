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
// f16 MFMAs per product instead of six bf16 ones.
//
// Staging is fused_bwd6.hip's: a lane owns 4 consecutive columns of ONE row in each 64-column block (16 lanes = 256 contiguous bytes of
// a row per load instruction: 8 cache lines per wave instruction; the first version of this kernel packed ROW PAIRS per lane as
// wgrad_x6_kernel does -- 16 rows x 64 bytes per instruction, a quarter of each line's tag lookups useful -- and its critical waves
// spent a third of their cycles issuing loads), the two fp16 planes go to LDS ROW-major ([32 rows][256 B] images, XOR-swizzled) and
// the reduction over rows reaches the MFMA through transposing reads (ds_read_b64_tr_b16, 32 x 32 x 16 MFMAs: fused_bwd6.hip's S3).
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
using f32x16w_t = __attribute__((ext_vector_type(16))) float;
typedef short wf_v4s_t __attribute__((ext_vector_type(4)));
union WfFrag { uint4 u; f16x8w_t v; struct { wf_v4s_t lo, hi; } t; };
constexpr int kWfBlock = 512;
constexpr int kWfRows = 32;                          // rows per stage
constexpr int kWfTO = 256, kWfTI = 128;              // the workgroup's tile of gW
constexpr int kWfPlane = kWfRows * 256;              // bytes per fp16 plane of a 128-column image: [32 rows][256 B]
constexpr int kWfImg = 2 * kWfPlane;                 // one image: planes h, l (16 KB)
constexpr int kWfBuf = 3 * kWfImg;                   // a stage: ga columns 0..127, ga columns 128..255, u (48 KB)
constexpr int kWfEMin = 20;                          // floor of a stage's biased exponent (stages below 2^-107 are treated as that small)

struct WfArgs {
  const float* gy; int64_t ldg;
  const uint32_t* mask; int mask_nh; float keep_out;
  const float* x; int64_t ldx; const float* stats; const float* gamma; const float* beta;
  int relu_in; float p_in; uint64_t seed_in; const uint64_t* seed_base;
  float* part_w; float* part_b; int64_t pw_stride, pb_stride;
  int64_t n; int O, I, tiles_i; int64_t rows_per_slice; int n_slices;
};

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
using wf_lds_u8 = __attribute__((address_space(3))) uint8_t;
__device__ __forceinline__ uint32_t wf_lds_off(const void* p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((wf_lds_u8*)(p))); }
// ds_max_u32 without the compiler's wave-reduction loop around an atomic on a uniform address (four lanes post per wave)
__device__ __forceinline__ void wf_lds_max(uint32_t* p, uint32_t v) {
  asm volatile("ds_max_u32 %0, %1" :: "v"(wf_lds_off(p)), "v"(v) : "memory");
}
// 2^(field - 127) for a biased exponent field; fields <= 0 give 0.0
__device__ __forceinline__ float wf_pow2(int field) { return field > 0 ? __uint_as_float(static_cast<uint32_t>(field) << 23) : 0.f; }
// byte offset of (row, column byte) in a [32][256 B] 16-bit plane (fused_bwd6.hip img_off_s: conflict-free for the row-wise 8-byte
// stores and for the transposing reads)
__device__ __forceinline__ int wf_img_off(int row, int colbyte) {
  return row * 256 + ((((colbyte >> 6) ^ row) & 3) << 6) + (((((colbyte >> 4) & 3) ^ (row >> 2)) & 3) << 4) + (colbyte & 15);
}
__device__ __forceinline__ f16x8w_t wf_tr_frag(uint32_t lo, uint32_t hi) {
  WfFrag f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) wf_v4s_t*>(static_cast<uintptr_t>(lo)));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) wf_v4s_t*>(static_cast<uintptr_t>(hi)));
  return f.v;
}

#ifdef ALLSET_ABL_WF_NOBAR            // (ablation builds: timing only, results wrong)
#define WF_SYNC() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define WF_SYNC() __syncthreads()
#endif
// MK: the source of the forward's "output > 0" test on gy: 0 none (the Linear has no relu / dropout epilogue), 2 its 1-bit activation mask.
// DROP: the input dropout's resolution (common.h drop_threshold): 0 none, 1 eight bits per element (one hash per four), 2 sixteen.
// (Template parameters, not flags: a branch inside the staging path splits it into basic blocks the scheduler cannot mix, and on a
// SIMD every vector instruction of that path is paid at the price of a quarter MFMA -- DESIGN.md 6.4.)
template <int MK, int DROP>
__global__ __launch_bounds__(kWfBlock) void wgrad_f16_kernel(WfArgs g) {
  __shared__ __attribute__((aligned(1024))) uint8_t sP[2 * kWfBuf];  // (1 KB-aligned: fragment addresses are formed by XOR on the low bits)
  __shared__ __attribute__((aligned(16))) float sGB[2 * kWfTI];      // gamma | beta of this tile's input columns
  __shared__ __attribute__((aligned(16))) float sRed[8 * kWfTO];     // the eight waves' bias column sums (epilogue)
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
  const bool want_b = tile_i == 0 && g.part_b != nullptr;
  if (r_begin >= r_end) {                                             // an empty slice still owns a (zero) partial
    for (int e = tid; e < kWfTO * kWfTI; e += kWfBlock) pw[static_cast<int64_t>(o_base + e / kWfTI) * I + i_base + e % kWfTI] = 0.f;
    if (want_b && tid < kWfTO) g.part_b[static_cast<int64_t>(slice) * g.pb_stride + o_base + tid] = 0.f;
    return;
  }
  // staging map: row lr of the 32-row stage (one row per 16 lanes), columns 64 hb + 4 c .. + 3 of every 64-column block hb:
  // four blocks of ga, two of u
  const int c = lane & 15, lr = 4 * wave + (lane >> 4);
  const float keep_in = DROP ? 1.f / (1.f - g.p_in) : 1.f;
  const uint32_t thr_in = drop_threshold(g.p_in);
  const uint64_t seed_in = resolve_seed(g.seed_base, g.seed_in);
  const float relu_floor = g.relu_in ? 0.f : -INFINITY;               // fmaxf(t, floor): the relu without a branch
  float4 bsum[4];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) bsum[hb] = make_float4(0.f, 0.f, 0.f, 0.f);

  // Addresses = a UNIFORM base per stage (scalar registers, scalar arithmetic) + a 32-bit per-thread offset that is a constant of the
  // thread in every full stage; only a slice's last, partial stage clamps its rows (a uniform branch).
  const uint32_t ldg = static_cast<uint32_t>(g.ldg), ldx = static_cast<uint32_t>(g.ldx);
  // "mask layout" (include/allset_hip_ext.h): block (row / 16, column / 64) of 32 dwords, dword (row % 16 / 4) * 8 + (row % 4) * 2 +
  // (column % 64) / 32, bit 8 q + (column % 32) / 4 for column + q: this lane's dword of block hb, its bits at (c & 7) + 8 q
  auto mask_off = [&](int rl) -> uint32_t {
    return static_cast<uint32_t>(((rl >> 4) * g.mask_nh + o_base / 64) * 32 + ((rl & 15) >> 2) * 8 + (rl & 3) * 2 + (c >> 3));
  };
  const uint32_t oa0 = lr * ldg + o_base + 4 * c, ox0 = lr * ldx + i_base + 4 * c, os0 = lr * 2, om0 = mask_off(lr);
  const int64_t r_last = (r_end - 1) & ~static_cast<int64_t>(31);    // first row of the slice's last stage
  struct Stage { float4 a[4], u[2]; float2 st; uint32_t m[4]; };
  auto load_stage = [&](Stage& sg, int64_t r0) {                      // issue only; stages past the end re-read the last one (never used)
    const int64_t rc = r0 < r_last ? r0 : r_last;
    const int rows_here = static_cast<int>(min(r_end - rc, static_cast<int64_t>(kWfRows)));
    const float* gp = g.gy + rc * g.ldg;
    const float* xp = g.x + rc * g.ldx;
    const float* sp = g.stats + rc * 2;
    const uint32_t* mp = MK == 2 ? g.mask + (rc >> 4) * (g.mask_nh * 32) : nullptr;
    uint32_t la = oa0, lx = ox0, ls = os0, lm = om0;
    if (rows_here < kWfRows) {                                        // (uniform) the partial stage: rows clamped to the last one
      const int rl = min(lr, rows_here - 1);
      la = rl * ldg + o_base + 4 * c; lx = rl * ldx + i_base + 4 * c; ls = rl * 2; lm = mask_off(rl);
    }
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      sg.a[hb] = *reinterpret_cast<const float4*>(gp + la + 64 * hb);
      if constexpr (MK == 2) sg.m[hb] = mp[lm + 32 * hb];
    }
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) sg.u[hb] = *reinterpret_cast<const float4*>(xp + lx + 64 * hb);
    sg.st = *reinterpret_cast<const float2*>(sp + ls);
  };
  // One stage ahead of the split: the forward's mask onto the registers (bit -> all-ones / zero -> AND with the float's bits), the
  // stage's largest |ga| keep_out into its slot (the mask only removes elements: any bound of the unmasked row would do), and -- in
  // the workgroups that own the bias partial -- the column sums of ga.
  auto post_max = [&](Stage& sg, int slot, int64_t r0) {
    float m = 0.f;
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      if constexpr (MK == 2) {
        const int w = static_cast<int>(sg.m[hb]);
#define WF_MSK(v, sh) v = __int_as_float(__float_as_int(v) & __builtin_amdgcn_sbfe(w, (c & 7) + (sh), 1))
        WF_MSK(sg.a[hb].x, 0); WF_MSK(sg.a[hb].y, 8); WF_MSK(sg.a[hb].z, 16); WF_MSK(sg.a[hb].w, 24);
#undef WF_MSK
      }
      m = wf_amax4(sg.a[hb], m);
    }
    m = wf_row16_max(m) * g.keep_out;
    if (c == 0) wf_lds_max(&sMax[slot], __float_as_uint(m));
    if (want_b) {                                                     // (uniform)
      const float kk = lr < r_end - r0 ? g.keep_out : 0.f;
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        bsum[hb].x = fmaf(sg.a[hb].x, kk, bsum[hb].x); bsum[hb].y = fmaf(sg.a[hb].y, kk, bsum[hb].y);
        bsum[hb].z = fmaf(sg.a[hb].z, kk, bsum[hb].z); bsum[hb].w = fmaf(sg.a[hb].w, kk, bsum[hb].w);
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
  float4 g4[2], be4[2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
    g4[hb] = *reinterpret_cast<const float4*>(sGB + 64 * hb + 4 * c);
    be4[hb] = *reinterpret_cast<const float4*>(sGB + kWfTI + 64 * hb + 4 * c);
  }
  int Q = kWfEMin;                                                    // the largest stage exponent met so far (uniform over the workgroup)
  // dropout counter (element index / 4) of (row, block hb) = row * (I / 4) + i_base / 4 + 16 hb + c
  const uint32_t hp_lane = static_cast<uint32_t>(lr * (I / 4) + i_base / 4 + c);
  const int wo0 = wf_img_off(lr, 8 * c), wo1 = wf_img_off(lr, 128 + 8 * c);     // column bytes 128 hb + 8 c of the lane's row

  auto store_stage = [&](Stage& sg, int buf, int slot, int64_t r0) -> int {   // returns the Q this stage was scaled against
    const int e = min(max(static_cast<int>(__builtin_amdgcn_readfirstlane(static_cast<int>(sMax[slot])) >> 23), kWfEMin), 254);
    Q = max(Q, e);
    const float sa = wf_pow2(267 - e);                                // 2^(140 - e)
    const float sk = wf_pow2(S + e - Q) * keep_in;                    // 2^(Su + e - Q), and the dropout's 1 / keep
    // rows past the slice's end were loaded from its last row: their ga is zeroed, so whatever u holds there is multiplied by zero
    const float ks = lr < r_end - r0 ? g.keep_out * sa : 0.f;
    uint8_t* img = sP + buf * kWfBuf;
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      const float4 a = sg.a[hb];
      uint32_t h0, l0, h1, l1;
      split2_f16c(a.x * ks, a.y * ks, h0, l0);
      split2_f16c(a.z * ks, a.w * ks, h1, l1);
      uint8_t* p = img + (hb >> 1) * kWfImg + ((hb & 1) ? wo1 : wo0);
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + kWfPlane) = make_uint2(l0, l1);
    }
    const float2 st = sg.st;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      float4 t = sg.u[hb];
      const float4 gs = make_float4(g4[hb].x * sk, g4[hb].y * sk, g4[hb].z * sk, g4[hb].w * sk);
      const float4 bs = make_float4(be4[hb].x * sk, be4[hb].y * sk, be4[hb].z * sk, be4[hb].w * sk);
      t.x = fmaxf(t.x, relu_floor); t.y = fmaxf(t.y, relu_floor); t.z = fmaxf(t.z, relu_floor); t.w = fmaxf(t.w, relu_floor);
      t.x = fmaf((t.x - st.x) * st.y, gs.x, bs.x); t.y = fmaf((t.y - st.x) * st.y, gs.y, bs.y);
      t.z = fmaf((t.z - st.x) * st.y, gs.z, bs.z); t.w = fmaf((t.w - st.x) * st.y, gs.w, bs.w);
      if constexpr (DROP != 0) {
        const int64_t quad = r0 * (I / 4) + static_cast<int64_t>(hp_lane + 16u * hb);     // a uniform 64-bit part plus a 32-bit one
        if constexpr (DROP == 1) {
          const uint32_t hsh = pair_hash(seed_in, quad), t8 = thr_in & 0xffu;
          t.x = (hsh & 0xffu) >= t8 ? t.x : 0.f; t.y = ((hsh >> 8) & 0xffu) >= t8 ? t.y : 0.f;
          t.z = ((hsh >> 16) & 0xffu) >= t8 ? t.z : 0.f; t.w = (hsh >> 24) >= t8 ? t.w : 0.f;
        } else {
          const float4 k = keep_scale4(seed_in, quad * 4, thr_in, 1.f);
          t.x *= k.x; t.y *= k.y; t.z *= k.z; t.w *= k.w;
        }
      }
      uint32_t h0, l0, h1, l1;
      split2_f16c(t.x, t.y, h0, l0);
      split2_f16c(t.z, t.w, h1, l1);
      uint8_t* p = img + 2 * kWfImg + (hb ? wo1 : wo0);
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + kWfPlane) = make_uint2(l0, l1);
    }
    return Q;
  };

  // ---- MFMA side (fused_bwd6.hip S3): 4 (o) x 2 (i) waves, wave tile 64 x 64 = 2 x 2 accumulators of 32 x 32; K = the stage's 32 rows
  // in two steps of 16; A = ga^T, B = u, both through transposing reads of the row-major images.
  //   row 16 kb + tr_row + 4 hi, column byte 64 (2 blk + tl) + tr_in:  base ^ (tl << 6) ^ (hi << 4), + 4096 kb + 1024 hi
  const int oh = wave >> 1, ih = wave & 1;                            // o in [64 oh, +64) of the tile's 256, i in [64 ih, +64) of its 128
  uint32_t base_a, base_u;
  {
    const int q4 = lane >> 4, tr_row = 8 * (q4 >> 1) + ((lane & 15) >> 2), tr_in = 32 * (q4 & 1) + 8 * (lane & 3);
    base_a = wf_lds_off(sP) + static_cast<uint32_t>((oh >> 1) * kWfImg + wf_img_off(tr_row, 64 * (2 * (oh & 1)) + tr_in));
    base_u = wf_lds_off(sP) + static_cast<uint32_t>(2 * kWfImg + wf_img_off(tr_row, 64 * (2 * ih) + tr_in));
  }
  f32x16w_t gw[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) gw[a][b][q] = 0.f;
  int Qacc = kWfEMin;                                                 // the accumulators are in units of 2^(140 + Su - Qacc)
  // The two waves that share a SIMD (w and w + 4: waves are dealt to SIMDs cyclically) run a half trip's two phases in opposite order
  // -- one multiplies while the other splits and stores (the phases touch different LDS buffers).
#ifdef ALLSET_ABL_WF_SAMEPHASE
  const bool mfma_first = true;
#else
  const bool mfma_first = wave < 4;
#endif
  auto mfma_stage = [&](int buf, int qb) {
    if (qb != Qacc) {                                                 // (uniform) the window moved up: bring the sums along
      const float s = wf_pow2(127 + Qacc - qb);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) gw[a][b] *= s;
      Qacc = qb;
    }
#ifdef ALLSET_ABL_WF_NOMFMA          // (ablation builds: timing only)
    return;
#endif
    const uint32_t ia = base_a + static_cast<uint32_t>(buf * kWfBuf), iu = base_u + static_cast<uint32_t>(buf * kWfBuf);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f16x8w_t wa[2][2], wb[2][2];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        const uint32_t a_lo = (ia ^ static_cast<uint32_t>(tl << 6)) + kb * 4096, a_hi = (ia ^ static_cast<uint32_t>((tl << 6) | 16)) + kb * 4096 + 1024;
        const uint32_t b_lo = (iu ^ static_cast<uint32_t>(tl << 6)) + kb * 4096, b_hi = (iu ^ static_cast<uint32_t>((tl << 6) | 16)) + kb * 4096 + 1024;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          wa[tl][pl] = wf_tr_frag(a_lo + pl * kWfPlane, a_hi + pl * kWfPlane);
          wb[tl][pl] = wf_tr_frag(b_lo + pl * kWfPlane, b_hi + pl * kWfPlane);
        }
      }
      constexpr int PA_[3] = {1, 0, 0}, PB_[3] = {0, 1, 0};          // l.h, h.l, h.h
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        gw[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[0][PA_[pr]], wb[0][PB_[pr]], gw[0][0], 0, 0, 0);
        gw[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[1][PA_[pr]], wb[0][PB_[pr]], gw[1][0], 0, 0, 0);
        gw[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[0][PA_[pr]], wb[1][PB_[pr]], gw[0][1], 0, 0, 0);
        gw[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[1][PA_[pr]], wb[1][PB_[pr]], gw[1][1], 0, 0, 0);
      }
    }
  };

  // stage k lives in LDS buffer k & 1 and was scaled from slot k % 3; register set s1 holds stage k + 1, s0 stage k + 2 (roles swap
  // every half trip).  Half trip k: multiply stage k | split stage k + 1 (its slot was posted a half trip ago) | request stage k + 3 |
  // post stage k + 2's maximum | clear stage k's slot -- three distinct slots, one barrier.
  int q0 = store_stage(s0, 0, 0, r_begin), q1 = Q;
  load_stage(s0, r_begin + 2 * kWfRows);
  __syncthreads();
  int sl0 = 0, sl1 = 1, sl2 = 2;                                      // slots of stages k, k + 1, k + 2
#ifdef ALLSET_ABL_WF_TIMING         // diagnostic builds only: cycles per segment of waves 0 and 4 of workgroup (0, 0)
  uint64_t tph[5] = {0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define WF_MARK(k) do { const uint64_t tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } while (0)
#else
#define WF_MARK(k) do {} while (0)
#endif
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * kWfRows) {
    const int qa = q0;
    if (mfma_first) mfma_stage(0, qa);
    WF_MARK(0);
    if (r0 + kWfRows < r_end) q1 = store_stage(s1, 1, sl1, r0 + kWfRows);
    WF_MARK(1);
    if (!mfma_first) mfma_stage(0, qa);
    WF_MARK(0);
    load_stage(s1, r0 + 3 * kWfRows);
    WF_MARK(2);
    post_max(s0, sl2, r0 + 2 * kWfRows);
    WF_MARK(3);
    if (tid == 0) sMax[sl0] = 0u;
    WF_SYNC();
    WF_MARK(4);
    if (r0 + kWfRows < r_end) {
      const int qb = q1;
      if (mfma_first) mfma_stage(1, qb);
      WF_MARK(0);
      if (r0 + 2 * kWfRows < r_end) q0 = store_stage(s0, 0, sl2, r0 + 2 * kWfRows);
      WF_MARK(1);
      if (!mfma_first) mfma_stage(1, qb);
      WF_MARK(0);
      load_stage(s0, r0 + 4 * kWfRows);
      WF_MARK(2);
      post_max(s1, sl0, r0 + 3 * kWfRows);
      WF_MARK(3);
      if (tid == 0) sMax[sl1] = 0u;
      WF_SYNC();
      WF_MARK(4);
    }
    const int t0 = sl0; sl0 = sl2; sl2 = sl1; sl1 = t0;               // two stages on: (k, k+1, k+2) -> (k+2, k+3, k+4) = slots (sl2, sl0, sl1)
  }

  // ---- epilogue: partial tile -> part_w[slice][O][I], unscaled; gw[a][b][q] is (o = 64 oh + 32 a + (q & 3) + 8 (q >> 2) + 4 (lane >> 5),
  // i = 64 ih + 32 b + (lane & 31))
  const float f1 = wf_pow2(Qacc - 13), f2 = wf_pow2(254 - S);         // 2^(Qacc - 140) 2^(-Su)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int o = o_base + 64 * oh + 32 * a + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        const int i = i_base + 64 * ih + 32 * b + (lane & 31);
        pw[static_cast<int64_t>(o) * I + i] = (gw[a][b][q] * f1) * f2;
      }
#ifdef ALLSET_ABL_WF_TIMING
  // wave 0 / wave 4 of the first workgroup: [0] MFMA phase, [1] split + LDS stores, [2] load issue, [3] maximum post, [4] barrier (cycles, all stages)
  __syncthreads();
  if (slice == 0 && tile == 0 && (tid == 0 || tid == 256)) for (int q = 0; q < 5; ++q) pw[(tid ? 8 : 0) + q] = static_cast<float>(tph[q]);
#endif
  // bias partial: the four row groups of a lane column fold by shuffles, the eight waves through LDS in a fixed order (only i-tile 0)
  if (want_b) {
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      float4 v = bsum[hb];
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
      }
      if (lane < 16) *reinterpret_cast<float4*>(&sRed[wave * kWfTO + 64 * hb + 4 * lane]) = v;
    }
    __syncthreads();
    if (tid < kWfTO) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += sRed[w * kWfTO + tid];
      g.part_b[static_cast<int64_t>(slice) * g.pb_stride + o_base + tid] = s;
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
