// The one-pass backward of the fused Linear (see fused_bwd.hip for the math and the operand images) at TWO waves per SIMD.
//
// fused_bwd.hip gives every wave its own complete gW accumulator (256 registers), so one wave owns a SIMD: its instruction
// stream issues at one instruction per ~5 cycles (tools/micro/mfma_filler.hip: a lone wave's VALU rate is 4.9 cycles per
// instruction, two waves on the SIMD reach 2.8) and nothing covers its LDS / HBM / MFMA-result waits -- the kernel spent
// about half of its 16.8 k cycles per 16-row chunk stalled (1819 instructions in the loop body).  Here TWO waves share a
// chunk ("pair"), each owning HALF of the input-feature columns i:
//     wave h of the pair:   gu[:, I/2 h .. ) = ga @ W[:, half h]     gW[:, half h] += ga^T @ u[:, half h]
// so its slice of gW is 128 accumulator registers, the other 128 are the wave's working set, 8 waves (4 pairs) are resident
// per CU and every per-row piece of vector work (mask, bf16 splits, LayerNorm backward, dropout hash, recomputation of u)
// is done on 64 instead of 128 columns per wave.  What the two waves must share is ga: each masks and splits its half of
// the gy columns into the pair's row-major image (the same swizzled [plane][16 rows][256 B] image as fused_bwd.hip), both
// read all of it -- row-major 16-byte fragments for backward-data, ds_read_b64_tr_b16 transposes for the weight gradient.
// u is private: wave h stores its half of the u planes in the very bytes its ga half occupied.  The LayerNorm backward's two
// row sums run over all I columns: each wave reduces its 64 and the pair swaps the partial sums through LDS.
// Three hand-offs per chunk, PAIR-local (LDS sequence numbers, see pair_signal / pair_wait; no workgroup barrier in the loop):
//     B1  ga image complete                      -> backward-data MFMAs, slab trips, partial row sums
//     B2  partial sums posted                    -> LayerNorm backward -> gx, u; transposed ga fragments into registers
//     B3  both waves are done with the ga image  -> u planes over it, weight-gradient MFMAs
// LDS: 96 KB W planes + 4 x 12 KB pair images + 8 x 1.25 KB slabs + gamma / beta + 8 sequence words = 155 KB.
#include <stdlib.h>

#include "common.h"

namespace allset {

using bf16x8p = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4p = __attribute__((ext_vector_type(4))) float;
using f32x16p = __attribute__((ext_vector_type(16))) float;
typedef short v4sp_t __attribute__((ext_vector_type(4)));
typedef __bf16 v2bfp_t __attribute__((ext_vector_type(2)));
union FragP { uint4 u; bf16x8p v; struct { v4sp_t lo, hi; } t; };
constexpr int kPBlock = 512;
constexpr int kPPairs = 4;

template <int CTRL>
__device__ __forceinline__ float dpp_fp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum_p(float v) {     // sum over the 16 lanes of a DPP row, result in every lane
  v += dpp_fp<0xB1>(v);
  v += dpp_fp<0x4E>(v);
  v += dpp_fp<0x141>(v);
  v += dpp_fp<0x140>(v);
  return v;
}
__device__ __forceinline__ bf16x8p tr_frag_p(const uint8_t* p, int half_stride) {
  FragP f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4sp_t*)(p));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4sp_t*)(p + half_stride));
  return f.v;
}
// byte offset of (row, column byte) in a [16][256 B] bf16 plane whose 64-byte chunks are XOR-swizzled by the row
__device__ __forceinline__ int img_off_p(int row, int colbyte) {
  return row * 256 + ((((colbyte >> 6) ^ row) & 3) << 6) + (colbyte & 63);
}
__device__ __forceinline__ uint32_t hash_mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; return x; }   // pair_hash's finaliser
// LDS-only barrier: global loads / stores stay in flight (s_barrier does not need them drained)
// One s_barrier per TICK.  A chunk is four segments of alternating kind,
//     S0 (vector)  ga: mask + bf16 planes into the pair's image
//     S1 (matrix)  backward-data MFMAs, slab trips, this wave's share of the LayerNorm row sums
//     S2 (vector)  LayerNorm backward -> gx, u recomputed, transposed ga fragments into registers
//     S3 (matrix)  u planes over the image, weight-gradient MFMAs
// and the two waves of a SIMD (waves w and w + 4 of the workgroup: pairs 0-1 and pairs 2-3) run the SAME program ONE TICK
// APART: while one is in a matrix segment its SIMD-mate is in a vector segment, in every tick.  That is the point of the
// exercise: tools/micro/mfma_valu_corun.hip shows an MFMA-only wave and a VALU-only wave on one SIMD run concurrently (1000 us
// of MFMAs + 570 us of v_fma finish in 1040 us), while waves that are all in the same phase -- what plain workgroup barriers
// or free-running pairs gave -- simply add their matrix and vector time (ablation: 0.25 + 0.24 ms + 0.10 ms of exposed
// load latency).  The tick barrier is also the pair's hand-off (B1 after S0, B2 after S1, B3 after S2).
// LDS-only wait in front of it: global loads / stores stay in flight across a tick.
#ifdef ALLSET_ABL2_NOBAR            // ablation builds only (tools/bwd_pair_ablation.py): timing without the barriers, results wrong
#define ALLSET_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define ALLSET_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// the lane id, re-derived where it is needed (two v_mbcnt) instead of living in a register across the row loop
#define ALLSET_FRESH_LANE_P(name) \
  int name = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); __asm__ volatile("" : "+v"(name))

template <bool HAS_LN, bool DROP_IN, bool RELU_IN, bool HAS_MASK>
__global__ __launch_bounds__(kPBlock, 2) void fused_linear_bwd_pair_kernel(
    const float* __restrict__ gy, int64_t ldg, const uint32_t* __restrict__ mask, float p_out, const float* __restrict__ W,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float p_in, uint64_t seed_in, float* gx, int64_t ldgx,
    float* __restrict__ part_ln, float* __restrict__ part_w, float* __restrict__ part_b, int64_t n,
    const uint64_t* __restrict__ seed_base, int64_t pstride_w, int64_t pstride_b, int64_t pstride_ln) {
  constexpr int OD = 128, ID = 128;
  constexpr int OQ = OD / 4, OQD = OQ / 2, T = OQ / 8;       // k-quarter of 32 o's per 16-lane group, 4 k-steps of 8
  constexpr int GS = ID * OQD;                               // dwords per k-quarter of a W plane
  constexpr int HW = ID / 2, NT = HW / 16;                   // columns and 16-column tiles per wave
  constexpr int PA = 256, PLA = 16 * PA, IMAGE = 3 * PLA;    // the pair's image: 3 planes x 16 rows x 256 B
  constexpr int SP = 20;                                     // slab pitch (floats): one 16 x 16 tile a trip, rows 80 B apart
  constexpr int SLAB = 16 * SP * 4;
  constexpr int OT = OD / 32, ITL = HW / 32;                 // 32 x 32 tiles of this wave's gW slice: 4 x 2
  __shared__ __attribute__((aligned(16))) uint32_t sW[3 * 4 * GS];
  __shared__ __attribute__((aligned(16))) uint8_t sImg[kPPairs * IMAGE];
  __shared__ __attribute__((aligned(16))) uint8_t sSlab[2 * kPPairs * SLAB];
  __shared__ __attribute__((aligned(16))) float sG[ID];
  __shared__ __attribute__((aligned(16))) float sB[ID];
  seed_in = resolve_seed(seed_base, seed_in);
  const int tid = threadIdx.x;
  if (tid < ID) { sG[tid] = HAS_LN ? gamma[tid] : 1.f; sB[tid] = HAS_LN ? beta[tid] : 0.f; }
  {
    uint32_t* const sWh = sW;
    uint32_t* const sWm = sW + 4 * GS;
    uint32_t* const sWl = sW + 8 * GS;
    for (int idx = tid; idx < (OD / 2) * ID; idx += kPBlock) {
      const int o = 2 * (idx / ID), i = idx % ID;                      // threads run along i: coalesced reads of W
      uint32_t ph, pm, pl;
      split3_bf16(W[o * ID + i], W[(o + 1) * ID + i], ph, pm, pl);
      const int e = o % OQ;
      const int off = (o / OQ) * GS + i * OQD + 4 * ((e / 8) ^ ((i / (64 / OQD)) % (OQD / 4))) + (e % 8) / 2;
      sWh[off] = ph; sWm[off] = pm; sWl[off] = pl;
    }
  }
  __syncthreads();

  const int lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave >> 1, h = wave & 1;
  uint8_t* const img = sImg + pair * IMAGE;
  float* const sT = reinterpret_cast<float*>(sSlab + wave * SLAB);
  const float* const sTp = reinterpret_cast<const float*>(sSlab + (wave ^ 1) * SLAB);     // the partner's slab (row sums)
  const bool late = wave >= 4;                    // the SIMD-mates of waves 0-3: one tick behind them
  const float inv_i = 1.f / static_cast<float>(ID);
  const float keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  const float keep_in = DROP_IN ? 1.f / (1.f - p_in) : 1.f;
  const uint32_t thr_in = drop_threshold(p_in);
  const int64_t n_chunks = (n + 15) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kPPairs;

  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = make_float4(0.f, 0.f, 0.f, 0.f);
  float gbs[2] = {0.f, 0.f};                      // bias gradient of o-tiles 2h, 2h+1: column (lane & 31), rows 8 (lane >> 5) .. +7
  f32x16p gw[OT][ITL];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int it = 0; it < ITL; ++it)
#pragma unroll
      for (int k = 0; k < 16; ++k) gw[ot][it][k] = 0.f;

  float ag[16];
  uint32_t am_bits = 0;
  auto rows_here = [&](int64_t chunk) -> int {
    const int64_t left = n - chunk * 16;
    return left >= 16 ? 16 : (left > 0 ? static_cast<int>(left) : 0);
  };
  // this lane's gy: row ri = lane & 15, o-columns 64 h + 16 g .. +15 (g = lane >> 4); rows past n clamp to a valid row and
  // are zeroed where used; a pair that has run out of chunks re-reads the last chunk (never consumed)
  auto request_rows = [&](int64_t chunk, int lane) {
    const int ri = lane & 15, g = lane >> 4;
    const int nr = rows_here(chunk);
#ifdef ALLSET_ABL2_HOT              // ablation: every pair re-reads chunk (its pair id): L2-resident operands, no HBM latency
    const int64_t c0 = pair;
#else
    const int64_t c0 = nr > 0 ? chunk : n_chunks - 1;
#endif
    const int lr = min(ri, max(nr, 1) - 1);
    if constexpr (HAS_MASK)       // "mask layout" (include/allset_hip.h): block (chunk, 64-column half h), dword (row, 32-column group)
      am_bits = (mask + (c0 * (OD / 64) + h) * 32)[(lr >> 2) * 8 + (lr & 3) * 2 + (g >> 1)];
    const char* base = reinterpret_cast<const char*>(gy + c0 * 16 * ldg);
    const uint32_t off = static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldg) * 4u + static_cast<uint32_t>((h * 64 + g * 16) * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(base + off + 16 * q);
      ag[4 * q] = v.x; ag[4 * q + 1] = v.y; ag[4 * q + 2] = v.z; ag[4 * q + 3] = v.w;
    }
  };
  // epilogue inputs, row-major: lane = rows it*4 + (lane >> 4), columns 64 h + 4 (lane & 15) .. +3
  float4 xr[4];
  float2 st[4];
  auto request_x = [&](int64_t chunk, int lane) {
    const int nr = rows_here(chunk);
#ifdef ALLSET_ABL2_HOT
    const int64_t c0 = pair;
#else
    const int64_t c0 = nr > 0 ? chunk : n_chunks - 1;
#endif
    const int nrc = max(nr, 1);
    const char* xb = reinterpret_cast<const char*>(x + c0 * 16 * ldx);
    const char* sb = reinterpret_cast<const char*>(stats + c0 * 16 * 2);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lr = min(it * 4 + (lane >> 4), nrc - 1);
      const uint32_t off = static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldx) * 4u + static_cast<uint32_t>((h * HW + (lane & 15) * 4) * 4);
      if constexpr (HAS_LN) st[it] = *reinterpret_cast<const float2*>(sb + lr * 8);
      xr[it] = *reinterpret_cast<const float4*>(xb + off);
    }
  };

  int64_t chunk = static_cast<int64_t>(blockIdx.x) * kPPairs + pair;
  request_rows(chunk, lane0);
  const int64_t trips = (n_chunks + stride - 1) / stride;              // the same for every wave of the grid: barriers inside
  if (late) ALLSET_TICK();
  for (int64_t trip = 0; trip < trips; ++trip, chunk += stride) {
    ALLSET_FRESH_LANE_P(lane);
    const int ri = lane & 15, g = lane >> 4;
    const int nrows = rows_here(chunk);                  // 0: this pair has no chunk left (it still takes part in the ticks)
    const bool valid = ri < nrows;
    request_x(chunk, lane);
    __builtin_amdgcn_sched_barrier(0);
    // ---- ga: epilogue mask of the forward, bf16 planes into the pair's image (this wave's 64 o-columns)
    if constexpr (HAS_MASK) {
      const uint32_t bits = valid ? (am_bits >> (4 * (g & 1))) : 0u;
#pragma unroll
      for (int j = 0; j < 16; ++j) ag[j] = (bits & (1u << (8 * (j & 3) + (j >> 2)))) ? ag[j] * keep_out : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) ag[j] = valid ? ag[j] : 0.f;
    }
    {
      const int wa_off = img_off_p(ri, h * 128 + g * 32);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t ph[4], pm[4], pl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3_bf16(ag[8 * q + 2 * j], ag[8 * q + 2 * j + 1], ph[j], pm[j], pl[j]);
        *reinterpret_cast<uint4*>(img + 0 * PLA + wa_off + 16 * q) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        *reinterpret_cast<uint4*>(img + 1 * PLA + wa_off + 16 * q) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        *reinterpret_cast<uint4*>(img + 2 * PLA + wa_off + 16 * q) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
      }
    }
    ALLSET_TICK();                                                                           // B1: the pair's ga image is complete
    __builtin_amdgcn_sched_barrier(0);
    // ---- backward-data: gu[:, this wave's 64 columns] = ga @ W on the bf16 matrix pipe (six of nine plane products)
    f32x4p acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) acc[tl] = f32x4p{0.f, 0.f, 0.f, 0.f};
    {
      // A fragment of k-step t: ga[ri][32 g + 8 t .. +7]; B fragment: W plane piece of (k-quarter g, column 64 h + 16 tl + ri, t)
      const int a_base = ri * 256 + 16 * 0, a_chunk = g;                 // colbyte = 64 g + 16 t: chunk g, in-chunk 16 t
      const int a_off = a_base + (((a_chunk ^ ri) & 3) << 6);
      const int wb_base = g * GS + (h * HW + ri) * OQD, wb_swz = (ri / (64 / OQD)) % (OQD / 4);
#ifdef ALLSET_ABL2_NOMFMA
      for (int t = 0; t < (p_in == 123.f ? T : 0); ++t) {
#else
#pragma unroll
      for (int t = 0; t < T; ++t) {
#endif
        FragP fa[3];
        fa[0].u = *reinterpret_cast<const uint4*>(img + 0 * PLA + a_off + 16 * t);
        fa[1].u = *reinterpret_cast<const uint4*>(img + 1 * PLA + a_off + 16 * t);
        fa[2].u = *reinterpret_cast<const uint4*>(img + 2 * PLA + a_off + 16 * t);
        int xo = wb_base + 4 * (t ^ wb_swz);
        __asm__ volatile("" : "+v"(xo));
#pragma unroll
        for (int tl = 0; tl < NT; tl += 2) {
          FragP b[6];
          const uint32_t* p = sW + xo + tl * 16 * OQD;
          b[0].u = *reinterpret_cast<const uint4*>(p);
          b[1].u = *reinterpret_cast<const uint4*>(p + 4 * GS);
          b[2].u = *reinterpret_cast<const uint4*>(p + 8 * GS);
          b[3].u = *reinterpret_cast<const uint4*>(p + 16 * OQD);
          b[4].u = *reinterpret_cast<const uint4*>(p + 16 * OQD + 4 * GS);
          b[5].u = *reinterpret_cast<const uint4*>(p + 16 * OQD + 8 * GS);
          acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2].v, b[0].v, acc[tl], 0, 0, 0);
          acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2].v, b[3].v, acc[tl + 1], 0, 0, 0);
          acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0].v, b[2].v, acc[tl], 0, 0, 0);
          acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0].v, b[5].v, acc[tl + 1], 0, 0, 0);
          acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1].v, b[1].v, acc[tl], 0, 0, 0);
          acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1].v, b[4].v, acc[tl + 1], 0, 0, 0);
          acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1].v, b[0].v, acc[tl], 0, 0, 0);
          acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1].v, b[3].v, acc[tl + 1], 0, 0, 0);
          acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0].v, b[1].v, acc[tl], 0, 0, 0);
          acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0].v, b[4].v, acc[tl + 1], 0, 0, 0);
          acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0].v, b[0].v, acc[tl], 0, 0, 0);
          acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0].v, b[3].v, acc[tl + 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);       // one block's fragments at a time: the partner wave covers the LDS latency
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- gu to row-major through the wave's slab, one 16 x 16 tile a trip: lane = rows it*4 + (lane>>4), columns 4 (lane & 15) .. +3
    // (the lanes whose four columns lie in the tile read it -- exec-masked loads, no selects)
    const int c4 = (lane & 15) * 4, r4 = lane >> 4;
    float4 gz[4];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sT[(4 * g + r) * SP + ri] = acc[tl][r];
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if ((ri >> 2) == tl) {
#pragma unroll
        for (int it = 0; it < 4; ++it) gz[it] = *reinterpret_cast<const float4*>(&sT[(it * 4 + r4) * SP + (c4 & 15)]);
      }
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- dropout-in mask, first half of the LayerNorm backward: this wave's share of the two row sums
    const float4 gam = *reinterpret_cast<const float4*>(&sG[h * HW + c4]);      // (re-read per chunk: 8 registers less across the loop)
    const float4 bet = *reinterpret_cast<const float4*>(&sB[h * HW + c4]);
    const uint64_t chunk_pair = static_cast<uint64_t>(chunk) * (16 * ID / 2);
    const uint32_t chunk_pair_lo = static_cast<uint32_t>(chunk_pair);
    const uint32_t seed_lo = static_cast<uint32_t>(seed_in);
    const uint32_t chunk_hi_term = __umul24(static_cast<uint32_t>(chunk_pair >> 32), 0x5EBCA7U) + static_cast<uint32_t>(seed_in >> 32);
    uint32_t kbits = 0, xbits = 0;      // dropout-in keep flags / "raw x > 0" flags of this lane's 16 elements (bit 4 it + j): one
    float s1[4], s2[4];               // register each instead of 16 (the keep factor is keep_in or 0, the relu mask a sign)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lrow = it * 4 + r4;
      const bool live = lrow < nrows;
      float4 v = gz[it];
      if constexpr (DROP_IN) {
        // pair index of (row r, column 64 h + c4) = chunk * 1024 + (lrow * 128 + 64 h + c4) / 2: the second term is < 1024, so
        // the 64-bit part is the wave-uniform chunk term and the lane adds its 10 bits with an OR (common.h pair_hash)
        float4 kp;
        const uint32_t lo = chunk_pair_lo | static_cast<uint32_t>((lrow * ID + h * HW + c4) >> 1);
        const uint32_t h0 = hash_mix((lo ^ seed_lo) * 0x9E3779B1U + chunk_hi_term);
        const uint32_t h1 = hash_mix(((lo + 1u) ^ seed_lo) * 0x9E3779B1U + chunk_hi_term);
        kp.x = (h0 & 0xffffu) >= thr_in ? keep_in : 0.f; kp.y = (h0 >> 16) >= thr_in ? keep_in : 0.f;
        kp.z = (h1 & 0xffffu) >= thr_in ? keep_in : 0.f; kp.w = (h1 >> 16) >= thr_in ? keep_in : 0.f;
        v.x *= kp.x; v.y *= kp.y; v.z *= kp.z; v.w *= kp.w;
        kbits |= (kp.x != 0.f ? 1u : 0u) << (4 * it) | (kp.y != 0.f ? 2u : 0u) << (4 * it) | (kp.z != 0.f ? 4u : 0u) << (4 * it) |
                 (kp.w != 0.f ? 8u : 0u) << (4 * it);
      }
      float4 t = xr[it];
      if (RELU_IN) {
        xbits |= (t.x > 0.f ? 1u : 0u) << (4 * it) | (t.y > 0.f ? 2u : 0u) << (4 * it) | (t.z > 0.f ? 4u : 0u) << (4 * it) |
                 (t.w > 0.f ? 8u : 0u) << (4 * it);
        t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
      }
      s1[it] = 0.f; s2[it] = 0.f;
      if constexpr (HAS_LN) {
        const float mean = st[it].x, rstd = st[it].y;
        float4 xh = make_float4((t.x - mean) * rstd, (t.y - mean) * rstd, (t.z - mean) * rstd, (t.w - mean) * rstd);
        if (!live) xh = make_float4(0.f, 0.f, 0.f, 0.f);
        dg.x = fmaf(v.x, xh.x, dg.x); dg.y = fmaf(v.y, xh.y, dg.y); dg.z = fmaf(v.z, xh.z, dg.z); dg.w = fmaf(v.w, xh.w, dg.w);
        db.x += v.x; db.y += v.y; db.z += v.z; db.w += v.w;
        v.x *= gam.x; v.y *= gam.y; v.z *= gam.z; v.w *= gam.w;
        float a = (v.x + v.y) + (v.z + v.w);
        float b = fmaf(v.x, xh.x, fmaf(v.y, xh.y, fmaf(v.z, xh.z, v.w * xh.w)));
        s1[it] = row16_sum_p(a);
        s2[it] = row16_sum_p(b);
        t = xh;
      }
      xr[it] = t;
      gz[it] = v;
      // the flag words are PACKED here (an opaque use: left alone the compiler sinks the packing to the flags' use after B2 and
      // keeps -- spills -- the sixteen keep factors until then)
      if constexpr (DROP_IN) __asm__ volatile("" : "+v"(kbits));
      if constexpr (RELU_IN) __asm__ volatile("" : "+v"(xbits));
    }
    if constexpr (HAS_LN) {
      if ((lane & 15) == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<float2*>(&sT[(it * 4 + r4) * 2]) = make_float2(s1[it], s2[it]);
      }
    }
    ALLSET_TICK();                                                                           // B2: the partner's row sums are posted
    if constexpr (HAS_LN) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float2 o = *reinterpret_cast<const float2*>(&sTp[(it * 4 + r4) * 2]);
        // (both waves add in the same order: half 0 + half 1)
        s1[it] = (h == 0 ? s1[it] + o.x : o.x + s1[it]) * inv_i;
        s2[it] = (h == 0 ? s2[it] + o.y : o.y + s2[it]) * inv_i;
      }
    }
    // ---- second half: gx, and u = dropout_in(LN(relu_in(x))) in place of x
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lrow = it * 4 + r4;
      const bool live = lrow < nrows;
      float4 o = gz[it];
      if constexpr (HAS_LN) {
        const float rstd = st[it].y;
        const float4 xh = xr[it];
        o = make_float4(rstd * (o.x - s1[it] - xh.x * s2[it]), rstd * (o.y - s1[it] - xh.y * s2[it]),
                        rstd * (o.z - s1[it] - xh.z * s2[it]), rstd * (o.w - s1[it] - xh.w * s2[it]));
      }
      if (RELU_IN) {
        o.x = (xbits >> (4 * it)) & 1u ? o.x : 0.f; o.y = (xbits >> (4 * it)) & 2u ? o.y : 0.f;
        o.z = (xbits >> (4 * it)) & 4u ? o.z : 0.f; o.w = (xbits >> (4 * it)) & 8u ? o.w : 0.f;
      }
#ifdef ALLSET_ABL2_HOT
      if (live)
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(gx + static_cast<int64_t>(pair) * 16 * ldgx) +
                                   static_cast<uint32_t>(lrow) * static_cast<uint32_t>(ldgx) * 4u + (h * HW + c4) * 4) = o;
#elif defined(ALLSET_ABL2_NOSTORE)
      if (live && o.x == 123.456f)
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(gx + chunk * 16 * ldgx) +
                                   static_cast<uint32_t>(lrow) * static_cast<uint32_t>(ldgx) * 4u + (h * HW + c4) * 4) = o;
#else
      if (live)
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(gx + chunk * 16 * ldgx) +
                                   static_cast<uint32_t>(lrow) * static_cast<uint32_t>(ldgx) * 4u + (h * HW + c4) * 4) = o;
#endif
      float4 u = xr[it];
      if constexpr (HAS_LN)
        u = make_float4(fmaf(u.x, gam.x, bet.x), fmaf(u.y, gam.y, bet.y), fmaf(u.z, gam.z, bet.z), fmaf(u.w, gam.w, bet.w));
      if constexpr (DROP_IN) {
        u.x = (kbits >> (4 * it)) & 1u ? u.x * keep_in : 0.f; u.y = (kbits >> (4 * it)) & 2u ? u.y * keep_in : 0.f;
        u.z = (kbits >> (4 * it)) & 4u ? u.z * keep_in : 0.f; u.w = (kbits >> (4 * it)) & 8u ? u.w * keep_in : 0.f;
      }
      xr[it] = u;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- weight-gradient A operands: ga^T fragments (32 o-columns x the chunk's 16 rows), three planes, transpose-read from
    // the pair's image; the bias gradient of this wave's two o-tiles falls out of them (v_dot2 with ones)
    ALLSET_FRESH_LANE_P(lane_w);
    const int q4 = lane_w >> 4, tr_r = (lane_w & 15) >> 2, tr_row = 8 * (q4 >> 1) + tr_r, tr_in = 32 * (q4 & 1) + 8 * (lane_w & 3);
    bf16x8p wa[OT][3];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wa[ot][pl] = tr_frag_p(img + pl * PLA + tr_row * PA + (((ot ^ tr_r) & 3) << 6) + tr_in, 4 * PA);
    ALLSET_TICK();                                                                           // B3: both waves are done with the ga image -- u may go over it
    if (part_b != nullptr) {
      const v2bfp_t ones = __builtin_bit_cast(v2bfp_t, 0x3f803f80u);
      auto colsum = [&](const bf16x8p (&w3)[3], float& acc_b) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          FragP f; f.v = w3[pl];
          acc_b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bfp_t, f.u.x), ones, acc_b, false);
          acc_b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bfp_t, f.u.y), ones, acc_b, false);
          acc_b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bfp_t, f.u.z), ones, acc_b, false);
          acc_b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bfp_t, f.u.w), ones, acc_b, false);
        }
      };
      if (h == 0) { colsum(wa[0], gbs[0]); colsum(wa[1], gbs[1]); }       // (h is wave-uniform: a scalar branch, no selects)
      else        { colsum(wa[2], gbs[0]); colsum(wa[3], gbs[1]); }
    }
    // ---- u planes over this wave's half of the image: row it*4 + (lane>>4), columns 64 h + 4 (lane & 15) .. +3 (8 bytes a plane)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float4 u = xr[it];
      uint32_t h0, m0, l0, h1, m1, l1;
      split3_bf16(u.x, u.y, h0, m0, l0);
      split3_bf16(u.z, u.w, h1, m1, l1);
      const int wo = img_off_p(it * 4 + (lane_w >> 4), (h * HW + (lane_w & 15) * 4) * 2);
      *reinterpret_cast<uint2*>(img + 0 * PLA + wo) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(img + 1 * PLA + wo) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(img + 2 * PLA + wo) = make_uint2(l0, l1);
    }
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the next chunk's gy half, requested once x / u are dead: the weight-gradient MFMAs and the tick barrier lie before its use
    request_rows(chunk + stride, lane_w);
    __builtin_amdgcn_sched_barrier(0);
    // ---- weight gradient: gW[o][this wave's i] += sum over the chunk's 16 rows of ga[r][o] u[r][i]; K = 16 = the chunk
#pragma unroll
    for (int it = 0; it < ITL; ++it) {
      bf16x8p wb[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        wb[pl] = tr_frag_p(img + pl * PLA + tr_row * PA + ((((2 * h + it) ^ tr_r) & 3) << 6) + tr_in, 4 * PA);
      constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#ifdef ALLSET_ABL2_NOMFMA
      for (int pr = 0; pr < (it == 0 ? 1 : 0); ++pr)
#else
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#endif
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
          gw[ot][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ot][PA_[pr]], wb[PB_[pr]], gw[ot][it], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    ALLSET_TICK();                                                                           // end of S3
  }
  if (!late) ALLSET_TICK();                        // (every wave passes 4 * trips + 1 barriers)

  // ---- per-pair partials: gW [O][I] (each wave its 64 columns), gb [O] (each wave its 64 o's), LayerNorm (dgamma, dbeta) [2][I]
  const int64_t slice = static_cast<int64_t>(blockIdx.x) * kPPairs + pair;
  const int lane = lane0;
  float* pw = part_w + slice * pstride_w;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int it = 0; it < ITL; ++it)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int o = ot * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
        pw[o * ID + h * HW + it * 32 + (lane & 31)] = gw[ot][it][k];
      }
  if (part_b != nullptr) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float sum = gbs[k] + __shfl_xor(gbs[k], 32);
      if (lane < 32) part_b[slice * pstride_b + (2 * h + k) * 32 + lane] = sum;
    }
  }
  if constexpr (HAS_LN) {
    float* pl = part_ln + slice * pstride_ln;
    float4 a = dg, b = db;
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
      b.x += __shfl_xor(b.x, off); b.y += __shfl_xor(b.y, off); b.z += __shfl_xor(b.z, off); b.w += __shfl_xor(b.w, off);
    }
    if (lane < 16) {
      *reinterpret_cast<float4*>(pl + h * HW + lane * 4) = a;
      *reinterpret_cast<float4*>(pl + ID + h * HW + lane * 4) = b;
    }
  }
}

}  // namespace allset

using namespace allset;

// 1 = the pair kernel takes this call (O = I = 128, no acc_in; bf16x6 mode).  OFF unless ALLSET_BWD_PAIR=1: measured on the
// GPU it does not beat the one-wave kernel (0.51-0.64 ms against 0.50 at [1M,128] x [128,128]; DESIGN.md section 6a has the three
// synchronisation schemes that were tried and the ablation of each) -- kept as the comparison arm of that analysis.
int fused_linear_bwd_pair_supported(int64_t O, int64_t I, int has_acc) {
  const char* e = getenv("ALLSET_BWD_PAIR");
  if (!(e && e[0] == '1')) return 0;
  return (dense_mfma_x6() && O == 128 && I == 128 && !has_acc) ? 1 : 0;
}

// Called by allset_fused_linear_bwd_all (fused_bwd.hip) after its argument checks; same partial-buffer contract (one slice
// per pair: grid * 4 = allset_fused_linear_bwd_all_slices(n)).
int launch_fused_linear_bwd_pair(unsigned grid, hipStream_t st, bool ln, bool drop, bool relu, bool hm, const float* gy,
                                 int64_t ldg, const uint32_t* mask, float p_out, const float* W, const float* x, int64_t ldx,
                                 const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                                 float* gx, int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n,
                                 const uint64_t* seed_base, int64_t psw, int64_t psb, int64_t psl) {
#define ALLSET_PAIR_K(LN, DI, RI, HM)                                                                                          \
  fused_linear_bwd_pair_kernel<LN, DI, RI, HM><<<grid, kPBlock, 0, st>>>(gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta,    \
                                                                        p_in, seed_in, gx, ldgx, part_ln, part_w, part_b, n,    \
                                                                        seed_base, psw, psb, psl)
#define ALLSET_PAIR_M(LN, DI, RI) do { if (hm) ALLSET_PAIR_K(LN, DI, RI, true); else ALLSET_PAIR_K(LN, DI, RI, false); } while (0)
  if (!relu) { if (ln) ALLSET_PAIR_M(true, false, false); else ALLSET_PAIR_M(false, false, false); }
  else if (ln) { if (drop) ALLSET_PAIR_M(true, true, true); else ALLSET_PAIR_M(true, false, true); }
  else { if (drop) ALLSET_PAIR_M(false, true, true); else ALLSET_PAIR_M(false, false, true); }
#undef ALLSET_PAIR_M
#undef ALLSET_PAIR_K
  return 0;
}
