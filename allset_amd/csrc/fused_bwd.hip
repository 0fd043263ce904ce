// The whole backward of the fused tall-skinny Linear in ONE pass over gy and x (reference MLP.forward, layers.py:571-579,
// as autograd differentiates it):
//
//   ga = gy * (mask ? 1/(1-p_out) : 0)                       relu / dropout epilogue of the forward (1-bit mask), or gy
//   gu = ga @ W            -> dropout-in mask -> LayerNorm backward -> relu-in mask -> gx      (backward-data)
//   u  = dropout_in(LN(relu_in(x)))   recomputed               gW = ga^T @ u,  gb = sum_rows ga (weight / bias gradient)
//   dgamma, dbeta                                                                              (LayerNorm parameters)
//
// Until round 2 these were two kernels (fused_linear_bwd_x6 + wgrad_x6) that each read gy, the mask, x and the row
// statistics: 1.6 + 1.08 GB per [1M,128] Linear where one pass needs 1.6 GB.  Why they were separate: the weight gradient
// reduces over ROWS, so its MFMA operands are 8 consecutive rows of one column -- the transpose of what backward-data holds
// in registers -- and a workgroup-level transpose stage next to the 96 KB of LDS-resident W planes does not fit in 160 KB.
// What makes one kernel possible on gfx950:
//   * ONE wave per SIMD (256-thread workgroups, one per CU): the wave owns the 512-register file.  It keeps its OWN full
//     gW [O x I] accumulator (256 registers at 128 x 128, the accumulator half of the file) and accumulates into it with
//     v_mfma_f32_32x32x16_bf16, whose K = 16 is exactly the 16 rows the wave processes per chunk -- no cross-wave
//     reduction, no barrier anywhere in the row loop; partial gW per wave, summed by allset_reduce_partials;
//   * the transposes are LDS READS: the wave writes the three bf16 planes of ga (already split for backward-data) and of u
//     row-major into its private 12 KB image and reads them back with ds_read_b64_tr_b16 (hardware 4x4 transpose:
//     lane c of a 16-lane group receives column c of a [4 rows][16 columns] block), i.e. no second split of ga, no
//     shuffles, no 16-bit scatter writes.  The image also serves as the slab that turns the backward-data accumulators
//     row-major (the three uses never overlap in time; one wave's LDS operations execute in order).
// Per 16-row chunk and wave: 192 v_mfma_f32_16x16x32_bf16 (backward-data, bf16x6) + 96 v_mfma_f32_32x32x16_bf16 (weight
// gradient, bf16x6): 6144 matrix cycles, against the ~1.6 KB/row of HBM traffic the chunk moves.
#include <stdlib.h>

#include "common.h"

namespace allset {

using bf16x8m = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4m = __attribute__((ext_vector_type(4))) float;
using f32x16m = __attribute__((ext_vector_type(16))) float;
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __bf16 v2bf_t __attribute__((ext_vector_type(2)));
union FragM { uint4 u; bf16x8m v; struct { v4s_t lo, hi; } t; };
constexpr int kMBlock = 256;
constexpr int kMWaves = kMBlock / kWave;

// dword offset of 16-byte piece t of (k-quarter g, column j) inside a W plane (same image as fused_mlp.hip)
template <int KQD, int GS>
__device__ __forceinline__ int mplane_off(int g, int j, int t) {
  constexpr int PIECES = KQD / 4, ROWS64 = 64 / KQD;
  return g * GS + j * KQD + 4 * (t ^ ((j / ROWS64) % PIECES));
}

// byte offset of (row, column byte) in a [16][PITCH bytes] row-major bf16 plane whose 64-byte chunks are XOR-swizzled by
// the row: the four rows a transpose-read touches land in four different bank quarters.
template <int PITCH>
__device__ __forceinline__ int img_off(int row, int colbyte) {
  constexpr int NCH = PITCH / 64;
  return row * PITCH + ((((colbyte >> 6) ^ row) & (NCH - 1)) << 6) + (colbyte & 63);
}

// sum over the 16 lanes of a DPP row (= the 16 lanes that share one matrix row in the row-major epilogue), result in
// every lane; four VALU instructions with DPP operands instead of four LDS permutes with their address registers
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);      // row_half_mirror
  v += dpp_f<0x140>(v);      // row_mirror
  return v;
}

__device__ __forceinline__ bf16x8m tr_frag(const uint8_t* p, int half_stride) {
  FragM f;
  f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(p));
  f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(p + half_stride));
  return f.v;
}

template <int OD, int ID, bool HAS_LN, bool DROP_IN, bool RELU_IN, bool HAS_MASK, bool HAS_ACC>
__global__ __launch_bounds__(kMBlock) void fused_linear_bwd_all_kernel(
    const float* __restrict__ gy, int64_t ldg, const uint32_t* __restrict__ mask, float p_out, const float* __restrict__ W,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float p_in, uint64_t seed_in, float* gx, int64_t ldgx,
    float* __restrict__ part_ln, float* __restrict__ part_w, float* __restrict__ part_b, int64_t n,
    const uint64_t* __restrict__ seed_base, const float* acc_in, int64_t ldacc, int64_t pstride_w, int64_t pstride_b,
    int64_t pstride_ln, float ln_inv) {                  // ln_inv: 1 / I (LayerNorm) or 0 (column affine): fused_bwd4.hip
  seed_in = resolve_seed(seed_base, seed_in);
  constexpr int OQ = OD / 4, OQD = OQ / 2, T = OQ / 8;
  constexpr int GS = ID * OQD;
  constexpr int NTILE = ID / 16, NH = ID / 64;
  constexpr int OT = OD / 32, IT = ID / 32;            // 32 x 32 tiles of gW
  constexpr int PA = OD * 2, PB = ID * 2;              // row pitch (bytes) of the ga / u images
  constexpr int PLA = 16 * PA, PLB = 16 * PB;          // bytes per plane
  constexpr int IMAGE = 3 * PLA > 3 * PLB ? 3 * PLA : 3 * PLB;       // ga planes, later u planes (never both at once)
  constexpr int SLAB = 16 * 32 * 4;                                  // 16 rows x 32 columns of fp32: accumulators -> row-major
  constexpr int REGION = IMAGE + SLAB;
  __shared__ __attribute__((aligned(16))) uint32_t sW[3 * 4 * GS];           // planes h, m, l of W, one after the other
  uint32_t* const sWh = sW;
  uint32_t* const sWm = sW + 4 * GS;
  uint32_t* const sWl = sW + 8 * GS;
  __shared__ __attribute__((aligned(16))) uint8_t sReg[kMWaves * REGION];
  __shared__ __attribute__((aligned(16))) float sG[ID];
  __shared__ __attribute__((aligned(16))) float sB[ID];
  const int tid = threadIdx.x;
  {   // all of a thread's weight pairs requested before the first is split and written (a plain loop kept ONE in flight: see fused_mlp.hip)
    constexpr int NPW = ((OD / 2) * ID + kMBlock - 1) / kMBlock;
    constexpr int NB = NPW > 16 ? 16 : NPW;                          // (batches of 16 pairs: 32 registers)
    for (int b0 = 0; b0 < NPW; b0 += NB) {
      float w0[NB], w1[NB];
#pragma unroll
      for (int it = 0; it < NB; ++it) {
        const int idx = tid + (b0 + it) * kMBlock;
        const int o = 2 * (idx / ID), i = idx % ID;                  // threads run along i: coalesced reads of W
        if (idx < (OD / 2) * ID) { w0[it] = W[o * ID + i]; w1[it] = W[(o + 1) * ID + i]; }
      }
#pragma unroll
      for (int it = 0; it < NB; ++it) {
        const int idx = tid + (b0 + it) * kMBlock;
        const int o = 2 * (idx / ID), i = idx % ID;
        if (idx < (OD / 2) * ID) {
          uint32_t ph, pm, pl;
          split3_bf16(w0[it], w1[it], ph, pm, pl);
          const int e = o % OQ;
          const int off = mplane_off<OQD, GS>(o / OQ, i, e / 8) + (e % 8) / 2;
          sWh[off] = ph; sWm[off] = pm; sWl[off] = pl;
        }
      }
    }
  }
  for (int idx = tid; idx < ID; idx += kMBlock) {
    sG[idx] = HAS_LN ? gamma[idx] : 1.f;
    sB[idx] = HAS_LN ? beta[idx] : 0.f;
  }
  __syncthreads();

  const int lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: chunk bases live in scalar registers
  const float inv_i = ln_inv;
  const float keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  const float keep_in = DROP_IN ? 1.f / (1.f - p_in) : 1.f;
  const uint32_t thr_in = drop_threshold(p_in);
  constexpr bool relu_in = RELU_IN;
  constexpr int NHO = OD / 64;
  const int64_t n_chunks = (n + 15) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kMWaves;
  uint8_t* reg = sReg + wave * REGION;
  float* sT = reinterpret_cast<float*>(reg + IMAGE);
  static_assert((16 / (64 / OQD)) % (OQD / 4) == 0, "column tiles must be whole swizzle periods");
  // One wave per SIMD leaves 256 registers for everything that is not the gW accumulator.  Lane-derived addresses (two dozen of
  // them) are therefore NOT loop invariants kept in registers: every phase of the row loop re-derives the few it needs from an
  // opaque copy of the lane id (a handful of integer instructions per 16-row chunk); hoisted, they were spilled to scratch and
  // every reload drained the prefetch queue (one in-order vmcnt for loads, stores and scratch on gfx9).
#define ALLSET_FRESH_LANE(name) int name = lane0; __asm__ volatile("" : "+v"(name))

  float4 dg[NH], db[NH];
#pragma unroll
  for (int hb = 0; hb < NH; ++hb) { dg[hb] = make_float4(0.f, 0.f, 0.f, 0.f); db[hb] = make_float4(0.f, 0.f, 0.f, 0.f); }
  float gbs[OT];                                   // bias gradient: column ot*32 + (lane & 31), rows 8 (lane >> 5) .. +7 of every chunk
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) gbs[ot] = 0.f;
  f32x16m gw[OT][IT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int k = 0; k < 16; ++k) gw[ot][it][k] = 0.f;

  float ag[OQ];
  uint32_t am_bits = 0;
  // Addresses: the chunk is wave-uniform, so a chunk's base pointers are scalar 64-bit values and a lane adds a 32-bit byte
  // offset (global_load ... v_off, s[base] form).  Rows past n are clamped to the last valid row of the chunk (unconditional
  // loads; dead rows are masked where used) -- on 32-bit local indices, not 64-bit row numbers.
  auto rows_here = [&](int64_t chunk) -> int {            // valid rows of this chunk, 0..16 (0: the wave has run out of chunks)
    const int64_t left = n - chunk * 16;
    return left >= 16 ? 16 : (left > 0 ? static_cast<int>(left) : 0);
  };
  auto request_rows = [&](int64_t chunk, int lane) {
    const int ri = lane & 15, g = lane >> 4;
    const int nr = rows_here(chunk);
    const int64_t c0 = nr > 0 ? chunk : n_chunks - 1;    // past the end: re-read the last chunk (never consumed)
    const int lr = min(ri, max(nr, 1) - 1);
    if constexpr (HAS_MASK) {
      // this lane's word of the activation mask: row ri of the chunk, columns g*OQ .. +OQ-1 (include/allset_hip.h "mask layout")
      const int m_l15 = ((g * OQ) % 64) / 4;
      const int m_word = (((g * OQ) / 64) * 4 + (lr >> 2)) * 8 + (lr & 3) * 2 + (m_l15 >> 3);
      am_bits = (mask + c0 * (NHO * 32))[m_word];
    }
#ifdef ALLSET_ABL_NOLOAD       // ablation builds only (tools/bwd_all_ablation.py)
    if (p_in == 123.f) {
#endif
    const char* base = reinterpret_cast<const char*>(gy + c0 * 16 * ldg);
    const uint32_t off = static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldg) * 4u + static_cast<uint32_t>(g * OQ * 4);
#pragma unroll
    for (int q = 0; q < OQ / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(base + off + 16 * q);
      ag[4 * q] = v.x; ag[4 * q + 1] = v.y; ag[4 * q + 2] = v.z; ag[4 * q + 3] = v.w;
    }
#ifdef ALLSET_ABL_NOLOAD
    }
#endif
  };

  // epilogue inputs, row-major: lane = rows it*4 + (lane>>4), columns hb*64 + c4 .. +3.  Requested at the very top of the chunk:
  // with one wave per SIMD the only latency hiding is distance, and a request from the previous chunk's weight-gradient phase
  // (40 more registers live there) tips the allocator into spilling an accumulator.
  float4 xr[NH][4];
  float2 st[4];
  auto request_x = [&](int64_t chunk, int lane) {
    const int c4 = (lane & 15) * 4;
    const int nr = rows_here(chunk);
    const char* xb = reinterpret_cast<const char*>(x + chunk * 16 * ldx);
    const char* sb = reinterpret_cast<const char*>(stats + chunk * 16 * 2);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lr = min(it * 4 + (lane >> 4), nr - 1);   // clamped, unconditional; dead rows are masked where used
      const uint32_t off = static_cast<uint32_t>(lr) * static_cast<uint32_t>(ldx) * 4u + static_cast<uint32_t>(c4 * 4);
#ifdef ALLSET_ABL_NOLOAD
      if (p_in == 123.f) {
#endif
      if constexpr (HAS_LN) st[it] = *reinterpret_cast<const float2*>(sb + lr * 8);
#pragma unroll
      for (int hb = 0; hb < NH; ++hb) xr[hb][it] = *reinterpret_cast<const float4*>(xb + off + hb * 256);
#ifdef ALLSET_ABL_NOLOAD
      }
#endif
    }
  };

  int64_t chunk = static_cast<int64_t>(blockIdx.x) * kMWaves + wave;
  request_rows(chunk, lane0);
  for (; chunk < n_chunks; chunk += stride) {
    // ---- ga: epilogue mask of the forward, bf16 planes
    ALLSET_FRESH_LANE(lane);
    const int ri = lane & 15, g = lane >> 4, c4 = ri * 4;
    const int m_shift = (((g * OQ) % 64) / 4) & 7;
    // writer of ga (backward-data layout): row ri, columns g*OQ .. +OQ-1 = OQ*2 bytes inside one 64-byte chunk of the image
    const int wa_off = img_off<PA>(ri, g * OQ * 2);
    const int wb_base = g * GS + ri * OQD, wb_swz = (ri / (64 / OQD)) % (OQD / 4);      // W-plane fragments: see load_b below
    const int nrows = rows_here(chunk);                 // 1..16 inside the loop
    const bool valid = ri < nrows;
    request_x(chunk, lane);                             // this chunk's x rows: the mask / split phase and the matrix phase cover them
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HAS_MASK) {
      const uint32_t bits = valid ? (am_bits >> m_shift) : 0u;
#pragma unroll
      for (int j = 0; j < OQ; ++j) ag[j] = (bits & (1u << (8 * (j & 3) + (j >> 2)))) ? ag[j] * keep_out : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < OQ; ++j) ag[j] = valid ? ag[j] : 0.f;
    }
    // the three bf16 planes go straight into the wave's row-major image, 16 bytes per plane and k-step: backward-data reads
    // its A fragments back from there (its own bytes, step by step -- 48 registers less across the matrix phase) and the
    // weight gradient reads them transposed
#pragma unroll
    for (int t = 0; t < T; ++t) {
      uint32_t ph[4], pm[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3_bf16(ag[8 * t + 2 * j], ag[8 * t + 2 * j + 1], ph[j], pm[j], pl[j]);
      *reinterpret_cast<uint4*>(reg + 0 * PLA + wa_off + 16 * t) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
      *reinterpret_cast<uint4*>(reg + 1 * PLA + wa_off + 16 * t) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
      *reinterpret_cast<uint4*>(reg + 2 * PLA + wa_off + 16 * t) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- backward-data: gu = ga @ W on the bf16 matrix pipe (six of nine plane products)
    // B fragments (W planes, LDS) are fetched ONE block ahead by hand: two sets of six 16-byte fragments, so the matrix pipe
    // never waits on LDS and the register cost of the look-ahead is fixed (one wave per SIMD: nobody else hides it)
    f32x4m acc[NTILE];
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) acc[tl] = f32x4m{0.f, 0.f, 0.f, 0.f};
    constexpr int NBLK = T * (NTILE / 2);
    FragM bw[2][6];
    // fragment address of (k-quarter g, column tl*16 + ri, k-step t) = wb_base + 4 (t ^ wb_swz) + tl * 16 * OQD dwords: the
    // swizzle term does not depend on the column tile (16 columns are a whole number of swizzle periods), so a k-step needs
    // ONE address register and the tiles are immediate offsets -- spelled out because the generic mplane_off() form made the
    // compiler hoist 2 x 16 per-block addresses out of the row loop and spill them
    int xo[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      xo[t] = wb_base + 4 * (t ^ wb_swz);
      __asm__ volatile("" : "+v"(xo[t]));             // opaque: one register per k-step, everything else immediates
    }
    auto load_b = [&](FragM (&b)[6], int blk) {
      const int t = blk / (NTILE / 2), tl = 2 * (blk % (NTILE / 2));
      const uint32_t* p = sW + xo[t] + tl * 16 * OQD;
      b[0].u = *reinterpret_cast<const uint4*>(p);
      b[1].u = *reinterpret_cast<const uint4*>(p + 4 * GS);
      b[2].u = *reinterpret_cast<const uint4*>(p + 8 * GS);
      b[3].u = *reinterpret_cast<const uint4*>(p + 16 * OQD);
      b[4].u = *reinterpret_cast<const uint4*>(p + 16 * OQD + 4 * GS);
      b[5].u = *reinterpret_cast<const uint4*>(p + 16 * OQD + 8 * GS);
    };
    FragM fa[2][3];                                     // A fragments of k-step t: [t & 1][h, m, l]
    auto load_a = [&](FragM (&a)[3], int t) {
      a[0].u = *reinterpret_cast<const uint4*>(reg + 0 * PLA + wa_off + 16 * t);
      a[1].u = *reinterpret_cast<const uint4*>(reg + 1 * PLA + wa_off + 16 * t);
      a[2].u = *reinterpret_cast<const uint4*>(reg + 2 * PLA + wa_off + 16 * t);
    };
    load_a(fa[0], 0);
    load_b(bw[0], 0);
#ifdef ALLSET_ABL_NOBD
    for (int blk = 0; blk < 1; ++blk) {
#else
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
#endif
      const int t = blk / (NTILE / 2), tl = 2 * (blk % (NTILE / 2));
      if (blk + 1 < NBLK) load_b(bw[(blk + 1) & 1], blk + 1);
      if (tl == NTILE - 2 && t + 1 < T) load_a(fa[(t + 1) & 1], t + 1);
      const FragM &fa_h = fa[t & 1][0], &fa_m = fa[t & 1][1], &fa_l = fa[t & 1][2];
      const FragM (&b)[6] = bw[blk & 1];
      acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l.v, b[0].v, acc[tl], 0, 0, 0);
      acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l.v, b[3].v, acc[tl + 1], 0, 0, 0);
      acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b[2].v, acc[tl], 0, 0, 0);
      acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b[5].v, acc[tl + 1], 0, 0, 0);
      acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b[1].v, acc[tl], 0, 0, 0);
      acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b[4].v, acc[tl + 1], 0, 0, 0);
      acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b[0].v, acc[tl], 0, 0, 0);
      acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b[3].v, acc[tl + 1], 0, 0, 0);
      acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b[1].v, acc[tl], 0, 0, 0);
      acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b[4].v, acc[tl + 1], 0, 0, 0);
      acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b[0].v, acc[tl], 0, 0, 0);
      acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b[3].v, acc[tl + 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next chunk's gy rows: requested here, where the operand planes are dead -- one wave per SIMD has no registers
    // for them during the matrix phase above; the epilogue and the weight-gradient phase (~2 us) cover the latency
    request_rows(chunk + stride, lane);
    __builtin_amdgcn_sched_barrier(0);
    // ---- gu to row-major through the wave's slab, 32 columns a trip: lane = rows it*4 + (lane>>4), columns hb*64 + c4 .. +3
    float4 gz[NH][4];
#pragma unroll
    for (int sx = 0; sx < ID / 32; ++sx) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sT[(4 * g + r) * 32 + tt * 16 + ri] = acc[sx * 2 + tt][r];
      // one wave, in-order LDS queue: no barrier needed, but the compiler must not move the vector reads above the
      // scalar writes (different access types) nor the next trip's writes above these reads
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const bool mine = (ri >> 3) == (sx & 1);            // the half of the lanes whose 4 columns lie in this trip
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float4 v = *reinterpret_cast<const float4*>(&sT[(it * 4 + (lane >> 4)) * 32 + (c4 & 31)]);
        if ((sx & 1) == 0) gz[sx >> 1][it] = v;                 // second trip of a half: keep what the first one delivered
        else gz[sx >> 1][it] = make_float4(mine ? v.x : gz[sx >> 1][it].x, mine ? v.y : gz[sx >> 1][it].y,
                                            mine ? v.z : gz[sx >> 1][it].z, mine ? v.w : gz[sx >> 1][it].w);
      }
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- dropout-in mask, LayerNorm backward, relu-in mask -> gx;  u = dropout_in(LN(relu_in(x))) replaces x in registers
    float4 gam[NH], bet[NH];
#pragma unroll
    for (int hb = 0; hb < NH; ++hb) {
      gam[hb] = *reinterpret_cast<const float4*>(&sG[hb * 64 + c4]);
      bet[hb] = *reinterpret_cast<const float4*>(&sB[hb * 64 + c4]);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lrow = it * 4 + (lane >> 4);
      const int64_t r = chunk * 16 + lrow;               // (the dropout hash is keyed by the global element index)
      const bool live = lrow < nrows;
      float s1 = 0.f, s2 = 0.f;
      float4 xraw[NH], kp[NH];                           // the raw x (relu-in mask) and the dropout-in keep factors of this row group
#pragma unroll
      for (int hb = 0; hb < NH; ++hb) {
        float4 v = gz[hb][it];
        if constexpr (DROP_IN) {
          keep_scale2(seed_in, r * ID + hb * 64 + c4, thr_in, keep_in, kp[hb].x, kp[hb].y);
          keep_scale2(seed_in, r * ID + hb * 64 + c4 + 2, thr_in, keep_in, kp[hb].z, kp[hb].w);
          v.x *= kp[hb].x; v.y *= kp[hb].y; v.z *= kp[hb].z; v.w *= kp[hb].w;
        }
        float4 t = xr[hb][it];
        xraw[hb] = t;
        if (relu_in) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
        if constexpr (HAS_LN) {
          const float mean = st[it].x, rstd = st[it].y;
          float4 xh = make_float4((t.x - mean) * rstd, (t.y - mean) * rstd, (t.z - mean) * rstd, (t.w - mean) * rstd);
          if (!live) xh = make_float4(0.f, 0.f, 0.f, 0.f);
          dg[hb].x = fmaf(v.x, xh.x, dg[hb].x); dg[hb].y = fmaf(v.y, xh.y, dg[hb].y);
          dg[hb].z = fmaf(v.z, xh.z, dg[hb].z); dg[hb].w = fmaf(v.w, xh.w, dg[hb].w);
          db[hb].x += v.x; db[hb].y += v.y; db[hb].z += v.z; db[hb].w += v.w;
          v.x *= gam[hb].x; v.y *= gam[hb].y; v.z *= gam[hb].z; v.w *= gam[hb].w;      // gh
          s1 += (v.x + v.y) + (v.z + v.w);
          s2 = fmaf(v.x, xh.x, s2); s2 = fmaf(v.y, xh.y, s2); s2 = fmaf(v.z, xh.z, s2); s2 = fmaf(v.w, xh.w, s2);
          t = xh;
        }
        xr[hb][it] = t;                                  // xhat (LayerNorm) or relu_in(x)
        gz[hb][it] = v;
      }
      if constexpr (HAS_LN) {
        s1 = row16_sum(s1) * inv_i;
        s2 = row16_sum(s2) * inv_i;
        const float rstd = st[it].y;
#pragma unroll
        for (int hb = 0; hb < NH; ++hb) {
          const float4 gh = gz[hb][it], xh = xr[hb][it];
          gz[hb][it] = make_float4(rstd * (gh.x - s1 - xh.x * s2), rstd * (gh.y - s1 - xh.y * s2),
                                   rstd * (gh.z - s1 - xh.z * s2), rstd * (gh.w - s1 - xh.w * s2));
        }
      }
      if (live) {
#pragma unroll
        for (int hb = 0; hb < NH; ++hb) {
          float4 o = gz[hb][it];
          if (relu_in) {
            o.x = xraw[hb].x > 0.f ? o.x : 0.f; o.y = xraw[hb].y > 0.f ? o.y : 0.f;
            o.z = xraw[hb].z > 0.f ? o.z : 0.f; o.w = xraw[hb].w > 0.f ? o.w : 0.f;
          }
          if constexpr (HAS_ACC) {        // gx = acc_in + ...: a second gradient branch of the same tensor, summed here
            const float4 ai = *reinterpret_cast<const float4*>(
                reinterpret_cast<const char*>(acc_in + chunk * 16 * ldacc) + static_cast<uint32_t>(lrow) * static_cast<uint32_t>(ldacc) * 4u +
                (hb * 64 + c4) * 4);                                                            // (may alias gx)
            o.x += ai.x; o.y += ai.y; o.z += ai.z; o.w += ai.w;
          }
#ifdef ALLSET_ABL_NOSTORE
          if (o.x == 123.456f)
#endif
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(gx + chunk * 16 * ldgx) +
                                     static_cast<uint32_t>(lrow) * static_cast<uint32_t>(ldgx) * 4u + (hb * 64 + c4) * 4) = o;
        }
      }
      // the Linear's input for the weight gradient, in place of x
#pragma unroll
      for (int hb = 0; hb < NH; ++hb) {
        float4 u = xr[hb][it];
        if constexpr (HAS_LN)
          u = make_float4(fmaf(u.x, gam[hb].x, bet[hb].x), fmaf(u.y, gam[hb].y, bet[hb].y),
                          fmaf(u.z, gam[hb].z, bet[hb].z), fmaf(u.w, gam[hb].w, bet[hb].w));
        if constexpr (DROP_IN) { u.x *= kp[hb].x; u.y *= kp[hb].y; u.z *= kp[hb].z; u.w *= kp[hb].w; }
        xr[hb][it] = u;
      }
      __builtin_amdgcn_sched_barrier(0);          // one row group at a time (measured: letting the four interleave is 2.5 % slower)
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- weight-gradient A operands: ga^T fragments (32 columns of o x the chunk's 16 rows), three planes, by transpose-
    // reads of the image; the bias gradient falls out of them (sum of the planes = ga exactly; v_dot2c with ones)
    // Transpose-reader: 16-lane group q4 -> k-half kh = q4 >> 1, column half ch = q4 & 1 of a 32-column tile; lane i of the group
    // supplies the address of row 8 kh + (i >> 2) [+4 for the second read], 4 columns at 4 (i & 3).
    ALLSET_FRESH_LANE(lane_w);
    const int q4 = lane_w >> 4, tr_r = (lane_w & 15) >> 2, tr_row = 8 * (q4 >> 1) + tr_r, tr_in = 32 * (q4 & 1) + 8 * (lane_w & 3);
    int ta_off[OT], tb_off[IT];
#pragma unroll
    for (int t = 0; t < OT; ++t) ta_off[t] = tr_row * PA + (((t ^ tr_r) & (PA / 64 - 1)) << 6) + tr_in;
#pragma unroll
    for (int t = 0; t < IT; ++t) tb_off[t] = tr_row * PB + (((t ^ tr_r) & (PB / 64 - 1)) << 6) + tr_in;
    bf16x8m wa[OT][3];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wa[ot][pl] = tr_frag(reg + pl * PLA + ta_off[ot], 4 * PA);
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the image is overwritten next (one wave: in-order LDS)
    {
      const v2bf_t ones = __builtin_bit_cast(v2bf_t, 0x3f803f80u);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          FragM f; f.v = wa[ot][pl];
          gbs[ot] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf_t, f.u.x), ones, gbs[ot], false);
          gbs[ot] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf_t, f.u.y), ones, gbs[ot], false);
          gbs[ot] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf_t, f.u.z), ones, gbs[ot], false);
          gbs[ot] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf_t, f.u.w), ones, gbs[ot], false);
        }
    }
    // ---- u planes into the image: row it*4 + (lane>>4), columns hb*64 + c4 .. +3 (8 bytes a plane)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int hb = 0; hb < NH; ++hb) {
        const float4 u = xr[hb][it];
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_bf16(u.x, u.y, h0, m0, l0);
        split3_bf16(u.z, u.w, h1, m1, l1);
        const int wo = img_off<PB>(it * 4 + (lane_w >> 4), (hb * 64 + (lane_w & 15) * 4) * 2);
        *reinterpret_cast<uint2*>(reg + 0 * PLB + wo) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(reg + 1 * PLB + wo) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(reg + 2 * PLB + wo) = make_uint2(l0, l1);
      }
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- weight gradient: gW[o][i] += sum over the chunk's 16 rows of ga[r][o] u[r][i]; 32 x 32 tiles, K = 16 = the chunk
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      bf16x8m wb[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wb[pl] = tr_frag(reg + pl * PLB + tb_off[it], 4 * PB);
      // the six plane products of an i-tile, each over ALL o-tiles before the next: OT independent accumulator chains between
      // two MFMAs on the same accumulator (a 32x32x16 result is not ready for 16 passes; one wave per SIMD has no neighbour to
      // fill the gap, so two interleaved chains -- enough at two waves per SIMD -- leave the pipe waiting)
      constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#ifdef ALLSET_ABL_NOWG
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
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the next chunk rewrites the image
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- per-wave partials: gW [O][I], gb [O], LayerNorm (dgamma, dbeta) [2][I]
  const int64_t slice = static_cast<int64_t>(blockIdx.x) * kMWaves + wave;
  const int lane = lane0, c4 = (lane0 & 15) * 4;
  float* pw = part_w + slice * pstride_w;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int o = ot * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
        pw[o * ID + it * 32 + (lane & 31)] = gw[ot][it][k];
      }
  if (part_b != nullptr) {                           // the two 8-row halves of a chunk live in lanes l and l + 32
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const float sum = gbs[ot] + __shfl_xor(gbs[ot], 32);
      if (lane < 32) part_b[slice * pstride_b + ot * 32 + lane] = sum;
    }
  }
  if constexpr (HAS_LN) {     // the 4 row groups of a lane column fold first
    float* pl = part_ln + slice * pstride_ln;
#pragma unroll
    for (int hb = 0; hb < NH; ++hb) {
      float4 a = dg[hb], b = db[hb];
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
        b.x += __shfl_xor(b.x, off); b.y += __shfl_xor(b.y, off); b.z += __shfl_xor(b.z, off); b.w += __shfl_xor(b.w, off);
      }
      if (lane < 16) {
        *reinterpret_cast<float4*>(pl + hb * 64 + c4) = a;
        *reinterpret_cast<float4*>(pl + ID + hb * 64 + c4) = b;
      }
    }
  }
}

}  // namespace allset

using namespace allset;

// fused_bwd4.hip: the same pass with the waves split by role (four vector waves, four matrix waves, one of each per SIMD)
int fused_linear_bwd_roles_supported(int64_t O, int64_t I, int has_acc);
unsigned fused_linear_bwd_roles_grid(int64_t n);
int launch_fused_linear_bwd_roles(unsigned grid, hipStream_t st, bool ln, bool drop, bool relu, bool hm, const float* gy,
                                  int64_t ldg, const uint32_t* mask, float p_out, const float* W, const float* x, int64_t ldx,
                                  const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                                  float* gx, int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n,
                                  const uint64_t* seed_base, int64_t psw, int64_t psb, int64_t psl, const float* acc_in,
                                  int64_t ldacc, const float* aux_g, const float* aux_w, int64_t gcb, int64_t xcb, int64_t gxcb,
                                  float ln_inv);

// fused_bwd6.hip: the split-role pass on two fp16 planes per operand (three MFMAs per product instead of six), eight vector
// waves: O = I = 128, every prologue (LayerNorm, column-affine, none) and acc_in, not the auxiliary columns; same grid and
// partial slices as the kernel above
int fused_linear_bwd_f16x3_supported(int64_t O, int64_t I, int has_ln, int norm_mode, int has_acc, int has_aux);
int launch_fused_linear_bwd_f16x3(unsigned grid, hipStream_t st, bool ln, bool drop, bool relu, bool hm, const float* gy, int64_t ldg,
                                  const uint32_t* mask, float p_out, const float* W, const float* x, int64_t ldx,
                                  const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                                  float* gx, int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n,
                                  const uint64_t* seed_base, int64_t psw, int64_t psb, int64_t psl, int64_t gcb, int64_t xcb,
                                  int64_t gxcb, const float* acc_in, int64_t ldacc, float ln_inv);

static inline unsigned bwd_all_grid(int64_t n) {
  int64_t blocks = ((n + 15) / 16 + kMWaves - 1) / kMWaves;
  return static_cast<unsigned>(blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks));      // one persistent workgroup per CU
}

// Built combinations (what the module surface of allset_amd/layers.py produces; anything else -> 0 and the caller keeps
// the two-kernel path): relu_in comes with a LayerNorm or alone (the Linear after the first one of an MLP), dropout_in only
// behind a relu_in; acc_in only on the plain Linear (PMA's residual block).
static inline bool bwd_all_combo(bool ln, bool drop, bool relu, bool mask, bool acc) {
  if (drop && !relu) return false;
  if (acc && (ln || drop || relu || mask)) return false;
  return true;
}

extern "C" int allset_fused_linear_bwd_all_supported(int64_t O, int64_t I, int has_ln, int drop_in, int relu_in, int has_mask,
                                                     int has_acc) {
  return ((O == 64 || O == 128) && (I == 64 || I == 128) &&
          bwd_all_combo(has_ln != 0, drop_in != 0, relu_in != 0, has_mask != 0, has_acc != 0)) ? 1 : 0;
}

extern "C" int allset_fused_linear_bwd_all_slices(int64_t n, int64_t* n_slices) {
  clear_error();
  ALLSET_REQUIRE(n_slices != nullptr && n >= 0, "fused_linear_bwd_all_slices: bad argument");
  *n_slices = static_cast<int64_t>(bwd_all_grid(n)) * kMWaves;
  return ALLSET_OK;
}

// The slice count of the kernel allset_fused_linear_bwd_all launches for these widths -- a pure function of its arguments: one
// per WORKGROUP for the split-role kernel (fused_bwd4.hip, O = I = 128: a single gW accumulator per CU), one per wave otherwise.
extern "C" int allset_fused_linear_bwd_all_slices_for(int64_t n, int64_t O, int64_t I, int has_acc, int64_t* n_slices) {
  clear_error();
  ALLSET_REQUIRE(n_slices != nullptr && n >= 0, "fused_linear_bwd_all_slices_for: bad argument");
  if (fused_linear_bwd_roles_supported(O, I, has_acc)) *n_slices = static_cast<int64_t>(fused_linear_bwd_roles_grid(n));
  else *n_slices = static_cast<int64_t>(bwd_all_grid(n)) * kMWaves;
  return ALLSET_OK;
}

template <int OD, int ID>
static void launch_bwd_all(unsigned grid, hipStream_t st, bool ln, bool drop, bool relu, bool hm, bool ha, const float* gy,
                           int64_t ldg, const uint32_t* mask, float p_out, const float* W, const float* x, int64_t ldx,
                           const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in, float* gx,
                           int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n, const uint64_t* seed_base,
                           const float* acc_in, int64_t ldacc, int64_t psw, int64_t psb, int64_t psl, float ln_inv) {
#define ALLSET_BWD_ALL_K(LN, DI, RI, HM, HA)                                                                                 \
  fused_linear_bwd_all_kernel<OD, ID, LN, DI, RI, HM, HA><<<grid, kMBlock, 0, st>>>(gy, ldg, mask, p_out, W, x, ldx, stats,     \
                                                                                   gamma, beta, p_in, seed_in, gx, ldgx,      \
                                                                                   part_ln, part_w, part_b, n, seed_base,     \
                                                                                   acc_in, ldacc, psw, psb, psl, ln_inv)
  if (ha) { ALLSET_BWD_ALL_K(false, false, false, false, true); return; }
#define ALLSET_BWD_ALL_M(LN, DI, RI)                                                               \
  do { if (hm) ALLSET_BWD_ALL_K(LN, DI, RI, true, false); else ALLSET_BWD_ALL_K(LN, DI, RI, false, false); } while (0)
  if (!relu) { if (ln) ALLSET_BWD_ALL_M(true, false, false); else ALLSET_BWD_ALL_M(false, false, false); }
  else if (ln) { if (drop) ALLSET_BWD_ALL_M(true, true, true); else ALLSET_BWD_ALL_M(true, false, true); }
  else { if (drop) ALLSET_BWD_ALL_M(false, true, true); else ALLSET_BWD_ALL_M(false, false, true); }
#undef ALLSET_BWD_ALL_M
#undef ALLSET_BWD_ALL_K
}

static bool bwd_block_cols_ok(int64_t cb, int64_t K, int64_t ld) {
  return cb >= 4 && cb <= K / 2 && (cb & (cb - 1)) == 0 && K % cb == 0 && ld == cb;
}

static int fused_linear_bwd_all_impl(const float* gy, int64_t ldg, const uint32_t* mask, float p_out, const float* W,
                                     const float* x, int64_t ldx, const float* stats, const float* gamma,
                                     const float* beta, int relu_in, float p_in, uint64_t seed_in, float* gx,
                                     int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n_slices,
                                     int64_t n, int64_t O, int64_t I, const uint64_t* seed_base, const float* acc_in,
                                     int64_t ldacc, int64_t part_stride, void* stream, int64_t gcb, int64_t xcb, int64_t gxcb,
                                     int norm_mode = ALLSET_NORM_LAYER, int arith = ALLSET_ARITH_AUTO) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "fused_linear_bwd_all: negative size");
  ALLSET_REQUIRE(arith == ALLSET_ARITH_AUTO || arith == ALLSET_ARITH_BF16X6 || arith == ALLSET_ARITH_FP16X3,
                 "fused_linear_bwd_all: arith must be ALLSET_ARITH_AUTO, ALLSET_ARITH_BF16X6 or ALLSET_ARITH_FP16X3");
  ALLSET_REQUIRE(norm_mode == ALLSET_NORM_LAYER || norm_mode == ALLSET_NORM_COLUMN_AFFINE, "fused_linear_bwd_all: norm_mode must be ALLSET_NORM_LAYER or ALLSET_NORM_COLUMN_AFFINE");
  ALLSET_REQUIRE(norm_mode == ALLSET_NORM_LAYER || stats != nullptr, "fused_linear_bwd_all: the column-affine prologue needs the {0, 1} row statistics its forward wrote, gamma (scale) and beta (shift)");
  const float ln_inv = norm_mode == ALLSET_NORM_COLUMN_AFFINE ? 0.f : 1.f / static_cast<float>(I);
  // part_stride = 0: three dense arrays [n_slices][O*I], [n_slices][O], [n_slices][2*I]; > 0: the three pointers address
  // the sections of ONE [n_slices][part_stride] buffer (one reduction launch for all of them)
  ALLSET_REQUIRE(part_stride == 0 || part_stride >= O * I, "fused_linear_bwd_all: part_stride smaller than a gW partial");
  const int64_t psw = part_stride ? part_stride : O * I, psb = part_stride ? part_stride : O, psl = part_stride ? part_stride : 2 * I;
  ALLSET_REQUIRE(p_in >= 0.f && p_in < 1.f && p_out >= 0.f && p_out < 1.f, "fused_linear_bwd_all: dropout p must be in [0,1)");
  const bool has_ln = stats != nullptr;
  ALLSET_REQUIRE(has_ln == (gamma != nullptr) && has_ln == (beta != nullptr), "fused_linear_bwd_all: stats, gamma and beta must come together");
  if (!allset_fused_linear_bwd_all_supported(O, I, has_ln, p_in > 0.f, relu_in, mask != nullptr, acc_in != nullptr)) {
    set_error("fused_linear_bwd_all: out=%lld in=%lld with LayerNorm=%d dropout_in=%d relu_in=%d mask=%d acc_in=%d is not built "
              "(allset_fused_linear_bwd_all_supported)", static_cast<long long>(O), static_cast<long long>(I), int(has_ln),
              int(p_in > 0.f), relu_in, int(mask != nullptr), int(acc_in != nullptr));
    return ALLSET_ERR_UNSUPPORTED;
  }
  const bool roles_kernel = fused_linear_bwd_roles_supported(O, I, acc_in != nullptr) != 0;
  const bool blocked = gcb != 0 || xcb != 0 || gxcb != 0;
  if (blocked && (!roles_kernel || acc_in != nullptr)) {
    set_error("fused_linear_bwd_all_blocked: column-blocked operands are read / written by the O = I = 128 split-role kernel only "
              "(allset_fused_linear_blocked_supported), without acc_in");
    return ALLSET_ERR_UNSUPPORTED;
  }
  const unsigned grid = roles_kernel ? fused_linear_bwd_roles_grid(n) : bwd_all_grid(n);
  ALLSET_REQUIRE(part_w != nullptr && n_slices == static_cast<int64_t>(grid) * (roles_kernel ? 1 : kMWaves),
                 "fused_linear_bwd_all: part_w must hold allset_fused_linear_bwd_all_slices_for() slices of [O][I]");
  ALLSET_REQUIRE(!has_ln || part_ln != nullptr, "fused_linear_bwd_all: LayerNorm partials buffer missing");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (part_stride) {
      ALLSET_HIP_CHECK(hipMemsetAsync(part_w, 0, static_cast<size_t>(n_slices) * part_stride * sizeof(float), st));
      return ALLSET_OK;
    }
    ALLSET_HIP_CHECK(hipMemsetAsync(part_w, 0, static_cast<size_t>(n_slices) * O * I * sizeof(float), st));
    if (part_b) ALLSET_HIP_CHECK(hipMemsetAsync(part_b, 0, static_cast<size_t>(n_slices) * O * sizeof(float), st));
    if (has_ln) ALLSET_HIP_CHECK(hipMemsetAsync(part_ln, 0, static_cast<size_t>(n_slices) * 2 * I * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && W && x && gx, "fused_linear_bwd_all: null pointer (x is always needed: the weight gradient recomputes the "
                                     "Linear's input; a Linear whose input needs no gradient keeps the two-kernel path)");
  ALLSET_REQUIRE(aligned16(gy) && aligned16(W) && aligned16(x) && aligned16(gx), "fused_linear_bwd_all: gy / W / x / gx must be 16-byte aligned");
  ALLSET_REQUIRE(gcb ? bwd_block_cols_ok(gcb, O, ldg) : (ldg >= O && ldg % 4 == 0), "fused_linear_bwd_all: gy must be 16-byte aligned rows (or a valid block width with ldg == it)");
  ALLSET_REQUIRE(xcb ? bwd_block_cols_ok(xcb, I, ldx) : (ldx >= I && ldx % 4 == 0), "fused_linear_bwd_all: x must be 16-byte aligned rows (or a valid block width with ldx == it)");
  ALLSET_REQUIRE(gxcb ? bwd_block_cols_ok(gxcb, I, ldgx) : (ldgx >= I && ldgx % 4 == 0), "fused_linear_bwd_all: gx must be 16-byte aligned rows (or a valid block width with ldgx == it)");
  ALLSET_REQUIRE(!blocked || n * 128 * 4 < (int64_t{1} << 32), "fused_linear_bwd_all_blocked: a blocked operand must stay below 4 GiB (32-bit lane offsets)");
  ALLSET_REQUIRE(acc_in == nullptr || (ldacc >= I && ldacc % 4 == 0 && aligned16(acc_in)),
                 "fused_linear_bwd_all: acc_in must be 16-byte aligned rows");
  ALLSET_REQUIRE(stats == nullptr || (reinterpret_cast<uintptr_t>(stats) & 7u) == 0, "fused_linear_bwd_all: stats must be 8-byte aligned");
  ALLSET_REQUIRE(ldg < (1 << 24) && ldx < (1 << 24) && ldgx < (1 << 24) && ldacc < (1 << 24),
                 "fused_linear_bwd_all: leading dimensions must stay below 2^24 elements (32-bit offsets inside a 16-row chunk)");
  const bool drop = p_in > 0.f, relu = relu_in != 0, hm = mask != nullptr, ha = acc_in != nullptr;
  const bool f16x3_built = roles_kernel && fused_linear_bwd_f16x3_supported(O, I, has_ln, norm_mode, ha, 0);
  if (arith == ALLSET_ARITH_FP16X3 && !f16x3_built) {
    set_error("fused_linear_bwd_all: ALLSET_ARITH_FP16X3 is built for O = I = 128 only (allset_fused_linear_arith_supported)");
    return ALLSET_ERR_UNSUPPORTED;
  }
#ifndef ALLSET_NO_F16X3
  if (f16x3_built && arith != ALLSET_ARITH_BF16X6) {
    launch_fused_linear_bwd_f16x3(grid, st, has_ln, drop, relu, hm, gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta, p_in, seed_in,
                                  gx, ldgx, part_ln, part_w, part_b, n, seed_base, psw, psb, psl, gcb, xcb, gxcb, acc_in, ldacc, ln_inv);
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
#endif
  if (roles_kernel) {                                            // one partial per workgroup
    launch_fused_linear_bwd_roles(grid, st, has_ln, drop, relu, hm, gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta, p_in,
                                  seed_in, gx, ldgx, part_ln, part_w, part_b, n, seed_base, psw, psb, psl, acc_in, ldacc,
                                  nullptr, nullptr, gcb, xcb, gxcb, ln_inv);
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
#define ALLSET_BWD_ALL_ARGS grid, st, has_ln, drop, relu, hm, ha, gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta, p_in, seed_in, \
                            gx, ldgx, part_ln, part_w, part_b, n, seed_base, acc_in, ldacc, psw, psb, psl, ln_inv
  // (O = I = 128 never arrives here: the split-role kernels above take it in either arithmetic.  Its 13 instantiations of this kernel
  //  were still built until round 6 -- profiles/r06_kernel_audit.md found them never launched, the dispatcher says why -- and are gone.)
  if (O == 128 && I == 64) launch_bwd_all<128, 64>(ALLSET_BWD_ALL_ARGS);
  else if (O == 64 && I == 128) launch_bwd_all<64, 128>(ALLSET_BWD_ALL_ARGS);
  else launch_bwd_all<64, 64>(ALLSET_BWD_ALL_ARGS);
#undef ALLSET_BWD_ALL_ARGS
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

// ---- the plain Linear with four auxiliary output columns (PMA's value projection + folded logits, fused_mlp.hip aux_out) ----
// gx = gy W + aux_g aux_w;  gW = gy^T x, gb = colsum(gy);  gaux_w = aux_g^T x [4, I], gaux_b = colsum(aux_g) [4] -- ONE pass over gy
// and x (the split-role kernel of fused_bwd4.hip) where the two-kernel path reads x three times.
extern "C" int allset_fused_linear_bwd_all_aux_supported(int64_t O, int64_t I) {
  return fused_linear_bwd_roles_supported(O, I, 0);
}

extern "C" int allset_fused_linear_bwd_all_aux(const float* gy, int64_t ldg, const float* W, const float* x, int64_t ldx,
                                               const float* aux_g, const float* aux_w, float* gx, int64_t ldgx, float* part,
                                               int64_t part_stride, int64_t n_slices, int64_t n, int64_t O, int64_t I,
                                               void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "fused_linear_bwd_all_aux: negative size");
  if (!allset_fused_linear_bwd_all_aux_supported(O, I)) {
    set_error("fused_linear_bwd_all_aux: out=%lld in=%lld is not built (allset_fused_linear_bwd_all_aux_supported)",
              static_cast<long long>(O), static_cast<long long>(I));
    return ALLSET_ERR_UNSUPPORTED;
  }
  // part: [n_slices][part_stride], sections gW [O*I] | gb [O] | gaux_w [4*I] | gaux_b [4]
  ALLSET_REQUIRE(part != nullptr && part_stride >= O * I + O + 4 * I + 4, "fused_linear_bwd_all_aux: part_stride smaller than O*I + O + 4*I + 4");
  const unsigned grid = fused_linear_bwd_roles_grid(n);
  ALLSET_REQUIRE(n_slices == static_cast<int64_t>(grid), "fused_linear_bwd_all_aux: part must hold allset_fused_linear_bwd_all_slices_for() slices");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(part, 0, static_cast<size_t>(n_slices) * part_stride * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && W && x && gx && aux_g && aux_w, "fused_linear_bwd_all_aux: null pointer");
  ALLSET_REQUIRE(ldg >= O && ldg % 4 == 0 && aligned16(gy) && aligned16(W), "fused_linear_bwd_all_aux: gy / W must be 16-byte aligned rows");
  ALLSET_REQUIRE(ldx >= I && ldx % 4 == 0 && aligned16(x), "fused_linear_bwd_all_aux: x must be 16-byte aligned rows");
  ALLSET_REQUIRE(ldgx >= I && ldgx % 4 == 0 && aligned16(gx), "fused_linear_bwd_all_aux: gx must be 16-byte aligned rows");
  ALLSET_REQUIRE(aligned16(aux_g) && aligned16(aux_w), "fused_linear_bwd_all_aux: aux_g [n,4] / aux_w [4,I] must be 16-byte aligned and dense");
  ALLSET_REQUIRE(ldg < (1 << 24) && ldx < (1 << 24) && ldgx < (1 << 24), "fused_linear_bwd_all_aux: leading dimensions must stay below 2^24 elements");
  launch_fused_linear_bwd_roles(grid, st, false, false, false, false, gy, ldg, nullptr, 0.f, W, x, ldx, nullptr, nullptr, nullptr, 0.f, 0,
                                gx, ldgx, part + O * I + O, part, part + O * I, n, nullptr, part_stride, part_stride, part_stride,
                                nullptr, 0, aux_g, aux_w, 0, 0, 0, 1.f / static_cast<float>(I));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_fused_linear_bwd_all(const float* gy, int64_t ldg, const uint32_t* mask, float p_out, const float* W,
                                           const float* x, int64_t ldx, const float* stats, const float* gamma,
                                           const float* beta, int relu_in, float p_in, uint64_t seed_in, float* gx,
                                           int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n_slices,
                                           int64_t n, int64_t O, int64_t I, const uint64_t* seed_base, const float* acc_in,
                                           int64_t ldacc, int64_t part_stride, void* stream) {
  return fused_linear_bwd_all_impl(gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta, relu_in, p_in, seed_in, gx, ldgx, part_ln,
                                   part_w, part_b, n_slices, n, O, I, seed_base, acc_in, ldacc, part_stride, stream, 0, 0, 0);
}

// allset_fused_linear_bwd_all with a choice of what the (stats, gamma, beta) prologue was in the forward (allset_fused_linear_fwd_nm):
// ALLSET_NORM_COLUMN_AFFINE -> gx = (gy W) * gamma through the relu / dropout masks, no row-mean terms; part_ln = per-slice column
// sums of gu * x and gu (the gradients of the caller's per-column scale and shift).
extern "C" int allset_fused_linear_bwd_all_nm(const float* gy, int64_t ldg, const uint32_t* mask, float p_out, const float* W,
                                              const float* x, int64_t ldx, const float* stats, const float* gamma,
                                              const float* beta, int norm_mode, int relu_in, float p_in, uint64_t seed_in, float* gx,
                                              int64_t ldgx, float* part_ln, float* part_w, float* part_b, int64_t n_slices,
                                              int64_t n, int64_t O, int64_t I, const uint64_t* seed_base, int64_t part_stride,
                                              void* stream) {
  return fused_linear_bwd_all_impl(gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta, relu_in, p_in, seed_in, gx, ldgx, part_ln,
                                   part_w, part_b, n_slices, n, O, I, seed_base, nullptr, 0, part_stride, stream, 0, 0, 0, norm_mode);
}


// Superset entry (ABI 11): every option of the three entries above in one call, plus the choice of arithmetic.
extern "C" int allset_fused_linear_bwd_all_ex(const float* gy, int64_t ldg, int64_t gy_block_cols, const uint32_t* mask, float p_out,
                                              const float* W, const float* x, int64_t ldx, int64_t x_block_cols, const float* stats,
                                              const float* gamma, const float* beta, int norm_mode, int relu_in, float p_in,
                                              uint64_t seed_in, float* gx, int64_t ldgx, int64_t gx_block_cols, float* part_ln,
                                              float* part_w, float* part_b, int64_t n_slices, int64_t n, int64_t O, int64_t I,
                                              const uint64_t* seed_base, const float* acc_in, int64_t ldacc, int64_t part_stride,
                                              int arith, void* stream) {
  return fused_linear_bwd_all_impl(gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta, relu_in, p_in, seed_in, gx, ldgx, part_ln,
                                   part_w, part_b, n_slices, n, O, I, seed_base, acc_in, ldacc, part_stride, stream, gy_block_cols,
                                   x_block_cols, gx_block_cols, norm_mode, arith);
}

// 1 = the fused Linear of this direction (0 forward, 1 one-pass backward) has a kernel in arithmetic `arith` at these widths
// (AUTO and BF16X6: wherever the entry itself is built; FP16X3: the O = I = 128 split-role kernels -- the forward behind a LayerNorm
// prologue or without a norm, not in the column-affine mode, and not with auxiliary output columns).
extern "C" int allset_fused_linear_arith_supported(int direction, int64_t K, int64_t N, int has_ln, int norm_mode, int arith) {
  if (!((K == 64 || K == 128) && (N == 64 || N == 128))) return 0;
  if (arith == ALLSET_ARITH_AUTO || arith == ALLSET_ARITH_BF16X6) return 1;
  if (arith != ALLSET_ARITH_FP16X3) return 0;
#ifdef ALLSET_NO_F16X3
  return 0;
#else
  if (K != 128 || N != 128) return 0;
  return direction == 0 ? ((!has_ln || norm_mode == ALLSET_NORM_LAYER) ? 1 : 0) : 1;      // (forward: not in the column-affine mode)
#endif
}

// The same pass with gy / x / gx COLUMN-BLOCKED ([cols / cb][n][cb], ld == cb; 0 = row-major): see allset_fused_linear_fwd_blocked.
extern "C" int allset_fused_linear_bwd_all_blocked(const float* gy, int64_t ldg, int64_t gy_block_cols, const uint32_t* mask,
                                                   float p_out, const float* W, const float* x, int64_t ldx, int64_t x_block_cols,
                                                   const float* stats, const float* gamma, const float* beta, int relu_in,
                                                   float p_in, uint64_t seed_in, float* gx, int64_t ldgx, int64_t gx_block_cols,
                                                   float* part_ln, float* part_w, float* part_b, int64_t n_slices, int64_t n,
                                                   int64_t O, int64_t I, const uint64_t* seed_base, int64_t part_stride,
                                                   void* stream) {
  return fused_linear_bwd_all_impl(gy, ldg, mask, p_out, W, x, ldx, stats, gamma, beta, relu_in, p_in, seed_in, gx, ldgx, part_ln,
                                   part_w, part_b, n_slices, n, O, I, seed_base, nullptr, 0, part_stride, stream, gy_block_cols,
                                   x_block_cols, gx_block_cols);
}
