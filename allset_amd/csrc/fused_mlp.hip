// Fused tall-skinny Linear for the AllSet dense tail on gfx950 (reference MLP.forward, layers.py:571-579).
//
// Two kernel families live here.  The DEFAULT is the bf16x6 family further down (fused_linear_*_x6_kernel: fp32-accurate
// arithmetic on the bf16 matrix pipe, HBM-bound); the native fp32-MFMA family that this header describes first is the
// comparison arm (ALLSET_DENSE_MFMA=f32) and the documentation of the row-organised scheme both share.
//
//   y = epilogue( prologue(x) @ W^T + b )
//     prologue : [relu] -> [LayerNorm(gamma, beta)] -> [dropout p_in]      applied to the A operand IN REGISTERS
//     epilogue : [relu] -> [dropout p_out]                                   applied to the MFMA accumulators
//
// Shape regime: n ~ 1e6 rows, K = in features and N = out features in {64, 128}.  At these widths a row block of
// the activation matrix is tiny and the weight matrix (<= 64 KiB) is shared by everything, so the kernel is
// organised around ROWS, not output tiles:
//   * persistent workgroups of 8 waves; W^T is staged once per workgroup in LDS (pitch K+1: the B-operand read
//     W[o0 + (l&31)][k] is conflict-free);
//   * one wave owns 32 complete rows at a time.  v_mfma_f32_32x32x2_f32 wants A[i = l&31][k = l>>5]: lane l
//     supplies row (l & 31) for the k-slot (l >> 5) of each step.  Which two columns a step covers is free as long
//     as A and B agree, so step j covers columns j (lower half-wave) and K/2 + j (upper half-wave): every lane
//     keeps a CONTIGUOUS half of its row in registers -- K/8 fully-used 16-byte loads landing directly in the
//     operand registers, LayerNorm statistics one xor-32 shuffle away, and no LDS spent on A;
//   * the wave then issues K/2 k-steps x (N/32) MFMAs (exact fp32, the 157 TF class) and runs the epilogue on its
//     accumulators; waves never synchronise with each other after the W load, so one wave's global loads and
//     epilogue overlap the other wave's MFMAs on the same SIMD (2 waves / SIMD).
// HBM traffic is 1 read + 1 write of the activation matrix per Linear instead of the 3 reads + 3 writes of the
// unfused norm -> GEMM -> activation chain.
#include <stdlib.h>

#include "common.h"

namespace allset {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kFusedBlock = 512;
constexpr int kFusedWaves = kFusedBlock / kWave;

// ---- the same forward on the bf16 matrix pipe (bf16x6, see common.h) -----------------------------------------------
// v_mfma_f32_16x16x32_bf16: lane l supplies A[i = l&15][k = 8*(l>>4) .. +7] as 8 packed bf16 (4 VGPRs), B likewise for
// column (l&15); C/D: column l&15, rows 4*(l>>4) + 0..3.  The k-order is free, so lane (i, g = l>>4) keeps the
// CONTIGUOUS quarter g of row i in registers (K/4 floats -> three planes of K/8 dwords) and step t uses its local
// columns 8t..8t+7.  One wave owns 16 complete rows at a time -- half the registers of the 32-row fp32 kernel, which
// buys what that kernel could not afford at two waves per SIMD: the rows of the next TWO chunks are always in flight
// into two register buffers (2 x 8 KiB per wave, 32 MiB chip-wide), so the activation stream never drains while a
// wave splits, multiplies and stores.
//   * W^T is split once per workgroup into three bf16 planes in LDS, laid out [k-quarter][column][K/8 dwords] with the
//     16-byte piece index XOR-swizzled by the column, so a B fragment is one conflict-free ds_read_b128;
//   * the accumulators (one column x 4 rows per lane) take one trip through the wave's LDS slab and leave as 16-byte
//     stores of 4 consecutive columns (dword stores from the MFMA layout are issue-bound at ~5 B/clk/CU); bias / relu /
//     dropout run on the row-major side, where one hash covers two neighbours.
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
union Frag8 { uint4 u; bf16x8 v; };
constexpr int kX6Block = 512;
constexpr int kX6Waves = kX6Block / kWave;

// dword offset of 16-byte piece t of (k-quarter g, column j) inside a plane
template <int KQD, int GS>
__device__ __forceinline__ int plane_off(int g, int j, int t) {
  constexpr int PIECES = KQD / 4, ROWS64 = 64 / KQD;
  return g * GS + j * KQD + 4 * (t ^ ((j / ROWS64) % PIECES));
}

// Waves per workgroup (round 3): 12 = three per SIMD for the variants with a LayerNorm prologue -- their vector work (statistics,
// normalisation, dropout hash, three-plane split, epilogue) is 7x the matrix pipe's issue slots, and a third wave per SIMD fills
// what two leave idle when both sit in the same kind of phase (same box: 0.251 -> 0.241 ms at [1M,128]); at 168 registers per
// wave there is room for ONE chunk of prefetch, requested after the matrix phase.  The plain variants keep 8 waves and two
// chunks in flight: with little vector work they are bound by bytes in flight (8 -> 12 waves measured 0.217 -> 0.228 there).
template <bool HAS_LN>
constexpr int fwd_x6_waves() {
#ifdef ALLSET_FWD_WAVES
  return ALLSET_FWD_WAVES;
#else
  return HAS_LN ? 12 : 8;
#endif
}
template <int KD, int ND, bool HAS_LN, bool DROP_IN, bool DROP_OUT>
__global__ __launch_bounds__(fwd_x6_waves<HAS_LN>() * kWave) void fused_linear_fwd_x6_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, int relu_in, float p_in, uint64_t seed_in, const float* __restrict__ W,
    const float* __restrict__ bias, int relu_out, float p_out, uint64_t seed_out, float* __restrict__ y,
    int64_t ldy, float* __restrict__ stats, int64_t n, const uint64_t* __restrict__ seed_base,
    uint8_t* __restrict__ mask_out, const float* __restrict__ aux_w, const float* __restrict__ aux_b,
    float* __restrict__ aux_out, float ln_inv) {          // ln_inv: 1 / K (LayerNorm) or 0 (column affine, eps = 1): fused_fwd2.hip
  seed_in = resolve_seed(seed_base, seed_in);
  seed_out = resolve_seed(seed_base, seed_out);
  constexpr int kF6Waves = fwd_x6_waves<HAS_LN>();
  constexpr int kF6Block = kF6Waves * kWave;
  constexpr int kF6Depth = kF6Waves > 8 ? 1 : 2;   // chunks in flight per wave
  constexpr int KQ = KD / 4;                       // columns per lane
  constexpr int KQD = KQ / 2;                      // dwords per (quarter, column) row of a plane
  constexpr int T = KQ / 8;                        // MFMA k-steps
  constexpr int GS = ND * KQD;
  constexpr int NTILE = ND / 16;
  __shared__ __attribute__((aligned(16))) uint32_t sWh[4 * GS];
  __shared__ __attribute__((aligned(16))) uint32_t sWm[4 * GS];
  __shared__ __attribute__((aligned(16))) uint32_t sWl[4 * GS];
  __shared__ __attribute__((aligned(16))) float sG[KD];
  __shared__ __attribute__((aligned(16))) float sBeta[KD];
  __shared__ __attribute__((aligned(16))) float sBias[ND];
  __shared__ __attribute__((aligned(16))) float sTrans[kF6Waves * 16 * 64];
  __shared__ __attribute__((aligned(16))) float sAux[4 * KD + 4];         // 4 auxiliary output columns: weights, bias
  const int tid = threadIdx.x;
  if (aux_out != nullptr) {
    for (int idx = tid; idx < 4 * KD; idx += kF6Block) sAux[idx] = aux_w[idx];
    if (tid < 4) sAux[4 * KD + tid] = aux_b ? aux_b[tid] : 0.f;
  }
  // ALL of a thread's weight pairs are requested before the first is split and written (as a plain loop the compiler kept ONE in
  // flight -- load, s_waitcnt vmcnt(0), three ds_writes, branch -- 4 to 16 dependent L2 round trips at the head of a kernel that,
  // at dataset scale, runs for ~5 us in all; found in the ISA of csrc/fused_bf16.hip's copy, round 6)
  {
    constexpr int NPW = (ND * KD / 2 + kF6Block - 1) / kF6Block;
    float2 wreg[NPW];
#pragma unroll
    for (int it = 0; it < NPW; ++it) {
      const int idx = tid + it * kF6Block;
      const int j = idx / (KD / 2), k = 2 * (idx % (KD / 2));
      if (ND * KD / 2 % kF6Block == 0 || idx < ND * KD / 2) wreg[it] = *reinterpret_cast<const float2*>(W + j * KD + k);
    }
#pragma unroll
    for (int it = 0; it < NPW; ++it) {
      const int idx = tid + it * kF6Block;
      const int j = idx / (KD / 2), k = 2 * (idx % (KD / 2));
      if (ND * KD / 2 % kF6Block == 0 || idx < ND * KD / 2) {
        uint32_t ph, pm, pl;
        split3_bf16(wreg[it].x, wreg[it].y, ph, pm, pl);
        const int e = k % KQ;
        const int off = plane_off<KQD, GS>(k / KQ, j, e / 8) + (e % 8) / 2;
        sWh[off] = ph; sWm[off] = pm; sWl[off] = pl;
      }
    }
  }
  for (int idx = tid; idx < KD; idx += kF6Block) {
    sG[idx] = HAS_LN ? gamma[idx] : 1.f;
    sBeta[idx] = HAS_LN ? beta[idx] : 0.f;
  }
  for (int idx = tid; idx < ND; idx += kF6Block) sBias[idx] = bias ? bias[idx] : 0.f;
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int ri = lane & 15, g = lane >> 4;
  const float inv_k = ln_inv;
  const float keep_in = DROP_IN ? 1.f / (1.f - p_in) : 1.f;
  const float keep_out = DROP_OUT ? 1.f / (1.f - p_out) : 1.f;
  const uint32_t thr_in = drop_threshold(p_in), thr_out = drop_threshold(p_out);
  const int64_t n_chunks = (n + 15) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kF6Waves;
  float* sT = sTrans + wave * (16 * 64);

  // Loads are unconditional on a clamped row (no exec-mask branches around memory instructions, whose joins cost
  // s_waitcnt vmcnt(0) and drain the prefetch); rows past n are zeroed when the operand is consumed.
  auto request_row = [&](float (&a)[KQ], int64_t chunk) {
    int64_t row = chunk * 16 + ri;
    row = row < n ? row : n - 1;
#ifdef ALLSET_ABLATE_NOLOAD
    if (p_in == 123.f) {
#endif
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx + g * KQ);
#pragma unroll
    for (int q = 0; q < KQ / 4; ++q) {
      const float4 v = xr[q];
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
#ifdef ALLSET_ABLATE_NOLOAD
    }
#endif
  };

  auto process = [&](float (&a)[KQ], int64_t chunk) {
    const int64_t row = chunk * 16 + ri;
    const bool valid = row < n;
    if (!valid) {
#pragma unroll
      for (int j = 0; j < KQ; ++j) a[j] = 0.f;
    }
    if (relu_in) {
#pragma unroll
      for (int j = 0; j < KQ; ++j) a[j] = fmaxf(a[j], 0.f);
    }
    if constexpr (HAS_LN) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < KQ; ++j) s += a[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s * inv_k;
      float q2 = 0.f;
#pragma unroll
      for (int j = 0; j < KQ; ++j) { a[j] -= mean; q2 = fmaf(a[j], a[j], q2); }
      q2 += __shfl_xor(q2, 16);
      q2 += __shfl_xor(q2, 32);
      const float rstd = rsqrtf(q2 * inv_k + eps);
      if (valid && g == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
#pragma unroll
      for (int jb = 0; jb < KQ; jb += 4) {
        const float4 g0 = *reinterpret_cast<const float4*>(&sG[g * KQ + jb]);
        const float4 b0 = *reinterpret_cast<const float4*>(&sBeta[g * KQ + jb]);
        a[jb + 0] = fmaf(a[jb + 0] * rstd, g0.x, b0.x); a[jb + 1] = fmaf(a[jb + 1] * rstd, g0.y, b0.y);
        a[jb + 2] = fmaf(a[jb + 2] * rstd, g0.z, b0.z); a[jb + 3] = fmaf(a[jb + 3] * rstd, g0.w, b0.w);
      }
    }
    if constexpr (DROP_IN) {
#pragma unroll
      for (int j = 0; j < KQ; j += 2) {
        float k0, k1;
        keep_scale2(seed_in, row * KD + g * KQ + j, thr_in, keep_in, k0, k1);
        a[j] *= k0; a[j + 1] *= k1;
      }
    }
    if (aux_out != nullptr) {
      // four extra output columns in plain fp32 FMAs (PMA's folded attention logits ride along with the value
      // projection: a [n,K] x [K,4] product is all bandwidth, and the rows are already in registers here)
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int j = 0; j < KQ; j += 4) {
        const float4 w0 = *reinterpret_cast<const float4*>(&sAux[0 * KD + g * KQ + j]);
        const float4 w1 = *reinterpret_cast<const float4*>(&sAux[1 * KD + g * KQ + j]);
        const float4 w2 = *reinterpret_cast<const float4*>(&sAux[2 * KD + g * KQ + j]);
        const float4 w3 = *reinterpret_cast<const float4*>(&sAux[3 * KD + g * KQ + j]);
        s0 = fmaf(a[j], w0.x, fmaf(a[j + 1], w0.y, fmaf(a[j + 2], w0.z, fmaf(a[j + 3], w0.w, s0))));
        s1 = fmaf(a[j], w1.x, fmaf(a[j + 1], w1.y, fmaf(a[j + 2], w1.z, fmaf(a[j + 3], w1.w, s1))));
        s2 = fmaf(a[j], w2.x, fmaf(a[j + 1], w2.y, fmaf(a[j + 2], w2.z, fmaf(a[j + 3], w2.w, s2))));
        s3 = fmaf(a[j], w3.x, fmaf(a[j + 1], w3.y, fmaf(a[j + 2], w3.z, fmaf(a[j + 3], w3.w, s3))));
      }
      s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16); s3 += __shfl_xor(s3, 16);
      s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 += __shfl_xor(s3, 32);
      if (valid && g == 0)
        *reinterpret_cast<float4*>(aux_out + row * 4) =
            make_float4(s0 + sAux[4 * KD], s1 + sAux[4 * KD + 1], s2 + sAux[4 * KD + 2], s3 + sAux[4 * KD + 3]);
    }
    uint32_t ah[KQD], am[KQD], al[KQD];
#pragma unroll
    for (int j = 0; j < KQD; ++j) split3_bf16(a[2 * j], a[2 * j + 1], ah[j], am[j], al[j]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kF6Depth == 2) request_row(a, chunk + 2 * stride);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[NTILE];
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) acc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef ALLSET_ABLATE_NOMFMA
    for (int t = 0; t < 1; ++t) {
#else
#pragma unroll
    for (int t = 0; t < T; ++t) {
#endif
      Frag8 fa_h, fa_m, fa_l;
      fa_h.u = make_uint4(ah[4 * t], ah[4 * t + 1], ah[4 * t + 2], ah[4 * t + 3]);
      fa_m.u = make_uint4(am[4 * t], am[4 * t + 1], am[4 * t + 2], am[4 * t + 3]);
      fa_l.u = make_uint4(al[4 * t], al[4 * t + 1], al[4 * t + 2], al[4 * t + 3]);
#pragma unroll
      for (int tl = 0; tl < NTILE; tl += 2) {       // two column tiles: two independent accumulator chains
        const int o0 = plane_off<KQD, GS>(g, tl * 16 + ri, t), o1 = plane_off<KQD, GS>(g, tl * 16 + 16 + ri, t);
        Frag8 b0h, b0m, b0l, b1h, b1m, b1l;
        b0h.u = *reinterpret_cast<const uint4*>(&sWh[o0]);
        b0m.u = *reinterpret_cast<const uint4*>(&sWm[o0]);
        b0l.u = *reinterpret_cast<const uint4*>(&sWl[o0]);
        b1h.u = *reinterpret_cast<const uint4*>(&sWh[o1]);
        b1m.u = *reinterpret_cast<const uint4*>(&sWm[o1]);
        b1l.u = *reinterpret_cast<const uint4*>(&sWl[o1]);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l.v, b0h.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l.v, b1h.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b0l.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b1l.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b0m.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b1m.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b0h.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b1h.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b0m.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b1m.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b0h.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b1h.v, acc[tl + 1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kF6Depth == 1) request_row(a, chunk + stride);     // (no registers for it during the matrix phase at 168)
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue, 64 columns (4 tiles) per trip through the slab
    const int c4 = (lane & 15) * 4;
#pragma unroll
    for (int hb = 0; hb < ND / 64; ++hb) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sT[(4 * g + r) * 64 + tt * 16 + ri] = acc[hb * 4 + tt][r];
      // one wave, in-order LDS queue: no barrier needed, but the compiler must not move the vector reads above the
      // scalar writes (different access types) nor the next trip's writes above these reads
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float4 bv = *reinterpret_cast<const float4*>(&sBias[hb * 64 + c4]);
      uint64_t bal[4][4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rl = it * 4 + (lane >> 4);
        const int64_t r = chunk * 16 + rl;
        float4 v = *reinterpret_cast<const float4*>(&sT[rl * 64 + c4]);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if constexpr (DROP_OUT) {
          float k0, k1, k2, k3;
          keep_scale2(seed_out, r * ND + hb * 64 + c4, thr_out, keep_out, k0, k1);
          keep_scale2(seed_out, r * ND + hb * 64 + c4 + 2, thr_out, keep_out, k2, k3);
          v.x *= k0; v.y *= k1; v.z *= k2; v.w *= k3;
        }
#ifdef ALLSET_ABLATE_NOSTORE
        if (r < n && v.x == 123.456f) *reinterpret_cast<float4*>(y + r * ldy + hb * 64 + c4) = v;
#else
        if (r < n) *reinterpret_cast<float4*>(y + r * ldy + hb * 64 + c4) = v;
#endif
        if (mask_out != nullptr) {     // four ballots per row group (one per column-of-quad c), combined below
          bal[it][0] = __ballot(v.x > 0.f); bal[it][1] = __ballot(v.y > 0.f);
          bal[it][2] = __ballot(v.z > 0.f); bal[it][3] = __ballot(v.w > 0.f);
        }
      }
      if (mask_out != nullptr) {
        // activation mask for the backward kernels, 1 bit per element (include/allset_hip.h "mask layout"): the 16 ballots
        // of this 16 x 64 block, byte-transposed so that every consumer needs ONE dword, leave as one 128-byte store:
        // lane L < 32 builds dword it*8 + idx (it = L>>3, idx = L&7) = bytes idx of the four ballots of row group it
        const int mit = (lane >> 3) & 3, sh = 8 * (lane & 7);
        uint32_t word = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint64_t bsel = mit == 0 ? bal[0][c] : (mit == 1 ? bal[1][c] : (mit == 2 ? bal[2][c] : bal[3][c]));
          word |= static_cast<uint32_t>((bsel >> sh) & 0xffu) << (8 * c);
        }
        if (lane < 32) reinterpret_cast<uint32_t*>(mask_out)[(chunk * (ND / 64) + hb) * 32 + lane] = word;
      }
      __asm__ volatile("" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  float a0[KQ];
  int64_t chunk = static_cast<int64_t>(blockIdx.x) * kF6Waves + wave;
  request_row(a0, chunk);
  if constexpr (kF6Depth == 1) {              // three waves per SIMD: one chunk ahead (168 registers per wave)
    for (; chunk < n_chunks; chunk += stride) process(a0, chunk);
    return;
  }
  float a1[KQ];
  request_row(a1, chunk + stride);
  // Two chunks per trip, the odd last chunk PEELED: with "if (chunk + stride < n_chunks) process(a1, ...)" inside the loop the
  // compiler's s_waitcnt insertion follows the path that skips the second half and puts s_waitcnt vmcnt(0) at the loop header --
  // every trip then waited for the buffer it had just re-requested, i.e. the two-deep prefetch covered one chunk of latency
  // instead of two (found in round 3 with the same construct in fused_bwd4.hip).
  for (; chunk + stride < n_chunks; chunk += 2 * stride) {
    process(a0, chunk);
    process(a1, chunk + stride);
  }
  if (chunk < n_chunks) process(a0, chunk);
}

// ---- backward-data on the bf16 matrix pipe (same scheme as fused_linear_fwd_x6_kernel) ---------------------------------
//   ga = gy * (y > 0 ? keep_out : 0)     lane (i, g) holds quarter g of row i: O/4 values of gy and of y, 16-byte loads;
//                                        the NEXT chunk's gy / y rows are in flight while this one is processed
//   gu = ga @ W                          W is split into three bf16 planes in LDS, transposed on the way in
//                                        ([o-quarter][in-column][O/8 dwords], swizzled as in the forward)
//   epilogue on the ROW-MAJOR side of the LDS trip (lane = 4 consecutive columns of 4 rows): dropout-in mask,
//   LayerNorm backward (row sums = 16-lane reductions), relu-in mask, 16-byte stores of gx; x and the row statistics
//   for the epilogue are requested before the MFMAs.  dgamma/dbeta: per-lane column sums, one partial row per wave.
template <int OD, int ID, bool HAS_LN, bool DROP_IN, bool HAS_AUX>
__global__ __launch_bounds__(kX6Block) void fused_linear_bwd_x6_kernel(
    const float* __restrict__ gy, int64_t ldg, const float* __restrict__ y, int64_t ldy, float p_out,
    const float* __restrict__ W, const float* __restrict__ x, int64_t ldx, const float* __restrict__ stats,
    const float* __restrict__ gamma, int relu_in, float p_in, uint64_t seed_in, float* gx,
    int64_t ldgx, float* __restrict__ part, int64_t n, const uint64_t* __restrict__ seed_base,
    const uint32_t* __restrict__ mask, const float* acc_in, int64_t ldacc, const float* __restrict__ aux_g,
    const float* __restrict__ aux_w) {
  seed_in = resolve_seed(seed_base, seed_in);
  constexpr int OQ = OD / 4, OQD = OQ / 2, T = OQ / 8;
  constexpr int GS = ID * OQD;
  constexpr int NTILE = ID / 16, NH = ID / 64;
  __shared__ __attribute__((aligned(16))) uint32_t sWh[4 * GS];
  __shared__ __attribute__((aligned(16))) uint32_t sWm[4 * GS];
  __shared__ __attribute__((aligned(16))) uint32_t sWl[4 * GS];
  __shared__ __attribute__((aligned(16))) float sG[ID];
  __shared__ __attribute__((aligned(16))) float sTrans[kX6Waves * 16 * 64];
  const int tid = threadIdx.x;
  {                                                                   // (all requests before the first split: see the forward kernel)
    constexpr int NPW = ((OD / 2) * ID + kX6Block - 1) / kX6Block;
    float w0[NPW], w1[NPW];
#pragma unroll
    for (int it = 0; it < NPW; ++it) {
      const int idx = tid + it * kX6Block;
      const int o = 2 * (idx / ID), i = idx % ID;                    // threads run along i: coalesced reads of W
      if ((OD / 2) * ID % kX6Block == 0 || idx < (OD / 2) * ID) { w0[it] = W[o * ID + i]; w1[it] = W[(o + 1) * ID + i]; }
    }
#pragma unroll
    for (int it = 0; it < NPW; ++it) {
      const int idx = tid + it * kX6Block;
      const int o = 2 * (idx / ID), i = idx % ID;
      if ((OD / 2) * ID % kX6Block == 0 || idx < (OD / 2) * ID) {
        uint32_t ph, pm, pl;
        split3_bf16(w0[it], w1[it], ph, pm, pl);
        const int e = o % OQ;
        const int off = plane_off<OQD, GS>(o / OQ, i, e / 8) + (e % 8) / 2;
        sWh[off] = ph; sWm[off] = pm; sWl[off] = pl;
      }
    }
  }
  for (int idx = tid; idx < ID; idx += kX6Block) sG[idx] = HAS_LN ? gamma[idx] : 1.f;
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int ri = lane & 15, g = lane >> 4;
  const int c4 = (lane & 15) * 4;
  const float inv_i = 1.f / static_cast<float>(ID);
  const float keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  const float keep_in = DROP_IN ? 1.f / (1.f - p_in) : 1.f;
  const uint32_t thr_in = drop_threshold(p_in);
  const bool has_mask = mask != nullptr;
  const bool has_y = y != nullptr && !has_mask;
  const bool need_x = HAS_LN || relu_in;
  // this lane's word of the activation mask: row ri of the chunk, columns g*OQ .. +OQ-1
  constexpr int NHO = OD / 64;
  const int m_l15 = ((g * OQ) % 64) / 4;
  const int m_word = (((g * OQ) / 64) * 4 + (ri >> 2)) * 8 + (ri & 3) * 2 + (m_l15 >> 3);
  const int m_shift = m_l15 & 7;
  const int64_t n_chunks = (n + 15) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kX6Waves;
  float* sT = sTrans + wave * (16 * 64);
  float4 gam[NH];
#pragma unroll
  for (int hb = 0; hb < NH; ++hb) gam[hb] = *reinterpret_cast<const float4*>(&sG[hb * 64 + c4]);
  float4 dg[NH], db[NH];
#pragma unroll
  for (int hb = 0; hb < NH; ++hb) { dg[hb] = make_float4(0.f, 0.f, 0.f, 0.f); db[hb] = make_float4(0.f, 0.f, 0.f, 0.f); }
  // rank-4 update gx += aux_g[n,4] @ aux_w[4,I] (the gradient of four auxiliary output columns of the forward)
  // (a template flag, and aux_w read from LDS where it is used: its 32 registers must not weigh on the epilogue)
  __shared__ __attribute__((aligned(16))) float sAuxW[HAS_AUX ? 4 * ID : 4];
  if constexpr (HAS_AUX) {
    for (int idx = tid; idx < 4 * ID; idx += kX6Block) sAuxW[idx] = aux_w[idx];
    __syncthreads();
  }

  float ag[OQ];
  uint32_t am_bits = 0;
  auto request_rows = [&](int64_t chunk) {               // unconditional, clamped (see the forward kernel)
    int64_t row = chunk * 16 + ri;
    row = row < n ? row : n - 1;
    if (has_mask) am_bits = mask[(row >> 4) * (NHO * 32) + m_word];
#ifdef ALLSET_ABLATE_NOLOAD
    if (p_in == 123.f) {
#endif
    const float4* gr = reinterpret_cast<const float4*>(gy + row * ldg + g * OQ);
#pragma unroll
    for (int q = 0; q < OQ / 4; ++q) {
      const float4 v = gr[q];
      ag[4 * q] = v.x; ag[4 * q + 1] = v.y; ag[4 * q + 2] = v.z; ag[4 * q + 3] = v.w;
    }
#ifdef ALLSET_ABLATE_NOLOAD
    }
#endif
  };

  int64_t chunk = static_cast<int64_t>(blockIdx.x) * kX6Waves + wave;
  request_rows(chunk);
  for (; chunk < n_chunks; chunk += stride) {
    // ---- A operand
    const bool valid = chunk * 16 + ri < n;
    if (has_mask) {
      const uint32_t bits = valid ? (am_bits >> m_shift) : 0u;
#pragma unroll
      for (int j = 0; j < OQ; ++j) ag[j] = (bits & (1u << (8 * (j & 3) + (j >> 2)))) ? ag[j] * keep_out : 0.f;
    } else if (has_y) {
      // legacy source of the epilogue mask (callers without the 1-bit mask): y is read here, not prefetched, so that
      // the mask path does not carry a second 32-register landing buffer
      const int64_t yrow = valid ? chunk * 16 + ri : n - 1;
      const float4* yr = reinterpret_cast<const float4*>(y + yrow * ldy + g * OQ);
#pragma unroll
      for (int q = 0; q < OQ / 4; ++q) {
        const float4 v = yr[q];
        ag[4 * q] = (valid && v.x > 0.f) ? ag[4 * q] * keep_out : 0.f;
        ag[4 * q + 1] = (valid && v.y > 0.f) ? ag[4 * q + 1] * keep_out : 0.f;
        ag[4 * q + 2] = (valid && v.z > 0.f) ? ag[4 * q + 2] * keep_out : 0.f;
        ag[4 * q + 3] = (valid && v.w > 0.f) ? ag[4 * q + 3] * keep_out : 0.f;
      }
    } else if (!valid) {
#pragma unroll
      for (int j = 0; j < OQ; ++j) ag[j] = 0.f;
    }
    uint32_t ah[OQD], am[OQD], al[OQD];
#pragma unroll
    for (int j = 0; j < OQD; ++j) split3_bf16(ag[2 * j], ag[2 * j + 1], ah[j], am[j], al[j]);
    __builtin_amdgcn_sched_barrier(0);
    request_rows(chunk + stride);
    // ---- epilogue inputs, row-major: lane = rows it*4 + (lane>>4), columns hb*64 + c4 .. +3
    float4 xr[NH][4];
    float2 st[4];
    if (need_x) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        int64_t r = chunk * 16 + it * 4 + (lane >> 4);
        r = r < n ? r : n - 1;                          // clamped, unconditional; dead rows are masked where used
#ifdef ALLSET_ABLATE_NOLOAD
        if (p_in == 123.f) {
#endif
        if constexpr (HAS_LN) st[it] = *reinterpret_cast<const float2*>(stats + r * 2);
#pragma unroll
        for (int hb = 0; hb < NH; ++hb) xr[hb][it] = *reinterpret_cast<const float4*>(x + r * ldx + hb * 64 + c4);
#ifdef ALLSET_ABLATE_NOLOAD
        }
#endif
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[NTILE];
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) acc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef ALLSET_ABLATE_NOMFMA
    for (int t = 0; t < 1; ++t) {
#else
#pragma unroll
    for (int t = 0; t < T; ++t) {
#endif
      Frag8 fa_h, fa_m, fa_l;
      fa_h.u = make_uint4(ah[4 * t], ah[4 * t + 1], ah[4 * t + 2], ah[4 * t + 3]);
      fa_m.u = make_uint4(am[4 * t], am[4 * t + 1], am[4 * t + 2], am[4 * t + 3]);
      fa_l.u = make_uint4(al[4 * t], al[4 * t + 1], al[4 * t + 2], al[4 * t + 3]);
#pragma unroll
      for (int tl = 0; tl < NTILE; tl += 2) {
        const int o0 = plane_off<OQD, GS>(g, tl * 16 + ri, t), o1 = plane_off<OQD, GS>(g, tl * 16 + 16 + ri, t);
        Frag8 b0h, b0m, b0l, b1h, b1m, b1l;
        b0h.u = *reinterpret_cast<const uint4*>(&sWh[o0]);
        b0m.u = *reinterpret_cast<const uint4*>(&sWm[o0]);
        b0l.u = *reinterpret_cast<const uint4*>(&sWl[o0]);
        b1h.u = *reinterpret_cast<const uint4*>(&sWh[o1]);
        b1m.u = *reinterpret_cast<const uint4*>(&sWm[o1]);
        b1l.u = *reinterpret_cast<const uint4*>(&sWl[o1]);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l.v, b0h.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_l.v, b1h.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b0l.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b1l.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b0m.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b1m.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b0h.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_m.v, b1h.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b0m.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b1m.v, acc[tl + 1], 0, 0, 0);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b0h.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_h.v, b1h.v, acc[tl + 1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- to row-major through the wave's LDS slab
    float4 gz[NH][4];
#pragma unroll
    for (int hb = 0; hb < NH; ++hb) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sT[(4 * g + r) * 64 + tt * 16 + ri] = acc[hb * 4 + tt][r];
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) gz[hb][it] = *reinterpret_cast<const float4*>(&sT[(it * 4 + (lane >> 4)) * 64 + c4]);
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- dropout-in mask, LayerNorm backward, relu-in mask
#ifdef ALLSET_EXP_WAIT
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t r = chunk * 16 + it * 4 + (lane >> 4);
      const bool live = r < n;
      float s1 = 0.f, s2 = 0.f;
      uint32_t pos = 0;                                  // x > 0 bits (relu-in mask), 4 per 64-column half
#pragma unroll
      for (int hb = 0; hb < NH; ++hb) {
        float4 v = gz[hb][it];
        if constexpr (DROP_IN) {
          float k0, k1, k2, k3;
          keep_scale2(seed_in, r * ID + hb * 64 + c4, thr_in, keep_in, k0, k1);
          keep_scale2(seed_in, r * ID + hb * 64 + c4 + 2, thr_in, keep_in, k2, k3);
          v.x *= k0; v.y *= k1; v.z *= k2; v.w *= k3;
        }
        if (need_x) {
          const float4 xv = xr[hb][it];
          pos |= ((xv.x > 0.f ? 1u : 0u) | (xv.y > 0.f ? 2u : 0u) | (xv.z > 0.f ? 4u : 0u) | (xv.w > 0.f ? 8u : 0u)) << (4 * hb);
        }
        if constexpr (HAS_LN) {
          float4 t = xr[hb][it];
          if (relu_in) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
          const float mean = st[it].x, rstd = st[it].y;
          float4 xh = make_float4((t.x - mean) * rstd, (t.y - mean) * rstd, (t.z - mean) * rstd, (t.w - mean) * rstd);
          if (!live) xh = make_float4(0.f, 0.f, 0.f, 0.f);
          dg[hb].x = fmaf(v.x, xh.x, dg[hb].x); dg[hb].y = fmaf(v.y, xh.y, dg[hb].y);
          dg[hb].z = fmaf(v.z, xh.z, dg[hb].z); dg[hb].w = fmaf(v.w, xh.w, dg[hb].w);
          db[hb].x += v.x; db[hb].y += v.y; db[hb].z += v.z; db[hb].w += v.w;
          v.x *= gam[hb].x; v.y *= gam[hb].y; v.z *= gam[hb].z; v.w *= gam[hb].w;      // gh
          s1 += (v.x + v.y) + (v.z + v.w);
          s2 = fmaf(v.x, xh.x, s2); s2 = fmaf(v.y, xh.y, s2); s2 = fmaf(v.z, xh.z, s2); s2 = fmaf(v.w, xh.w, s2);
          xr[hb][it] = xh;
        }
        gz[hb][it] = v;
      }
      if constexpr (HAS_LN) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        s1 *= inv_i; s2 *= inv_i;
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
            const uint32_t m = pos >> (4 * hb);
            if (!(m & 1u)) o.x = 0.f;
            if (!(m & 2u)) o.y = 0.f;
            if (!(m & 4u)) o.z = 0.f;
            if (!(m & 8u)) o.w = 0.f;
          }
          if (acc_in != nullptr) {        // gx = acc_in + ...: a second gradient branch of the same tensor, summed here
            const float4 ai = *reinterpret_cast<const float4*>(acc_in + r * ldacc + hb * 64 + c4);   // (may alias gx)
            o.x += ai.x; o.y += ai.y; o.z += ai.z; o.w += ai.w;
          }
          if constexpr (HAS_AUX) {
            const float4 q = *reinterpret_cast<const float4*>(aux_g + r * 4);
            const float4 w0 = *reinterpret_cast<const float4*>(&sAuxW[0 * ID + hb * 64 + c4]);
            const float4 w1 = *reinterpret_cast<const float4*>(&sAuxW[1 * ID + hb * 64 + c4]);
            const float4 w2 = *reinterpret_cast<const float4*>(&sAuxW[2 * ID + hb * 64 + c4]);
            const float4 w3 = *reinterpret_cast<const float4*>(&sAuxW[3 * ID + hb * 64 + c4]);
            o.x = fmaf(q.x, w0.x, fmaf(q.y, w1.x, fmaf(q.z, w2.x, fmaf(q.w, w3.x, o.x))));
            o.y = fmaf(q.x, w0.y, fmaf(q.y, w1.y, fmaf(q.z, w2.y, fmaf(q.w, w3.y, o.y))));
            o.z = fmaf(q.x, w0.z, fmaf(q.y, w1.z, fmaf(q.z, w2.z, fmaf(q.w, w3.z, o.z))));
            o.w = fmaf(q.x, w0.w, fmaf(q.y, w1.w, fmaf(q.z, w2.w, fmaf(q.w, w3.w, o.w))));
          }
#ifdef ALLSET_ABLATE_NOSTORE
          if (o.x == 123.456f)
#endif
          if (gx != nullptr) *reinterpret_cast<float4*>(gx + r * ldgx + hb * 64 + c4) = o;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (HAS_LN) {     // one partial row per wave: part[wave_global][0|1][I]; the 4 row groups of a lane column fold first
    float* pw = part + (static_cast<int64_t>(blockIdx.x) * kX6Waves + wave) * 2 * ID;
#pragma unroll
    for (int hb = 0; hb < NH; ++hb) {
      float4 a = dg[hb], b = db[hb];
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
        b.x += __shfl_xor(b.x, off); b.y += __shfl_xor(b.y, off); b.z += __shfl_xor(b.z, off); b.w += __shfl_xor(b.w, off);
      }
      if (lane < 16) {
        *reinterpret_cast<float4*>(pw + hb * 64 + c4) = a;
        *reinterpret_cast<float4*>(pw + ID + hb * 64 + c4) = b;
      }
    }
  }
}

}  // namespace allset

using namespace allset;


extern "C" int64_t allset_fused_linear_mask_words(int64_t n, int64_t N) {
  // 1 bit per output element in 16-row x 64-column blocks of 32 dwords; 0 = this build/mode has no mask support
  if (n <= 0 || N % 64 != 0) return 0;
  return ((n + 15) / 16) * (N / 64) * 32;
}

extern "C" int allset_fused_linear_supported(int64_t K, int64_t N) {
  return ((K == 64 || K == 128) && (N == 64 || N == 128)) ? 1 : 0;
}

// fused_fwd2.hip: the forward with the waves split by role (eight vector waves, four matrix waves; K = N = 128)
int fused_linear_fwd_roles_supported(int64_t K, int64_t N, int has_aux);
int launch_fused_linear_fwd_roles(hipStream_t st, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                  int relu_in, float p_in, uint64_t seed_in, const float* W, const float* bias, int relu_out,
                                  float p_out, uint64_t seed_out, float* y, int64_t ldy, float* stats, int64_t n,
                                  const uint64_t* seed_base, uint32_t* mask_out, int64_t xcb, int64_t ycb, float ln_inv, int arith);
int launch_fused_linear_fwd_roles_aux(hipStream_t st, const float* x, int64_t ldx, int relu_in, const float* W, const float* bias,
                                      int relu_out, float* y, int64_t ldy, int64_t n, uint32_t* mask_out, const float* aux_w,
                                      const float* aux_b, float* aux_out, int arith);

// 1 = cb is a usable column-block width for a K-column operand (a power of two, 4 <= cb <= K / 2, the operand's ld == cb)
static bool block_cols_ok(int64_t cb, int64_t K, int64_t ld) {
  return cb >= 4 && cb <= K / 2 && (cb & (cb - 1)) == 0 && K % cb == 0 && ld == cb;
}

static int fused_linear_fwd_impl(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                 int relu_in, float p_in, uint64_t seed_in, const float* W, const float* bias,
                                 int relu_out, float p_out, uint64_t seed_out, float* y, int64_t ldy,
                                 float* stats, int64_t n, int64_t K, int64_t N, const uint64_t* seed_base,
                                 uint32_t* mask_out, const float* aux_w, const float* aux_b, float* aux_out,
                                 void* stream, int64_t xcb, int64_t ycb, int norm_mode = ALLSET_NORM_LAYER,
                                 int arith = ALLSET_ARITH_AUTO) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "fused_linear_fwd: negative size");
  ALLSET_REQUIRE(arith == ALLSET_ARITH_AUTO || arith == ALLSET_ARITH_BF16X6 || arith == ALLSET_ARITH_FP16X3,
                 "fused_linear_fwd: arith must be ALLSET_ARITH_AUTO, ALLSET_ARITH_BF16X6 or ALLSET_ARITH_FP16X3");
  ALLSET_REQUIRE(norm_mode == ALLSET_NORM_LAYER || norm_mode == ALLSET_NORM_COLUMN_AFFINE, "fused_linear_fwd: norm_mode must be ALLSET_NORM_LAYER or ALLSET_NORM_COLUMN_AFFINE");
  ALLSET_REQUIRE(norm_mode == ALLSET_NORM_LAYER || gamma != nullptr, "fused_linear_fwd: the column-affine prologue needs gamma (scale) and beta (shift)");
  // column affine = the LayerNorm prologue with the row statistics switched off: mean = s * 0, rstd = rsqrt(q * 0 + 1)
  const float ln_inv = norm_mode == ALLSET_NORM_COLUMN_AFFINE ? 0.f : 1.f / static_cast<float>(K);
  if (norm_mode == ALLSET_NORM_COLUMN_AFFINE) eps = 1.f;
  ALLSET_REQUIRE(aux_out == nullptr || (aux_w != nullptr && aligned16(aux_out)), "fused_linear_fwd: aux_out needs aux_w and 16-byte alignment");
  ALLSET_REQUIRE(p_in >= 0.f && p_in < 1.f && p_out >= 0.f && p_out < 1.f, "fused_linear_fwd: dropout p must be in [0,1)");
  if (!allset_fused_linear_supported(K, N)) {
    set_error("fused_linear_fwd: K=%lld N=%lld not built (K, N in {64,128})", static_cast<long long>(K), static_cast<long long>(N));
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(x && W && y, "fused_linear_fwd: null pointer");
  ALLSET_REQUIRE((gamma == nullptr) == (beta == nullptr), "fused_linear_fwd: gamma and beta must come together");
  ALLSET_REQUIRE(gamma == nullptr || stats != nullptr, "fused_linear_fwd: LayerNorm prologue needs a stats buffer");
  ALLSET_REQUIRE(aligned16(x) && aligned16(y), "fused_linear_fwd: x / y must be 16-byte aligned");
  ALLSET_REQUIRE(xcb != 0 || (ldx >= K && ldx % 4 == 0), "fused_linear_fwd: x must be 16-byte aligned rows");
  ALLSET_REQUIRE(ycb != 0 || (ldy >= N && ldy % 4 == 0), "fused_linear_fwd: y must be 16-byte aligned rows");
  ALLSET_REQUIRE(xcb == 0 || block_cols_ok(xcb, K, ldx), "fused_linear_fwd_blocked: x_block_cols must be a power of two in [4, K/2] and ldx == x_block_cols");
  ALLSET_REQUIRE(ycb == 0 || block_cols_ok(ycb, N, ldy), "fused_linear_fwd_blocked: y_block_cols must be a power of two in [4, N/2] and ldy == y_block_cols");
  ALLSET_REQUIRE((xcb == 0 && ycb == 0) || n * 128 * 4 < (int64_t{1} << 32), "fused_linear_fwd_blocked: a blocked operand must stay below 4 GiB (32-bit lane offsets)");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int has_ln = gamma != nullptr;
  // (auxiliary columns ride in the split-role kernel for the plain Linear -- PMA's value projection; any other prologue / epilogue with
  //  them keeps the symmetric kernel below)
  const bool aux_plain = aux_out == nullptr || (!has_ln && p_in == 0.f && p_out == 0.f && xcb == 0 && ycb == 0 && aligned16(aux_w));
  const bool roles = fused_linear_fwd_roles_supported(K, N, aux_out != nullptr) && aux_plain && aligned16(W) && ldx < (1 << 24) &&
                     ldy < (1 << 24) && (reinterpret_cast<uintptr_t>(stats) & 7u) == 0;
  if (arith == ALLSET_ARITH_FP16X3 && !(roles && (!has_ln || norm_mode == ALLSET_NORM_LAYER))) {
    set_error("fused_linear_fwd: ALLSET_ARITH_FP16X3 is built for K = N = 128 behind a LayerNorm prologue or none (auxiliary columns: the "
              "plain Linear only) (allset_fused_linear_arith_supported)");
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (roles && aux_out != nullptr) {
    launch_fused_linear_fwd_roles_aux(st, x, ldx, relu_in, W, bias, relu_out, y, ldy, n, reinterpret_cast<uint32_t*>(mask_out), aux_w, aux_b,
                                      aux_out, arith);
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
  if (roles) {                                                            // K = N = 128: the split-role kernel (fused_fwd2.hip)
    launch_fused_linear_fwd_roles(st, x, ldx, gamma, beta, eps, relu_in, p_in, seed_in, W, bias, relu_out, p_out, seed_out, y, ldy,
                                  stats, n, seed_base, reinterpret_cast<uint32_t*>(mask_out), xcb, ycb, ln_inv, arith);
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
  if (xcb != 0 || ycb != 0) {
    set_error("fused_linear_fwd_blocked: column-blocked operands are read / written by the K = N = 128 split-role kernel only "
              "(allset_fused_linear_blocked_supported)");
    return ALLSET_ERR_UNSUPPORTED;
  }
  const int waves_x6 = has_ln ? fwd_x6_waves<true>() : fwd_x6_waves<false>();
  int64_t blocks_x6 = ((n + 15) / 16 + waves_x6 - 1) / waves_x6;
  if (blocks_x6 > 256) blocks_x6 = 256;                   // one persistent workgroup per CU
  const unsigned grid_x6 = static_cast<unsigned>(blocks_x6);
#define ALLSET_FUSED_FWD_F(KD, NT, LN, DI, DO)                                                                           \
  fused_linear_fwd_x6_kernel<KD, 32 * NT, LN, DI, DO><<<grid_x6, fwd_x6_waves<LN>() * kWave, 0, st>>>(                     \
      x, ldx, gamma, beta, eps, relu_in, p_in, seed_in, W, bias, relu_out, p_out, seed_out, y, ldy, stats, n,            \
      seed_base, reinterpret_cast<uint8_t*>(mask_out), aux_w, aux_b, aux_out, ln_inv)
#define ALLSET_FUSED_FWD(KD, NT)                                                      \
  do {                                                                                \
    const int v = (has_ln ? 4 : 0) | (p_in > 0.f ? 2 : 0) | (p_out > 0.f ? 1 : 0);    \
    switch (v) {                                                                      \
      case 0: ALLSET_FUSED_FWD_F(KD, NT, false, false, false); break;                 \
      case 1: ALLSET_FUSED_FWD_F(KD, NT, false, false, true); break;                  \
      case 2: ALLSET_FUSED_FWD_F(KD, NT, false, true, false); break;                  \
      case 3: ALLSET_FUSED_FWD_F(KD, NT, false, true, true); break;                   \
      case 4: ALLSET_FUSED_FWD_F(KD, NT, true, false, false); break;                  \
      case 5: ALLSET_FUSED_FWD_F(KD, NT, true, false, true); break;                   \
      case 6: ALLSET_FUSED_FWD_F(KD, NT, true, true, false); break;                   \
      default: ALLSET_FUSED_FWD_F(KD, NT, true, true, true); break;                   \
    }                                                                                 \
  } while (0)
  if (K == 128 && N == 128) ALLSET_FUSED_FWD(128, 4);
  else if (K == 128 && N == 64) ALLSET_FUSED_FWD(128, 2);
  else if (K == 64 && N == 128) ALLSET_FUSED_FWD(64, 4);
  else ALLSET_FUSED_FWD(64, 2);
#undef ALLSET_FUSED_FWD_F
#undef ALLSET_FUSED_FWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static inline unsigned fused_grid(int64_t n) {                 // 16-row chunks, one persistent 8-wave workgroup per CU
  int64_t blocks = ((n + 15) / 16 + kX6Waves - 1) / kX6Waves;
  return static_cast<unsigned>(blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks));
}

extern "C" int allset_fused_linear_bwd_partials(int64_t n, int64_t* n_partials) {
  clear_error();
  ALLSET_REQUIRE(n_partials != nullptr && n >= 0, "fused_linear_bwd_partials: bad argument");
  *n_partials = static_cast<int64_t>(fused_grid(n)) * kFusedWaves;
  return ALLSET_OK;
}

extern "C" int allset_fused_linear_bwd(const float* gy, int64_t ldg, const float* y, int64_t ldy, float p_out,
                                       const float* W, const float* x, int64_t ldx, const float* stats,
                                       const float* gamma, int relu_in, float p_in, uint64_t seed_in, float* gx,
                                       int64_t ldgx, float* partials, int64_t n_partials, int64_t n, int64_t O,
                                       int64_t I, const uint64_t* seed_base, const uint32_t* mask, const float* acc_in,
                                       int64_t ldacc, const float* aux_g, const float* aux_w, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "fused_linear_bwd: negative size");
  ALLSET_REQUIRE(aux_g == nullptr || (aux_w != nullptr && aligned16(aux_g) && aligned16(aux_w)), "fused_linear_bwd: aux_g needs aux_w, both 16-byte aligned");
  ALLSET_REQUIRE(acc_in == nullptr || (ldacc >= I && ldacc % 4 == 0 && aligned16(acc_in)), "fused_linear_bwd: acc_in must be 16-byte aligned rows");
  ALLSET_REQUIRE(p_in >= 0.f && p_in < 1.f && p_out >= 0.f && p_out < 1.f, "fused_linear_bwd: dropout p must be in [0,1)");
  if (!allset_fused_linear_supported(I, O)) {
    set_error("fused_linear_bwd: in=%lld out=%lld not built (both in {64,128})", static_cast<long long>(I), static_cast<long long>(O));
    return ALLSET_ERR_UNSUPPORTED;
  }
  const bool has_ln = stats != nullptr;
  ALLSET_REQUIRE(has_ln == (gamma != nullptr), "fused_linear_bwd: stats and gamma must come together");
  const unsigned grid = fused_grid(n);
  ALLSET_REQUIRE(!has_ln || (partials != nullptr && n_partials == static_cast<int64_t>(grid) * kFusedWaves),
                 "fused_linear_bwd: partials buffer must hold allset_fused_linear_bwd_partials() rows");
  if (n == 0) {
    if (has_ln) ALLSET_HIP_CHECK(hipMemsetAsync(partials, 0, static_cast<size_t>(n_partials) * 2 * I * sizeof(float), static_cast<hipStream_t>(stream)));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && W, "fused_linear_bwd: null pointer");
  ALLSET_REQUIRE(gx != nullptr || stats != nullptr, "fused_linear_bwd: gx may be NULL only to get the LayerNorm partials alone");
  ALLSET_REQUIRE((has_ln || relu_in) ? x != nullptr : true, "fused_linear_bwd: x required for LayerNorm / relu backward");
  ALLSET_REQUIRE(ldg >= O && ldg % 4 == 0 && aligned16(gy) && aligned16(W), "fused_linear_bwd: gy/W must be 16-byte aligned rows");
  ALLSET_REQUIRE(y == nullptr || (ldy >= O && ldy % 4 == 0 && aligned16(y)), "fused_linear_bwd: y must be 16-byte aligned rows");
  ALLSET_REQUIRE(ldgx >= I && (x == nullptr || ldx >= I), "fused_linear_bwd: leading dimension too small");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  ALLSET_REQUIRE((gx == nullptr || (ldgx % 4 == 0 && aligned16(gx))) && (x == nullptr || (ldx % 4 == 0 && aligned16(x))),
                 "fused_linear_bwd: gx / x must be 16-byte aligned rows");
  ALLSET_REQUIRE(stats == nullptr || (reinterpret_cast<uintptr_t>(stats) & 7u) == 0, "fused_linear_bwd: stats must be 8-byte aligned");
#define ALLSET_FUSED_BWD_F(OD, IT, LN, DI)                                                                                   \
  fused_linear_bwd_x6_kernel<OD, 32 * IT, LN, DI, false><<<grid, kX6Block, 0, st>>>(                                         \
      gy, ldg, y, ldy, p_out, W, x, ldx, stats, gamma, relu_in, p_in, seed_in, gx, ldgx, partials, n, seed_base, mask,       \
      acc_in, ldacc, nullptr, nullptr)
#define ALLSET_FUSED_BWD(OD, IT)                                          \
  do {                                                                    \
    if (has_ln) { if (p_in > 0.f) ALLSET_FUSED_BWD_F(OD, IT, true, true); else ALLSET_FUSED_BWD_F(OD, IT, true, false); }     \
    else        { if (p_in > 0.f) ALLSET_FUSED_BWD_F(OD, IT, false, true); else ALLSET_FUSED_BWD_F(OD, IT, false, false); }   \
  } while (0)
  if (aux_g != nullptr) {     // the rank-4 update exists for the plain Linear only (PMA's value projection)
    if (has_ln || p_in > 0.f) {
      set_error("fused_linear_bwd: aux_g is supported without LayerNorm / dropout prologue only");
      return ALLSET_ERR_UNSUPPORTED;
    }
#define ALLSET_FUSED_BWD_AUX(OD, ID)                                                                                         \
  fused_linear_bwd_x6_kernel<OD, ID, false, false, true><<<grid, kX6Block, 0, st>>>(                                         \
      gy, ldg, y, ldy, p_out, W, x, ldx, stats, gamma, relu_in, p_in, seed_in, gx, ldgx, partials, n, seed_base, mask,       \
      acc_in, ldacc, aux_g, aux_w)
    if (O == 128 && I == 128) ALLSET_FUSED_BWD_AUX(128, 128);
    else if (O == 128 && I == 64) ALLSET_FUSED_BWD_AUX(128, 64);
    else if (O == 64 && I == 128) ALLSET_FUSED_BWD_AUX(64, 128);
    else ALLSET_FUSED_BWD_AUX(64, 64);
#undef ALLSET_FUSED_BWD_AUX
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
  if (O == 128 && I == 128) ALLSET_FUSED_BWD(128, 4);
  else if (O == 128 && I == 64) ALLSET_FUSED_BWD(128, 2);
  else if (O == 64 && I == 128) ALLSET_FUSED_BWD(64, 4);
  else ALLSET_FUSED_BWD(64, 2);
#undef ALLSET_FUSED_BWD
#undef ALLSET_FUSED_BWD_F
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_fused_linear_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                       int relu_in, float p_in, uint64_t seed_in, const float* W, const float* bias,
                                       int relu_out, float p_out, uint64_t seed_out, float* y, int64_t ldy,
                                       float* stats, int64_t n, int64_t K, int64_t N, const uint64_t* seed_base,
                                       uint32_t* mask_out, const float* aux_w, const float* aux_b, float* aux_out,
                                       void* stream) {
  return fused_linear_fwd_impl(x, ldx, gamma, beta, eps, relu_in, p_in, seed_in, W, bias, relu_out, p_out, seed_out, y, ldy, stats, n,
                               K, N, seed_base, mask_out, aux_w, aux_b, aux_out, stream, 0, 0);
}

// The same forward with a choice of what the (gamma, beta) prologue means: ALLSET_NORM_LAYER = LayerNorm (row statistics computed
// here and written to `stats`), ALLSET_NORM_COLUMN_AFFINE = relu?(x) * gamma + beta per column, no row statistics (`stats` is
// filled with {0, 1} so that the backward entries read what they expect) -- training-mode BatchNorm1d with the batch statistics
// folded into (gamma, beta) by the caller (csrc/batchnorm.hip).
extern "C" int allset_fused_linear_fwd_nm(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int norm_mode,
                                          int relu_in, float p_in, uint64_t seed_in, const float* W, const float* bias,
                                          int relu_out, float p_out, uint64_t seed_out, float* y, int64_t ldy, float* stats,
                                          int64_t n, int64_t K, int64_t N, const uint64_t* seed_base, uint32_t* mask_out,
                                          void* stream) {
  return fused_linear_fwd_impl(x, ldx, gamma, beta, eps, relu_in, p_in, seed_in, W, bias, relu_out, p_out, seed_out, y, ldy, stats, n,
                               K, N, seed_base, mask_out, nullptr, nullptr, nullptr, stream, 0, 0, norm_mode);
}

// Superset entry (ABI 11): every option of allset_fused_linear_fwd / _nm / _blocked in one call, plus the choice of arithmetic.
extern "C" int allset_fused_linear_fwd_ex(const float* x, int64_t ldx, int64_t x_block_cols, const float* gamma, const float* beta,
                                          float eps, int norm_mode, int relu_in, float p_in, uint64_t seed_in, const float* W,
                                          const float* bias, int relu_out, float p_out, uint64_t seed_out, float* y, int64_t ldy,
                                          int64_t y_block_cols, float* stats, int64_t n, int64_t K, int64_t N,
                                          const uint64_t* seed_base, uint32_t* mask_out, const float* aux_w, const float* aux_b,
                                          float* aux_out, int arith, void* stream) {
  return fused_linear_fwd_impl(x, ldx, gamma, beta, eps, relu_in, p_in, seed_in, W, bias, relu_out, p_out, seed_out, y, ldy, stats, n,
                               K, N, seed_base, mask_out, aux_w, aux_b, aux_out, stream, x_block_cols, y_block_cols, norm_mode, arith);
}

// 1 = allset_fused_linear_fwd_blocked / allset_fused_linear_bwd_all_blocked take column-blocked operands at these widths
extern "C" int allset_fused_linear_blocked_supported(int64_t K, int64_t N) {
  return fused_linear_fwd_roles_supported(K, N, 0);
}

// The same forward with x and / or y COLUMN-BLOCKED: an operand with block width cb is stored [cols / cb][n][cb] (ld == cb) -- the
// send / receive layout of the column-sharded layer's all-to-all (allset_amd/dist.py _rows_to_cols / _cols_to_rows), so that no
// pack / unpack pass stands between the Linear and the exchange.  x_block_cols / y_block_cols = 0: that operand is row-major.
extern "C" int allset_fused_linear_fwd_blocked(const float* x, int64_t ldx, int64_t x_block_cols, const float* gamma,
                                               const float* beta, float eps, int relu_in, float p_in, uint64_t seed_in,
                                               const float* W, const float* bias, int relu_out, float p_out, uint64_t seed_out,
                                               float* y, int64_t ldy, int64_t y_block_cols, float* stats, int64_t n, int64_t K,
                                               int64_t N, const uint64_t* seed_base, uint32_t* mask_out, void* stream) {
  return fused_linear_fwd_impl(x, ldx, gamma, beta, eps, relu_in, p_in, seed_in, W, bias, relu_out, p_out, seed_out, y, ldy, stats, n,
                               K, N, seed_base, mask_out, nullptr, nullptr, nullptr, stream, x_block_cols, y_block_cols);
}
