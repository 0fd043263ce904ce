// The Linear of the bf16 regime (BASELINE configs[4]: bf16 activations and parameters, fp32 accumulation) with what
// surrounds it in MLP.forward / PMA.forward (reference layers.py:571-579, 120-157) folded into ONE pass per direction:
//
//   forward        y = act(x W^T + b)                       act = relu or identity;            x, W, b, y: bf16
//                  aux[n, 4] = x aux_w^T + aux_b            PMA's folded attention logits (fp32 out), optional
//   backward-data  gx = (gy (.) [ymask > 0]) W  [+ acc_in]  [+ galpha aux_w]                    all bf16 except galpha (fp32)
//                  ga_out = gy (.) [ymask > 0]              the masked gradient, kept for the weight-gradient kernel
//   round 6: the relu mask as ONE BIT per element -- the forward's epilogue writes it (`mask_out`, N / 8 bytes per row), the
//   backward-data kernel and the weight-gradient kernel (dense.hip, wgrad_bf16_tr_kernel) both apply it to gy as they stage it:
//   the 2-byte-per-element read of the saved activation and the write + re-read of the masked gradient are gone
//   (640 -> 392 MB per masked backward at [250k, 256]).  Bit layout (private to these three kernels; allset_hip_ext.h):
//   a row is four words of N / 32 bytes; bit (16 hb + j) of word sq is column 64 hb + 16 sq + j.
//
// The library GEMM these replace already runs near the traffic floor (62 us for [250k, 256] x [256, 256]); what a fused
// kernel removes is the element-wise traffic AROUND it -- relu forward, relu backward, the gradient-branch sums, the skinny
// logits GEMM -- and their launches (the configs[4] per-GPU step is launch-bound).
//
// Organisation (both directions are the same kernel; backward-data stages W transposed):
//   * the whole weight is resident in LDS as ONE bf16 image (128 KB at 256 x 256), laid out [k-quarter][column][K/8 dwords]
//     with the 16-byte pieces XOR-swizzled by the column: a B fragment of v_mfma_f32_16x16x32_bf16 is one conflict-free
//     ds_read_b128.  The k-order of an MFMA is free as long as both operands agree, so lane (i, g) of a wave owns the
//     CONTIGUOUS quarter g of activation row i (128 B at K = 256: whole cache lines per lane) and k-step t multiplies its
//     local columns 8t .. 8t+7 -- activation fragments are the registers exactly as loaded, no conversion, no LDS;
//   * 8 waves per workgroup (two per SIMD), one persistent workgroup per CU, 16 rows per wave and step, the rows of the
//     next two steps always in flight in registers;
//   * operands are SWAPPED in the MFMA (A = weight fragment, B = activation fragment), so a lane's four accumulators of
//     a tile are four CONSECUTIVE output columns of one row: bias / relu / the gradient-branch sum / the rank-4 logits
//     term are applied there in fp32, the four values are packed to bf16 once, leave through a 16 x 64 per-wave LDS slab
//     as one 8-byte write and reach memory as whole 128-byte row segments.
#include "common.h"

#include <stdint.h>
#include <type_traits>

namespace allset {

using bf16x8_b = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4_b = __attribute__((ext_vector_type(4))) float;
union FragB { uint4 u; bf16x8_b v; };
typedef unsigned swap2_t __attribute__((ext_vector_type(2)));

constexpr int kBfBlock = 512;
constexpr int kBfWaves = kBfBlock / kWave;
constexpr int kSlabPitch = 36;                 // dwords per slab row: 32 of data (64 bf16) + 4 of padding (bank spread)

// dword offset of 16-byte piece t of (k-quarter g, column j) inside the weight image
template <int KQD, int GS>
__device__ __forceinline__ int wimg_off(int g, int j, int t) {
  constexpr int PIECES = KQD / 4, ROWS64 = (64 / KQD) > 0 ? (64 / KQD) : 1;
  return g * GS + j * KQD + 4 * (t ^ ((j / ROWS64) % PIECES));
}

__device__ __forceinline__ uint32_t keep_where_positive(uint32_t v, uint32_t y) {
  // per 16-bit half: keep v where the bf16 in y is > 0 (relu output: +0 or positive, so "magnitude bits set" is the test
  // torch's threshold_backward applies to the saved bf16 activation)
  const uint32_t lo = (y & 0x7fffu) != 0u && (y & 0x8000u) == 0u ? 0x0000ffffu : 0u;
  const uint32_t hi = (y & 0x7fff0000u) != 0u && (y & 0x80000000u) == 0u ? 0xffff0000u : 0u;
  return v & (lo | hi);
}

typedef unsigned short bf_us2_t __attribute__((ext_vector_type(2)));
// per 16-bit half: 1 where the half is not +0.  The halves are relu outputs max(acc + bias, +0): never negative, and never -0
// either (an fp32 sum that starts from the accumulator's +0 cannot round to -0), so "any bit set" is "> 0".
__device__ __forceinline__ uint32_t nz_halves(uint32_t v) {
  uint32_t r;                        // (as inline asm: __builtin_elementwise_min on a ushort2 compiles to two v_cmp + two v_cndmask)
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(v), "v"(0x00010001u));
  return r;
}
// a (two bf16) with each half kept where the matching half of m (0 or 1) is 1: one packed multiply, exact
__device__ __forceinline__ uint32_t keep_halves(uint32_t a, uint32_t m) {
  uint32_t r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(m));
  return r;
}
// 8 relu outputs (two per dword, columns in order) -> their 8 "is positive" bits, column j at bit j
__device__ __forceinline__ uint32_t relu_bits8(const uint4& o) {
  uint32_t x = nz_halves(o.x);
  x |= nz_halves(o.y) << 2; x |= nz_halves(o.z) << 4; x |= nz_halves(o.w) << 6;
  return (x & 0xffu) | ((x >> 16) << 1 & 0xffu);
}
// 16 relu outputs (+0 or positive bf16, two per dword, columns in order) -> their 16 "is positive" bits, column j at bit j
__device__ __forceinline__ uint32_t relu_bits16(const uint4& o0, const uint4& o1) {
  uint32_t x = nz_halves(o0.x);
  x |= nz_halves(o0.y) << 2; x |= nz_halves(o0.z) << 4; x |= nz_halves(o0.w) << 6;
  x |= nz_halves(o1.x) << 8; x |= nz_halves(o1.y) << 10; x |= nz_halves(o1.z) << 12; x |= nz_halves(o1.w) << 14;
  return (x & 0xffffu) | ((x >> 16) << 1);          // low halves sit at even bits, high halves at 16 + even
}

// KD: reduction width (columns of the activation rows), ND: output width.
// MASK: 0 none, 1 the saved bf16 activation (`ymask`), 2 the forward's bit mask (`ymask` points at the bit rows).
// TRANS_W: W is [KD, ND] row-major (backward-data: the layer's [out, in] weight, reduced over its rows);
//          otherwise [ND, KD] (forward).
// AUX: forward -- the four auxiliary output columns (aux_out); backward-data -- the logits' gradient term (aux_in).
// EXTRA: forward -- the relu bit mask as a second output (mask_out); backward-data -- the other gradient branch (acc_in).
// Compile-time on purpose (round 6): a run-time `if (acc_in != nullptr)` per tile, uniform as it is, cut the epilogue into
// basic blocks the scheduler cannot overlap -- the per-tile switches of round 5 cost 12 us of a 69-us launch
// (tools/linear_bf16_ablation.py).  The legacy MASK == 1 form keeps its run-time switches (one instantiation).
template <int KD, int ND, bool TRANS_W, int MASK, bool AUX, bool EXTRA>
__global__ __launch_bounds__(kBfBlock) void linear_bf16_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ ymask, int64_t ldm,
    uint16_t* __restrict__ a_out, int64_t lda, const uint16_t* __restrict__ W, const uint16_t* __restrict__ bias,
    int relu_out, const uint16_t* __restrict__ aux_w, const uint16_t* __restrict__ aux_b, float* __restrict__ aux_out,
    const float* __restrict__ aux_in, const uint16_t* __restrict__ acc_in, int64_t ldacc, uint16_t* __restrict__ y,
    int64_t ldy, uint8_t* __restrict__ mask_out, int64_t n) {
  constexpr bool HAS_MASK = MASK == 1;             // (the bf16-activation form; MASK == 2 has its own staging below)
  constexpr int MG = KD / 64;                      // 16-column groups per lane (bit-mask form)
  constexpr int KQ = KD / 4;                       // bf16 columns per lane
  constexpr int KQD = KQ / 2;                      // dwords per lane
  constexpr int T = KQ / 8;                        // MFMA k-steps
  constexpr int GS = ND * KQD;
  constexpr int NTILE = ND / 16;
  constexpr bool AUX_OUT = !TRANS_W && AUX;
  constexpr bool RT = MASK == 1;                   // the legacy form: run-time switches
  constexpr int AUXW = AUX_OUT ? KD : ND;          // width of the 4 auxiliary weight rows this instantiation keeps
  __shared__ __attribute__((aligned(16))) uint32_t sW[4 * GS];
  __shared__ __attribute__((aligned(16))) float sBias[ND];
  __shared__ __attribute__((aligned(16))) float sAux[4 * AUXW + 4];
#ifdef ALLSET_BF16_SLAB_EPILOGUE
  __shared__ __attribute__((aligned(16))) uint32_t sSlab[kBfWaves * 16 * kSlabPitch];
#endif
  const int tid = threadIdx.x;
  const bool has_aux_in = TRANS_W && (RT ? aux_in != nullptr : AUX);      // (the logits' gradient: backward-data only)
  const bool has_acc = TRANS_W && (RT ? acc_in != nullptr : EXTRA);       // (the other gradient branch: backward-data only)
  constexpr bool MASK_OUT = !TRANS_W && EXTRA;
  // relu without a branch per tile and without touching a no-relu output (a max against -inf would turn a NaN into -inf): the max is
  // always computed and a bit-select on a uniform all-ones / zero mask keeps it or the input (v_bfi_b32)
  const uint32_t relu_sel = (!TRANS_W && relu_out) ? 0xffffffffu : 0u;

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: chunk indices and row bases stay scalar
  const int ri = lane & 15, g = lane >> 4;
  const int64_t n_chunks = (n + 15) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBfWaves;
#ifdef ALLSET_BF16_SLAB_EPILOGUE
  uint32_t* slab = sSlab + wave * (16 * kSlabPitch);
#endif

  // Addresses are a scalar row base per step (64-bit, in SGPRs) plus a 32-bit lane offset: 64-bit per-lane pointers for the
  // six row-major operands cost ~20 VGPRs, which this kernel does not have.  Loads are unconditional on a clamped row
  // (a branch around a load costs s_waitcnt vmcnt(0) at the join, which drains the prefetch); rows past n are zeroed when
  // consumed.  `chunk` may run past the end (prefetch): clamped to the last step.
  auto rows_here = [&](int64_t chunk) -> int {                 // rows of this step that exist (1..16); chunk < n_chunks
    const int64_t left = n - chunk * 16;
    return left < 16 ? static_cast<int>(left) : 16;
  };
  auto request = [&](uint32_t (&a)[KQD], const uint16_t* __restrict__ src, int64_t ld, int64_t chunk) {
    const int64_t c = chunk < n_chunks ? chunk : n_chunks - 1;
    const int rh = rows_here(c);
    const uint32_t rr = ri < rh ? ri : rh - 1;
    const uint16_t* base = src + c * 16 * ld;
    const uint4* p = reinterpret_cast<const uint4*>(base + (rr * static_cast<uint32_t>(ld) + g * KQ));
#pragma unroll
    for (int q = 0; q < KQD / 4; ++q) {
#ifdef ALLSET_BF16_ABL_NOLOAD              // (tools/linear_bf16_ablation.py: the kernel without its row loads)
      const uint4 v = make_uint4(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)) + q, 0x3f803f80u, 0u, 0x3f003f00u);
#elif defined(ALLSET_BF16_ABL_PIECEMAP)    // (timing probe only, results wrong: the four lanes of a row read 64 contiguous bytes per instruction)
      const uint4 v = (p - g * (KQD / 4))[4 * q + g];
#else
      const uint4 v = p[q];
#endif
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
  };

  // bit-mask form: the lane's columns g KQ + 16 q .. + 15 are bits 16 hb .. of word sq of the row (hb = column / 64,
  // sq = (column % 64) / 16): one 2-byte load per 16 columns, MG of them per step, requested as soon as the previous ones are consumed
  constexpr int MN = MASK == 1 ? KQD : (MASK == 2 ? MG : 1);
  uint32_t bit_off[MG];
#pragma unroll
  for (int q = 0; q < MG; ++q) {
    const int col0 = g * KQ + 16 * q;
    bit_off[q] = ((col0 & 63) >> 4) * (KD / 32) + (col0 >> 6) * 2;
  }
  const uint8_t* bitrows = reinterpret_cast<const uint8_t*>(ymask);
  auto request_bits = [&](uint32_t (&m)[MN], int64_t chunk) {
    const int64_t c = chunk < n_chunks ? chunk : n_chunks - 1;
    const int rh = rows_here(c);
    const uint32_t rr = ri < rh ? ri : rh - 1;
    const uint8_t* base = bitrows + c * 16 * (KD / 8);
#pragma unroll
    for (int q = 0; q < MG; ++q) m[q] = *reinterpret_cast<const uint16_t*>(base + (rr * static_cast<uint32_t>(KD / 8) + bit_off[q]));
  };

  // DEPTH register sets of rows in flight.  Measured (round 6, tools/linear_bf16_ablation.py): a third set in the forward
  // (184 -> 222 registers) is SLOWER, 63 -> 70 us, and so is requesting the first rows before the weight image is staged
  // (-DALLSET_BF16_EARLY_REQUEST: 54 -> 63 us) -- both kept as ablation arms only.
#ifdef ALLSET_BF16_DEPTH3
  constexpr int DEPTH = TRANS_W ? 2 : 3;
#else
  constexpr int DEPTH = 2;
#endif
  uint32_t a0[KQD], a1[KQD], a2[DEPTH == 3 ? KQD : 1];
  uint32_t m0[MN];
  int64_t chunk = static_cast<int64_t>(blockIdx.x) * kBfWaves + wave;
  auto first_requests = [&]() {
    request(a0, x, ldx, chunk);
    request(a1, x, ldx, chunk + stride);
    if constexpr (DEPTH == 3) request(a2, x, ldx, chunk + 2 * stride);
    if constexpr (HAS_MASK) request(m0, ymask, ldm, chunk);
    if constexpr (MASK == 2) request_bits(m0, chunk);
  };
#ifdef ALLSET_BF16_EARLY_REQUEST
  first_requests();
#endif

  if constexpr (!TRANS_W) {
    // W[j][k]: 16-byte pieces copy straight into the image.  ALL of a thread's pieces are requested before the first one is
    // written (round 6, second session: as a plain loop the compiler kept ONE piece in flight -- load, s_waitcnt vmcnt(0), ds_write,
    // branch: 16 dependent L2 round trips, most of the 8 us this copy took of a 77-us launch)
    constexpr int NPW = ND * KD / 8 / kBfBlock;
    static_assert(ND * KD / 8 % kBfBlock == 0, "weight image: whole pieces per thread");
    uint4 wreg[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int idx = tid + i * kBfBlock;
      const int j = idx / (KD / 8), k0 = 8 * (idx % (KD / 8));
      wreg[i] = *reinterpret_cast<const uint4*>(W + static_cast<int64_t>(j) * KD + k0);
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int idx = tid + i * kBfBlock;
      const int j = idx / (KD / 8), k0 = 8 * (idx % (KD / 8));
      *reinterpret_cast<uint4*>(&sW[wimg_off<KQD, GS>(k0 / KQ, j, (k0 % KQ) / 8)]) = wreg[i];
    }
  } else {
    // W[o][i] (the layer's [out, in] weight), reduction over o: the image row of column i holds its 256 o-values, so the copy is
    // a transpose.  A thread takes an 8 x 8 block (8 rows o, one 16-byte piece of 8 columns each), transposes it in registers
    // (v_perm_b32: dword m of column c = {W[o0 + 2m][c], W[o0 + 2m + 1][c]}) and writes eight 16-byte pieces -- each exactly one
    // piece (k-quarter, column, 8 k) of the image.  Round 6, second session: the former copy wrote 64 single dwords per thread
    // behind one load at a time, 15 of the launch's ~78 us.  A wave-sized group of 64 blocks is 8 o-blocks x 8 column blocks: a
    // load instruction reads 128 contiguous bytes of 8 rows, a write instruction spreads over 8 distinct 16-byte bank groups.
    constexpr int NBLK = (KD / 8) * (ND / 8), NBT = (NBLK + kBfBlock - 1) / kBfBlock;
    uint4 wreg[NBT][8];
#pragma unroll
    for (int sblk = 0; sblk < NBT; ++sblk) {
      const int b = tid + sblk * kBfBlock, grp = b >> 6;
      const int ob = (b & 7) + 8 * (grp % (KD / 64)), ib = ((b >> 3) & 7) + 8 * (grp / (KD / 64));
      if (NBLK % kBfBlock == 0 || b < NBLK) {
#pragma unroll
        for (int r = 0; r < 8; ++r) wreg[sblk][r] = *reinterpret_cast<const uint4*>(W + static_cast<int64_t>(8 * ob + r) * ND + 8 * ib);
      }
    }
#pragma unroll
    for (int sblk = 0; sblk < NBT; ++sblk) {
      const int b = tid + sblk * kBfBlock, grp = b >> 6;
      const int ob = (b & 7) + 8 * (grp % (KD / 64)), ib = ((b >> 3) & 7) + 8 * (grp / (KD / 64));
      if (NBLK % kBfBlock == 0 || b < NBLK) {
        const int gq = (8 * ob) / KQ, tp = ((8 * ob) % KQ) / 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          uint32_t d[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const uint4 &ra = wreg[sblk][2 * m], &rb = wreg[sblk][2 * m + 1];
            const uint32_t a = c < 2 ? ra.x : c < 4 ? ra.y : c < 6 ? ra.z : ra.w, bb = c < 2 ? rb.x : c < 4 ? rb.y : c < 6 ? rb.z : rb.w;
            d[m] = (c & 1) ? __builtin_amdgcn_perm(bb, a, 0x07060302u) : __builtin_amdgcn_perm(bb, a, 0x05040100u);
          }
          *reinterpret_cast<uint4*>(&sW[wimg_off<KQD, GS>(gq, 8 * ib + c, tp)]) = make_uint4(d[0], d[1], d[2], d[3]);
        }
      }
    }
  }
  for (int idx = tid; idx < ND; idx += kBfBlock) sBias[idx] = bias ? bf16_to_f32(bias[idx]) : 0.f;
  if (AUX_OUT || has_aux_in) {
    constexpr int NA = (4 * AUXW + kBfBlock - 1) / kBfBlock;        // (both requests before the first write, as the weight image)
    uint16_t ar[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) ar[i] = (4 * AUXW % kBfBlock == 0 || tid + i * kBfBlock < 4 * AUXW) ? aux_w[tid + i * kBfBlock] : uint16_t{0};
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (4 * AUXW % kBfBlock == 0 || tid + i * kBfBlock < 4 * AUXW) sAux[tid + i * kBfBlock] = bf16_to_f32(ar[i]);
    if (tid < 4) sAux[4 * AUXW + tid] = (AUX_OUT && aux_b) ? bf16_to_f32(aux_b[tid]) : 0.f;
  }
  __syncthreads();

  // ---- what happens to the rows of a step before the matrix phase: the relu mask (and the next mask request: `mask_next` is
  // the step this mask buffer serves next), zeros for rows past the end, the four auxiliary output columns
  auto prologue = [&](uint32_t (&a)[KQD], uint32_t (&m)[MN], int64_t chunk, int64_t mask_next) {
    const int rh = rows_here(chunk);
    const bool valid = ri < rh;
    if constexpr (MASK == 2) {
#pragma unroll
      for (int q = 0; q < MG; ++q) {
        // bit 2 i of the field -> bit 0, bit 2 i + 1 -> bit 16 of (sp >> 2 i): the two halves' 0 / 1 factors of dword i
        const uint32_t sp = m[q] | (m[q] << 15);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[8 * q + i] = keep_halves(a[8 * q + i], (sp >> (2 * i)) & 0x00010001u);
      }
      __builtin_amdgcn_sched_barrier(0);
      request_bits(m, mask_next);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (HAS_MASK) {
#pragma unroll
      for (int j = 0; j < KQD; ++j) a[j] = keep_where_positive(a[j], m[j]);
      if (a_out != nullptr && valid) {
        uint4* p = reinterpret_cast<uint4*>(a_out + chunk * 16 * lda + (ri * static_cast<uint32_t>(lda) + g * KQ));
#pragma unroll
        for (int q = 0; q < KQD / 4; ++q) p[q] = make_uint4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
      }
      // ONE mask buffer: the next step's rows are requested as soon as this step's are consumed and have the whole matrix
      // phase and epilogue to arrive (the activation rows themselves are two steps ahead in two buffers)
      __builtin_amdgcn_sched_barrier(0);
      request(m, ymask, ldm, mask_next);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!valid) {
#pragma unroll
      for (int j = 0; j < KQD; ++j) a[j] = 0u;
    }
    if constexpr (AUX_OUT) {
      {
        // four extra output columns in plain fp32 FMAs from the rows already in registers ([n, K] x [K, 4] is all bandwidth)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int j = 0; j < KQD; j += 2) {
          const float v0 = __uint_as_float(a[j] << 16), v1 = __uint_as_float(a[j] & 0xffff0000u);
          const float v2 = __uint_as_float(a[j + 1] << 16), v3 = __uint_as_float(a[j + 1] & 0xffff0000u);
          const float4 w0 = *reinterpret_cast<const float4*>(&sAux[0 * KD + g * KQ + 2 * j]);
          const float4 w1 = *reinterpret_cast<const float4*>(&sAux[1 * KD + g * KQ + 2 * j]);
          const float4 w2 = *reinterpret_cast<const float4*>(&sAux[2 * KD + g * KQ + 2 * j]);
          const float4 w3 = *reinterpret_cast<const float4*>(&sAux[3 * KD + g * KQ + 2 * j]);
          s0 = fmaf(v0, w0.x, fmaf(v1, w0.y, fmaf(v2, w0.z, fmaf(v3, w0.w, s0))));
          s1 = fmaf(v0, w1.x, fmaf(v1, w1.y, fmaf(v2, w1.z, fmaf(v3, w1.w, s1))));
          s2 = fmaf(v0, w2.x, fmaf(v1, w2.y, fmaf(v2, w2.z, fmaf(v3, w2.w, s2))));
          s3 = fmaf(v0, w3.x, fmaf(v1, w3.y, fmaf(v2, w3.z, fmaf(v3, w3.w, s3))));
        }
        s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16); s3 += __shfl_xor(s3, 16);
        s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 += __shfl_xor(s3, 32);
        if (valid && g == 0)
          *reinterpret_cast<float4*>(aux_out + chunk * 64 + ri * 4) =
              make_float4(s0 + sAux[4 * KD], s1 + sAux[4 * KD + 1], s2 + sAux[4 * KD + 2], s3 + sAux[4 * KD + 3]);
      }
    }
  };

  // ---- matrix phase.  Swapped operands: D[i][j] = sum_k Wimg[col 16 tl + i][k] * x[row j][k]; a lane holds row (lane & 15),
  // columns 4 g .. + 3 of every tile.  (Round 6, measured and dropped: the rows of TWO steps sharing every weight fragment --
  // half the LDS reads -- left the matrix phase alone at 26.4 us against 28.6 and the whole launch 15 us SLOWER (the rows of the
  // next step can only be requested once both steps' fragments are consumed): the phase is MFMA issue + the fixed weight staging,
  // not LDS rate.  tools/linear_bf16_ablation.py, profiles/r06_bf16_linear_ablation.txt.)
  auto matrix = [&](const uint32_t (&a)[KQD], f32x4_b (&acc)[NTILE]) {
#ifdef ALLSET_BF16_ABL_NOMFMA               // (ablation: no matrix phase -- the accumulators take the rows' bits so that they stay live)
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl)
      acc[tl] = f32x4_b{__uint_as_float(a[(2 * tl) % KQD]), __uint_as_float(a[(2 * tl + 1) % KQD]), __uint_as_float(a[(2 * tl + 7) % KQD]), 0.f};
#else
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) acc[tl] = f32x4_b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t) {
      FragB fa;
      fa.u = make_uint4(a[4 * t], a[4 * t + 1], a[4 * t + 2], a[4 * t + 3]);
#pragma unroll
      for (int tl = 0; tl < NTILE; tl += 2) {       // two column tiles: two independent accumulator chains
        FragB b0, b1;
        b0.u = *reinterpret_cast<const uint4*>(&sW[wimg_off<KQD, GS>(g, tl * 16 + ri, t)]);
        b1.u = *reinterpret_cast<const uint4*>(&sW[wimg_off<KQD, GS>(g, tl * 16 + 16 + ri, t)]);
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0.v, fa.v, acc[tl], 0, 0, 0);
        acc[tl + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1.v, fa.v, acc[tl + 1], 0, 0, 0);
      }
    }
#endif
  };
  // ---- epilogue: bias / the logits' rank-4 term / the other gradient branch / relu in fp32 on the accumulators, ONE rounding
  // to bf16, then (round 6) the lanes of a row trade their 8-byte pieces in registers -- v_permlane16_swap pairs up the tiles of
  // neighbouring 16-lane rows, v_permlane32_swap the pairs -- so that lane (ri, g) ends up with the 16 consecutive columns
  // 64 hb + 16 g .. of row ri and a row leaves as whole 128-byte segments.  No LDS slab, no lgkmcnt round trips (the slab form,
  // -DALLSET_BF16_SLAB_EPILOGUE, cost 13 us of LDS waits per launch beside the stores themselves).
  auto epilogue = [&](auto with_acc, f32x4_b (&acc)[NTILE], int64_t chunk, const uint2 (&accv)[decltype(with_acc)::value ? NTILE : 1], const float4& ga4) {
    constexpr bool WITH_ACC = decltype(with_acc)::value;
    const int rh = rows_here(chunk);
#ifdef ALLSET_BF16_ABL_NOEPI                // (ablation: no epilogue; one never-true store keeps the accumulators live)
    {
      float sacc = 0.f;
#pragma unroll
      for (int tl = 0; tl < NTILE; ++tl) sacc += acc[tl][0] + acc[tl][1] + acc[tl][2] + acc[tl][3];
      if (sacc == 1.2345e-30f) y[chunk] = 1;
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
#endif
    uint32_t mword[ND / 128] = {}, mword2[ND / 128] = {};
#ifdef ALLSET_BF16_HALF_SECTOR_STORES
    constexpr bool kFullSector = false;
#else
    constexpr bool kFullSector = true;
#endif
#ifdef ALLSET_BF16_SLAB_EPILOGUE
    const int srow = lane >> 2, sq = lane & 3;
#else
    const int srow = ri, sq = g;
#endif
#pragma unroll
    for (int hb = 0; hb < ND / 64; ++hb) {
      uint2 pk[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int tl = hb * 4 + tt;
        const int c = tl * 16 + 4 * g;
        float v0 = acc[tl][0], v1 = acc[tl][1], v2 = acc[tl][2], v3 = acc[tl][3];
        if constexpr (!TRANS_W) {                  // (backward-data has no bias and no relu)
          const float4 bv = *reinterpret_cast<const float4*>(&sBias[c]);
          v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
        }
        if (has_aux_in) {
          const float4 w0 = *reinterpret_cast<const float4*>(&sAux[0 * AUXW + c]);
          const float4 w1 = *reinterpret_cast<const float4*>(&sAux[1 * AUXW + c]);
          const float4 w2 = *reinterpret_cast<const float4*>(&sAux[2 * AUXW + c]);
          const float4 w3 = *reinterpret_cast<const float4*>(&sAux[3 * AUXW + c]);
          v0 = fmaf(ga4.x, w0.x, fmaf(ga4.y, w1.x, fmaf(ga4.z, w2.x, fmaf(ga4.w, w3.x, v0))));
          v1 = fmaf(ga4.x, w0.y, fmaf(ga4.y, w1.y, fmaf(ga4.z, w2.y, fmaf(ga4.w, w3.y, v1))));
          v2 = fmaf(ga4.x, w0.z, fmaf(ga4.y, w1.z, fmaf(ga4.z, w2.z, fmaf(ga4.w, w3.z, v2))));
          v3 = fmaf(ga4.x, w0.w, fmaf(ga4.y, w1.w, fmaf(ga4.z, w2.w, fmaf(ga4.w, w3.w, v3))));
        }
        if constexpr (WITH_ACC) {
          if (has_acc) {
            v0 += __uint_as_float(accv[tl].x << 16); v1 += __uint_as_float(accv[tl].x & 0xffff0000u);
            v2 += __uint_as_float(accv[tl].y << 16); v3 += __uint_as_float(accv[tl].y & 0xffff0000u);
          }
        }
        if constexpr (!TRANS_W) {
          auto relu_if = [&](float v) {
            const uint32_t a = __float_as_uint(fmaxf(v, 0.f)), b = __float_as_uint(v);
            return __uint_as_float((a & relu_sel) | (b & ~relu_sel));
          };
          v0 = relu_if(v0); v1 = relu_if(v1); v2 = relu_if(v2); v3 = relu_if(v3);
        }
        pk[tt] = make_uint2(cvt_pk_bf16(v0, v1), cvt_pk_bf16(v2, v3));
      }
      uint4 o0, o1;
#ifdef ALLSET_BF16_SLAB_EPILOGUE
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<uint2*>(&slab[ri * kSlabPitch + tt * 8 + 2 * g]) = pk[tt];
      // one wave, in-order LDS queue: no barrier needed, but the compiler must not move the reads above the writes nor the
      // next trip's writes above these reads
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      o0 = *reinterpret_cast<const uint4*>(&slab[srow * kSlabPitch + sq * 8]);
      o1 = *reinterpret_cast<const uint4*>(&slab[srow * kSlabPitch + sq * 8 + 4]);
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
      {
        // level 1: odd 16-lane rows of tile 0 (2) <-> even rows of tile 1 (3): lane g = 0 / 2 keeps tile 0's columns 0-7 / 8-15,
        // lane g = 1 / 3 tile 1's (likewise tiles 2, 3)
        const swap2_t x01 = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
        const swap2_t y01 = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
        const swap2_t x23 = __builtin_amdgcn_permlane16_swap(pk[2].x, pk[3].x, false, false);
        const swap2_t y23 = __builtin_amdgcn_permlane16_swap(pk[2].y, pk[3].y, false, false);
        // q01 = {x01[0], y01[0], x01[1], y01[1]}: 8 consecutive columns -- lane g holds columns 8 (g >> 1) .. + 7 of tile (g & 1) in
        // q01 and of tile 2 + (g & 1) in q23.  Stored like that (round 6, second session), the four lanes of a row write 64 CONTIGUOUS
        // bytes per store instruction -- whole 64-byte sectors -- where the former second level (v_permlane32_swap: lane g = tile g,
        // 32 contiguous bytes per lane) left every sector half-written by each of the two instructions.
        if constexpr (kFullSector) {
          o0 = make_uint4(x01[0], y01[0], x01[1], y01[1]);
          o1 = make_uint4(x23[0], y23[0], x23[1], y23[1]);
        } else {
          // level 2: the upper 32 lanes of (tiles 0, 1) <-> the lower 32 of (tiles 2, 3): lane g holds tile g: columns 0-7 in o0, 8-15 in o1
          const swap2_t s0 = __builtin_amdgcn_permlane32_swap(x01[0], x23[0], false, false);
          const swap2_t s1 = __builtin_amdgcn_permlane32_swap(y01[0], y23[0], false, false);
          const swap2_t s2 = __builtin_amdgcn_permlane32_swap(x01[1], x23[1], false, false);
          const swap2_t s3 = __builtin_amdgcn_permlane32_swap(y01[1], y23[1], false, false);
          o0 = make_uint4(s0[0], s1[0], s2[0], s3[0]);
          o1 = make_uint4(s0[1], s1[1], s2[1], s3[1]);
        }
      }
#endif
#ifdef ALLSET_BF16_ABL_NOSTORE              // (ablation: the epilogue without its global stores)
      if (srow < rh && o0.x == 0x12345678u && o1.w == 0x9abcdef0u) y[chunk] = 1;
#else
      if (srow < rh) {
#ifndef ALLSET_BF16_SLAB_EPILOGUE
        if constexpr (kFullSector) {
          uint4* dst = reinterpret_cast<uint4*>(y + chunk * 16 * ldy + (srow * static_cast<uint32_t>(ldy) + (sq & 1) * 16 + (sq >> 1) * 8 + hb * 64));
          dst[0] = o0;
          dst[4] = o1;                              // (+ 32 columns)
        } else
#endif
        {
          uint4* dst = reinterpret_cast<uint4*>(y + chunk * 16 * ldy + (srow * static_cast<uint32_t>(ldy) + sq * 16 + hb * 64));
          dst[0] = o0;
          dst[1] = o1;
        }
      }
#endif
      if constexpr (MASK_OUT) {
#ifndef ALLSET_BF16_SLAB_EPILOGUE
        if constexpr (kFullSector) {
          // this lane's 8 columns of tile (sq & 1) are bits 16 hb + 8 (sq >> 1) .. of word (sq & 1), those of tile 2 + (sq & 1) the
          // same bits of word 2 + (sq & 1); the two lanes of a word trade halves after the loop
          const uint32_t sh = 16 * (hb & 1) + 8 * (sq >> 1);
          const uint32_t ba = relu_bits8(o0) << sh, bb = relu_bits8(o1) << sh;
          if (hb & 1) { mword[hb >> 1] |= ba; mword2[hb >> 1] |= bb; }
          else { mword[hb >> 1] = ba; mword2[hb >> 1] = bb; }
        } else
#endif
        {                                           // the relu mask of these 16 columns: bits 16 hb .. of this lane's word sq
          const uint32_t b16 = relu_bits16(o0, o1);
          if (hb & 1) mword[hb >> 1] |= b16 << 16;
          else mword[hb >> 1] = b16;
        }
      }
    }
    if constexpr (MASK_OUT) {
#ifndef ALLSET_BF16_SLAB_EPILOGUE
      if constexpr (kFullSector) {
        // lanes g and g ^ 2 (32 lanes apart) hold the two halves of words (g & 1) and 2 + (g & 1): after the swap the lower 32 lanes
        // have both halves of the first, the upper 32 both halves of the second -- lane g ends up with word g, as before
#pragma unroll
        for (int w = 0; w < ND / 128; ++w) {
          const swap2_t t = __builtin_amdgcn_permlane32_swap(mword[w], mword2[w], false, false);
          mword[w] = t[0] | t[1];
        }
      }
#endif
      if (srow < rh) {
        uint8_t* mp = mask_out + chunk * 16 * (ND / 8) + (srow * static_cast<uint32_t>(ND / 8) + sq * (ND / 32));
        if constexpr (ND == 256) *reinterpret_cast<uint2*>(mp) = make_uint2(mword[0], mword[1]);
        else *reinterpret_cast<uint32_t*>(mp) = mword[0];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  auto load_ga4 = [&](int64_t chunk) -> float4 {
    if (!has_aux_in) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int rh = rows_here(chunk);
    const uint32_t rr = ri < rh ? ri : rh - 1;
    return *reinterpret_cast<const float4*>(aux_in + chunk * 64 + rr * 4);
  };

  // one step: prologue, the other gradient branch's rows requested before the matrix phase, matrix phase, the rows of the step
  // after next requested into the registers the matrix phase has just freed, epilogue
  auto process = [&](uint32_t (&a)[KQD], uint32_t (&m)[MN], int64_t chunk, int ahead) {
    prologue(a, m, chunk, chunk + stride);
    uint2 accv[(TRANS_W && (RT || EXTRA)) ? NTILE : 1];
    if constexpr (TRANS_W && (RT || EXTRA)) {   // (the other gradient branch: backward-data only)
      if (has_acc) {
        const int rh = rows_here(chunk);
        const uint32_t rr = ri < rh ? ri : rh - 1;
        const uint16_t* ap = acc_in + chunk * 16 * ldacc + (rr * static_cast<uint32_t>(ldacc) + 4 * g);
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl) accv[tl] = *reinterpret_cast<const uint2*>(ap + tl * 16);
      }
    }
    const float4 ga4 = load_ga4(chunk);
    f32x4_b acc[NTILE];
    matrix(a, acc);
    __builtin_amdgcn_sched_barrier(0);
    request(a, x, ldx, chunk + ahead * stride);
    __builtin_amdgcn_sched_barrier(0);
    epilogue(std::integral_constant<bool, TRANS_W && (RT || EXTRA)>{}, acc, chunk, accv, ga4);
  };
#ifndef ALLSET_BF16_EARLY_REQUEST
  first_requests();
#endif
  // (DEPTH chunks per trip, the last ones peeled: a conditional second half makes the compiler wait vmcnt(0) at the loop
  // header -- see fused_mlp.hip)
  if constexpr (DEPTH == 3) {
    for (; chunk + 2 * stride < n_chunks; chunk += 3 * stride) {
      process(a0, m0, chunk, DEPTH);
      process(a1, m0, chunk + stride, DEPTH);
      process(a2, m0, chunk + 2 * stride, DEPTH);
    }
    if (chunk < n_chunks) process(a0, m0, chunk, DEPTH);
    if (chunk + stride < n_chunks) process(a1, m0, chunk + stride, DEPTH);
  } else {
    for (; chunk + stride < n_chunks; chunk += 2 * stride) {
      process(a0, m0, chunk, DEPTH);
      process(a1, m0, chunk + stride, DEPTH);
    }
    if (chunk < n_chunks) process(a0, m0, chunk, DEPTH);
  }
}

static inline bool bf16_width(int64_t w) { return w == 128 || w == 256; }

}  // namespace allset

using namespace allset;

extern "C" int allset_linear_bf16_supported(int64_t in_features, int64_t out_features) {
  return (bf16_width(in_features) && bf16_width(out_features)) ? 1 : 0;
}

template <int KD, int ND, bool TRANS_W>
static void launch_bf16(int mask_mode, bool aux_out, unsigned grid, hipStream_t st, const uint16_t* x, int64_t ldx,
                        const uint16_t* ymask, int64_t ldm, uint16_t* a_out, int64_t lda, const uint16_t* W,
                        const uint16_t* bias, int relu_out, const uint16_t* aux_w, const uint16_t* aux_b, float* aux_o,
                        const float* aux_in, const uint16_t* acc_in, int64_t ldacc, uint16_t* y, int64_t ldy, uint8_t* mask_out,
                        int64_t n) {
#define ALLSET_BF16_GO(MASK, AUXF, EXTRA)                                                                               \
  linear_bf16_kernel<KD, ND, TRANS_W, MASK, AUXF, EXTRA><<<grid, kBfBlock, 0, st>>>(x, ldx, ymask, ldm, a_out, lda, W,  \
                                                                                     bias, relu_out, aux_w, aux_b,      \
                                                                                     aux_o, aux_in, acc_in, ldacc, y,   \
                                                                                     ldy, mask_out, n)
  if constexpr (!TRANS_W) {
    if (aux_out) ALLSET_BF16_GO(0, true, false);
    else if (mask_out != nullptr) ALLSET_BF16_GO(0, false, true);
    else ALLSET_BF16_GO(0, false, false);
  } else {
    const bool acc = acc_in != nullptr, axi = aux_in != nullptr;
    if (mask_mode == 2) { if (acc) ALLSET_BF16_GO(2, false, true); else ALLSET_BF16_GO(2, false, false); }
    else if (mask_mode == 1) ALLSET_BF16_GO(1, false, false);
    else if (acc && axi) ALLSET_BF16_GO(0, true, true);
    else if (acc) ALLSET_BF16_GO(0, false, true);
    else if (axi) ALLSET_BF16_GO(0, true, false);
    else ALLSET_BF16_GO(0, false, false);
  }
#undef ALLSET_BF16_GO
}

static inline unsigned bf16_grid(int64_t n) {
  int64_t blocks = ((n + 15) / 16 + kBfWaves - 1) / kBfWaves;
  if (blocks > 256) blocks = 256;                    // one persistent 8-wave workgroup per CU
  return static_cast<unsigned>(blocks < 1 ? 1 : blocks);
}

static inline bool rows_ok(const void* p, int64_t ld, int64_t w) {     // 32-bit lane offsets: 16 rows x ld must stay below 2^31
  return p && aligned16(p) && ld >= w && ld % 8 == 0 && ld < (int64_t{1} << 26);
}

// bytes per row of the relu bit mask of an N-wide Linear (rows are dense: pitch = this)
extern "C" int64_t allset_linear_bf16_mask_pitch(int64_t N) { return bf16_width(N) ? N / 8 : 0; }

static int linear_bf16_fwd_impl(const void* x, int64_t ldx, const void* W, const void* bias, int relu_out, const void* aux_w,
                                const void* aux_b, float* aux_out, void* y, int64_t ldy, void* mask_out, int64_t n, int64_t K,
                                int64_t N, void* stream) {
  ALLSET_REQUIRE(n >= 0, "linear_bf16_fwd: negative size");
  if (!allset_linear_bf16_supported(K, N)) {
    set_error("linear_bf16_fwd: K=%lld N=%lld not built (K, N in {128,256})", static_cast<long long>(K), static_cast<long long>(N));
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(rows_ok(x, ldx, K) && rows_ok(y, ldy, N), "linear_bf16_fwd: x and y must be 16-byte aligned rows (ld a multiple of 8)");
  ALLSET_REQUIRE(W && aligned16(W), "linear_bf16_fwd: W must be 16-byte aligned");
  ALLSET_REQUIRE(aux_out == nullptr || (aux_w != nullptr && aligned16(aux_out)), "linear_bf16_fwd: aux_out needs aux_w and 16-byte alignment");
  ALLSET_REQUIRE(mask_out == nullptr || (relu_out && aux_out == nullptr && aligned16(mask_out)),
                 "linear_bf16_fwd: mask_out needs relu_out, no auxiliary columns and 16-byte alignment");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = bf16_grid(n);
  const uint16_t *xp = static_cast<const uint16_t*>(x), *Wp = static_cast<const uint16_t*>(W), *bp = static_cast<const uint16_t*>(bias);
  const uint16_t *awp = static_cast<const uint16_t*>(aux_w), *abp = static_cast<const uint16_t*>(aux_b);
  uint16_t* yp = static_cast<uint16_t*>(y);
  uint8_t* mo = static_cast<uint8_t*>(mask_out);
#define ALLSET_BF16_FWD(KD, ND) \
  launch_bf16<KD, ND, false>(0, aux_out != nullptr, grid, st, xp, ldx, nullptr, 0, nullptr, 0, Wp, bp, relu_out, awp, abp, aux_out, nullptr, nullptr, 0, yp, ldy, mo, n)
  if (K == 256 && N == 256) ALLSET_BF16_FWD(256, 256);
  else if (K == 256 && N == 128) ALLSET_BF16_FWD(256, 128);
  else if (K == 128 && N == 256) ALLSET_BF16_FWD(128, 256);
  else ALLSET_BF16_FWD(128, 128);
#undef ALLSET_BF16_FWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_linear_bf16_fwd(const void* x, int64_t ldx, const void* W, const void* bias, int relu_out,
                                      const void* aux_w, const void* aux_b, float* aux_out, void* y, int64_t ldy,
                                      int64_t n, int64_t K, int64_t N, void* stream) {
  clear_error();
  return linear_bf16_fwd_impl(x, ldx, W, bias, relu_out, aux_w, aux_b, aux_out, y, ldy, nullptr, n, K, N, stream);
}

// the same with the relu bit mask of y as a second output (mask_out: n rows of allset_linear_bf16_mask_pitch(N) bytes)
extern "C" int allset_linear_bf16_fwd_mask(const void* x, int64_t ldx, const void* W, const void* bias, void* y, int64_t ldy,
                                           void* mask_out, int64_t n, int64_t K, int64_t N, void* stream) {
  clear_error();
  ALLSET_REQUIRE(mask_out != nullptr, "linear_bf16_fwd_mask: null mask_out");
  return linear_bf16_fwd_impl(x, ldx, W, bias, 1, nullptr, nullptr, nullptr, y, ldy, mask_out, n, K, N, stream);
}

static int linear_bf16_bwd_impl(const void* gy, int64_t ldg, const void* ymask, int64_t ldm, int mask_mode, void* ga_out,
                                int64_t lda, const void* W, const float* galpha, const void* aux_w, const void* acc_in,
                                int64_t ldacc, void* gx, int64_t ldgx, int64_t n, int64_t O, int64_t I, void* stream) {
  ALLSET_REQUIRE(n >= 0, "linear_bf16_bwd: negative size");
  if (!allset_linear_bf16_supported(I, O)) {
    set_error("linear_bf16_bwd: O=%lld I=%lld not built (O, I in {128,256})", static_cast<long long>(O), static_cast<long long>(I));
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(rows_ok(gy, ldg, O) && rows_ok(gx, ldgx, I), "linear_bf16_bwd: gy and gx must be 16-byte aligned rows (ld a multiple of 8)");
  ALLSET_REQUIRE(W && aligned16(W), "linear_bf16_bwd: W must be 16-byte aligned");
  ALLSET_REQUIRE(mask_mode != 1 || rows_ok(ymask, ldm, O), "linear_bf16_bwd: ymask must be 16-byte aligned rows");
  ALLSET_REQUIRE(mask_mode != 2 || (ymask != nullptr && (reinterpret_cast<uintptr_t>(ymask) & 7u) == 0), "linear_bf16_bwd_bits: the bit mask must be 8-byte aligned");
  ALLSET_REQUIRE(ga_out == nullptr || (mask_mode == 1 && rows_ok(ga_out, lda, O)), "linear_bf16_bwd: ga_out needs ymask and 16-byte aligned rows");
  ALLSET_REQUIRE(acc_in == nullptr || (ldacc >= I && ldacc % 4 == 0 && ldacc < (int64_t{1} << 26) && (reinterpret_cast<uintptr_t>(acc_in) & 7u) == 0),
                 "linear_bf16_bwd: acc_in must be 8-byte aligned rows");
  ALLSET_REQUIRE(galpha == nullptr || (aux_w != nullptr && aligned16(galpha)), "linear_bf16_bwd: galpha needs aux_w and 16-byte alignment");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = bf16_grid(n);
  const uint16_t *gp = static_cast<const uint16_t*>(gy), *mp = static_cast<const uint16_t*>(ymask), *Wp = static_cast<const uint16_t*>(W);
  const uint16_t *awp = static_cast<const uint16_t*>(aux_w), *ap = static_cast<const uint16_t*>(acc_in);
  uint16_t *gap = static_cast<uint16_t*>(ga_out), *gxp = static_cast<uint16_t*>(gx);
#define ALLSET_BF16_BWD(KD, ND) \
  launch_bf16<KD, ND, true>(mask_mode, false, grid, st, gp, ldg, mp, ldm, gap, lda, Wp, nullptr, 0, awp, nullptr, nullptr, galpha, ap, ldacc, gxp, ldgx, nullptr, n)
  if (O == 256 && I == 256) ALLSET_BF16_BWD(256, 256);
  else if (O == 256 && I == 128) ALLSET_BF16_BWD(256, 128);
  else if (O == 128 && I == 256) ALLSET_BF16_BWD(128, 256);
  else ALLSET_BF16_BWD(128, 128);
#undef ALLSET_BF16_BWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_linear_bf16_bwd(const void* gy, int64_t ldg, const void* ymask, int64_t ldm, void* ga_out,
                                      int64_t lda, const void* W, const float* galpha, const void* aux_w,
                                      const void* acc_in, int64_t ldacc, void* gx, int64_t ldgx, int64_t n, int64_t O,
                                      int64_t I, void* stream) {
  clear_error();
  return linear_bf16_bwd_impl(gy, ldg, ymask, ldm, ymask != nullptr ? 1 : 0, ga_out, lda, W, galpha, aux_w, acc_in, ldacc, gx, ldgx, n, O, I, stream);
}

// backward-data behind a relu whose mask is the forward's bit mask (allset_linear_bf16_fwd_mask): gx = (gy where bit) W [+ acc_in]
extern "C" int allset_linear_bf16_bwd_bits(const void* gy, int64_t ldg, const void* bits, const void* W, const void* acc_in,
                                           int64_t ldacc, void* gx, int64_t ldgx, int64_t n, int64_t O, int64_t I, void* stream) {
  clear_error();
  ALLSET_REQUIRE(bits != nullptr, "linear_bf16_bwd_bits: null bit mask");
  return linear_bf16_bwd_impl(gy, ldg, bits, 0, 2, nullptr, 0, W, nullptr, nullptr, acc_in, ldacc, gx, ldgx, n, O, I, stream);
}
