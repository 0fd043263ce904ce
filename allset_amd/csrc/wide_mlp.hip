// Linear layers wider than one LDS-resident weight (reference layers.py:571-579 with MLP_hidden 256 / 512, the widths of
// src/run_AllSetTransformer.sh): a tiled GEMM on the bf16 matrix pipe with fp32-accurate arithmetic ("bf16x6", common.h).
//
//   out[r, n] = epi( sum_k pro(A)[r, k] * B[n, k] + bias[n] )          A: [rows, K] fp32,  B: [N, K] (weight, or its transpose)
//
// fused_mlp.hip keeps the three bf16 planes of a whole <=128 x 128 weight in LDS and streams rows past it; a 256 x 256
// weight is 384 KB of planes, so here the weight streams too: a workgroup owns a 128-row x 256-column output tile (fp32
// accumulators: 64 VGPRs per lane) and walks K in steps of 32, staging per step the A slab (split into planes on the fly,
// with the prologue -- mask / relu / LayerNorm-apply / dropout -- applied per element as it passes) and the matching
// 48 KB image of pre-split weight planes (allset_gemm_x6_planes lays them out as the LDS wants them, the six 1-KB pieces a
// wave copies per step side by side: gx_image_at; 384 KB per 256 x 256 weight, L2-resident).  Double-buffered: 144 KB of the 160 KB LDS, one workgroup of 8 waves per CU.
// Per step and wave: 24 ds_read_b128 feed 96 MFMAs (16 accumulator tiles x 6 plane products) -- the kernel is meant to
// be matrix-pipe-bound: 2*rows*N*K*6 flop at the bf16 rate is ~1.6x the HBM time of its operands at K = N = 256.
#include <type_traits>

#include "common.h"

namespace allset {

constexpr int kGxBM = 128, kGxBN = 256, kGxKS = 32, kGxThreads = 512;

// LDS rows are 32 bf16 = four 16-byte pieces; an MFMA fragment read (ds_read_b128) has lane (fr = l & 15, fg = l >> 4) take
// piece fg of row fr.  gfx950 services that instruction in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32 for
// the upper half; MI355X_MICROARCH.md, LDS): per group 8 lanes of one piece on rows {0-3,12-15} and 8 lanes of the next
// piece on rows {4-11}, and the 16 of them must land on the 16 distinct 16-byte slots of the 256-byte bank line.  Storing
// piece p of row r at position p ^ h((r >> 2) & 3) with h = {0, 2, 3, 1} does exactly that (slot = 4 * (r & 3) + position).
__host__ __device__ __forceinline__ int gx_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }
using bf16x8_t = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4_t = __attribute__((ext_vector_type(4))) float;
union GxFrag { uint4 u; bf16x8_t v; };

// Where dword `at` of a K step's LDS slab of B (3 planes x 256 n x 32 k bf16 = 3072 pieces of 16 bytes; thread t of the
// 512 copies pieces t, t + 512, ..., t + 2560) sits in the global image: the six pieces of one WAVE are adjacent (6 x 1 KB),
// so one 64-bit address per K step reaches all six through the +-4 KB immediate of global_load, each load still 1 KB contiguous.
template <int NPC = 6>       // pieces a thread copies per K step: 6 (three bf16 planes) or 4 (two fp16 planes)
__host__ __device__ __forceinline__ int gx_image_at(int at) {
  const int j = at >> 2, piece = j >> 9, t = j & 511;
  return (((t >> 6) * NPC + piece) * 64 + (t & 63)) * 4 + (at & 3);
}

// ---- weight planes: [n_tile][k_step] slabs of [plane][256 n][32 k] bf16 (zero-padded in n), pieces permuted by gx_image_at ------------------------------------------
__global__ __launch_bounds__(kBlock) void gemm_x6_planes_kernel(const float* __restrict__ W, int64_t ldw, int transpose,
                                                                uint32_t* __restrict__ planes, int N, int K) {
  const int n_pad = (N + kGxBN - 1) / kGxBN * kGxBN;
  const int64_t pairs = static_cast<int64_t>(n_pad) * (K / 2);
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < pairs;
       idx += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int n = static_cast<int>(idx / (K / 2));
    const int k = 2 * static_cast<int>(idx - static_cast<int64_t>(n) * (K / 2));
    float w0 = 0.f, w1 = 0.f;
    if (n < N) {
      if (transpose) { w0 = W[static_cast<int64_t>(k) * ldw + n]; w1 = W[static_cast<int64_t>(k + 1) * ldw + n]; }
      else { w0 = W[static_cast<int64_t>(n) * ldw + k]; w1 = W[static_cast<int64_t>(n) * ldw + k + 1]; }
    }
    uint32_t ph, pm, pl;
    split3_bf16(w0, w1, ph, pm, pl);
    const int nt = n / kGxBN, nn = n % kGxBN, ks = k / kGxKS, kk = k % kGxKS;
    const int64_t image = (static_cast<int64_t>(nt) * (K / kGxKS) + ks) * 3 * (kGxBN * kGxKS / 2);   // dwords
    const int at = nn * (kGxKS / 2) + (((kk >> 3) ^ gx_swz(nn)) << 2) + ((kk & 7) >> 1);   // swizzled piece (dword of the LDS slab)
    planes[image + gx_image_at(at)] = ph;
    planes[image + gx_image_at(at + kGxBN * kGxKS / 2)] = pm;
    planes[image + gx_image_at(at + 2 * (kGxBN * kGxKS / 2))] = pl;
  }
}

// ---- the same weight in TWO fp16 planes ("fp16x3", common.h split2_f16c) -----------------------------------------------------------
// Row n of B is scaled by 2^(140 - e_n), e_n = the biased exponent of its largest element (so that it lands in [2^13, 2^14)); the
// inverse 2^(e_n - 140) per output column is stored behind the planes (float[n_pad]) and applied in the GEMM's epilogue.
// ONE launch (round 6, second session: at dataset scale a step rebuilds ten plane images -- W and W^T of five wide Linears -- and the
// former pair of launches per image, scales then planes, was 107 us of a 1-ms step): a wave owns output row n, finds its largest
// element, then reads the row again (out of L1 / L2) and writes the scaled planes.
__device__ __forceinline__ void gemm_f16_planes_row(const float* __restrict__ W, int64_t ldw, int transpose, uint32_t* __restrict__ planes,
                                                    float* __restrict__ bscale, int N, int n_pad, int K, int n, int lane) {
  if (n >= n_pad) return;
  float amax = 0.f;
  if (n < N)
    for (int k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(transpose ? W[static_cast<int64_t>(k) * ldw + n] : W[static_cast<int64_t>(n) * ldw + k]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  const int e = min(max(static_cast<int>(__float_as_uint(amax) >> 23), 20), 254);
  if (lane == 0) bscale[n] = __uint_as_float(static_cast<uint32_t>(e - 13) << 23);        // 2^(e - 140)
  const float sc = __uint_as_float(static_cast<uint32_t>(267 - e) << 23);                 // 2^(140 - e): field 13 .. 247
  const int nt = n / kGxBN, nn = n % kGxBN;
  for (int k = 2 * lane; k < K; k += 128) {
    float w0 = 0.f, w1 = 0.f;
    if (n < N) {
      if (transpose) { w0 = W[static_cast<int64_t>(k) * ldw + n]; w1 = W[static_cast<int64_t>(k + 1) * ldw + n]; }
      else { w0 = W[static_cast<int64_t>(n) * ldw + k]; w1 = W[static_cast<int64_t>(n) * ldw + k + 1]; }
    }
    uint32_t ph, pl;
    split2_f16c(w0 * sc, w1 * sc, ph, pl);
    const int ks = k / kGxKS, kk = k % kGxKS;
    const int64_t image = (static_cast<int64_t>(nt) * (K / kGxKS) + ks) * 2 * (kGxBN * kGxKS / 2);   // dwords
    const int at = nn * (kGxKS / 2) + (((kk >> 3) ^ gx_swz(nn)) << 2) + ((kk & 7) >> 1);
    planes[image + gx_image_at<4>(at)] = ph;
    planes[image + gx_image_at<4>(at + kGxBN * kGxKS / 2)] = pl;
  }
}

__global__ __launch_bounds__(kBlock) void gemm_f16_planes_fused_kernel(const float* __restrict__ W, int64_t ldw, int transpose,
                                                                       uint32_t* __restrict__ planes, float* __restrict__ bscale,
                                                                       int N, int n_pad, int K) {
  gemm_f16_planes_row(W, ldw, transpose, planes, bscale, N, n_pad, K, blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), threadIdx.x & 63);
}

// ... and the plane images of MANY weights in one launch (ABI 14): a dataset-scale training step at the reference's tuned widths
// rebuilds ten of them -- W and W^T of five wide Linears -- each a ~5-us launch in front of its GEMM on the step's dependent chain; all
// weights are known when the step starts.  Each workgroup finds its weight by a scan of the table's block offsets.
constexpr int kPlaneBatchMax = 32;
struct PlaneBatchTable {
  const float* W[kPlaneBatchMax];
  uint32_t* planes[kPlaneBatchMax];
  int32_t ldw[kPlaneBatchMax], N[kPlaneBatchMax], n_pad[kPlaneBatchMax], K[kPlaneBatchMax], transpose[kPlaneBatchMax];
  int32_t first_block[kPlaneBatchMax + 1];
  int32_t count;
};
__global__ __launch_bounds__(kBlock) void gemm_f16_planes_batched_kernel(PlaneBatchTable tb) {
  const int b = blockIdx.x;
  int t = 0;
  while (t + 1 < tb.count && tb.first_block[t + 1] <= b) ++t;
  float* bscale = reinterpret_cast<float*>(reinterpret_cast<char*>(tb.planes[t]) + static_cast<int64_t>(tb.n_pad[t]) * tb.K[t] * 2 * 2);
  gemm_f16_planes_row(tb.W[t], tb.ldw[t], tb.transpose[t], tb.planes[t], bscale, tb.N[t], tb.n_pad[t], tb.K[t],
                      (b - tb.first_block[t]) * (kBlock / 64) + (threadIdx.x >> 6), threadIdx.x & 63);
}

struct GxPro {
  const float* y;          // backward: A = gy * (y > 0 ? 1/(1-p_mask) : 0), y [rows, K] (the forward's output); or NULL
  int64_t ldy;
  const uint32_t* mask;    // the same test from the forward's 1-bit activation mask ("mask layout", K % 64 == 0) instead of y; or NULL
  float p_mask;
  int relu_in;
  const float* stats;      // LayerNorm-apply: (a - mean) * rstd * gamma[k] + beta[k]; or NULL
  const float* gamma;
  const float* beta;
  float p_in;              // dropout on the prologue's result, index r*K + k
  uint64_t seed_in;
};

struct GxEpi {
  const float* bias;       // [N] or NULL
  int relu_out;
  float p_out;             // dropout on the output, index r*N + n
  uint64_t seed_out;
  // LayerNorm-backward epilogue (N <= 256: a tile holds whole rows): the GEMM's result is the gradient of
  // u = dropout_p(LN(relu_in ? relu(x) : x)); what is stored is the gradient of x, and the workgroup's sums of
  // dgamma / dbeta go to lnb_part[workgroup][0|1][N].  Active iff lnb_x != NULL (then bias / relu_out / p_out are unused).
  uint32_t* mask_out;      // 1 bit per output element "out > 0" after the epilogue ("mask layout", N % 64 == 0); or NULL
  const float* lnb_x;
  int64_t lnb_ldx;
  const float* lnb_stats;
  const float* lnb_gamma;
  int lnb_relu_in;
  float lnb_p;
  uint64_t lnb_seed;
  float* lnb_part;
  // the row statistics the NEXT Linear's LayerNorm prologue needs ({mean, rstd} of stats_relu ? relu(out) : out; N == 256: a wave holds
  // whole rows), written by the row pass instead of a separate allset_row_stats pass over the output; or NULL
  float* stats_out;
  float stats_eps;
  int stats_relu;
  // backward-data of a Linear whose input was relu(x) (no LayerNorm in between: PMA's rFF): the result is stored where x > 0 and zero
  // elsewhere -- the relu's backward as the GEMM's epilogue instead of an elementwise pass over [rows, N] (ABI 15; the split-role
  // kernel's own instantiations: allset_gemm_wide_sgn_supported); or NULL
  const float* sgn_x;
  int64_t sgn_ldx;
};

// out = x > 0 ? out : 0, four columns
__device__ __forceinline__ float4 gx_sign_mask(float4 o, float4 x) {
  return make_float4(x.x > 0.f ? o.x : 0.f, x.y > 0.f ? o.y : 0.f, x.z > 0.f ? o.z : 0.f, x.w > 0.f ? o.w : 0.f);
}

// Per-tile staging context of one thread: the row it stages (128 rows x 4 segments of 8 floats per K step).
struct GxRow {
  const float* a;          // A + row * lda + seg * 8 (row clamped: loads are unconditional)
  const float* y;          // mask source, or NULL
  const uint32_t* m;       // mask words of the row (bit-mask form), or NULL
  int64_t g_row;
  float mean, rstd;
  float asc;               // fp16x3 without a LayerNorm prologue: the power of two this row's A elements are scaled by (its inverse goes
  bool ok;                 // to sRowInv for the epilogue)
};
using f16x8_t = __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16;
union GxFragH { uint4 u; f16x8_t v; };
// sum over the 64 lanes of a wave, the same value in every lane: DPP sums inside each row of 16 lanes, then the four row totals by
// v_readlane (scalar registers) -- ~11 short-latency instructions where six __shfl_xor steps are six dependent ds_bpermute round trips
template <int CTRL>
__device__ __forceinline__ float gx_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float gx_wave_sum(float v) {
  v += gx_dpp<0xB1>(v);          // quad_perm [1,0,3,2]
  v += gx_dpp<0x4E>(v);          // quad_perm [2,3,0,1]
  v += gx_dpp<0x141>(v);         // row_half_mirror
  v += gx_dpp<0x140>(v);         // row_mirror
  const int b = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

// F16 (round 5, "fp16x3"): two fp16 planes per operand and three MFMAs per product instead of three bf16 planes and six -- half the
// matrix work of a kernel whose matrix pipe is its critical resource.  fp16's five exponent bits need every operand
// inside a window: B per output column (gemm_f16_planes_fused_kernel; inverse scales `bscale`), A behind a LayerNorm-apply prologue by ONE
// power of two for the launch (|u| <= (sqrt(K - 1) max|gamma| + max|beta|) keep -- folded into the LDS copy of gamma / beta, no
// instruction), any other A per ROW: the row's largest element to [2^13, 2^14).  That row maximum needs the whole row before its
// first K step: every thread reads, one TILE ahead, the segments it will stage for the next tile (same addresses: the second read
// is an L2 / Infinity Cache hit, HBM traffic unchanged) and folds their maximum; the four threads of a row combine by two shuffles
// when the tile changes.  The mask / relu of the prologue only shrink elements: the unmasked maximum is a valid bound.
// YM: where the prologue's "forward output > 0" test comes from -- 0 none, 1 the forward's fp32 output y (a second [rows, K] read),
// 2 its 1-bit activation mask (one dword per thread and K step; round 5).
template <int YM, bool LNB, bool F16 = false>
__global__ __launch_bounds__(kGxThreads) void gemm_x6_kernel(
    const float* __restrict__ A, int64_t lda, GxPro pro, const uint4* __restrict__ planes, GxEpi epi,
    float* __restrict__ out, int64_t ldo, int64_t rows, int N, int K, const uint64_t* __restrict__ seed_base,
    const float* __restrict__ bscale = nullptr) {
  // one arena: A slabs [buffer][plane][row][32 x 16 bit = 4 x 16 B], B slabs likewise, and -- after the K loop -- the fp32
  // output tile on its way from the MFMA layout to row-major 16-byte stores
  constexpr int NP = F16 ? 2 : 3, NPC = 2 * NP;
  // kSwap (round 6, the forward): the MFMA operands trade places -- the weight fragment is the A operand, the activation fragment B --
  // so a lane's four accumulator registers of a 16 x 16 tile are FOUR CONSECUTIVE OUTPUT COLUMNS of one row (lane (fr, fg): row fr,
  // columns 4 fg .. 4 fg + 3) instead of one column of four rows: bias / relu / dropout / the 1-bit mask / the next LayerNorm's row
  // statistics are applied where the MFMA left the numbers and the tile leaves as 16-byte stores -- no trip of the 133-KB output tile
  // through LDS, no epilogue barriers, and the staging arena is free for the next tile's first K step while slower waves finish their
  // epilogue.  The fragments in LDS are the same either way.  Measured (profiles/r06_wide_swap.txt): forward 0.655-0.68 -> 0.648 ms at
  // [1M, 256] x [256, 256] -- the row pass was never the 24-29 % its cycle share suggested: a tile takes ~21 us, of which the K loop's
  // eight ~2.3-us steps are 17-19.  The LayerNorm-BACKWARD form keeps the LDS trip: its epilogue needs the tile's 128 KB of x rows in
  // flight at once (64 registers per lane), which only fit because the accumulators have left for LDS by then -- the register-resident
  // form was built and measured (x rows streamed through two 16-register sets, twice: eight dependent round trips per tile): 0.94 ->
  // 1.43 ms, reverted.
  constexpr bool kSwap = !LNB;
  constexpr int kASlab = NP * kGxBM * 4, kBSlab = NP * kGxBN * 4, kOutPitch = kGxBN + 4;
  constexpr int kSlabs = 2 * kASlab + 2 * kBSlab, kTile16 = (kGxBM * kOutPitch * 4 + 15) / 16;
  __shared__ __attribute__((aligned(16))) uint4 smem[(LNB && kTile16 > kSlabs) ? kTile16 : kSlabs];   // (the forward's tile never visits LDS: kSwap)
  // fp16x3: what undoes the row's (or the launch's) A scale in the epilogue; by tile parity (kSwap: a wave may start the next tile --
  // and write its row scales -- while another still reads this tile's in its epilogue; they meet again at the next tile's first barrier)
  __shared__ float sRowInv[F16 ? 2 * kGxBM : 1];
  __shared__ __attribute__((aligned(16))) float sStat[kGxBM * 4];    // kSwap + stats_out: per row, the four column quarters' partial sums
  // gamma / beta of the LayerNorm-apply prologue, once per workgroup: fetched from global memory inside the staging path
  // they would sit behind an s_waitcnt vmcnt(0) -- which also drains the prefetch of the next A / B slabs -- four times a step
  __shared__ __attribute__((aligned(16))) float sGB[2 * 512];
  uint4 (*sA)[kASlab] = reinterpret_cast<uint4 (*)[kASlab]>(smem);
  uint4 (*sB)[kBSlab] = reinterpret_cast<uint4 (*)[kBSlab]>(smem + 2 * kASlab);
  float* sOut = reinterpret_cast<float*>(smem);
  const int n_tiles = (N + kGxBN - 1) / kGxBN;
  const int64_t total = (rows + kGxBM - 1) / kGxBM * n_tiles;           // tiles; a workgroup walks tiles b, b + grid, ...
  const int ksteps = K / kGxKS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the wave's index as a SCALAR: everything the epilogues derive from "this wave's row" (64-bit row offsets, dropout counters, the
  // mask word's address) is then scalar arithmetic -- as tid >> 6 the compiler keeps it per lane, and the forward's row pass spent as
  // many vector instructions per tile as the whole K loop, a third of them 64-bit address arithmetic (tools/gemm_wide_ablation.py)
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t seed_in = resolve_seed(seed_base, pro.seed_in), seed_out = resolve_seed(seed_base, epi.seed_out);
  const float inv_mask = pro.p_mask > 0.f ? 1.f / (1.f - pro.p_mask) : 1.f;
  const float inv_in = pro.p_in > 0.f ? 1.f / (1.f - pro.p_in) : 1.f;
  const uint32_t thr_in = drop_threshold(pro.p_in);
  const float inv_out = epi.p_out > 0.f ? 1.f / (1.f - epi.p_out) : 1.f;
  const uint32_t thr_out = drop_threshold(epi.p_out);

  // ---- staging roles
  const int s_row = tid >> 2, s_seg = tid & 3;
  const int a_st = s_row * 4 + (s_seg ^ gx_swz(s_row));                 // swizzled 16-byte piece (gx_swz)
  const int b_img = (tid >> 6) * (64 * NPC) + (tid & 63) + 32 * NPC;   // gx_image_at: this wave's NPC pieces, from the middle
  const bool rowsc = F16 && pro.stats == nullptr;                      // A scaled per row (no LayerNorm bound)
  const bool mask_fold = YM == 2 && rowsc;                             // the mask's 1 / keep multiplied into the row scale (nothing non-linear in between)
  auto row_ctx = [&](int64_t tile) {
    GxRow c;
    const int64_t row0 = tile / n_tiles * kGxBM;
    c.g_row = row0 + s_row;
    c.ok = c.g_row < rows;
    const int64_t cr = c.ok ? c.g_row : rows - 1;
    c.a = A + cr * lda + s_seg * 8;
    c.y = pro.y ? pro.y + cr * pro.ldy + s_seg * 8 : nullptr;
    c.m = pro.mask ? pro.mask + (cr >> 4) * (K / 64) * 32 + ((cr & 15) >> 2) * 8 + (cr & 3) * 2 : nullptr;
    c.mean = 0.f; c.rstd = 1.f; c.asc = 1.f;
    if (pro.stats) { c.mean = pro.stats[cr * 2]; c.rstd = pro.stats[cr * 2 + 1]; }
    return c;
  };
  // fp16x3, per-row window: the scale from the maximum of |A[row, :]| gathered by the row's four threads
  auto row_scale = [&](float amax) -> float {
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    const int e = min(max(static_cast<int>(__float_as_uint(amax * inv_mask * inv_in) >> 23), 20), 254);
    return __uint_as_float(static_cast<uint32_t>(267 - e) << 23);      // 2^(140 - e): the largest element -> [2^13, 2^14)
  };
  auto amax8 = [](float4 a, float4 b, float m) -> float {
    return fmaxf(fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                       fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))), m);
  };

  // One K step ahead of the MFMAs, in named registers -- not arrays: hipcc keeps indexed staging arrays in scratch, and a
  // scratch store right behind the global load is a full wait, exactly the latency the prefetch hides.
  float4 pa0, pa1, py0 = make_float4(1.f, 1.f, 1.f, 1.f), py1 = py0;
  uint4 pb0, pb1, pb2, pb3, pb4, pb5;
  uint32_t pm = 0;                                                     // YM == 2: the mask dword of this thread's row and K step
  float4 la0 = make_float4(0.f, 0.f, 0.f, 0.f), la1 = la0;            // fp16x3 per-row window: the next tile's segments, one tile ahead
  float nmax = 0.f;
#define GX_LOAD(ctx_, img_base_, ks_)                                                           \
  do {                                                                                          \
    const int k0_ = (ks_) * kGxKS;                                                              \
    pa0 = *reinterpret_cast<const float4*>((ctx_).a + k0_);                                     \
    pa1 = *reinterpret_cast<const float4*>((ctx_).a + k0_ + 4);                                 \
    if constexpr (YM == 1) {                                                                    \
      py0 = *reinterpret_cast<const float4*>((ctx_).y + k0_);                                   \
      py1 = *reinterpret_cast<const float4*>((ctx_).y + k0_ + 4);                               \
    }                                                                                           \
    if constexpr (YM == 2) pm = (ctx_).m[((ks_) >> 1) * 32 + ((ks_) & 1)];                      \
    const uint4* img_ = (img_base_) + static_cast<int64_t>(ks_) * kBSlab + b_img;               \
    if constexpr (F16) {                                                                        \
      pb0 = img_[-128]; pb1 = img_[-64]; pb2 = img_[0]; pb3 = img_[64];                         \
    } else {                                                                                    \
      pb0 = img_[-192]; pb1 = img_[-128]; pb2 = img_[-64];                                      \
      pb3 = img_[0]; pb4 = img_[64]; pb5 = img_[128];                                           \
    }                                                                                           \
  } while (0)

  // The prologue on the thread's 8 consecutive elements of a K step (columns kb .. kb + 7 of its row): ONE uniform branch per switch and
  // step (it was one per element pair), the dropout by two quad hashes (four pair hashes), its 1 / keep folded into the LDS copy of
  // gamma / beta, and no zeroing of rows past the end -- they re-read the last row and their output rows are never stored
  // (tools/gemm_wide_ablation.py: prologue + split + stores were a third of the critical waves' cycles).
  auto stage8 = [&](const GxRow& c, int kb, float (&e)[8]) {
    e[0] = pa0.x; e[1] = pa0.y; e[2] = pa0.z; e[3] = pa0.w; e[4] = pa1.x; e[5] = pa1.y; e[6] = pa1.z; e[7] = pa1.w;
    if constexpr (YM == 1) {
      const float y[8] = {py0.x, py0.y, py0.z, py0.w, py1.x, py1.y, py1.z, py1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = y[i] > 0.f ? e[i] * inv_mask : 0.f;
    }
    if constexpr (YM == 2) {
      // bits 8 q + 2 s_seg (+ 1) of the step's mask dword: columns kb + q (kb + 4 + q) of the row.  Bit -> all-ones / zero -> AND with
      // the float's bits: two instructions per element where a test, a select and a multiply on a float copy of the bit were six
      // (the masked backward GEMM ran 14 % behind the unmasked one); 1 / keep rides on the row scale where that commutes (mask_fold).
      const int w = static_cast<int>(pm);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = __int_as_float(__float_as_int(e[i]) & __builtin_amdgcn_sbfe(w, 2 * s_seg + 8 * i, 1));
        e[4 + i] = __int_as_float(__float_as_int(e[4 + i]) & __builtin_amdgcn_sbfe(w, 2 * s_seg + 8 * i + 1, 1));
      }
      if (!mask_fold) {
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] *= inv_mask;
      }
    }
    if (pro.relu_in) {
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = fmaxf(e[i], 0.f);
    }
    if (pro.stats) {
      const float4 g0 = *reinterpret_cast<const float4*>(sGB + kb), g1 = *reinterpret_cast<const float4*>(sGB + kb + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(sGB + 512 + kb), b1 = *reinterpret_cast<const float4*>(sGB + 512 + kb + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = fmaf((e[i] - c.mean) * c.rstd, g[i], b[i]);
      if (pro.p_in > 0.f) {                                            // (an input dropout comes with a LayerNorm: wide_linear_supported)
        const float4 k0 = keep_scale4(seed_in, c.g_row * K + kb, thr_in, 1.f), k1 = keep_scale4(seed_in, c.g_row * K + kb + 4, thr_in, 1.f);
        e[0] *= k0.x; e[1] *= k0.y; e[2] *= k0.z; e[3] *= k0.w; e[4] *= k1.x; e[5] *= k1.y; e[6] *= k1.z; e[7] *= k1.w;
      }
    }
  };
#define GX_STORE(ctx_, ks_, buf_)                                                               \
  do {                                                                                          \
    const int kb_ = (ks_) * kGxKS + s_seg * 8;                                                  \
    float ee_[8];                                                                               \
    stage8(ctx_, kb_, ee_);                                                                     \
    const float e0 = ee_[0], e1 = ee_[1], e2 = ee_[2], e3 = ee_[3], e4 = ee_[4], e5 = ee_[5], e6 = ee_[6], e7 = ee_[7]; \
    uint4 h_, m_, l_;                                                                           \
    if constexpr (F16) {                                                                        \
      const float sc_ = mask_fold ? (ctx_).asc * inv_mask : (ctx_).asc;                         \
      split2_f16c(e0 * sc_, e1 * sc_, h_.x, l_.x);                                              \
      split2_f16c(e2 * sc_, e3 * sc_, h_.y, l_.y);                                              \
      split2_f16c(e4 * sc_, e5 * sc_, h_.z, l_.z);                                              \
      split2_f16c(e6 * sc_, e7 * sc_, h_.w, l_.w);                                              \
      sA[buf_][a_st] = h_;                                                                      \
      sA[buf_][kGxBM * 4 + a_st] = l_;                                                          \
      sB[buf_][tid] = pb0; sB[buf_][tid + kGxThreads] = pb1; sB[buf_][tid + 2 * kGxThreads] = pb2; \
      sB[buf_][tid + 3 * kGxThreads] = pb3;                                                     \
    } else {                                                                                    \
      split3_bf16(e0, e1, h_.x, m_.x, l_.x);                                                    \
      split3_bf16(e2, e3, h_.y, m_.y, l_.y);                                                    \
      split3_bf16(e4, e5, h_.z, m_.z, l_.z);                                                    \
      split3_bf16(e6, e7, h_.w, m_.w, l_.w);                                                    \
      sA[buf_][a_st] = h_;                                                                      \
      sA[buf_][kGxBM * 4 + a_st] = m_;                                                          \
      sA[buf_][2 * kGxBM * 4 + a_st] = l_;                                                      \
      sB[buf_][tid] = pb0; sB[buf_][tid + kGxThreads] = pb1; sB[buf_][tid + 2 * kGxThreads] = pb2; \
      sB[buf_][tid + 3 * kGxThreads] = pb3; sB[buf_][tid + 4 * kGxThreads] = pb4;               \
      sB[buf_][tid + 5 * kGxThreads] = pb5;                                                     \
    }                                                                                           \
  } while (0)

  // ---- MFMA roles: wave (wr, wc) owns rows wr*64.., columns wc*64.. of the tile as 4 x 4 tiles of 16 x 16
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int sw = gx_swz(fr);                                           // tile rows start at multiples of 16
  const int a_at = (wr * 64 + fr) * 4 + (fg ^ sw), b_at = (wc * 64 + fr) * 4 + (fg ^ sw);
  // The two waves that share a SIMD (w and w + 4: waves are dealt to SIMDs cyclically) run a step's two halves in
  // opposite order -- one multiplies out of the current buffers while the other splits and stores the next ones -- so the
  // matrix pipe and the VALU / LDS-store path are busy at the same time instead of taking turns.
  const bool mfma_first = wave < 4;
  f32x4_t acc[4][4];

  auto compute = [&](int buf) {
    if constexpr (F16) {
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        GxFragH a[2][2];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int p = 0; p < 2; ++p) a[r2][p].u = sA[buf][p * kGxBM * 4 + a_at + (rh * 2 + r2) * 64];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          GxFragH b[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) b[p].u = sB[buf][p * kGxBN * 4 + b_at + ct * 64];
#define GX_MM16(X, Y, C) (kSwap ? __builtin_amdgcn_mfma_f32_16x16x32_f16(Y, X, C, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_f16(X, Y, C, 0, 0, 0))
#define GX_MFMAH(PA, PB)                                                                                                  \
  acc[rh * 2][ct] = GX_MM16(a[0][PA].v, b[PB].v, acc[rh * 2][ct]);                                                        \
  acc[rh * 2 + 1][ct] = GX_MM16(a[1][PA].v, b[PB].v, acc[rh * 2 + 1][ct])
          GX_MFMAH(1, 0); GX_MFMAH(0, 1); GX_MFMAH(0, 0);              // l.h, h.l, h.h
#undef GX_MFMAH
#undef GX_MM16
        }
      }
    } else {
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {                                   // two row tiles at a time: 24 + 12 fragment registers
      GxFrag a[2][3];
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[r2][p].u = sA[buf][p * kGxBM * 4 + a_at + (rh * 2 + r2) * 64];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        GxFrag b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p].u = sB[buf][p * kGxBN * 4 + b_at + ct * 64];
        // smallest products first; consecutive MFMAs alternate between two accumulators
#define GX_MMB(X, Y, C) (kSwap ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(Y, X, C, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(X, Y, C, 0, 0, 0))
#define GX_MFMA2(PA, PB)                                                                                                  \
  acc[rh * 2][ct] = GX_MMB(a[0][PA].v, b[PB].v, acc[rh * 2][ct]);                                                         \
  acc[rh * 2 + 1][ct] = GX_MMB(a[1][PA].v, b[PB].v, acc[rh * 2 + 1][ct])
        GX_MFMA2(1, 1); GX_MFMA2(2, 0); GX_MFMA2(0, 2); GX_MFMA2(1, 0); GX_MFMA2(0, 1); GX_MFMA2(0, 0);
#undef GX_MFMA2
#undef GX_MMB
      }
    }
    }
  };

#ifdef ALLSET_ABL_GX_TIMING         // diagnostic builds only (tools/gemm_wide_ablation.py): cycles per segment of waves 0 and 4 of workgroup 0
  uint64_t tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define GX_MARK(k) do { const uint64_t tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } while (0)
#else
#define GX_MARK(k) do {} while (0)
#endif
  const uint64_t lnb_seed = resolve_seed(seed_base, epi.lnb_seed);
  const float lnb_inv = epi.lnb_p > 0.f ? 1.f / (1.f - epi.lnb_p) : 1.f;
  const uint32_t lnb_thr = drop_threshold(epi.lnb_p);
  float4 lnb_dg = make_float4(0.f, 0.f, 0.f, 0.f), lnb_db = lnb_dg;     // (LNB only: dead otherwise)
  int64_t tile = blockIdx.x;
  if (tile >= total) {
    if constexpr (LNB)                                                 // an idle workgroup still owns a (zero) partial row
      for (int i = threadIdx.x; i < 2 * N; i += kGxThreads) epi.lnb_part[static_cast<int64_t>(blockIdx.x) * 2 * N + i] = 0.f;
    return;
  }
  float launch_inv = 1.f;                                              // fp16x3 behind a LayerNorm: 2^-Su, undone in the epilogue
  if (pro.stats) {
    for (int i = threadIdx.x; i < K; i += kGxThreads) { sGB[i] = pro.gamma[i] * inv_in; sGB[512 + i] = pro.beta[i] * inv_in; }     // (the dropout's 1 / keep rides along)
    __syncthreads();
    if constexpr (F16) {
      // |u| <= (sqrt(K - 1) max|gamma| + max|beta|) keep: one power of two 2^Su brings every A element below 2^14
      float g = 0.f, bm = 0.f;
      for (int i = lane; i < K; i += 64) { g = fmaxf(g, fabsf(sGB[i])); bm = fmaxf(bm, fabsf(sGB[512 + i])); }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { g = fmaxf(g, __shfl_xor(g, off)); bm = fmaxf(bm, __shfl_xor(bm, off)); }
      const float U = sqrtf(static_cast<float>(K)) * g + bm;           // (g, bm carry 1 / keep already)
      const int eU = static_cast<int>(__float_as_uint(U) >> 23);       // U < 2^(eU - 126)
      const int Su = min(max(140 - eU, -100), 100);
      const float su = __uint_as_float(static_cast<uint32_t>(127 + Su) << 23);
      launch_inv = __uint_as_float(static_cast<uint32_t>(127 - Su) << 23);
      __syncthreads();                                                 // (every wave has read the unscaled copy)
      for (int i = threadIdx.x; i < K; i += kGxThreads) { sGB[i] *= su; sGB[512 + i] *= su; }
      __syncthreads();
    }
  }
  GxRow cur = row_ctx(tile);
  if (rowsc) {                                                         // the first tile's row maxima: once per workgroup, not overlapped
    float m = 0.f;
    for (int ks = 0; ks < ksteps; ++ks)
      m = amax8(*reinterpret_cast<const float4*>(cur.a + ks * kGxKS), *reinterpret_cast<const float4*>(cur.a + ks * kGxKS + 4), m);
    cur.asc = row_scale(m);
  }
  const uint4* img_cur = planes + static_cast<int64_t>(tile % n_tiles) * ksteps * kBSlab;
  GX_LOAD(cur, img_cur, 0);
  int tpar = 0;                                                        // parity of the tile count (sRowInv)
  for (; tile < total; tile += gridDim.x, tpar ^= 1) {
    const int64_t next = tile + gridDim.x;
    const bool has_next = next < total;
    GxRow nxt = row_ctx(has_next ? next : tile);                       // (stats of the next tile are in flight early)
    if constexpr (F16) {                                               // what the epilogue multiplies a row of the tile by
      if (s_seg == 0) sRowInv[tpar * kGxBM + s_row] = rowsc ? 1.f / cur.asc : launch_inv;
      nmax = 0.f;
    }
    const uint4* img_nxt = planes + static_cast<int64_t>((has_next ? next : tile) % n_tiles) * ksteps * kBSlab;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    GX_MARK(4);
    GX_STORE(cur, 0, 0);                                               // step 0 of this tile (loaded during the previous one)
    GX_MARK(1);
    if (ksteps > 1) GX_LOAD(cur, img_cur, 1); else if (has_next) GX_LOAD(nxt, img_nxt, 0);
    GX_MARK(2);
    __syncthreads();
    GX_MARK(3);
    // MFMAs out of buffer ks & 1; the registers (step ks + 1) go to the other buffer and reload with step ks + 2 -- of the
    // next tile when this one has no such step; the last step only multiplies (the arena turns into the output tile next, the
    // registers keep the next tile's step 0).  Two steps per trip so that the buffer index is a compile-time constant: every
    // LDS address of the fragments and of the staging stores is then an immediate offset instead of integer arithmetic
    // (profiles/pmc_gemm_x6.json: a third of this kernel's VALU instructions were address arithmetic).
#define GX_STEP(KS_, BUF_)                                                                      \
  do {                                                                                          \
    const int ks_ = (KS_);                                                                      \
    const bool last_ = ks_ + 1 == ksteps;                                                       \
    if constexpr (F16) {                                                                        \
      if (rowsc && has_next) {        /* one tile ahead: the segments this thread will stage for the next tile */ \
        la0 = *reinterpret_cast<const float4*>(nxt.a + ks_ * kGxKS);                            \
        la1 = *reinterpret_cast<const float4*>(nxt.a + ks_ * kGxKS + 4);                        \
      }                                                                                         \
    }                                                                                           \
    GX_MARK(2);                                                                                 \
    if (mfma_first) compute(BUF_);                                                              \
    GX_MARK(0);                                                                                 \
    if (!last_) {                                                                               \
      GX_STORE(cur, ks_ + 1, (BUF_) ^ 1);                                                       \
      GX_MARK(1);                                                                               \
      if (ks_ + 2 < ksteps) GX_LOAD(cur, img_cur, ks_ + 2); else if (has_next) GX_LOAD(nxt, img_nxt, 0); \
      GX_MARK(2);                                                                               \
    }                                                                                           \
    if (!mfma_first) compute(BUF_);                                                             \
    GX_MARK(0);                                                                                 \
    if constexpr (F16) { if (rowsc && has_next) nmax = amax8(la0, la1, nmax); }                 \
    GX_MARK(5);                                                                                 \
    /* LDS traffic must have landed; the global loads just issued stay in flight ACROSS the barrier (__syncthreads()   \
       would drain them: s_waitcnt vmcnt(0), one exposed memory latency per step) */                                   \
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                         \
    GX_MARK(3);                                                                                 \
  } while (0)
    for (int ks = 0; ks < ksteps; ks += 2) {
      GX_STEP(ks, 0);
      if (ks + 1 < ksteps) GX_STEP(ks + 1, 1);
    }
#undef GX_STEP

    // ---- epilogue.  lane holds acc[rt][ct][r] = tile[wr*64 + rt*16 + 4*fg + r][wc*64 + ct*16 + fr]: dword stores from
    // that layout are issue-bound (~5 B/clk/CU), so the tile takes one trip through LDS (pitch 260 floats: the four row
    // groups of a wave land on disjoint banks) and leaves as whole rows of 16-byte stores -- which then drain while the
    // next tile's K loop runs; bias / relu / dropout on the row-major side.
    if constexpr (kSwap) {
      // ---- register epilogue.  acc[rt][ct] = out[row0 + wr*64 + rt*16 + fr][ntile*256 + wc*64 + ct*16 + 4*fg + (0..3)]
      const int64_t row0s = tile / n_tiles * kGxBM;
      const int ncol0 = static_cast<int>(tile % n_tiles) * kGxBN + wc * 64 + 4 * fg;      // + 16 ct
      float4 bv[4], cs[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int n = ncol0 + 16 * ct;
        bv[ct] = (epi.bias != nullptr && n < N) ? *reinterpret_cast<const float4*>(epi.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (F16) cs[ct] = *reinterpret_cast<const float4*>(bscale + n);           // (bscale has n_pad entries: always in range)
        else cs[ct] = make_float4(1.f, 1.f, 1.f, 1.f);
      }
      const float out_floor = epi.relu_out ? 0.f : -INFINITY;
      const bool has_drop = epi.p_out > 0.f, has_mask = epi.mask_out != nullptr;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int rr = wr * 64 + rt * 16 + fr;
        const int64_t row = row0s + rr;
        const float ri = F16 ? sRowInv[tpar * kGxBM + rr] : 1.f;
        uint32_t mbits[2] = {0u, 0u};                                  // this lane's bits of the row's two mask dwords (32-column halves)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int n = ncol0 + 16 * ct;
          float4 o = make_float4(acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]);
          if constexpr (F16) {
            o.x = fmaf(o.x, ri * cs[ct].x, bv[ct].x); o.y = fmaf(o.y, ri * cs[ct].y, bv[ct].y);
            o.z = fmaf(o.z, ri * cs[ct].z, bv[ct].z); o.w = fmaf(o.w, ri * cs[ct].w, bv[ct].w);
          } else {
            o.x += bv[ct].x; o.y += bv[ct].y; o.z += bv[ct].z; o.w += bv[ct].w;
          }
          o.x = fmaxf(o.x, out_floor); o.y = fmaxf(o.y, out_floor); o.z = fmaxf(o.z, out_floor); o.w = fmaxf(o.w, out_floor);
          if (has_drop) {
            const float4 k = keep_scale4(seed_out, row * N + n, thr_out, inv_out);
            o.x *= k.x; o.y *= k.y; o.z *= k.z; o.w *= k.w;
          }
          acc[rt][ct][0] = o.x; acc[rt][ct][1] = o.y; acc[rt][ct][2] = o.z; acc[rt][ct][3] = o.w;     // (stats_out reads them back)
          if (row < rows && n < N) *reinterpret_cast<float4*>(out + row * ldo + n) = o;
          if (has_mask) {
            // "mask layout" (include/allset_hip_ext.h): the dword of (row, 32-column half) has bit 8 q + (c % 8) for column 4 c + q of
            // its 64-column block; here c % 8 = 4 (ct & 1) + fg
            const int sh = 4 * (ct & 1) + fg;
            mbits[ct >> 1] |= ((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 0x100u : 0u) | (o.z > 0.f ? 0x10000u : 0u) | (o.w > 0.f ? 0x1000000u : 0u)) << sh;
          }
        }
        if (has_mask) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint32_t w = mbits[h];
            w |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(w), 16));
            w |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(w), 32));
            const int col = static_cast<int>(tile % n_tiles) * kGxBN + wc * 64 + 32 * h;
            if (fg == 0 && row < rows && col < N)
              (epi.mask_out + ((row >> 4) * (N / 64)) * 32 + ((row & 15) >> 2) * 8 + (row & 3) * 2)[(col >> 6) * 32 + ((col & 63) >> 5)] = w;
          }
        }
      }
      if (epi.stats_out != nullptr) {
        // {mean, rstd} of stats_relu ? relu(out) : out per row (N == 256: the row's four 64-column quarters sit in the four waves wc of a
        // row band): per-lane partial sums over its 16 columns, the four lanes fg of a row by two shuffles, the four waves through LDS;
        // two passes (mean, then the squared deviations) as the row pass had them
        const float sf = epi.stats_relu ? 0.f : -INFINITY;
        float mean4[4];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            float a = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float t = fmaxf(acc[rt][ct][q], sf);
                if (ph == 0) a += t; else { const float dlt = t - mean4[rt]; a = fmaf(dlt, dlt, a); }
              }
            a += __shfl_xor(a, 16);
            a += __shfl_xor(a, 32);
            if (fg == 0) sStat[(wr * 64 + rt * 16 + fr) * 4 + wc] = a;
          }
          __syncthreads();
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            const float4 p4 = *reinterpret_cast<const float4*>(&sStat[(wr * 64 + rt * 16 + fr) * 4]);
            const float tot = ((p4.x + p4.y) + (p4.z + p4.w)) * (1.f / kGxBN);
            if (ph == 0) mean4[rt] = tot;
            else {
              const int64_t row = row0s + wr * 64 + rt * 16 + fr;
              if (wc == 0 && fg == 0 && row < rows) *reinterpret_cast<float2*>(epi.stats_out + row * 2) = make_float2(mean4[rt], rsqrtf(tot + epi.stats_eps));
            }
          }
          __syncthreads();
        }
      }
      if constexpr (F16) { if (rowsc && has_next) nxt.asc = row_scale(nmax); }
      cur = nxt;
      img_cur = img_nxt;
      continue;
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sOut[(wr * 64 + rt * 16 + 4 * fg + r) * kOutPitch + wc * 64 + ct * 16 + fr] = acc[rt][ct][r];
    const int c4 = (tid & 63) * 4;
    const int n = static_cast<int>(tile % n_tiles) * kGxBN + c4;
    const int64_t row0 = tile / n_tiles * kGxBM;
    // LayerNorm-backward epilogue: the x rows and row statistics of the 16 rows this wave will finish, requested NOW (the accumulators
    // are dead, their registers hold the requests) so that their latency passes under the tile's trip through LDS instead of once
    // per row inside the loop below (round 5: that loop's dependent loads were about half of this kernel's tile time)
    constexpr int kLnbIter = kGxBM / (kGxThreads / 64);
    float4 lnb_xp[LNB ? kLnbIter : 1];
    float2 lnb_sp = make_float2(0.f, 1.f);                             // lane i < 16: the statistics of the wave's i-th row
    if constexpr (LNB) {
      static_assert(kLnbIter <= 64, "one lane per row of the wave");
#pragma unroll
      for (int i = 0; i < kLnbIter; ++i) {
        const int64_t row = row0 + wave_u + i * (kGxThreads / 64);
        const int64_t rc = row < rows ? row : rows - 1;
        lnb_xp[i] = n < N ? *reinterpret_cast<const float4*>(epi.lnb_x + rc * epi.lnb_ldx + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      {
        const int li = (tid & 63) < kLnbIter ? (tid & 63) : kLnbIter - 1;
        const int64_t row = row0 + wave_u + li * (kGxThreads / 64);
        lnb_sp = *reinterpret_cast<const float2*>(epi.lnb_stats + (row < rows ? row : rows - 1) * 2);
      }
    }
    GX_MARK(6);
    __syncthreads();
    GX_MARK(7);
    float4 cs4 = make_float4(1.f, 1.f, 1.f, 1.f);                      // fp16x3: the inverse scales of the lane's four B columns
    if constexpr (F16) cs4 = *reinterpret_cast<const float4*>(bscale + n);       // (bscale has n_pad entries: always in range)
    if constexpr (LNB) {
      // One wavefront per row (64 lanes x 4 columns = the whole row, N <= 256): the two row sums of the LayerNorm backward are wave
      // reductions; dgamma / dbeta accumulate in registers over all rows this lane ever sees.  The wave's 16 rows go in batches of
      // kLnbBatch as straight-line code -- per-lane partial sums of the batch's rows, their reductions side by side (DPP row sums +
      // four v_readlane each), then the outputs; as a loop with one row per trip (a `continue` past the end, the sums by six
      // dependent ds_bpermute steps) every row waited out its own LDS read and two shuffle chains.
      const bool act = n < N;
      const float inv_d = 1.f / static_cast<float>(N);
      float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (act) g4 = *reinterpret_cast<const float4*>(epi.lnb_gamma + n);
      const float relu_floor = epi.lnb_relu_in ? 0.f : -INFINITY;
      constexpr int kLnbBatch = 4;                                     // rows per batch: 8 reductions side by side, 16 registers of gh
      static_assert(kLnbIter % kLnbBatch == 0, "whole batches");
#pragma unroll
      for (int b0 = 0; b0 < kLnbIter; b0 += kLnbBatch) {
        float4 gh[kLnbBatch];                                          // gv * gamma of the batch's rows
        float a1[kLnbBatch], a2[kLnbBatch], rs[kLnbBatch];             // partial row sums (then the row means); rstd
        uint32_t pos = 0u;                                             // bit 4 j + q: raw x > 0 (the relu_in mask of the output)
#pragma unroll
        for (int j = 0; j < kLnbBatch; ++j) {
          const int it = b0 + j;
          const int rr = wave_u + it * (kGxThreads / 64);
          const int64_t row = row0 + rr;
          float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (act) gv = *reinterpret_cast<const float4*>(sOut + rr * kOutPitch + c4);
          float kz = row < rows ? 1.f : 0.f;                           // (wave-uniform) a row past the end contributes nothing
          if constexpr (F16) kz *= sRowInv[tpar * kGxBM + rr];
          gv.x *= kz * cs4.x; gv.y *= kz * cs4.y; gv.z *= kz * cs4.z; gv.w *= kz * cs4.w;
          const float mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lnb_sp.x), it));
          const float rstd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lnb_sp.y), it));
          rs[j] = rstd;
          const float4 xv = lnb_xp[it];
          pos |= ((xv.x > 0.f ? 1u : 0u) | (xv.y > 0.f ? 2u : 0u) | (xv.z > 0.f ? 4u : 0u) | (xv.w > 0.f ? 8u : 0u)) << (4 * j);
          float4 xh = make_float4((fmaxf(xv.x, relu_floor) - mu) * rstd, (fmaxf(xv.y, relu_floor) - mu) * rstd,
                                  (fmaxf(xv.z, relu_floor) - mu) * rstd, (fmaxf(xv.w, relu_floor) - mu) * rstd);
          if (!act) xh = make_float4(0.f, 0.f, 0.f, 0.f);
          if (epi.lnb_p > 0.f) {
            const float4 k = keep_scale4(lnb_seed, row * N + n, lnb_thr, lnb_inv);
            gv.x *= k.x; gv.y *= k.y; gv.z *= k.z; gv.w *= k.w;
          }
          lnb_dg.x = fmaf(gv.x, xh.x, lnb_dg.x); lnb_dg.y = fmaf(gv.y, xh.y, lnb_dg.y);
          lnb_dg.z = fmaf(gv.z, xh.z, lnb_dg.z); lnb_dg.w = fmaf(gv.w, xh.w, lnb_dg.w);
          lnb_db.x += gv.x; lnb_db.y += gv.y; lnb_db.z += gv.z; lnb_db.w += gv.w;
          gh[j] = make_float4(gv.x * g4.x, gv.y * g4.y, gv.z * g4.z, gv.w * g4.w);
          a1[j] = (gh[j].x + gh[j].y) + (gh[j].z + gh[j].w);
          a2[j] = fmaf(gh[j].x, xh.x, fmaf(gh[j].y, xh.y, fmaf(gh[j].z, xh.z, gh[j].w * xh.w)));
          lnb_xp[it] = xh;                                             // (the raw row is done with: its registers carry xhat on)
        }
#pragma unroll
        for (int j = 0; j < kLnbBatch; ++j) { a1[j] = gx_wave_sum(a1[j]) * inv_d; a2[j] = gx_wave_sum(a2[j]) * inv_d; }
#pragma unroll
        for (int j = 0; j < kLnbBatch; ++j) {
          const int it = b0 + j;
          const int rr = wave_u + it * (kGxThreads / 64);
          const int64_t row = row0 + rr;
          if (row < rows && act) {
            const float4 xh = lnb_xp[it];
            const float rstd = rs[j], s1 = a1[j], s2 = a2[j];
            float4 o = make_float4(rstd * (gh[j].x - s1 - xh.x * s2), rstd * (gh[j].y - s1 - xh.y * s2),
                                   rstd * (gh[j].z - s1 - xh.z * s2), rstd * (gh[j].w - s1 - xh.w * s2));
            if (epi.lnb_relu_in) {
              const uint32_t b = pos >> (4 * j);
              o.x = (b & 1u) ? o.x : 0.f; o.y = (b & 2u) ? o.y : 0.f; o.z = (b & 4u) ? o.z : 0.f; o.w = (b & 8u) ? o.w : 0.f;
            }
            *reinterpret_cast<float4*>(out + row * ldo + n) = o;
          }
        }
      }
    }
    if (!LNB && n < N) {                                               // N % 4 == 0: a packet is inside or outside
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (epi.bias) bv = *reinterpret_cast<const float4*>(epi.bias + n);
      const float out_floor = epi.relu_out ? 0.f : -INFINITY;          // fmaxf(v, floor): the relu without a branch
      // The wave's 16 rows as straight-line code (rows past the end are computed and not stored), the dropout and the mask output
      // as compile-time variants of the pass: as a loop with a `break` and two run-time switches inside, each row waited for its own
      // LDS read and the pass took a quarter of the forward's cycles (tools/gemm_wide_ablation.py).
      auto row_pass = [&](auto drop_tag, auto mask_tag, auto stats_tag) {
        constexpr bool kDrop = decltype(drop_tag)::value, kMask = decltype(mask_tag)::value, kStats = decltype(stats_tag)::value;
        constexpr int kRowsPerWave = kGxBM / (kGxThreads / 64);
        float4 v[kRowsPerWave];
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) v[i] = *reinterpret_cast<const float4*>(sOut + (wave_u + i * (kGxThreads / 64)) * kOutPitch + c4);
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
          const int rr = wave_u + i * (kGxThreads / 64);
          const int64_t row = row0 + rr;
          float4 o = v[i];
          if constexpr (F16) {
            const float ri = sRowInv[tpar * kGxBM + rr];
            o.x = fmaf(o.x, ri * cs4.x, bv.x); o.y = fmaf(o.y, ri * cs4.y, bv.y); o.z = fmaf(o.z, ri * cs4.z, bv.z); o.w = fmaf(o.w, ri * cs4.w, bv.w);
          } else {
            o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
          }
          o.x = fmaxf(o.x, out_floor); o.y = fmaxf(o.y, out_floor); o.z = fmaxf(o.z, out_floor); o.w = fmaxf(o.w, out_floor);
          if constexpr (kDrop) {
            const float4 k = keep_scale4(seed_out, row * N + n, thr_out, inv_out);
            o.x *= k.x; o.y *= k.y; o.z *= k.z; o.w *= k.w;
          }
          if constexpr (kStats) {                                      // (N == 256: all 64 lanes are here, the wave holds the row)
            const float sf = epi.stats_relu ? 0.f : -INFINITY;
            const float4 t = make_float4(fmaxf(o.x, sf), fmaxf(o.y, sf), fmaxf(o.z, sf), fmaxf(o.w, sf));
            const float mean = gx_wave_sum((t.x + t.y) + (t.z + t.w)) * (1.f / kGxBN);
            const float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
            const float var = gx_wave_sum(fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3)))) * (1.f / kGxBN);
            if (row < rows && (tid & 63) == 0) *reinterpret_cast<float2*>(epi.stats_out + row * 2) = make_float2(mean, rsqrtf(var + epi.stats_eps));
          }
          if (row < rows) {                                            // (wave-uniform)
            *reinterpret_cast<float4*>(out + row * ldo + n) = o;
            if constexpr (kMask) {
              // "mask layout" (include/allset_hip_ext.h): byte q of dword j = bits 8 j .. 8 j + 7 of the ballot of component q: lane j < 8
              // assembles the dword of columns 32 j .. 32 j + 31 of this tile's row (lanes past N are inactive: their bits are 0)
              const uint64_t b0 = __ballot(o.x > 0.f), b1 = __ballot(o.y > 0.f), b2 = __ballot(o.z > 0.f), b3 = __ballot(o.w > 0.f);
              const int lj = tid & 63, sh = 8 * (lj & 7);
              const uint32_t word = static_cast<uint32_t>((b0 >> sh) & 0xffu) | (static_cast<uint32_t>((b1 >> sh) & 0xffu) << 8) |
                                    (static_cast<uint32_t>((b2 >> sh) & 0xffu) << 16) | (static_cast<uint32_t>((b3 >> sh) & 0xffu) << 24);
              const int col = static_cast<int>(tile % n_tiles) * kGxBN + 32 * lj;
              if (lj < 8 && col < N)
                (epi.mask_out + ((row >> 4) * (N / 64)) * 32 + ((row & 15) >> 2) * 8 + (row & 3) * 2)[(col >> 6) * 32 + ((col & 63) >> 5)] = word;
            }
          }
        }
      };
      using gx_true = std::integral_constant<bool, true>;
      using gx_false = std::integral_constant<bool, false>;
      if (epi.stats_out != nullptr) {                                  // (a Linear that feeds the next one's LayerNorm has no relu / dropout epilogue of its own)
        if (epi.p_out > 0.f) { if (epi.mask_out != nullptr) row_pass(gx_true{}, gx_true{}, gx_true{}); else row_pass(gx_true{}, gx_false{}, gx_true{}); }
        else { if (epi.mask_out != nullptr) row_pass(gx_false{}, gx_true{}, gx_true{}); else row_pass(gx_false{}, gx_false{}, gx_true{}); }
      } else if (epi.p_out > 0.f) { if (epi.mask_out != nullptr) row_pass(gx_true{}, gx_true{}, gx_false{}); else row_pass(gx_true{}, gx_false{}, gx_false{}); }
      else { if (epi.mask_out != nullptr) row_pass(gx_false{}, gx_true{}, gx_false{}); else row_pass(gx_false{}, gx_false{}, gx_false{}); }
    }
    GX_MARK(4);
    __syncthreads();                                                   // the arena is free again
    if constexpr (F16) { if (rowsc && has_next) nxt.asc = row_scale(nmax); }
    cur = nxt;
    img_cur = img_nxt;
    GX_MARK(7);
  }
#ifdef ALLSET_ABL_GX_TIMING
  // wave 0 / wave 4 of workgroup 0: [0] MFMA phase, [1] prologue + split + LDS stores, [2] load issue, [3] K-loop barriers, [4] the epilogue's
  // row pass (LDS reads, arithmetic, global stores), [5] row-maximum lookahead, [6] accumulators -> LDS + the epilogue's requests, [7] the
  // epilogue's two barriers -- written over the first floats of the output (results wrong)
  if (blockIdx.x == 0 && (tid == 0 || tid == 256)) for (int q = 0; q < 8; ++q) out[(tid ? 8 : 0) + q] = static_cast<float>(tph[q]);
#endif
#undef GX_LOAD
#undef GX_STORE
  if constexpr (LNB) {
    // lane l of every wave holds the sums of columns 4l..4l+3 over that wave's rows: add the 8 waves up through LDS
    float* red = sOut;                                                 // [8 waves][2][256]
    const int c4 = (tid & 63) * 4;
    *reinterpret_cast<float4*>(red + (wave * 2 + 0) * kGxBN + c4) = lnb_dg;
    *reinterpret_cast<float4*>(red + (wave * 2 + 1) * kGxBN + c4) = lnb_db;
    __syncthreads();
    for (int i = tid; i < 2 * N; i += kGxThreads) {
      const int which = i / N, c = i - which * N;
      float acc_ = 0.f;
#pragma unroll
      for (int w = 0; w < kGxThreads / 64; ++w) acc_ += red[(w * 2 + which) * kGxBN + c];
      epi.lnb_part[static_cast<int64_t>(blockIdx.x) * 2 * N + i] = acc_;
    }
  }
}

// ---- the forward on fp16 planes with the CU's waves SPLIT BY ROLE (round 6) ------------------------------------------------------------
// gemm_x6_kernel<.., F16> has every wave do everything: per K step a wave multiplies (48 MFMAs, ~800 cycles), then runs the prologue
// on its share of the next A slab (LayerNorm-apply, dropout hash, fp16 split: ~1300 cycles of dependent vector work and LDS stores),
// then issues the loads of the step after -- 2750 cycles of serial work per wave and step, 3300-5500 measured with the barrier's skew
// (profiles/r05_gemm_wide_segments.txt), a tile of 128 rows in ~21 us where its MFMAs need 5 and its bytes 6.  Same cure as at width
// 128 (fused_fwd2.hip): 1024 threads, waves 0-7 do ALL the vector work (the staging thread map of the kernel above, unchanged: thread
// (row = t >> 2, segment = t & 3) of 512), waves 8-15 ALL the matrix work (the 64 x 64 sub-tile map of the kernel above: wave (wr, wc));
// every SIMD holds two vector waves, whose dependent chains hide each other's latency, and two matrix waves that issue nothing but
// fragment reads and MFMAs.  One barrier per K step ("tick"): at tick g the vector waves stage step g into buffer g & 1 -- out of
// registers they requested FOUR steps earlier (A: four register sets, 64 KB in flight per CU; the weight pieces, L2-resident: two)
// -- while the matrix waves multiply step g - 1 out of the other buffer.  The stream of steps runs across tiles: the vector
// waves stage the next tile's first steps while the matrix waves finish a tile and run its epilogue IN THEIR ACCUMULATOR REGISTERS
// (operands swapped: a lane holds four consecutive output columns of a row -- bias / relu / dropout / 1-bit mask / the next
// LayerNorm's row statistics -- after each wave has turned its patch row-major through a private 4-KB piece of LDS, no barrier: see
// `epilogue`).  LDS: two 48-KB stages + gamma / beta + the patches = 138 KB.
// Results are bit-identical to gemm_x6_kernel<.., F16>'s: the same planes, the same products in the same order.
constexpr int kGrThreads = 1024, kGrVThreads = 512, kGrPatchPitch = 68;

// ROWSC: no LayerNorm-apply prologue (pro.stats == NULL): A is scaled per row.  A template parameter, not a run-time switch: every
// global load of the vector waves' loop is then UNCONDITIONAL -- behind a (uniform) branch hipcc's wait-count bookkeeping gives up at
// the join and waits for vmcnt(0), i.e. for the requests issued four steps ahead: the prefetch is gone and every tick costs a memory
// latency (measured: 4 register sets instead of 2 changed nothing until the `if (r_ti < T)` around the requests went).
template <int YM, bool ROWSC, bool SGN = false>
__global__ __launch_bounds__(kGrThreads) void gemm_f16_roles_kernel(
    const float* __restrict__ A, int64_t lda, GxPro pro, const uint4* __restrict__ planes, GxEpi epi,
    float* __restrict__ out, int64_t ldo, int64_t rows, int N, int K, const uint64_t* __restrict__ seed_base,
    const float* __restrict__ bscale) {
  constexpr int NP = 2, NPC = 4;
  constexpr int kASlab = NP * kGxBM * 4, kBSlab = NP * kGxBN * 4;
  __shared__ __attribute__((aligned(16))) uint4 smem[2 * kASlab + 2 * kBSlab];
  __shared__ float sRowInv[2 * kGxBM];                // by tile parity: what undoes the row's (or the launch's) A scale in the epilogue
  __shared__ __attribute__((aligned(16))) float sStat[2 * kGxBM * 4];
  __shared__ __attribute__((aligned(16))) float sGB[2 * 512];
  // the matrix waves' private 16 x 64 patches (pitch 68 floats: the eight lanes of a 16-byte-store group land on distinct banks)
  __shared__ __attribute__((aligned(16))) float sPatch[8 * 16 * kGrPatchPitch];
  uint4 (*sA)[kASlab] = reinterpret_cast<uint4 (*)[kASlab]>(smem);
  uint4 (*sB)[kBSlab] = reinterpret_cast<uint4 (*)[kBSlab]>(smem + 2 * kASlab);
  const int n_tiles = (N + kGxBN - 1) / kGxBN;
  const int64_t total = (rows + kGxBM - 1) / kGxBM * n_tiles;
  const int ksteps = K / kGxKS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (static_cast<int64_t>(blockIdx.x) >= total) return;
  const int64_t T = (total - blockIdx.x + gridDim.x - 1) / gridDim.x;          // this workgroup's tiles: blockIdx.x + i * gridDim.x
  const int64_t total_steps = T * ksteps;
  const uint64_t seed_in = resolve_seed(seed_base, pro.seed_in), seed_out = resolve_seed(seed_base, epi.seed_out);
  const float inv_mask = pro.p_mask > 0.f ? 1.f / (1.f - pro.p_mask) : 1.f;
  const float inv_in = pro.p_in > 0.f ? 1.f / (1.f - pro.p_in) : 1.f;
  const uint32_t thr_in = drop_threshold(pro.p_in);
  const float inv_out = epi.p_out > 0.f ? 1.f / (1.f - epi.p_out) : 1.f;
  const uint32_t thr_out = drop_threshold(epi.p_out);
  constexpr bool rowsc = ROWSC;                                        // A scaled per row (no LayerNorm bound)
  float launch_inv = 1.f;                                              // behind a LayerNorm: 2^-Su, undone in the epilogue
  if constexpr (!ROWSC) {
    for (int i = tid; i < K; i += kGrThreads) { sGB[i] = pro.gamma[i] * inv_in; sGB[512 + i] = pro.beta[i] * inv_in; }     // (the dropout's 1 / keep rides along)
    __syncthreads();
    float g = 0.f, bm = 0.f;
    for (int i = lane; i < K; i += 64) { g = fmaxf(g, fabsf(sGB[i])); bm = fmaxf(bm, fabsf(sGB[512 + i])); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { g = fmaxf(g, __shfl_xor(g, off)); bm = fmaxf(bm, __shfl_xor(bm, off)); }
    const float U = sqrtf(static_cast<float>(K)) * g + bm;             // |u| <= (sqrt(K - 1) max|gamma| + max|beta|) keep
    const int eU = static_cast<int>(__float_as_uint(U) >> 23);
    const int Su = min(max(140 - eU, -100), 100);
    const float su = __uint_as_float(static_cast<uint32_t>(127 + Su) << 23);
    launch_inv = __uint_as_float(static_cast<uint32_t>(127 - Su) << 23);
    __syncthreads();                                                   // (every wave has read the unscaled copy)
    for (int i = tid; i < K; i += kGrThreads) { sGB[i] *= su; sGB[512 + i] *= su; }
    __syncthreads();
  }
#ifdef ALLSET_ABL_GR_NOBAR          // ablation builds only (tools/gemm_roles_ablation.py): timing without the barriers, results wrong
#define GR_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define GR_TICK() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
  const bool has_stats_out = epi.stats_out != nullptr;
#ifdef ALLSET_ABL_GR_TIMING         // diagnostic builds only (tools/gemm_roles_ablation.py): cycles per segment of waves 0 (vector) and 8 (matrix) of workgroup 0
  uint64_t tph[4] = {0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define GR_MARK(k) do { const uint64_t tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } while (0)
#else
#define GR_MARK(k) do {} while (0)
#endif

  if (wave_u < 8) {
    // =================================================== vector waves ===================================================
    const int s_row = tid >> 2, s_seg = tid & 3;
    const int a_st = s_row * 4 + (s_seg ^ gx_swz(s_row));
    const int b_img = (tid >> 6) * (64 * NPC) + (tid & 63) + 32 * NPC;
    const bool mask_fold = YM == 2 && rowsc;
    auto tile_of = [&](int64_t ti) __attribute__((always_inline)) -> int64_t { return blockIdx.x + ti * static_cast<int64_t>(gridDim.x); };
    auto crow_of = [&](int64_t ti) __attribute__((always_inline)) -> int64_t {                        // this thread's row of tile ti, clamped (loads are unconditional)
      const int64_t g_row = tile_of(ti) / n_tiles * kGxBM + s_row;
      return g_row < rows ? g_row : rows - 1;
    };
    auto row_scale = [&](float amax) __attribute__((always_inline)) -> float {
      amax = fmaxf(amax, __shfl_xor(amax, 1));
      amax = fmaxf(amax, __shfl_xor(amax, 2));
      const int e = min(max(static_cast<int>(__float_as_uint(amax * inv_mask * inv_in) >> 23), 20), 254);
      return __uint_as_float(static_cast<uint32_t>(267 - e) << 23);    // 2^(140 - e): the largest element -> [2^13, 2^14)
    };
    auto amax8 = [](float4 a, float4 b, float m) __attribute__((always_inline)) -> float {
      return fmaxf(fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                         fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))), m);
    };
    // One step of the stream in flight per register set (named registers, two sets "0" / "1": as a struct passed by reference hipcc
    // kept part of a set in an alloca promoted to LDS): the A segment, the mask source, the B pieces.  What depends on the TILE only --
    // the row's address, statistics and mask words, the weight image -- is recomputed when a stream position wraps to a new tile.
    // A segments (HBM) are requested FOUR steps ahead, the weight pieces (L2) two: bytes in flight bound this kernel -- with two sets of
    // each, 32 KB of A per CU = 8 MB chip-wide, it read at 3.1 TB/s whatever else it did (tools/gemm_roles_ablation.py: 325 us for the
    // K loop alone) and the reads drained while the matrix waves ran a tile's epilogue.
    float4 a0_0, a1_0, a0_1, a1_1, a0_2, a1_2, a0_3, a1_3;
    float4 y0_0 = make_float4(1.f, 1.f, 1.f, 1.f), y1_0 = y0_0, y0_1 = y0_0, y1_1 = y0_0, y0_2 = y0_0, y1_2 = y0_0, y0_3 = y0_0, y1_3 = y0_0;
    uint4 b0_0, b1_0, b2_0, b3_0, b0_1, b1_1, b2_1, b3_1;
    uint32_t pm_0 = 0, pm_1 = 0, pm_2 = 0, pm_3 = 0;
    float2 ms_0 = make_float2(0.f, 1.f), ms_1 = ms_0, ms_2 = ms_0, ms_3 = ms_0;      // the row's LayerNorm statistics, per A set
    // the REQUEST sides' tiles: A four steps ahead of the step being staged, the weight image two
    int64_t r_ti = 0; int r_ks = 0;
    const float* r_a = nullptr; const float* r_y = nullptr; const uint32_t* r_m = nullptr; const float* r_s = nullptr;
    auto r_enter = [&](int64_t ti) __attribute__((always_inline)) {
      const int64_t tile = tile_of(ti), rt_ = tile / n_tiles;
      const int64_t g_row = rt_ * kGxBM + s_row, cr = g_row < rows ? g_row : rows - 1;
      r_a = A + cr * lda + s_seg * 8;
      if constexpr (YM == 1) r_y = pro.y + cr * pro.ldy + s_seg * 8;
      if constexpr (YM == 2) r_m = pro.mask + (cr >> 4) * (K / 64) * 32 + ((cr & 15) >> 2) * 8 + (cr & 3) * 2;
      if constexpr (!ROWSC) r_s = pro.stats + cr * 2;
    };
    int64_t q_ti = 0; int q_ks = 0;
    const uint4* q_img = nullptr;
    auto q_enter = [&](int64_t ti) __attribute__((always_inline)) {
      const int64_t tile = tile_of(ti), rt_ = tile / n_tiles, nt_ = tile - rt_ * n_tiles;
      q_img = planes + nt_ * ksteps * kBSlab + b_img;
    };
#define GR_REQUEST_A(X)                                                                           \
  do {                                                                                            \
    a0_##X = *reinterpret_cast<const float4*>(r_a + r_ks * kGxKS);                                \
    a1_##X = *reinterpret_cast<const float4*>(r_a + r_ks * kGxKS + 4);                            \
    if constexpr (YM == 1) {                                                                      \
      y0_##X = *reinterpret_cast<const float4*>(r_y + r_ks * kGxKS);                              \
      y1_##X = *reinterpret_cast<const float4*>(r_y + r_ks * kGxKS + 4);                          \
    }                                                                                             \
    if constexpr (YM == 2) pm_##X = r_m[(r_ks >> 1) * 32 + (r_ks & 1)];                           \
    if constexpr (!ROWSC) ms_##X = *reinterpret_cast<const float2*>(r_s);                         \
    if (++r_ks == ksteps) { r_ks = 0; if (r_ti + 1 < T) { ++r_ti; r_enter(r_ti); } }  /* (past the end: the last tile again, never staged) */ \
  } while (0)
#define GR_REQUEST_B(X)                                                                           \
  do {                                                                                            \
    const uint4* img_ = q_img + static_cast<int64_t>(q_ks) * kBSlab;                              \
    b0_##X = img_[-128]; b1_##X = img_[-64]; b2_##X = img_[0]; b3_##X = img_[64];                 \
    if (++q_ks == ksteps) { q_ks = 0; if (q_ti + 1 < T) { ++q_ti; q_enter(q_ti); } }              \
  } while (0)
    // the STAGING side's tile
    int64_t s_ti = 0; int s_ks = 0;
    int64_t s_grow = 0;                                                // this thread's row of the tile being staged (unclamped: the dropout index)
    const float* s_next = nullptr;                                     // rowsc: this thread's segment base in the NEXT tile's row (the lookahead), or null
    float asc = 1.f, nmax = 0.f;                                       // per-row window (rowsc): this tile's scale, the next tile's maximum so far
    auto s_enter = [&](int64_t ti) __attribute__((always_inline)) {
      const int64_t tile = tile_of(ti), rt_ = tile / n_tiles;
      s_grow = rt_ * kGxBM + s_row;
      if constexpr (ROWSC) s_next = A + crow_of(ti + 1 < T ? ti + 1 : ti) * lda + s_seg * 8;     // (the last tile looks at itself: unused)
      if (s_seg == 0) sRowInv[(ti & 1) * kGxBM + s_row] = rowsc ? 1.f / asc : launch_inv;     // the tile's row scales for the epilogue, by tile parity
      nmax = 0.f;
    };
    auto stage = [&](float4 a0, float4 a1, float4 y0, float4 y1, uint32_t pm, float2 ms, uint4 q0, uint4 q1, uint4 q2, uint4 q3, int ks, int buf) __attribute__((always_inline)) {
      const int kb = ks * kGxKS + s_seg * 8;
#ifdef ALLSET_ABL_GR_NOSTAGE        // ablation builds only: the loads and the LDS stores stay, the prologue's arithmetic goes
      sA[buf][a_st] = make_uint4(__float_as_uint(a0.x), __float_as_uint(a0.y), __float_as_uint(a0.z), __float_as_uint(a0.w));
      sA[buf][kGxBM * 4 + a_st] = make_uint4(__float_as_uint(a1.x), __float_as_uint(a1.y), __float_as_uint(a1.z), __float_as_uint(a1.w) + kb);
      sB[buf][tid] = q0; sB[buf][tid + kGrVThreads] = q1; sB[buf][tid + 2 * kGrVThreads] = q2; sB[buf][tid + 3 * kGrVThreads] = q3;
      return;
#endif
      float e[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      if constexpr (YM == 1) {
        const float y[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = y[i] > 0.f ? e[i] * inv_mask : 0.f;
      }
      if constexpr (YM == 2) {
        const int w = static_cast<int>(pm);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[i] = __int_as_float(__float_as_int(e[i]) & __builtin_amdgcn_sbfe(w, 2 * s_seg + 8 * i, 1));
          e[4 + i] = __int_as_float(__float_as_int(e[4 + i]) & __builtin_amdgcn_sbfe(w, 2 * s_seg + 8 * i + 1, 1));
        }
        if (!mask_fold) {
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] *= inv_mask;
        }
      }
      if (pro.relu_in) {
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = fmaxf(e[i], 0.f);
      }
      if constexpr (!ROWSC) {
        const float4 g0 = *reinterpret_cast<const float4*>(sGB + kb), g1 = *reinterpret_cast<const float4*>(sGB + kb + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(sGB + 512 + kb), c1 = *reinterpret_cast<const float4*>(sGB + 512 + kb + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = fmaf((e[i] - ms.x) * ms.y, g[i], c[i]);
        if (pro.p_in > 0.f) {
          const float4 k0 = keep_scale4(seed_in, s_grow * K + kb, thr_in, 1.f), k1 = keep_scale4(seed_in, s_grow * K + kb + 4, thr_in, 1.f);
          e[0] *= k0.x; e[1] *= k0.y; e[2] *= k0.z; e[3] *= k0.w; e[4] *= k1.x; e[5] *= k1.y; e[6] *= k1.z; e[7] *= k1.w;
        }
      }
      const float sc = mask_fold ? asc * inv_mask : asc;
      uint4 h, l;
      split2_f16c(e[0] * sc, e[1] * sc, h.x, l.x);
      split2_f16c(e[2] * sc, e[3] * sc, h.y, l.y);
      split2_f16c(e[4] * sc, e[5] * sc, h.z, l.z);
      split2_f16c(e[6] * sc, e[7] * sc, h.w, l.w);
      sA[buf][a_st] = h;
      sA[buf][kGxBM * 4 + a_st] = l;
      sB[buf][tid] = q0; sB[buf][tid + kGrVThreads] = q1; sB[buf][tid + 2 * kGrVThreads] = q2; sB[buf][tid + 3 * kGrVThreads] = q3;
    };
    if constexpr (ROWSC) {                                             // the first tile's row maxima: once per workgroup, not overlapped
      const float* ap = A + crow_of(0) * lda + s_seg * 8;
      float m = 0.f;
      for (int ks = 0; ks < ksteps; ++ks) m = amax8(*reinterpret_cast<const float4*>(ap + ks * kGxKS), *reinterpret_cast<const float4*>(ap + ks * kGxKS + 4), m);
      asc = row_scale(m);
    }
    r_enter(0);
    q_enter(0);
    GR_REQUEST_A(0); GR_REQUEST_B(0);
    GR_REQUEST_A(1); GR_REQUEST_B(1);
    GR_REQUEST_A(2);
    GR_REQUEST_A(3);
    s_enter(0);
#define GR_ONE(X, XB, BUF)                                                                        \
  do {                                                                                            \
    float4 la0 = make_float4(0.f, 0.f, 0.f, 0.f), la1 = la0;                                      \
    if constexpr (ROWSC) {          /* one tile ahead: the segment this thread will stage for the next tile */ \
      la0 = *reinterpret_cast<const float4*>(s_next + s_ks * kGxKS);                              \
      la1 = *reinterpret_cast<const float4*>(s_next + s_ks * kGxKS + 4);                          \
    }                                                                                             \
    GR_MARK(3);                                                                                   \
    stage(a0_##X, a1_##X, y0_##X, y1_##X, pm_##X, ms_##X, b0_##XB, b1_##XB, b2_##XB, b3_##XB, s_ks, BUF); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    GR_MARK(0);                                                                                   \
    GR_REQUEST_A(X);               /* into the sets just consumed: A four steps ahead, */         \
    GR_REQUEST_B(XB);              /* the weight pieces two (unconditional: see ROWSC above) */   \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if constexpr (ROWSC) nmax = amax8(la0, la1, nmax);                                            \
    const bool first_of_tile_ = s_ks == 0 && s_ti > 0;                                            \
    if (++s_ks == ksteps) {         /* the tile is staged: the next one's row scale, statistics, lookahead base */ \
      s_ks = 0; ++s_ti;                                                                           \
      if constexpr (ROWSC) asc = row_scale(nmax);                                                 \
      if (s_ti < T) s_enter(s_ti);                                                                \
    }                                                                                             \
    GR_MARK(1);                                                                                   \
    GR_TICK();                                                                                    \
    GR_MARK(2);                                                                                   \
    if (first_of_tile_ && has_stats_out) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); } /* the matrix waves' epilogue of the previous tile */ \
  } while (0)
    for (int64_t g = 0; g < total_steps; g += 4) {                     // (K % 128 == 0: a multiple of four steps per tile)
      GR_ONE(0, 0, 0);
      GR_ONE(1, 1, 1);
      GR_ONE(2, 0, 0);
      GR_ONE(3, 1, 1);
    }
#undef GR_ONE
#undef GR_REQUEST_A
#undef GR_REQUEST_B
    GR_TICK();                                                         // the matrix waves' tick after the last step
    if (has_stats_out) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }                       // (the last tile's epilogue)
  } else {
    // =================================================== matrix waves ===================================================
    const int m = wave_u - 8;
    const int wr = m >> 2, wc = m & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = gx_swz(fr);
    const int a_at = (wr * 64 + fr) * 4 + (fg ^ sw), b_at = (wc * 64 + fr) * 4 + (fg ^ sw);
    f32x4_t acc[4][4];
    auto compute = [&](int buf) __attribute__((always_inline)) {
      // 128 registers per wave (16 waves per CU): 64 accumulators + the A fragments of two row tiles (16) + the B fragments of the
      // current and the NEXT column tile (16) -- requested one column tile ahead, explicitly: left to itself the scheduler hoists all of a
      // step's 24 fragment reads above its MFMAs and spills (132 dwords of scratch per lane around the MFMA loop)
      const uint4* pa = &sA[buf][a_at];
      const uint4* pb = &sB[buf][b_at];
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        GxFragH a[2][2], b[2][2];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int p = 0; p < 2; ++p) a[r2][p].u = pa[p * kGxBM * 4 + (rh * 2 + r2) * 64];
        b[0][0].u = pb[0]; b[0][1].u = pb[kGxBN * 4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          if (ct + 1 < 4) { b[(ct + 1) & 1][0].u = pb[(ct + 1) * 64]; b[(ct + 1) & 1][1].u = pb[kGxBN * 4 + (ct + 1) * 64]; }
          const GxFragH (&bc)[2] = b[ct & 1];
          // operands swapped (the weight fragment is A): acc[rt][ct][q] = out[row fr][column 4 fg + q] of the 16 x 16 tile; l.h, h.l, h.h
#define GR_MFMA(PA, PB)                                                                                                   \
  acc[rh * 2][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bc[PB].v, a[0][PA].v, acc[rh * 2][ct], 0, 0, 0);               \
  acc[rh * 2 + 1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bc[PB].v, a[1][PA].v, acc[rh * 2 + 1][ct], 0, 0, 0)
#ifdef ALLSET_ABL_GR_NOMFMA         // ablation builds only: the fragment reads stay, the MFMAs go
          acc[rh * 2][ct][0] += __builtin_bit_cast(float, bc[0].u.x ^ bc[1].u.y ^ a[0][0].u.z ^ a[0][1].u.w);
          acc[rh * 2 + 1][ct][0] += __builtin_bit_cast(float, bc[0].u.y ^ bc[1].u.x ^ a[1][0].u.z ^ a[1][1].u.w);
#else
          GR_MFMA(1, 0); GR_MFMA(0, 1); GR_MFMA(0, 0);
#endif
#undef GR_MFMA
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    auto epilogue = [&](int64_t ti) __attribute__((always_inline)) {
      // The MFMAs leave a lane with four consecutive columns of ONE row per 16 x 16 tile -- 16-byte pieces whose neighbours in a row sit
      // 16 lanes away: stored like that, a wave's store instruction is 64 separate 16-byte writes, and the tile's epilogue took 12-18
      // us of a 28-us tile (tools/gemm_roles_ablation.py: 865 us per launch, 325 without the epilogue).  So each wave turns its 16 x 64
      // patch of a row tile through a PRIVATE 4.3-KB LDS patch (no barrier: a wave's DS operations execute in order) into row-major
      // lanes -- lane (r4 = lane >> 4, c = lane & 15) holds columns 4 c .. 4 c + 3 of rows r4, r4 + 4, r4 + 8, r4 + 12 -- where a store
      // instruction writes four whole 256-byte row segments, the lane's bias / column scales are ONE float4 each, the 1-bit mask is
      // four ballots and a row's sum over the wave's 64 columns a DPP row sum.
      const int64_t tile = blockIdx.x + ti * static_cast<int64_t>(gridDim.x);
      const int tpar = static_cast<int>(ti & 1);
      const int64_t row0s = tile / n_tiles * kGxBM + wr * 64;
      const int ntile0 = static_cast<int>(tile % n_tiles) * kGxBN;
      const int r4 = lane >> 4, c = lane & 15;
      const int n = ntile0 + wc * 64 + 4 * c;                          // this lane's four columns
      const float out_floor = epi.relu_out ? 0.f : -INFINITY;
      const bool has_drop = epi.p_out > 0.f, has_mask = epi.mask_out != nullptr;
      constexpr bool has_sgn = SGN;                                    // (its own instantiations: the forward's keep their registers)
      const float4 bv = (epi.bias != nullptr && n < N) ? *reinterpret_cast<const float4*>(epi.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 cs = *reinterpret_cast<const float4*>(bscale + n);    // (bscale has n_pad entries: always in range)
      float* patch = sPatch + m * (16 * kGrPatchPitch);
      float4 o[4][4];                                                  // [rt][j]: row wr*64 + rt*16 + r4 + 4 j
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          *reinterpret_cast<float4*>(&patch[fr * kGrPatchPitch + 16 * ct + 4 * fg]) = make_float4(acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[rt][j] = *reinterpret_cast<const float4*>(&patch[(r4 + 4 * j) * kGrPatchPitch + 4 * c]);
        float4 sx[4];
        if constexpr (has_sgn) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t row = row0s + rt * 16 + r4 + 4 * j;
            sx[j] = (row < rows && n < N) ? *reinterpret_cast<const float4*>(epi.sgn_x + row * epi.sgn_ldx + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rr = wr * 64 + rt * 16 + r4 + 4 * j;
          const int64_t row = row0s + rt * 16 + r4 + 4 * j;
          const float ri = sRowInv[tpar * kGxBM + rr];
          float4 v = o[rt][j];
          v = make_float4(fmaf(v.x, ri * cs.x, bv.x), fmaf(v.y, ri * cs.y, bv.y), fmaf(v.z, ri * cs.z, bv.z), fmaf(v.w, ri * cs.w, bv.w));
          v.x = fmaxf(v.x, out_floor); v.y = fmaxf(v.y, out_floor); v.z = fmaxf(v.z, out_floor); v.w = fmaxf(v.w, out_floor);
          if (has_drop) {
            const float4 k = keep_scale4(seed_out, row * N + n, thr_out, inv_out);
            v.x *= k.x; v.y *= k.y; v.z *= k.z; v.w *= k.w;
          }
          if constexpr (has_sgn) v = gx_sign_mask(v, sx[j]);
          o[rt][j] = v;
          if (row < rows && n < N) *reinterpret_cast<float4*>(out + row * ldo + n) = v;
          if (has_mask) {
            // "mask layout" (include/allset_hip_ext.h): a ballot's bit 16 r4 + c is lane (c, r4): byte (2 r4 + h8) of the ballot for
            // component q is byte q of the dword of row r4, 32-column half h8 -- lane L < 8 assembles dword (r4 = L >> 1, h8 = L & 1)
            const bool in = n < N;
            const uint64_t b0 = __ballot(in && v.x > 0.f), b1 = __ballot(in && v.y > 0.f), b2 = __ballot(in && v.z > 0.f), b3 = __ballot(in && v.w > 0.f);
            const int sh = 8 * (lane & 7);
            const uint32_t word = static_cast<uint32_t>((b0 >> sh) & 0xffu) | (static_cast<uint32_t>((b1 >> sh) & 0xffu) << 8) |
                                  (static_cast<uint32_t>((b2 >> sh) & 0xffu) << 16) | (static_cast<uint32_t>((b3 >> sh) & 0xffu) << 24);
            const int64_t mrow = row0s + rt * 16 + (lane >> 1) + 4 * j;                      // (lanes < 8: row r4 = lane >> 1 of this store)
            const int col = ntile0 + wc * 64 + 32 * (lane & 1);
            if (lane < 8 && mrow < rows && col < N)
              (epi.mask_out + ((mrow >> 4) * (N / 64)) * 32 + ((mrow & 15) >> 2) * 8 + (mrow & 3) * 2)[(col >> 6) * 32 + ((col & 63) >> 5)] = word;
          }
        }
      }
      if (has_stats_out) {
        // {mean, rstd} of stats_relu ? relu(out) : out per row (N == 256): the wave's 64 columns of a row by a DPP row sum, the four
        // waves wc of a row band through LDS; two passes (mean, then the squared deviations).  The four barriers are matched by the
        // vector waves (GR_ONE above)
        const float sf = epi.stats_relu ? 0.f : -INFINITY;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          float* sS = sStat + ph * (kGxBM * 4);                        // the sums, then the squared deviations (the means are re-read from the first: no 16 registers of them)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int rr = wr * 64 + rt * 16 + r4 + 4 * j;
              const float4 t = make_float4(fmaxf(o[rt][j].x, sf), fmaxf(o[rt][j].y, sf), fmaxf(o[rt][j].z, sf), fmaxf(o[rt][j].w, sf));
              float a;
              if (ph == 0) a = (t.x + t.y) + (t.z + t.w);
              else {
                const float4 p4 = *reinterpret_cast<const float4*>(&sStat[rr * 4]);
                const float mu = ((p4.x + p4.y) + (p4.z + p4.w)) * (1.f / kGxBN);
                const float d0 = t.x - mu, d1 = t.y - mu, d2 = t.z - mu, d3 = t.w - mu;
                a = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3)));
              }
              a += gx_dpp<0xB1>(a); a += gx_dpp<0x4E>(a); a += gx_dpp<0x141>(a); a += gx_dpp<0x140>(a);
              if (c == 0) sS[rr * 4 + wc] = a;
            }
          __syncthreads();
          if (ph == 1) {
            // rows of the band wr: 64 rows, one per lane of the wave wc == 0
            if (wc == 0) {
              const int rr = wr * 64 + lane;
              const int64_t row = row0s + lane;
              const float4 p4 = *reinterpret_cast<const float4*>(&sStat[rr * 4]), q4 = *reinterpret_cast<const float4*>(&sS[rr * 4]);
              const float mu = ((p4.x + p4.y) + (p4.z + p4.w)) * (1.f / kGxBN), var = ((q4.x + q4.y) + (q4.z + q4.w)) * (1.f / kGxBN);
              if (row < rows) *reinterpret_cast<float2*>(epi.stats_out + row * 2) = make_float2(mu, rsqrtf(var + epi.stats_eps));
            }
          }
          __syncthreads();
        }
      }
    };
    GR_TICK();                                                         // step 0 is staged
    for (int64_t ti = 0; ti < T; ++ti) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < ksteps; ks += 2) {                         // (ksteps is even: the buffer index is a compile-time constant --
        GR_MARK(3);
        compute(0);                                                    //  behind a run-time parity the accumulators became loop phis with
        GR_MARK(0);
        GR_TICK();                                                     //  two sources and lived in registers twice: 160 VGPRs)
        GR_MARK(2);
        compute(1);
        GR_MARK(0);
        GR_TICK();
        GR_MARK(2);
      }
#ifdef ALLSET_ABL_GR_NOEPI          // ablation builds only: one store per lane instead of the epilogue
      if (lane == 0) out[ti] = acc[0][0][0] + acc[1][1][1] + acc[2][2][2] + acc[3][3][3];
#else
      epilogue(ti);
#endif
      GR_MARK(1);
    }
  }
#ifdef ALLSET_ABL_GR_TIMING
  // vector wave 0: [0] staging, [1] requests + bookkeeping, [2] tick wait; matrix wave 8: [0] fragment reads + MFMAs, [1] epilogue, [2] tick wait (cycles) -> over the first floats of the output
  if (blockIdx.x == 0 && (tid == 0 || tid == 512)) for (int q = 0; q < 4; ++q) out[(tid ? 4 : 0) + q] = static_cast<float>(tph[q]);
#endif
#undef GR_MARK
#undef GR_TICK
}

// ---- row statistics for the LayerNorm-apply prologue: stats[r] = {mean, rstd} of relu_in ? relu(x[r]) : x[r] ----------------
__global__ __launch_bounds__(kBlock) void row_stats_kernel(const float* __restrict__ x, int64_t ldx, int relu_in, float eps,
                                                           float* __restrict__ stats, int64_t rows, int d) {
  const int lane = lane_id();
  const float inv_d = 1.f / static_cast<float>(d);
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6); row < rows;
       row += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const float* p = x + row * ldx;
    float4 v[2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = (lane + i * kWave) * 4;
      v[i] = c < d ? *reinterpret_cast<const float4*>(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (relu_in) { v[i].x = fmaxf(v[i].x, 0.f); v[i].y = fmaxf(v[i].y, 0.f); v[i].z = fmaxf(v[i].z, 0.f); v[i].w = fmaxf(v[i].w, 0.f); }
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = (lane + i * kWave) * 4;
      if (c < d) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, e = v[i].w - mean;
        q += a * a + b * b + cc * cc + e * e;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rsqrtf(q * inv_d + eps); }
  }
}

}  // namespace allset

using namespace allset;

extern "C" int allset_gemm_x6_supported(int64_t N, int64_t K) {
  return (N >= 1 && K >= kGxKS && K % kGxKS == 0 && K <= 4096 && N <= 4096 && N % 4 == 0) ? 1 : 0;
}

extern "C" int64_t allset_gemm_x6_plane_bytes(int64_t N, int64_t K) {
  if (!allset_gemm_x6_supported(N, K)) return -1;
  return (N + kGxBN - 1) / kGxBN * kGxBN * K * 3 * 2;
}

extern "C" int allset_gemm_x6_planes(const float* W, int64_t ldw, int transpose, void* planes, int64_t N, int64_t K, void* stream) {
  clear_error();
  if (!allset_gemm_x6_supported(N, K)) { set_error("gemm_x6_planes: N=%lld K=%lld not supported (K %% 32 == 0, N %% 4 == 0)", (long long)N, (long long)K); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(W && planes && aligned16(planes), "gemm_x6_planes: null / misaligned pointer");
  ALLSET_REQUIRE(ldw >= (transpose ? N : K), "gemm_x6_planes: leading dimension too small");
  const int64_t pairs = (N + kGxBN - 1) / kGxBN * kGxBN * (K / 2);
  const int64_t want = (pairs + kBlock - 1) / kBlock;
  gemm_x6_planes_kernel<<<static_cast<unsigned>(want > 4096 ? 4096 : want), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      W, ldw, transpose, static_cast<uint32_t*>(planes), static_cast<int>(N), static_cast<int>(K));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_row_stats(const float* x, int64_t ldx, int relu_in, float eps, float* stats, int64_t rows, int64_t d,
                                void* stream) {
  clear_error();
  ALLSET_REQUIRE(rows >= 0 && d >= 4 && d <= 512 && d % 4 == 0, "row_stats: width must be a multiple of 4, <= 512");
  if (rows == 0) return ALLSET_OK;
  ALLSET_REQUIRE(x && stats && ldx >= d && ldx % 4 == 0 && aligned16(x), "row_stats: bad pointer / leading dimension");
  const int64_t want = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
  row_stats_kernel<<<static_cast<unsigned>(want > 65536 ? 65536 : want), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      x, ldx, relu_in, eps, stats, rows, static_cast<int>(d));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static int gemm_x6_impl(const float* A, int64_t lda, const float* mask_y, int64_t ldy, float p_mask, int relu_in,
                        const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                        const void* planes, GxEpi epi, float* out, int64_t ldo, int64_t rows, int64_t N, int64_t K,
                        const uint64_t* seed_base, void* stream, bool f16 = false, const uint32_t* mask_bits = nullptr);

// ---- the fp16x3 family: same contracts, the weight in two fp16 planes + per-column inverse scales (allset_gemm_f16x3_planes) ----
extern "C" int64_t allset_gemm_f16x3_plane_bytes(int64_t N, int64_t K) {
  if (!allset_gemm_x6_supported(N, K)) return -1;
  const int64_t n_pad = (N + kGxBN - 1) / kGxBN * kGxBN;
  return n_pad * K * 2 * 2 + n_pad * 4;
}

extern "C" int allset_gemm_f16x3_planes(const float* W, int64_t ldw, int transpose, void* planes, int64_t N, int64_t K, void* stream) {
  clear_error();
  if (!allset_gemm_x6_supported(N, K)) { set_error("gemm_f16x3_planes: N=%lld K=%lld not supported (K %% 32 == 0, N %% 4 == 0)", (long long)N, (long long)K); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(W && planes && aligned16(planes), "gemm_f16x3_planes: null / misaligned pointer");
  ALLSET_REQUIRE(ldw >= (transpose ? N : K), "gemm_f16x3_planes: leading dimension too small");
  const int64_t n_pad = (N + kGxBN - 1) / kGxBN * kGxBN;
  float* bscale = reinterpret_cast<float*>(static_cast<char*>(planes) + n_pad * K * 2 * 2);
  const hipStream_t st = static_cast<hipStream_t>(stream);
  gemm_f16_planes_fused_kernel<<<static_cast<unsigned>((n_pad + kBlock / 64 - 1) / (kBlock / 64)), kBlock, 0, st>>>(
      W, ldw, transpose, static_cast<uint32_t*>(planes), bscale, static_cast<int>(N), static_cast<int>(n_pad), static_cast<int>(K));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_gemm_f16x3_planes_batch_max(void) { return kPlaneBatchMax; }

// planes[k] = allset_gemm_f16x3_planes(Ws[k], ldws[k], transposes[k], ., Ns[k], Ks[k]) for k < count <= allset_gemm_f16x3_planes_batch_max(),
// in ONE launch (host arrays, copied into the kernel's arguments); bit-identical images.
extern "C" int allset_gemm_f16x3_planes_batched(const float* const* Ws, const int64_t* ldws, const int32_t* transposes, void* const* planes,
                                                const int64_t* Ns, const int64_t* Ks, int64_t count, void* stream) {
  clear_error();
  ALLSET_REQUIRE(count >= 0 && count <= kPlaneBatchMax, "gemm_f16x3_planes_batched: at most %d weights per call", kPlaneBatchMax);
  if (count == 0) return ALLSET_OK;
  ALLSET_REQUIRE(Ws && ldws && transposes && planes && Ns && Ks, "gemm_f16x3_planes_batched: null pointer");
  PlaneBatchTable tb;
  int64_t blocks = 0;
  for (int64_t k = 0; k < count; ++k) {
    if (!allset_gemm_x6_supported(Ns[k], Ks[k])) { set_error("gemm_f16x3_planes_batched: weight %lld: N=%lld K=%lld not supported", (long long)k, (long long)Ns[k], (long long)Ks[k]); return ALLSET_ERR_UNSUPPORTED; }
    ALLSET_REQUIRE(Ws[k] && planes[k] && aligned16(planes[k]) && ldws[k] >= (transposes[k] ? Ns[k] : Ks[k]) && ldws[k] < INT32_MAX,
                   "gemm_f16x3_planes_batched: weight %lld: null / misaligned pointer or leading dimension too small", static_cast<long long>(k));
    const int64_t n_pad = (Ns[k] + kGxBN - 1) / kGxBN * kGxBN;
    tb.W[k] = Ws[k]; tb.planes[k] = static_cast<uint32_t*>(planes[k]);
    tb.ldw[k] = static_cast<int32_t>(ldws[k]); tb.N[k] = static_cast<int32_t>(Ns[k]); tb.n_pad[k] = static_cast<int32_t>(n_pad);
    tb.K[k] = static_cast<int32_t>(Ks[k]); tb.transpose[k] = transposes[k] ? 1 : 0;
    tb.first_block[k] = static_cast<int32_t>(blocks);
    blocks += (n_pad + kBlock / 64 - 1) / (kBlock / 64);
  }
  tb.first_block[count] = static_cast<int32_t>(blocks);
  tb.count = static_cast<int32_t>(count);
  gemm_f16_planes_batched_kernel<<<static_cast<unsigned>(blocks), kBlock, 0, static_cast<hipStream_t>(stream)>>>(tb);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_gemm_f16x3_lnb(const float* G, int64_t ldg, const float* mask_y, int64_t ldy, float p_mask, const void* planes,
                                     const float* x, int64_t ldx, const float* stats, const float* gamma, int relu_in, float p,
                                     uint64_t seed, float* gx, int64_t ldgx, float* partials, int64_t n_partials, int64_t rows,
                                     int64_t N, int64_t K, const uint64_t* seed_base, void* stream) {
  clear_error();
  ALLSET_REQUIRE(N >= 4 && N <= kGxBN, "gemm_f16x3_lnb: the LayerNorm row (N) must fit one 256-column tile");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "gemm_f16x3_lnb: dropout p must be in [0,1)");
  ALLSET_REQUIRE(partials != nullptr && n_partials == allset_gemm_x6_lnb_partials(rows), "gemm_f16x3_lnb: partials must hold allset_gemm_x6_lnb_partials(rows) x 2 x N floats");
  if (rows == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(partials, 0, static_cast<size_t>(n_partials) * 2 * N * sizeof(float), static_cast<hipStream_t>(stream)));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(x && stats && gamma && gx, "gemm_f16x3_lnb: null pointer");
  ALLSET_REQUIRE(ldx >= N && ldx % 4 == 0 && aligned16(x) && aligned16(gamma), "gemm_f16x3_lnb: x rows / gamma must be 16-byte aligned");
  GxEpi epi{nullptr, 0, 0.f, 0, nullptr, x, ldx, stats, gamma, relu_in, p, seed, partials};
  return gemm_x6_impl(G, ldg, mask_y, ldy, p_mask, 0, nullptr, nullptr, nullptr, 0.f, 0, planes, epi, gx, ldgx, rows, N, K, seed_base, stream, true);
}

extern "C" int allset_gemm_f16x3(const float* A, int64_t lda, const float* mask_y, int64_t ldy, float p_mask, int relu_in,
                                 const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                                 const void* planes, const float* bias, int relu_out, float p_out, uint64_t seed_out,
                                 float* out, int64_t ldo, int64_t rows, int64_t N, int64_t K, const uint64_t* seed_base,
                                 void* stream) {
  clear_error();
  ALLSET_REQUIRE(p_out >= 0.f && p_out < 1.f, "gemm_f16x3: dropout p must be in [0,1)");
  GxEpi epi{bias, relu_out, p_out, seed_out, nullptr, nullptr, 0, nullptr, nullptr, 0, 0.f, 0, nullptr};
  ALLSET_REQUIRE(bias == nullptr || aligned16(bias), "gemm_f16x3: bias must be 16-byte aligned");
  return gemm_x6_impl(A, lda, mask_y, ldy, p_mask, relu_in, stats, gamma, beta, p_in, seed_in, planes, epi, out, ldo, rows, N, K, seed_base, stream, true);
}

extern "C" int64_t allset_gemm_x6_lnb_partials(int64_t rows) {
  const int64_t tiles = (rows + kGxBM - 1) / kGxBM;
  return tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256;
}

extern "C" int allset_gemm_x6_lnb(const float* G, int64_t ldg, const float* mask_y, int64_t ldy, float p_mask, const void* planes,
                                  const float* x, int64_t ldx, const float* stats, const float* gamma, int relu_in, float p,
                                  uint64_t seed, float* gx, int64_t ldgx, float* partials, int64_t n_partials, int64_t rows,
                                  int64_t N, int64_t K, const uint64_t* seed_base, void* stream) {
  clear_error();
  ALLSET_REQUIRE(N >= 4 && N <= kGxBN, "gemm_x6_lnb: the LayerNorm row (N) must fit one 256-column tile");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "gemm_x6_lnb: dropout p must be in [0,1)");
  ALLSET_REQUIRE(partials != nullptr && n_partials == allset_gemm_x6_lnb_partials(rows), "gemm_x6_lnb: partials must hold allset_gemm_x6_lnb_partials(rows) x 2 x N floats");
  if (rows == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(partials, 0, static_cast<size_t>(n_partials) * 2 * N * sizeof(float), static_cast<hipStream_t>(stream)));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(x && stats && gamma && gx, "gemm_x6_lnb: null pointer");
  ALLSET_REQUIRE(ldx >= N && ldx % 4 == 0 && aligned16(x) && aligned16(gamma), "gemm_x6_lnb: x rows / gamma must be 16-byte aligned");
  GxEpi epi{nullptr, 0, 0.f, 0, nullptr, x, ldx, stats, gamma, relu_in, p, seed, partials};
  return gemm_x6_impl(G, ldg, mask_y, ldy, p_mask, 0, nullptr, nullptr, nullptr, 0.f, 0, planes, epi, gx, ldgx, rows, N, K, seed_base, stream);
}

extern "C" int allset_gemm_x6(const float* A, int64_t lda, const float* mask_y, int64_t ldy, float p_mask, int relu_in,
                              const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                              const void* planes, const float* bias, int relu_out, float p_out, uint64_t seed_out,
                              float* out, int64_t ldo, int64_t rows, int64_t N, int64_t K, const uint64_t* seed_base,
                              void* stream) {
  clear_error();
  ALLSET_REQUIRE(p_out >= 0.f && p_out < 1.f, "gemm_x6: dropout p must be in [0,1)");
  GxEpi epi{bias, relu_out, p_out, seed_out, nullptr, nullptr, 0, nullptr, nullptr, 0, 0.f, 0, nullptr};
  ALLSET_REQUIRE(bias == nullptr || aligned16(bias), "gemm_x6: bias must be 16-byte aligned");
  return gemm_x6_impl(A, lda, mask_y, ldy, p_mask, relu_in, stats, gamma, beta, p_in, seed_in, planes, epi, out, ldo, rows, N, K, seed_base, stream);
}

extern "C" int allset_gemm_wide_sgn_supported(int arith, int64_t N, int64_t K) {
  return (arith == ALLSET_ARITH_FP16X3 && allset_gemm_x6_supported(N, K) && K % (4 * kGxKS) == 0) ? 1 : 0;
}

static int gemm_x6_impl(const float* A, int64_t lda, const float* mask_y, int64_t ldy, float p_mask, int relu_in,
                        const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                        const void* planes, GxEpi epi, float* out, int64_t ldo, int64_t rows, int64_t N, int64_t K,
                        const uint64_t* seed_base, void* stream, bool f16, const uint32_t* mask_bits) {
  ALLSET_REQUIRE(mask_bits == nullptr || (mask_y == nullptr && K % 64 == 0), "gemm_wide: the 1-bit mask needs K % 64 == 0 and replaces mask_y");
  ALLSET_REQUIRE(epi.mask_out == nullptr || (N % 64 == 0 && epi.lnb_x == nullptr), "gemm_wide: mask_out needs N % 64 == 0 (and no LayerNorm-backward epilogue)");
  ALLSET_REQUIRE(epi.sgn_x == nullptr || (epi.lnb_x == nullptr && epi.sgn_ldx >= N && epi.sgn_ldx % 4 == 0 && aligned16(epi.sgn_x)),
                 "gemm_wide_sgn: the relu's input must have 16-byte aligned rows of at least N floats");
  if (epi.sgn_x != nullptr && !(f16 && stats == nullptr && allset_gemm_wide_sgn_supported(ALLSET_ARITH_FP16X3, N, K))) {
    set_error("gemm_wide_sgn: built on the split-role kernel only (fp16x3 planes, K %% 128 == 0, no LayerNorm prologue); N=%lld K=%lld", (long long)N, (long long)K);
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (!allset_gemm_x6_supported(N, K)) { set_error("gemm_x6: N=%lld K=%lld not supported (K %% 32 == 0, N %% 4 == 0)", (long long)N, (long long)K); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(rows >= 0, "gemm_x6: bad row count");
  ALLSET_REQUIRE(p_in >= 0.f && p_in < 1.f && p_mask >= 0.f && p_mask < 1.f, "gemm_x6: dropout p must be in [0,1)");
  if (rows == 0) return ALLSET_OK;
  ALLSET_REQUIRE(A && planes && out, "gemm_x6: null pointer");
  ALLSET_REQUIRE(lda >= K && lda % 4 == 0 && aligned16(A) && aligned16(planes), "gemm_x6: A rows / planes must be 16-byte aligned");
  ALLSET_REQUIRE(ldo >= N && ldo % 4 == 0 && aligned16(out), "gemm_x6: output rows must be 16-byte aligned");
  ALLSET_REQUIRE(mask_y == nullptr || (ldy >= K && ldy % 4 == 0 && aligned16(mask_y)), "gemm_x6: mask source rows must be 16-byte aligned");
  ALLSET_REQUIRE(stats == nullptr || (gamma && beta && K <= 512), "gemm_x6: LayerNorm-apply needs gamma and beta, K <= 512");
  GxPro pro{mask_y, ldy, mask_bits, p_mask, relu_in, stats, gamma, beta, p_in, seed_in};
  const int64_t tiles = (rows + kGxBM - 1) / kGxBM * ((N + kGxBN - 1) / kGxBN);
  const int64_t blocks = tiles < 256 ? tiles : 256;                     // one workgroup per CU (144 KB of LDS), walking its tiles
  const int64_t n_pad = (N + kGxBN - 1) / kGxBN * kGxBN;
  const float* bscale = f16 ? reinterpret_cast<const float*>(static_cast<const char*>(planes) + n_pad * K * 2 * 2) : nullptr;
  if (f16 && epi.lnb_x == nullptr && K % (4 * kGxKS) == 0) {           // the forward on fp16 planes: split-role kernel (four steps of A in flight: a multiple of four K steps)
#define GR_LAUNCH2(Y, R) gemm_f16_roles_kernel<Y, R><<<static_cast<unsigned>(blocks), kGrThreads, 0, static_cast<hipStream_t>(stream)>>>( \
      A, lda, pro, static_cast<const uint4*>(planes), epi, out, ldo, rows, static_cast<int>(N), static_cast<int>(K), seed_base, bscale)
#define GR_LAUNCH(Y) do { if (stats == nullptr) GR_LAUNCH2(Y, true); else GR_LAUNCH2(Y, false); } while (0)
#define GR_LAUNCH_SGN(Y) gemm_f16_roles_kernel<Y, true, true><<<static_cast<unsigned>(blocks), kGrThreads, 0, static_cast<hipStream_t>(stream)>>>( \
      A, lda, pro, static_cast<const uint4*>(planes), epi, out, ldo, rows, static_cast<int>(N), static_cast<int>(K), seed_base, bscale)
    if (epi.sgn_x != nullptr) { if (mask_bits) GR_LAUNCH_SGN(2); else if (mask_y) GR_LAUNCH_SGN(1); else GR_LAUNCH_SGN(0); }
    else if (mask_bits) GR_LAUNCH(2); else if (mask_y) GR_LAUNCH(1); else GR_LAUNCH(0);
#undef GR_LAUNCH_SGN
#undef GR_LAUNCH2
#undef GR_LAUNCH
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
#define GX_LAUNCH(Y, L, H) gemm_x6_kernel<Y, L, H><<<static_cast<unsigned>(blocks), kGxThreads, 0, static_cast<hipStream_t>(stream)>>>( \
      A, lda, pro, static_cast<const uint4*>(planes), epi, out, ldo, rows, static_cast<int>(N), static_cast<int>(K), seed_base, bscale)
#define GX_LAUNCH2(Y, L) do { if (f16) GX_LAUNCH(Y, L, true); else GX_LAUNCH(Y, L, false); } while (0)
#define GX_LAUNCH3(L) do { if (mask_bits) GX_LAUNCH2(2, L); else if (mask_y) GX_LAUNCH2(1, L); else GX_LAUNCH2(0, L); } while (0)
  if (epi.lnb_x != nullptr) GX_LAUNCH3(true); else GX_LAUNCH3(false);
#undef GX_LAUNCH3
#undef GX_LAUNCH2
#undef GX_LAUNCH
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

// ---- superset entries (ABI 11): the arithmetic as an argument (1 = ALLSET_ARITH_BF16X6 planes, 2 = ALLSET_ARITH_FP16X3 planes) and the
// forward's 1-bit activation mask -- written by the forward (mask_out; "mask layout", N % 64 == 0), read by the backward-data GEMMs
// (mask_bits, K % 64 == 0; replaces the [rows, K] fp32 mask source mask_y: one dword per thread and K step instead of 32 bytes) ----
extern "C" int allset_gemm_wide(int arith, const float* A, int64_t lda, const float* mask_y, int64_t ldy, const uint32_t* mask_bits,
                                float p_mask, int relu_in, const float* stats, const float* gamma, const float* beta, float p_in,
                                uint64_t seed_in, const void* planes, const float* bias, int relu_out, float p_out, uint64_t seed_out,
                                uint32_t* mask_out, float* stats_out, float stats_eps, int stats_relu, float* out, int64_t ldo,
                                int64_t rows, int64_t N, int64_t K, const uint64_t* seed_base, void* stream) {
  clear_error();
  ALLSET_REQUIRE(arith == ALLSET_ARITH_BF16X6 || arith == ALLSET_ARITH_FP16X3, "gemm_wide: arith names the planes' format: ALLSET_ARITH_BF16X6 or ALLSET_ARITH_FP16X3");
  ALLSET_REQUIRE(p_out >= 0.f && p_out < 1.f, "gemm_wide: dropout p must be in [0,1)");
  ALLSET_REQUIRE(bias == nullptr || aligned16(bias), "gemm_wide: bias must be 16-byte aligned");
  if (stats_out != nullptr && N != kGxBN) {
    set_error("gemm_wide: stats_out needs N == 256 (one wave holds a whole output row)");
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(stats_out == nullptr || (stats_eps > 0.f && (reinterpret_cast<uintptr_t>(stats_out) & 7u) == 0), "gemm_wide: stats_out needs eps > 0 and an 8-byte aligned buffer");
  GxEpi epi{bias, relu_out, p_out, seed_out, mask_out, nullptr, 0, nullptr, nullptr, 0, 0.f, 0, nullptr, stats_out, stats_eps, stats_relu};
  return gemm_x6_impl(A, lda, mask_y, ldy, p_mask, relu_in, stats, gamma, beta, p_in, seed_in, planes, epi, out, ldo, rows, N, K, seed_base,
                      stream, arith == ALLSET_ARITH_FP16X3, mask_bits);
}

extern "C" int allset_gemm_wide_sgn(int arith, const float* G, int64_t ldg, const float* mask_y, int64_t ldy, const uint32_t* mask_bits,
                                    float p_mask, const void* planes, const float* x, int64_t ldx, float* gx, int64_t ldgx, int64_t rows,
                                    int64_t N, int64_t K, const uint64_t* seed_base, void* stream) {
  clear_error();
  ALLSET_REQUIRE(arith == ALLSET_ARITH_BF16X6 || arith == ALLSET_ARITH_FP16X3, "gemm_wide_sgn: arith names the planes' format: ALLSET_ARITH_BF16X6 or ALLSET_ARITH_FP16X3");
  ALLSET_REQUIRE(rows == 0 || x != nullptr, "gemm_wide_sgn: null x");
  GxEpi epi{nullptr, 0, 0.f, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0.f, 0, nullptr, nullptr, 0.f, 0, x, ldx};
  return gemm_x6_impl(G, ldg, mask_y, ldy, p_mask, 0, nullptr, nullptr, nullptr, 0.f, 0, planes, epi, gx, ldgx, rows, N, K, seed_base, stream,
                      arith == ALLSET_ARITH_FP16X3, mask_bits);
}

extern "C" int allset_gemm_wide_lnb(int arith, const float* G, int64_t ldg, const float* mask_y, int64_t ldy, const uint32_t* mask_bits,
                                    float p_mask, const void* planes, const float* x, int64_t ldx, const float* stats,
                                    const float* gamma, int relu_in, float p, uint64_t seed, float* gx, int64_t ldgx, float* partials,
                                    int64_t n_partials, int64_t rows, int64_t N, int64_t K, const uint64_t* seed_base, void* stream) {
  clear_error();
  ALLSET_REQUIRE(arith == ALLSET_ARITH_BF16X6 || arith == ALLSET_ARITH_FP16X3, "gemm_wide_lnb: arith names the planes' format: ALLSET_ARITH_BF16X6 or ALLSET_ARITH_FP16X3");
  ALLSET_REQUIRE(N >= 4 && N <= kGxBN, "gemm_wide_lnb: the LayerNorm row (N) must fit one 256-column tile");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "gemm_wide_lnb: dropout p must be in [0,1)");
  ALLSET_REQUIRE(partials != nullptr && n_partials == allset_gemm_x6_lnb_partials(rows), "gemm_wide_lnb: partials must hold allset_gemm_x6_lnb_partials(rows) x 2 x N floats");
  if (rows == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(partials, 0, static_cast<size_t>(n_partials) * 2 * N * sizeof(float), static_cast<hipStream_t>(stream)));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(x && stats && gamma && gx, "gemm_wide_lnb: null pointer");
  ALLSET_REQUIRE(ldx >= N && ldx % 4 == 0 && aligned16(x) && aligned16(gamma), "gemm_wide_lnb: x rows / gamma must be 16-byte aligned");
  GxEpi epi{nullptr, 0, 0.f, 0, nullptr, x, ldx, stats, gamma, relu_in, p, seed, partials};
  return gemm_x6_impl(G, ldg, mask_y, ldy, p_mask, 0, nullptr, nullptr, nullptr, 0.f, 0, planes, epi, gx, ldgx, rows, N, K, seed_base, stream,
                      arith == ALLSET_ARITH_FP16X3, mask_bits);
}
