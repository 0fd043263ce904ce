// The loss of the reference's training loop (train.py:479-480: log_softmax over classes, NLLLoss over the train split) as two
// small kernels -- at dataset scale (Cora: 2708 x 7 logits) the torch composition is ~8 launches of 5-16 us each, a tenth of
// the whole graphed training step.  Rows are selected by a 0/1 weight per row (the train split as a mask), so the gradient
// kernel writes every row of d loss / d logits (zeros outside the split) and nothing needs a scatter.
//   loss = inv_count * sum_r w[r] * (logsumexp(logits[r, :]) - logits[r, y[r]])
#include "common.h"

#include <float.h>

namespace allset {

constexpr int kLossBlock = 256;

__device__ __forceinline__ float row_lse(const float* __restrict__ row, int C) {
  float m = -FLT_MAX;
  for (int c = 0; c < C; ++c) m = fmaxf(m, row[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += __expf(row[c] - m);
  return m + __logf(s);
}

__global__ __launch_bounds__(kLossBlock) void nll_fwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                            const int64_t* __restrict__ y, const float* __restrict__ w,
                                                            float inv_count, float* __restrict__ partials, int64_t n, int C,
                                                            unsigned* __restrict__ ticket, float* __restrict__ total) {
  __shared__ float red[kLossBlock / kWave];
  float acc = 0.f;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kLossBlock + threadIdx.x; r < n; r += static_cast<int64_t>(gridDim.x) * kLossBlock) {
    const float wr = w ? w[r] : 1.f;
    if (wr != 0.f) {
      const float* row = logits + r * ld;
      const int64_t t = y[r];
      // a label outside [0, C) (an unlabeled row that reached a split, a class count that does not match the head) contributes
      // NaN: loud in the loss, where torch's nll_loss would raise -- never an out-of-bounds read
      acc += wr * (row_lse(row, C) - ((t >= 0 && t < C) ? row[t] : __builtin_nanf("")));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLossBlock / kWave; ++k) s += red[k];
    partials[blockIdx.x] = s * inv_count;
    if (ticket != nullptr) {
      // the LAST workgroup to arrive adds the partials in index order (the same sum whichever workgroup that is) and re-arms the
      // ticket: the loss leaves this launch as one number, not as a buffer for a reduction launch of its own
      __threadfence();
      const unsigned prev = atomicAdd(ticket, 1u);
      if (prev == gridDim.x - 1) {
        __threadfence();
        float t = 0.f;
        for (unsigned b = 0; b < gridDim.x; ++b) t += __hip_atomic_load(partials + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        total[0] = t;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

__global__ __launch_bounds__(kLossBlock) void nll_bwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                            const int64_t* __restrict__ y, const float* __restrict__ w,
                                                            float inv_count, const float* __restrict__ gout,
                                                            float* __restrict__ glogits, int64_t ldg, int64_t n, int C) {
  const float scale = inv_count * (gout ? gout[0] : 1.f);
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kLossBlock + threadIdx.x; r < n; r += static_cast<int64_t>(gridDim.x) * kLossBlock) {
    const float wr = (w ? w[r] : 1.f) * scale;
    float* g = glogits + r * ldg;
    if (wr == 0.f) {
      for (int c = 0; c < C; ++c) g[c] = 0.f;
      continue;
    }
    const float* row = logits + r * ld;
    const float lse = row_lse(row, C);
    const int64_t t = y[r];
    const float bad = (t >= 0 && t < C) ? 0.f : __builtin_nanf("");        // out-of-range label: NaN gradient (see nll_fwd_kernel)
    for (int c = 0; c < C; ++c) g[c] = wr * (__expf(row[c] - lse) - (c == t ? 1.f : 0.f)) + bad;
  }
}

// Per-split accuracy and loss of the reference's evaluate() (train.py:169-199: eval_acc = mean(argmax == y), NLLLoss of
// log_softmax, each over the train / valid / test rows) in ONE pass over the logits: partials[block][6] = {correct_0..2, nll_0..2}
// summed over the block's rows of each split (split[r] in {0, 1, 2}, anything else = in no split).  The driver keeps the six
// numbers of every epoch on the device and reads them back once per run -- the reference's per-epoch .cpu() calls are host
// round trips that dwarf a 0.4 ms training step at dataset scale.
__global__ __launch_bounds__(kLossBlock) void split_metrics_kernel(const float* __restrict__ logits, int64_t ld,
                                                                  const int64_t* __restrict__ y, const int8_t* __restrict__ split,
                                                                  float* __restrict__ partials, int64_t n, int C) {
  __shared__ float red[kLossBlock / kWave][6];
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kLossBlock + threadIdx.x; r < n; r += static_cast<int64_t>(gridDim.x) * kLossBlock) {
    const int sp = split[r];
    if (sp < 0 || sp > 2) continue;
    const float* row = logits + r * ld;
    float m = row[0];
    int am = 0;
    for (int c = 1; c < C; ++c) if (row[c] > m) { m = row[c]; am = c; }      // first maximum, as torch.argmax
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += __expf(row[c] - m);
    const int64_t t = y[r];
    const float nll = (m + __logf(s)) - ((t >= 0 && t < C) ? row[t] : __builtin_nanf(""));
    const float ok = am == t ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { acc[k] += sp == k ? ok : 0.f; acc[3 + k] += sp == k ? nll : 0.f; }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kLossBlock / kWave; ++w) s += red[w][threadIdx.x];
    partials[blockIdx.x * 6 + threadIdx.x] = s;
  }
}

static inline unsigned loss_grid(int64_t n) {
  int64_t b = (n + kLossBlock - 1) / kLossBlock;
  if (b > 256) b = 256;
  return static_cast<unsigned>(b < 1 ? 1 : b);
}

}  // namespace allset

using namespace allset;

extern "C" int allset_nll_partials(int64_t n, int64_t* n_partials) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && n_partials != nullptr, "nll_partials: bad argument");
  *n_partials = loss_grid(n);
  return ALLSET_OK;
}

extern "C" int allset_nll_logsoftmax_fwd(const float* logits, int64_t ld, const int64_t* y, const float* w, float inv_count,
                                         float* partials, int64_t n_partials, int64_t n, int64_t C, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && C >= 1 && C < INT32_MAX && ld >= C, "nll_logsoftmax_fwd: bad size");
  ALLSET_REQUIRE(partials != nullptr && n_partials == loss_grid(n), "nll_logsoftmax_fwd: partials must hold allset_nll_partials(n) floats");
  ALLSET_REQUIRE(n == 0 || (logits && y), "nll_logsoftmax_fwd: null pointer");
  nll_fwd_kernel<<<loss_grid(n), kLossBlock, 0, static_cast<hipStream_t>(stream)>>>(logits, ld, y, w, inv_count, partials, n, static_cast<int>(C),
                                                                                     nullptr, nullptr);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_nll_logsoftmax_fwd_total(const float* logits, int64_t ld, const int64_t* y, const float* w, float inv_count,
                                               float* partials, int64_t n_partials, uint32_t* ticket, float* total, int64_t n,
                                               int64_t C, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && C >= 1 && C < INT32_MAX && ld >= C, "nll_logsoftmax_fwd_total: bad size");
  ALLSET_REQUIRE(partials != nullptr && n_partials == loss_grid(n), "nll_logsoftmax_fwd_total: partials must hold allset_nll_partials(n) floats");
  ALLSET_REQUIRE(ticket != nullptr && total != nullptr, "nll_logsoftmax_fwd_total: ticket (a zeroed uint32 the launch re-arms) and total are required");
  ALLSET_REQUIRE(n == 0 || (logits && y), "nll_logsoftmax_fwd_total: null pointer");
  nll_fwd_kernel<<<loss_grid(n), kLossBlock, 0, static_cast<hipStream_t>(stream)>>>(logits, ld, y, w, inv_count, partials, n, static_cast<int>(C),
                                                                                     ticket, total);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_nll_logsoftmax_bwd(const float* logits, int64_t ld, const int64_t* y, const float* w, float inv_count,
                                         const float* gout, float* glogits, int64_t ldg, int64_t n, int64_t C, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && C >= 1 && C < INT32_MAX && ld >= C && ldg >= C, "nll_logsoftmax_bwd: bad size");
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(logits && y && glogits, "nll_logsoftmax_bwd: null pointer");
  nll_bwd_kernel<<<loss_grid(n), kLossBlock, 0, static_cast<hipStream_t>(stream)>>>(logits, ld, y, w, inv_count, gout, glogits, ldg, n, static_cast<int>(C));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_split_metrics(const float* logits, int64_t ld, const int64_t* y, const int8_t* split, float* partials,
                                    int64_t n_partials, int64_t n, int64_t C, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && C >= 1 && C < INT32_MAX && ld >= C, "split_metrics: bad size");
  ALLSET_REQUIRE(partials != nullptr && n_partials == loss_grid(n), "split_metrics: partials must hold 6 x allset_nll_partials(n) floats");
  ALLSET_REQUIRE(n == 0 || (logits && y && split), "split_metrics: null pointer");
  split_metrics_kernel<<<loss_grid(n), kLossBlock, 0, static_cast<hipStream_t>(stream)>>>(logits, ld, y, split, partials, n, static_cast<int>(C));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

// ---- PMA's folded attention logits: alpha = K . att_r is linear in x, so the layer multiplies x by
//   w[h, k] = sum_c W_K[h C + c, k] att_r[h, c],   b[h] = sum_c b_K[h C + c] att_r[h, c]
// instead of forming K (reference layers.py:126-131 computes the full [n, H C] projection).  As torch ops the fold and its
// backward are ~10 tiny launches per direction; at dataset scale that is a tenth of a replayed AllSetTransformer step.
namespace allset {

// T = float, or uint16_t for bf16 parameters (configs[4] regime: fp32 arithmetic, one rounding per output -- the torch
// expression it replaces rounded every product to bf16 first and cost ~12 launches per conv and step)
template <typename T> __device__ __forceinline__ float fold_ld(const T* p, int64_t i);
template <> __device__ __forceinline__ float fold_ld<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float fold_ld<uint16_t>(const uint16_t* p, int64_t i) { return bf16_to_f32(p[i]); }
__device__ __forceinline__ void fold_st(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void fold_st(uint16_t* p, int64_t i, float v) { p[i] = static_cast<uint16_t>(cvt_pk_bf16(v, 0.f) & 0xffffu); }

// One workgroup per (h, 64 columns): 64 column lanes x 4 row groups, each thread sums C / 4 rows with independent loads, the four
// partial sums meet in LDS (round 6: the one-thread-per-output form walked C dependent loads -- 21 us at C = 64, K = 256, twice
// per AllSetTransformer layer and step).  The last workgroup (blockIdx.x == H * ceil(K / 64)) folds the bias.
template <typename T>
__global__ __launch_bounds__(256) void pma_fold_fwd_kernel(const T* __restrict__ Wk, const T* __restrict__ bk,
                                                          const T* __restrict__ att, T* __restrict__ w,
                                                          T* __restrict__ b, int H, int C, int K) {
  __shared__ float red[4][64];
  const int kb = (K + 63) / 64;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  if (static_cast<int>(blockIdx.x) == H * kb) {              // b[h] = sum_c bk[h C + c] att[h, c]: wave `grp` takes heads grp, grp + 4, ..
    for (int h = grp; h < H; h += 4) {
      float s = 0.f;
      if (bk != nullptr)
        for (int c = lane; c < C; c += 64) s = fmaf(fold_ld(bk, h * C + c), fold_ld(att, h * C + c), s);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0) fold_st(b, h, s);
    }
    return;
  }
  const int h = blockIdx.x / kb, k = (blockIdx.x % kb) * 64 + lane;
  float s = 0.f;
  if (k < K) {
    int c = grp;
    for (; c + 12 < C; c += 16) {                           // four independent loads in flight per thread
      const float w0 = fold_ld(Wk, static_cast<int64_t>(h * C + c) * K + k), w1 = fold_ld(Wk, static_cast<int64_t>(h * C + c + 4) * K + k);
      const float w2 = fold_ld(Wk, static_cast<int64_t>(h * C + c + 8) * K + k), w3 = fold_ld(Wk, static_cast<int64_t>(h * C + c + 12) * K + k);
      s = fmaf(w0, fold_ld(att, h * C + c), s); s = fmaf(w1, fold_ld(att, h * C + c + 4), s);
      s = fmaf(w2, fold_ld(att, h * C + c + 8), s); s = fmaf(w3, fold_ld(att, h * C + c + 12), s);
    }
    for (; c < C; c += 4) s = fmaf(fold_ld(Wk, static_cast<int64_t>(h * C + c) * K + k), fold_ld(att, h * C + c), s);
  }
  red[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && k < K) fold_st(w, h * K + k, (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// gWk[h C + c, k] = gw[h, k] att[h, c];  gbk[h C + c] = gb[h] att[h, c];  gatt[h, c] = sum_k gw[h, k] Wk[h C + c, k] + gb[h] bk[h C + c]
template <typename T>
__global__ __launch_bounds__(256) void pma_fold_bwd_kernel(const T* __restrict__ Wk, const T* __restrict__ bk,
                                                          const T* __restrict__ att, const T* __restrict__ gw,
                                                          const T* __restrict__ gb, T* __restrict__ gWk,
                                                          T* __restrict__ gbk, T* __restrict__ gatt, int H, int C, int K) {
  // one workgroup per row (h, c) of W_K: writes its gWk row and reduces its gatt entry
  const int row = blockIdx.x, h = row / C;
  const float a = fold_ld(att, row);
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float g = fold_ld(gw, h * K + k);
    fold_st(gWk, static_cast<int64_t>(row) * K + k, g * a);
    s = fmaf(g, fold_ld(Wk, static_cast<int64_t>(row) * K + k), s);
  }
  __shared__ float red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = red[0] + red[1] + red[2] + red[3];
    const float g = gb ? fold_ld(gb, h) : 0.f;
    if (bk != nullptr) { t = fmaf(g, fold_ld(bk, row), t); fold_st(gbk, row, g * a); }
    fold_st(gatt, row, t);
  }
}

}  // namespace allset

template <typename T>
static int pma_fold_fwd_impl(const T* Wk, const T* bk, const T* att, T* w, T* b, int64_t H, int64_t C, int64_t K, void* stream) {
  ALLSET_REQUIRE(H >= 1 && C >= 1 && K >= 1 && H * K < (int64_t{1} << 30), "pma_fold_fwd: bad size");
  ALLSET_REQUIRE(Wk && att && w && b, "pma_fold_fwd: null pointer");
  const unsigned grid = static_cast<unsigned>(H * ((K + 63) / 64) + 1);
  allset::pma_fold_fwd_kernel<T><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(Wk, bk, att, w, b, static_cast<int>(H), static_cast<int>(C),
                                                                               static_cast<int>(K));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

template <typename T>
static int pma_fold_bwd_impl(const T* Wk, const T* bk, const T* att, const T* gw, const T* gb, T* gWk, T* gbk, T* gatt, int64_t H,
                             int64_t C, int64_t K, void* stream) {
  ALLSET_REQUIRE(H >= 1 && C >= 1 && K >= 1 && H * C < (int64_t{1} << 30), "pma_fold_bwd: bad size");
  ALLSET_REQUIRE(Wk && att && gw && gWk && gatt && (bk == nullptr || gbk != nullptr), "pma_fold_bwd: null pointer");
  allset::pma_fold_bwd_kernel<T><<<static_cast<unsigned>(H * C), 256, 0, static_cast<hipStream_t>(stream)>>>(
      Wk, bk, att, gw, gb, gWk, gbk, gatt, static_cast<int>(H), static_cast<int>(C), static_cast<int>(K));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_pma_fold_fwd(const float* Wk, const float* bk, const float* att, float* w, float* b, int64_t H, int64_t C,
                                   int64_t K, void* stream) {
  clear_error();
  return pma_fold_fwd_impl<float>(Wk, bk, att, w, b, H, C, K, stream);
}

extern "C" int allset_pma_fold_bwd(const float* Wk, const float* bk, const float* att, const float* gw, const float* gb, float* gWk,
                                   float* gbk, float* gatt, int64_t H, int64_t C, int64_t K, void* stream) {
  clear_error();
  return pma_fold_bwd_impl<float>(Wk, bk, att, gw, gb, gWk, gbk, gatt, H, C, K, stream);
}

// the same for bf16 parameters (every pointer bf16; fp32 arithmetic, one rounding per output element)
extern "C" int allset_pma_fold_fwd_bf16(const void* Wk, const void* bk, const void* att, void* w, void* b, int64_t H, int64_t C,
                                        int64_t K, void* stream) {
  clear_error();
  return pma_fold_fwd_impl<uint16_t>(static_cast<const uint16_t*>(Wk), static_cast<const uint16_t*>(bk), static_cast<const uint16_t*>(att),
                                     static_cast<uint16_t*>(w), static_cast<uint16_t*>(b), H, C, K, stream);
}

extern "C" int allset_pma_fold_bwd_bf16(const void* Wk, const void* bk, const void* att, const void* gw, const void* gb, void* gWk,
                                        void* gbk, void* gatt, int64_t H, int64_t C, int64_t K, void* stream) {
  clear_error();
  return pma_fold_bwd_impl<uint16_t>(static_cast<const uint16_t*>(Wk), static_cast<const uint16_t*>(bk), static_cast<const uint16_t*>(att),
                                     static_cast<const uint16_t*>(gw), static_cast<const uint16_t*>(gb), static_cast<uint16_t*>(gWk),
                                     static_cast<uint16_t*>(gbk), static_cast<uint16_t*>(gatt), H, C, K, stream);
}
