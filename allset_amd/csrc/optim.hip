// Adam over a LIST of parameter tensors in one launch (reference train.py:469 uses torch.optim.Adam; same arithmetic, torch's
// default non-amsgrad form with L2 weight decay).  torch's own fused multi-tensor Adam in capturable mode takes ~40 us for the
// ~20 small tensors of a Cora-sized SetGNN -- a tenth of a hipGraph-replayed training step; this kernel needs ~5.
//   g = grad + wd * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t is read from DEVICE memory (one float step counter per tensor, which the caller increments before the launch -- one
// _foreach_add_), so a captured hipGraph advances the bias corrections on every replay.
#include "common.h"

namespace allset {

constexpr int kAdamMaxTensors = 48;
constexpr int kAdamBlock = 256;
constexpr int kAdamChunk = 2048;            // elements per workgroup

struct AdamTable {
  void* p[kAdamMaxTensors];                  // fp32 or bf16 (parameter, gradient and both moments share the dtype, as in torch)
  const void* g[kAdamMaxTensors];
  void* m[kAdamMaxTensors];
  void* v[kAdamMaxTensors];
  const float* step[kAdamMaxTensors];        // per-tensor step counter t >= 1 (device float), as torch's capturable Adam keeps it
  int64_t numel[kAdamMaxTensors];
  int32_t first_chunk[kAdamMaxTensors + 1];  // prefix sums of ceil(numel / kAdamChunk)
  int32_t count;
};

template <typename T>
__device__ __forceinline__ float adam_ld(const T* p, int64_t i) {
  if constexpr (sizeof(T) == 4) return p[i];
  else return bf16_to_f32(p[i]);
}
template <typename T>
__device__ __forceinline__ void adam_st(T* p, int64_t i, float v) {
  if constexpr (sizeof(T) == 4) p[i] = v;
  else p[i] = static_cast<uint16_t>(f32_to_bf16(v));
}

template <typename T>      // float, or uint16_t = bf16 bits (fp32 arithmetic, each stored value rounded once)
__global__ __launch_bounds__(kAdamBlock) void adam_kernel(AdamTable tb, float lr, float b1, float b2, float eps, float wd) {
  // which tensor does this workgroup's chunk belong to (<= 48 entries: a linear scan of scalars)
  const int chunk = blockIdx.x;
  int t = 0;
  while (t + 1 < tb.count && tb.first_chunk[t + 1] <= chunk) ++t;
  const int64_t base = static_cast<int64_t>(chunk - tb.first_chunk[t]) * kAdamChunk;
  const int64_t n = tb.numel[t];
  T* __restrict__ p = static_cast<T*>(tb.p[t]);
  const T* __restrict__ g = static_cast<const T*>(tb.g[t]);
  T* __restrict__ m = static_cast<T*>(tb.m[t]);
  T* __restrict__ v = static_cast<T*>(tb.v[t]);
  const float tt = tb.step[t][0];
  const float bc1 = 1.f - powf(b1, tt), bc2 = 1.f - powf(b2, tt);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  for (int64_t i = base + threadIdx.x; i < base + kAdamChunk && i < n; i += kAdamBlock) {
    const float pv = adam_ld(p, i);
    const float gv = adam_ld(g, i) + wd * pv;
    const float mv = b1 * adam_ld(m, i) + (1.f - b1) * gv;
    const float vv = b2 * adam_ld(v, i) + (1.f - b2) * gv * gv;
    adam_st(m, i, mv);
    adam_st(v, i, vv);
    adam_st(p, i, pv - step_size * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  }
}

}  // namespace allset

using namespace allset;

extern "C" int allset_adam_max_tensors(void) { return kAdamMaxTensors; }

static int adam_impl(int dtype, void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                     const float* const* steps, const int64_t* numel, int64_t count, float lr, float beta1, float beta2,
                     float eps, float weight_decay, void* stream);

extern "C" int allset_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                const float* const* steps, const int64_t* numel, int64_t count, float lr, float beta1, float beta2,
                                float eps, float weight_decay, void* stream) {
  clear_error();
  return adam_impl(ALLSET_F32, reinterpret_cast<void* const*>(params), reinterpret_cast<const void* const*>(grads),
                   reinterpret_cast<void* const*>(exp_avg), reinterpret_cast<void* const*>(exp_avg_sq), steps, numel, count, lr, beta1,
                   beta2, eps, weight_decay, stream);
}

extern "C" int allset_adam_step_dtype(int dtype, void* const* params, const void* const* grads, void* const* exp_avg,
                                      void* const* exp_avg_sq, const float* const* steps, const int64_t* numel, int64_t count,
                                      float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
  clear_error();
  ALLSET_REQUIRE(dtype == ALLSET_F32 || dtype == ALLSET_BF16, "adam_step_dtype: dtype must be ALLSET_F32 or ALLSET_BF16");
  return adam_impl(dtype, params, grads, exp_avg, exp_avg_sq, steps, numel, count, lr, beta1, beta2, eps, weight_decay, stream);
}

static int adam_impl(int dtype, void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                     const float* const* steps, const int64_t* numel, int64_t count, float lr, float beta1, float beta2,
                     float eps, float weight_decay, void* stream) {
  ALLSET_REQUIRE(count >= 0 && count <= kAdamMaxTensors, "adam_step: at most %d tensors per call", kAdamMaxTensors);
  if (count == 0) return ALLSET_OK;
  ALLSET_REQUIRE(params && grads && exp_avg && exp_avg_sq && steps && numel, "adam_step: null pointer");
  AdamTable tb;
  int32_t chunks = 0;
  for (int64_t k = 0; k < count; ++k) {
    ALLSET_REQUIRE(numel[k] >= 0 && params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k] && steps[k], "adam_step: null tensor %lld", static_cast<long long>(k));
    tb.p[k] = params[k]; tb.g[k] = grads[k]; tb.m[k] = exp_avg[k]; tb.v[k] = exp_avg_sq[k]; tb.step[k] = steps[k];
    tb.numel[k] = numel[k];
    tb.first_chunk[k] = chunks;
    const int64_t c = (numel[k] + kAdamChunk - 1) / kAdamChunk;
    ALLSET_REQUIRE(chunks + c < (int64_t{1} << 30), "adam_step: too many elements");
    chunks += static_cast<int32_t>(c);
  }
  tb.first_chunk[count] = chunks;
  tb.count = static_cast<int32_t>(count);
  if (chunks == 0) return ALLSET_OK;
  if (dtype == ALLSET_BF16)
    adam_kernel<uint16_t><<<static_cast<unsigned>(chunks), kAdamBlock, 0, static_cast<hipStream_t>(stream)>>>(tb, lr, beta1, beta2, eps, weight_decay);
  else
    adam_kernel<float><<<static_cast<unsigned>(chunks), kAdamBlock, 0, static_cast<hipStream_t>(stream)>>>(tb, lr, beta1, beta2, eps, weight_decay);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
