// The FIRST Linear of a model: `LayerNorm(raw features) -> Linear` on an input that needs no gradient (reference
// models.py:473-476: dropout(0.2) of data.x, then V2EConvs[0].f_enc = MLP(num_features -> hidden, InputNorm) -- layers.py:571-579).
// Raw-feature widths (1433 Cora, 3703 Citeseer) fit none of the LDS-resident-weight kernels, and at dataset scale the step is a
// chain of short kernels, so the aim here is FEWER of them.  With x_hat the LayerNorm of a row WITHOUT its affine part,
//
//   y = (x_hat * gamma + beta) W^T + b = [x_hat | 1] [W * gamma | b + W beta]^T            (one GEMM, bias folded in as a column)
//
// and, because the input needs no gradient, every parameter gradient follows from ONE product M = gy^T [x_hat | 1]  ([O, d + 1]):
//
//   gW[k,j] = gamma_j M[k,j] + beta_j M[k,d]      gb[k] = M[k,d]
//   ggamma_j = sum_k W[k,j] M[k,j]                gbeta_j = sum_k W[k,j] M[k,d]
//
// -- the [n, d] input gradient of the Linear, the LayerNorm backward pass over it and its column reductions are never formed
// (three of the six longest kernels of a Cora-shaped step).
//
//   xhat_rows_kernel     [dropout ->] row statistics -> x_hat, a ones column, zero padding to the GEMM's K     1 read + 1 write
//   fold_ln_kernel       [W * gamma | b + W beta | 0]   (one workgroup per output row; fixed-order sums)        O x K, tiny
//   unfold_ln_kernel     M -> gW, gb, ggamma, gbeta      (64 columns x 4 row groups per workgroup)              O x K, tiny
//
// The two GEMMs ([n,K] x [K,O] and [O,n] x [n,K], K = d + 1 padded to 16) stay on the library through torch.
#include "common.h"

namespace allset {

__device__ __forceinline__ float wave_sum(float v) {      // butterfly: every lane ends with the same sum, bit-identical
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int NS>
__global__ __launch_bounds__(kBlock) void xhat_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int d, float eps,
                                                           float p_pre, uint64_t seed, const uint64_t* __restrict__ seed_base,
                                                           float* __restrict__ xh, int64_t ldxh) {
  seed = resolve_seed(seed_base, seed);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = lane_id();
  const float* xr = x + row * ldx;
  const float inv_keep = p_pre > 0.f ? 1.f / (1.f - p_pre) : 1.f;
  const uint32_t thr = drop_threshold(p_pre);
  float v[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int c = k * kWave + lane;
    v[k] = c < d ? xr[c] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int c = k * kWave + lane;
    if (p_pre > 0.f && c < d) v[k] *= keep_scale(seed, row * d + c, thr, inv_keep);
    s += v[k];
  }
  const float mean = wave_sum(s) / static_cast<float>(d);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int c = k * kWave + lane;
    v[k] = c < d ? v[k] - mean : 0.f;
    q = fmaf(v[k], v[k], q);
  }
  const float rstd = rsqrtf(wave_sum(q) / static_cast<float>(d) + eps);
  float* out = xh + row * ldxh;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int c = k * kWave + lane;
    if (c < d) out[c] = v[k] * rstd;
  }
  for (int c = d + lane; c < ldxh; c += kWave) out[c] = c == d ? 1.f : 0.f;
}

__global__ __launch_bounds__(kBlock) void fold_ln_kernel(const float* __restrict__ W, int64_t ldw, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ b, int d,
                                                         float* __restrict__ Wp, int64_t ldwp) {
  __shared__ float red[kBlock];
  const int k = blockIdx.x, t = threadIdx.x;
  const float* w = W + static_cast<int64_t>(k) * ldw;
  float* o = Wp + static_cast<int64_t>(k) * ldwp;
  float acc = 0.f;
  for (int j = t; j < d; j += kBlock) {
    const float wv = w[j];
    o[j] = wv * gamma[j];
    acc = fmaf(wv, beta[j], acc);
  }
  red[t] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  for (int c = d + t; c < ldwp; c += kBlock) o[c] = c == d ? red[0] + (b ? b[k] : 0.f) : 0.f;
}

// 64 columns x 4 groups of output rows per workgroup (thread t: column t % 64, rows k = t / 64, +4, ...): the loads of a thread's
// rows are independent (unrolled by 8), the four groups' ggamma / gbeta terms meet in LDS in a fixed order.
__global__ __launch_bounds__(kBlock) void unfold_ln_kernel(const float* __restrict__ M, int64_t ldm, const float* __restrict__ W,
                                                           int64_t ldw, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int O, int d, float* __restrict__ gW, int64_t ldgw, float* __restrict__ gb,
                                                           float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                           const float* __restrict__ su_part, int n_slices) {
  __shared__ float red[2][3][kWave];
  __shared__ float s_s[kBlock], u_s[kBlock];                 // su_part form (sparse_input.hip): O <= 256
  const int t = threadIdx.x, jl = t & 63, kq = t >> 6;
  const int j = blockIdx.x * kWave + jl;
  if (su_part != nullptr) {      // M holds gy^T v-hat WITHOUT the rows' shared offset: M_true = M - u, ones column = s (slice sums)
    if (t < O) {
      float s = 0.f, u = 0.f;
      for (int s0 = 0; s0 < n_slices; s0 += 8) {             // eight slices' loads in flight, summed in slice order
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int sl = min(s0 + i, n_slices - 1);
          a[i] = su_part[(2 * sl) * O + t]; b[i] = su_part[(2 * sl + 1) * O + t];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (s0 + i < n_slices) { s += a[i]; u += b[i]; }
      }
      s_s[t] = s; u_s[t] = u;
    }
    __syncthreads();
    if (gb && kq == 0 && j < O) gb[j] = s_s[j];
  } else if (gb && kq == 0 && j < O) {
    gb[j] = M[static_cast<int64_t>(j) * ldm + d];
  }
  const bool live = j < d;
  const float g = live ? gamma[j] : 0.f, bt = live ? beta[j] : 0.f;
  float gg = 0.f, gbt = 0.f;
  if (live) {
#pragma unroll 8
    for (int k = kq; k < O; k += 4) {
      const float m = M[static_cast<int64_t>(k) * ldm + j] - (su_part ? u_s[k] : 0.f);
      const float sk = su_part ? s_s[k] : M[static_cast<int64_t>(k) * ldm + d];        // (uniform across the wave)
      const float w = W[static_cast<int64_t>(k) * ldw + j];
      gW[static_cast<int64_t>(k) * ldgw + j] = fmaf(g, m, bt * sk);
      gg = fmaf(w, m, gg);
      gbt = fmaf(w, sk, gbt);
    }
  }
  if (kq > 0) { red[0][kq - 1][jl] = gg; red[1][kq - 1][jl] = gbt; }
  __syncthreads();
  if (kq == 0 && live) {
    ggamma[j] = ((gg + red[0][0][jl]) + red[0][1][jl]) + red[0][2][jl];
    gbeta[j] = ((gbt + red[1][0][jl]) + red[1][1][jl]) + red[1][2][jl];
  }
}

}  // namespace allset

using namespace allset;

extern "C" int64_t allset_input_linear_k(int64_t d) {     // leading dimension of [x_hat | 1 | 0...]: d + 1 up to 64-byte rows
  return (d + 1 + 15) / 16 * 16;
}

extern "C" int allset_input_linear_supported(int64_t d) { return d >= 1 && d <= 64 * kWave; }

extern "C" int allset_xhat_rows(const float* x, int64_t ldx, int64_t n, int64_t d, float eps, float p_pre, uint64_t seed,
                                const uint64_t* seed_base, float* xh, int64_t ldxh, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && allset_input_linear_supported(d), "xhat_rows: 1 <= d <= 4096");
  ALLSET_REQUIRE(p_pre >= 0.f && p_pre < 1.f, "xhat_rows: dropout p must be in [0,1)");
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(x && xh, "xhat_rows: null pointer");
  ALLSET_REQUIRE(ldx >= d && ldxh >= d + 1, "xhat_rows: leading dimension too small (ldxh >= d + 1)");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = static_cast<unsigned>((n + kWavesPerBlock - 1) / kWavesPerBlock);
  const int di = static_cast<int>(d);
#define ALLSET_XHAT(NS) xhat_rows_kernel<NS><<<grid, kBlock, 0, st>>>(x, ldx, n, di, eps, p_pre, seed, seed_base, xh, ldxh)
  if (d <= 8 * kWave) ALLSET_XHAT(8);
  else if (d <= 16 * kWave) ALLSET_XHAT(16);
  else if (d <= 24 * kWave) ALLSET_XHAT(24);
  else if (d <= 32 * kWave) ALLSET_XHAT(32);
  else if (d <= 48 * kWave) ALLSET_XHAT(48);
  else ALLSET_XHAT(64);
#undef ALLSET_XHAT
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_fold_ln_linear(const float* W, int64_t ldw, const float* gamma, const float* beta, const float* b,
                                     int64_t O, int64_t d, float* Wp, int64_t ldwp, void* stream) {
  clear_error();
  ALLSET_REQUIRE(O >= 1 && O < INT32_MAX && d >= 1 && d < INT32_MAX, "fold_ln_linear: bad size");
  ALLSET_REQUIRE(W && gamma && beta && Wp, "fold_ln_linear: null pointer");
  ALLSET_REQUIRE(ldw >= d && ldwp >= d + 1, "fold_ln_linear: leading dimension too small (ldwp >= d + 1)");
  fold_ln_kernel<<<static_cast<unsigned>(O), kBlock, 0, static_cast<hipStream_t>(stream)>>>(W, ldw, gamma, beta, b,
                                                                                              static_cast<int>(d), Wp, ldwp);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static int unfold_impl(const float* M, int64_t ldm, const float* W, int64_t ldw, const float* gamma, const float* beta, int64_t O,
                       int64_t d, float* gW, int64_t ldgw, float* gb, float* ggamma, float* gbeta, const float* su_part,
                       int64_t n_slices, void* stream);

extern "C" int allset_unfold_ln_linear(const float* M, int64_t ldm, const float* W, int64_t ldw, const float* gamma,
                                       const float* beta, int64_t O, int64_t d, float* gW, int64_t ldgw, float* gb,
                                       float* ggamma, float* gbeta, void* stream) {
  clear_error();
  return unfold_impl(M, ldm, W, ldw, gamma, beta, O, d, gW, ldgw, gb, ggamma, gbeta, nullptr, 0, stream);
}

extern "C" int allset_unfold_ln_linear_ex(const float* M, int64_t ldm, const float* W, int64_t ldw, const float* gamma,
                                          const float* beta, int64_t O, int64_t d, float* gW, int64_t ldgw, float* gb,
                                          float* ggamma, float* gbeta, const float* su_part, int64_t n_slices, void* stream) {
  clear_error();
  ALLSET_REQUIRE(su_part == nullptr || (O <= kBlock && n_slices >= 1 && n_slices < 4096), "unfold_ln_linear_ex: su_part needs O <= 256 and n_slices >= 1");
  return unfold_impl(M, ldm, W, ldw, gamma, beta, O, d, gW, ldgw, gb, ggamma, gbeta, su_part, su_part ? n_slices : 0, stream);
}

static int unfold_impl(const float* M, int64_t ldm, const float* W, int64_t ldw, const float* gamma, const float* beta, int64_t O,
                       int64_t d, float* gW, int64_t ldgw, float* gb, float* ggamma, float* gbeta, const float* su_part,
                       int64_t n_slices, void* stream) {
  ALLSET_REQUIRE(O >= 1 && O < INT32_MAX && d >= 1 && d < INT32_MAX, "unfold_ln_linear: bad size");
  ALLSET_REQUIRE(M && W && gamma && beta && gW && ggamma && gbeta, "unfold_ln_linear: null pointer");
  ALLSET_REQUIRE(ldm >= d + (su_part ? 0 : 1) && ldw >= d && ldgw >= d, "unfold_ln_linear: leading dimension too small (ldm >= d + 1)");
  const int64_t cols = d > O ? d : O;
  unfold_ln_kernel<<<static_cast<unsigned>((cols + kWave - 1) / kWave), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      M, ldm, W, ldw, gamma, beta, static_cast<int>(O), static_cast<int>(d), gW, ldgw, gb, ggamma, gbeta, su_part,
      static_cast<int>(n_slices));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
