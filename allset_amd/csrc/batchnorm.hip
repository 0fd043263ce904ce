// Training-mode BatchNorm1d of the reference MLP (`Normalization='bn'`, the constructor default: layers.py:499-517, applied at
// layers.py:571-579) as ONE column-moment pass + a column-affine prologue of the fused Linear kernels:
//
//   mean_c = sum_r f(x)[r,c] / n,   var_c = sum_r (f(x)[r,c] - mean_c)^2 / n      f = relu (behind a Linear) or identity
//   BatchNorm(f(x)) = f(x) * a_c + b_c,   a_c = gamma_c * rsqrt(var_c + eps),  b_c = beta_c - mean_c * a_c
//
// The affine map is the operand prologue of the following Linear (allset_fused_linear_fwd_nm / _bwd_all_nm with norm_mode =
// ALLSET_NORM_COLUMN_AFFINE: the LayerNorm prologue with the row statistics switched off), so no normalised tensor is ever
// written; the dependence of (mean, var) on x comes back as one more column-affine term of the input gradient:
//
//   gx[r,c] += [f = relu ? x > 0 : 1] * (f(x)[r,c] * s_c + t_c),    s_c = 2 dvar_c / n,   t_c = dmean_c / n - s_c * mean_c
//
// (allset_amd/dense.py `_BatchNormLinear` has the closed forms of dmean / dvar from the kernels' column sums).
//
//   col_moments2_kernel     per-slice fp64 sums of f(x) and f(x)^2: both moments from ONE read    1 read, HBM-bound streaming
//   col_moments_kernel      per-slice fp32 sums of f(x) or of (f(x) - center)^2 (two-pass form)   1 read each
//   col_affine_add_kernel   gx += mask * (f(x) * s + t), in place                                 2 reads + 1 write, HBM-bound
//
// Partials are summed in a fixed order -- no atomics, bitwise reproducible.
#include "common.h"

namespace allset {

constexpr int kCmRowsPerSlice = 1024;          // rows one workgroup walks (at most); slices = ceil(n / this), capped
constexpr int kCmMaxSlices = 2048;

// Thread t owns column quad q = t % Q (Q = d / 4 <= 256) of the rows rg, rg + RG, ... of its slice (rg = t / Q, RG = 256 / Q).
__global__ __launch_bounds__(kBlock) void col_moments_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int d,
                                                             int relu_in, const float* __restrict__ center,
                                                             float* __restrict__ part, int64_t rows_per_slice) {
  __shared__ float4 red[kBlock];
  const int Q = d >> 2;
  const int RG = kBlock / Q;
  const int t = threadIdx.x;
  const int q = t % Q, rg = t / Q;
  const bool active = rg < RG;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_slice;
  const int64_t r1 = min(r0 + rows_per_slice, n);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool centred = center != nullptr;
    if (centred) c4 = *reinterpret_cast<const float4*>(center + 4 * q);
    const float* xp = x + 4 * q;
    // four independent 16-byte loads in flight per thread
    int64_t r = r0 + rg;
    for (; r + 3 * RG < r1; r += 4 * RG) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xp + (r + u * RG) * ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float4 w = v[u];
        if (relu_in) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
        if (centred) {
          w.x -= c4.x; w.y -= c4.y; w.z -= c4.z; w.w -= c4.w;
          acc.x = fmaf(w.x, w.x, acc.x); acc.y = fmaf(w.y, w.y, acc.y); acc.z = fmaf(w.z, w.z, acc.z); acc.w = fmaf(w.w, w.w, acc.w);
        } else {
          acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
        }
      }
    }
    for (; r < r1; r += RG) {
      float4 w = *reinterpret_cast<const float4*>(xp + r * ldx);
      if (relu_in) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
      if (centred) {
        w.x -= c4.x; w.y -= c4.y; w.z -= c4.z; w.w -= c4.w;
        acc.x = fmaf(w.x, w.x, acc.x); acc.y = fmaf(w.y, w.y, acc.y); acc.z = fmaf(w.z, w.z, acc.z); acc.w = fmaf(w.w, w.w, acc.w);
      } else {
        acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
      }
    }
  }
  red[t] = acc;
  __syncthreads();
  if (t < Q) {                                   // fold the row groups in a fixed order
    float4 s = red[t];
    for (int g = 1; g < RG; ++g) { const float4 o = red[g * Q + t]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *reinterpret_cast<float4*>(part + static_cast<int64_t>(blockIdx.x) * d + 4 * t) = s;
  }
}

// Both raw moments in ONE read: per-slice sums of f(x) and f(x)^2 accumulated in fp64 (the kernel is a streaming read; the fp64
// adds ride under it), so that var = E[f^2] - mean^2 is taken in double precision by the caller -- cancellation costs
// (mean^2 / var) * 2^-53 there, nothing at any realistic scale.  part: f64 [slice][2][d].
__global__ __launch_bounds__(kBlock) void col_moments2_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int d,
                                                              int relu_in, double* __restrict__ part, int64_t rows_per_slice) {
  __shared__ double red[2][kBlock][4];
  const int Q = d >> 2;
  const int RG = kBlock / Q;
  const int t = threadIdx.x;
  const int q = t % Q, rg = t / Q;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_slice;
  const int64_t r1 = min(r0 + rows_per_slice, n);
  double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
  if (rg < RG) {
    const float* xp = x + 4 * q;
    int64_t r = r0 + rg;
    for (; r + 3 * RG < r1; r += 4 * RG) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xp + (r + u * RG) * ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double a = static_cast<double>(relu_in ? fmaxf(w[k], 0.f) : w[k]);
          s1[k] += a; s2[k] = fma(a, a, s2[k]);
        }
      }
    }
    for (; r < r1; r += RG) {
      const float4 v = *reinterpret_cast<const float4*>(xp + r * ldx);
      float w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double a = static_cast<double>(relu_in ? fmaxf(w[k], 0.f) : w[k]);
        s1[k] += a; s2[k] = fma(a, a, s2[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][t][k] = s1[k]; red[1][t][k] = s2[k]; }
  __syncthreads();
  if (t < Q) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      double a[4] = {red[m][t][0], red[m][t][1], red[m][t][2], red[m][t][3]};
      for (int g = 1; g < RG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += red[m][g * Q + t][k];
      double* o = part + (static_cast<int64_t>(blockIdx.x) * 2 + m) * d + 4 * t;
      o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    }
  }
}

__global__ __launch_bounds__(kBlock) void col_affine_add_kernel(float* __restrict__ gx, int64_t ldgx, const float* __restrict__ x,
                                                                int64_t ldx, const float* __restrict__ s,
                                                                const float* __restrict__ tt, int relu_mask, int64_t n, int d) {
  const int Q = d >> 2;
  const int64_t total = n * Q;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / Q;
    const int q = static_cast<int>(i - r * Q);
    const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + 4 * q);
    float4 g = *reinterpret_cast<const float4*>(gx + r * ldgx + 4 * q);
    const float4 sv = *reinterpret_cast<const float4*>(s + 4 * q);
    const float4 tv = *reinterpret_cast<const float4*>(tt + 4 * q);
    // relu_mask: the statistics were taken of relu(x); the term reaches x only where x > 0 (and relu(x) = x there)
    g.x += (!relu_mask || xv.x > 0.f) ? fmaf(xv.x, sv.x, tv.x) : 0.f;
    g.y += (!relu_mask || xv.y > 0.f) ? fmaf(xv.y, sv.y, tv.y) : 0.f;
    g.z += (!relu_mask || xv.z > 0.f) ? fmaf(xv.z, sv.z, tv.z) : 0.f;
    g.w += (!relu_mask || xv.w > 0.f) ? fmaf(xv.w, sv.w, tv.w) : 0.f;
    *reinterpret_cast<float4*>(gx + r * ldgx + 4 * q) = g;
  }
}

}  // namespace allset

using namespace allset;

extern "C" int allset_col_moments_supported(int64_t d) { return (d >= 4 && d <= 1024 && d % 4 == 0) ? 1 : 0; }

extern "C" int allset_col_moments_slices(int64_t n, int64_t* n_slices) {
  clear_error();
  ALLSET_REQUIRE(n_slices != nullptr && n >= 0, "col_moments_slices: bad argument");
  int64_t s = (n + kCmRowsPerSlice - 1) / kCmRowsPerSlice;
  if (s > kCmMaxSlices) s = kCmMaxSlices;
  *n_slices = s < 1 ? 1 : s;
  return ALLSET_OK;
}

extern "C" int allset_col_moments(const float* x, int64_t ldx, int64_t n, int64_t d, int relu_in, const float* center,
                                  float* part, int64_t n_slices, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "col_moments: negative size");
  if (!allset_col_moments_supported(d)) {
    set_error("col_moments: width %lld not built (4 <= d <= 1024, d %% 4 == 0)", static_cast<long long>(d));
    return ALLSET_ERR_UNSUPPORTED;
  }
  int64_t want = 0;
  allset_col_moments_slices(n, &want);
  ALLSET_REQUIRE(part != nullptr && n_slices == want, "col_moments: part must hold allset_col_moments_slices(n) rows of d floats");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(part, 0, static_cast<size_t>(n_slices) * d * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(x != nullptr && ldx >= d && ldx % 4 == 0 && aligned16(x) && aligned16(part) && (center == nullptr || aligned16(center)),
                 "col_moments: x rows, part and center must be 16-byte aligned");
  const int64_t rows_per_slice = (n + n_slices - 1) / n_slices;
  col_moments_kernel<<<static_cast<unsigned>(n_slices), kBlock, 0, st>>>(x, ldx, n, static_cast<int>(d), relu_in, center, part,
                                                                         rows_per_slice);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_col_moments2(const float* x, int64_t ldx, int64_t n, int64_t d, int relu_in, double* part, int64_t n_slices,
                                   void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "col_moments2: negative size");
  if (!allset_col_moments_supported(d)) {
    set_error("col_moments2: width %lld not built (4 <= d <= 1024, d %% 4 == 0)", static_cast<long long>(d));
    return ALLSET_ERR_UNSUPPORTED;
  }
  int64_t want = 0;
  allset_col_moments_slices(n, &want);
  ALLSET_REQUIRE(part != nullptr && n_slices == want, "col_moments2: part must hold allset_col_moments_slices(n) x 2 rows of d doubles");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(part, 0, static_cast<size_t>(n_slices) * 2 * d * sizeof(double), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(x != nullptr && ldx >= d && ldx % 4 == 0 && aligned16(x) && (reinterpret_cast<uintptr_t>(part) & 7u) == 0,
                 "col_moments2: x rows must be 16-byte aligned, part 8-byte aligned");
  const int64_t rows_per_slice = (n + n_slices - 1) / n_slices;
  col_moments2_kernel<<<static_cast<unsigned>(n_slices), kBlock, 0, st>>>(x, ldx, n, static_cast<int>(d), relu_in, part, rows_per_slice);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_col_affine_add(float* gx, int64_t ldgx, const float* x, int64_t ldx, const float* s, const float* t,
                                     int relu_mask, int64_t n, int64_t d, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "col_affine_add: negative size");
  if (!allset_col_moments_supported(d)) {
    set_error("col_affine_add: width %lld not built (4 <= d <= 1024, d %% 4 == 0)", static_cast<long long>(d));
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(gx && x && s && t, "col_affine_add: null pointer");
  ALLSET_REQUIRE(ldgx >= d && ldx >= d && ldgx % 4 == 0 && ldx % 4 == 0 && aligned16(gx) && aligned16(x) && aligned16(s) && aligned16(t),
                 "col_affine_add: rows and the column vectors must be 16-byte aligned");
  const int64_t total = n * (d / 4);
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  col_affine_add_kernel<<<static_cast<unsigned>(blocks), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      gx, ldgx, x, ldx, s, t, relu_mask, n, static_cast<int>(d));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
