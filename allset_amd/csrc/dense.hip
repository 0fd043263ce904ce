// Dense tail of the AllSet layer (reference MLP.forward, layers.py:571-579, and its autograd) for gfx950.
//
// The profile of round 1 (profiles/r01a_bench_kernel_stats.csv) showed the torch tail at [1M,128] fp32 spending
// ~1 ms per LayerNorm pass (1 GB of traffic -> ~1 TB/s) and 1.8 ms per weight-gradient GEMM
// (a [128 x 1M] x [1M x 128] product hipBLASLt tiles badly).  These kernels replace exactly those pieces:
//
//   ln_fwd_kernel      y = dropout( LayerNorm( relu?(x) ) )            1 read + 1 write, row statistics saved
//   ln_bwd_kernel      gx, per-block partial (dgamma, dbeta)            2 reads + 1 write; mask regenerated
//   relu_dropout_*     y = dropout(relu(x)) and its backward            elementwise, 16 B per lane
//   wgrad_x6_kernel    gW = ga^T @ u, gb = colsum(ga)                   split-K over rows on the bf16 matrix pipe (bf16x6,
//                                                                       fp32-accurate), per-slice partials (deterministic)
//
// The LayerNorm/elementwise kernels are HBM-bound streaming kernels; wgrad is the only MFMA user.  Dropout masks come from a counter-based
// hash of (seed, element index), so the backward regenerates them instead of storing a mask tensor.
#include <stdlib.h>

#include "common.h"

namespace allset {

template <int W>
__device__ __forceinline__ float group_sum(float v) {   // sum over aligned groups of W lanes (W power of two)
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int kLnRowsPerGroup = 4;   // rows in flight per lane group (independent loads)

// y = dropout(LN(relu?(x))).  LPR lanes x 16 B cover a row (d <= 4*LPR, d % 4 == 0); NS = 64/LPR rows per wave.
template <int LPR>
__global__ __launch_bounds__(kBlock) void ln_fwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, int relu_in, float p, uint64_t seed, float* __restrict__ y, int64_t ldy,
    float* __restrict__ stats, int64_t n, int d, const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  constexpr int NS = kWave / LPR;
  const int lane = lane_id();
  const int grp = (threadIdx.x >> 6) * NS + lane / LPR;            // row group inside the block
  const int li = lane % LPR;
  const int c0 = li * 4;
  const bool active = c0 < d;
  constexpr int kGroups = kWavesPerBlock * NS;
  const float inv_d = 1.f / static_cast<float>(d);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  float4 g4 = make_float4(0, 0, 0, 0), b4 = make_float4(0, 0, 0, 0);
  if (active) {
    g4 = *reinterpret_cast<const float4*>(gamma + c0);
    b4 = *reinterpret_cast<const float4*>(beta + c0);
  }
  const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * kGroups + grp) * kLnRowsPerGroup;
  float4 v[kLnRowsPerGroup];
#pragma unroll
  for (int r = 0; r < kLnRowsPerGroup; ++r) {
    const int64_t row = row0 + r;
    v[r] = make_float4(0, 0, 0, 0);
    if (active && row < n) v[r] = *reinterpret_cast<const float4*>(x + row * ldx + c0);
  }
#pragma unroll
  for (int r = 0; r < kLnRowsPerGroup; ++r) {
    const int64_t row = row0 + r;
    float4 t = v[r];
    if (relu_in) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
    const float mean = group_sum<LPR>(t.x + t.y + t.z + t.w) * inv_d;
    float4 c = make_float4(t.x - mean, t.y - mean, t.z - mean, t.w - mean);
    if (!active) c = make_float4(0, 0, 0, 0);
    const float var = group_sum<LPR>(c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) * inv_d;
    const float rstd = rsqrtf(var + eps);
    if (row < n) {
      if (active) {
        float4 o = make_float4(c.x * rstd * g4.x + b4.x, c.y * rstd * g4.y + b4.y, c.z * rstd * g4.z + b4.z,
                               c.w * rstd * g4.w + b4.w);
        if (p > 0.f) {
          const int64_t e = row * d + c0;
          float k0, k1, k2, k3;
          keep_scale2(seed, e, thr, inv_keep, k0, k1); keep_scale2(seed, e + 2, thr, inv_keep, k2, k3);
          o.x *= k0; o.y *= k1; o.z *= k2; o.w *= k3;
        }
        *reinterpret_cast<float4*>(y + row * ldy + c0) = o;
      }
      if (li == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    }
  }
}

// Widths the 16-byte kernels do not take (d > 256 or d % 4 != 0, e.g. 1433 raw features) up to 64 * NS columns: one wave
// per row, the whole row in NS registers per lane -- every load of the row is issued before the first is consumed, the
// statistics come from the registers, one pass over memory (the scalar kernel below makes three dependent passes).
template <int NS>
__global__ __launch_bounds__(kBlock) void ln_fwd_rows_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, int relu_in, float p, uint64_t seed, float* __restrict__ y, int64_t ldy,
    float* __restrict__ stats, int64_t n, int d, const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = lane_id();
  const float* xr = x + row * ldx;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  float v[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int c = k * kWave + lane;
    v[k] = c < d ? xr[c] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NS; ++k) { if (relu_in) v[k] = fmaxf(v[k], 0.f); s += v[k]; }
  const float mean = group_sum<kWave>(s) / static_cast<float>(d);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int c = k * kWave + lane;
    v[k] = c < d ? v[k] - mean : 0.f;
    q = fmaf(v[k], v[k], q);
  }
  const float rstd = rsqrtf(group_sum<kWave>(q) / static_cast<float>(d) + eps);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int c = k * kWave + lane;
    if (c < d) {
      float o = fmaf(v[k] * rstd, gamma[c], beta[c]);
      if (p > 0.f) o *= keep_scale(seed, row * d + c, thr, inv_keep);
      y[row * ldy + c] = o;
    }
  }
  if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}

// Its backward: persistent waves, one row per wave and trip, gy and x rows in registers, dgamma / dbeta accumulated per lane
// over the wave's rows; the block's waves are combined through LDS in a fixed order (deterministic, no atomics).
template <int NS>
__global__ __launch_bounds__(kBlock) void ln_bwd_rows_kernel(
    const float* __restrict__ gy, int64_t ldg, const float* __restrict__ x, int64_t ldx,
    const float* __restrict__ stats, const float* __restrict__ gamma, int relu_in, float p, uint64_t seed,
    float* __restrict__ gx, int64_t ldgx, float* __restrict__ part, int64_t n, int d, const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  const float inv_d = 1.f / static_cast<float>(d);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  __shared__ float red[kWavesPerBlock][2][NS * kWave];
  float dg[NS], db[NS], gm[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int c = k * kWave + lane;
    dg[k] = 0.f; db[k] = 0.f;
    gm[k] = c < d ? gamma[c] : 0.f;
  }
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave; row < n; row += stride) {
    const float* xr = x + row * ldx;
    const float* gr = gy + row * ldg;
    float xv[NS], gv[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int c = k * kWave + lane;
      xv[k] = c < d ? xr[c] : 0.f;
      gv[k] = c < d ? gr[c] : 0.f;
    }
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int c = k * kWave + lane;
      const bool pos = xv[k] > 0.f;
      const float t = relu_in ? fmaxf(xv[k], 0.f) : xv[k];
      const float xh = c < d ? (t - mean) * rstd : 0.f;
      if (p > 0.f && c < d) gv[k] *= keep_scale(seed, row * d + c, thr, inv_keep);
      dg[k] = fmaf(gv[k], xh, dg[k]);
      db[k] += gv[k];
      const float gh = gv[k] * gm[k];
      s1 += gh;
      s2 = fmaf(gh, xh, s2);
      gv[k] = gh;
      xv[k] = (relu_in && !pos) ? __int_as_float(0x7fc00000) : xh;      // NaN marks "relu closed": gx = 0 there
    }
    if (gx != nullptr) {
      s1 = group_sum<kWave>(s1) * inv_d;
      s2 = group_sum<kWave>(s2) * inv_d;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int c = k * kWave + lane;
        if (c < d) gx[row * ldgx + c] = (xv[k] != xv[k]) ? 0.f : rstd * (gv[k] - s1 - xv[k] * s2);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) { red[wave][0][k * kWave + lane] = dg[k]; red[wave][1][k * kWave + lane] = db[k]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * NS * kWave; t += kBlock) {
    const int which = t / (NS * kWave), c = t % (NS * kWave);
    if (c < d) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < kWavesPerBlock; ++w) sum += red[w][which][c];
      part[(static_cast<int64_t>(blockIdx.x) * 2 + which) * d + c] = sum;
    }
  }
}

// generic width: one wave per row, three passes over the row (L1/L2 resident), scalar accesses
__global__ __launch_bounds__(kBlock) void ln_fwd_generic_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, int relu_in, float p, uint64_t seed, float* __restrict__ y, int64_t ldy,
    float* __restrict__ stats, int64_t n, int d, const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = lane_id();
  const float* xr = x + row * ldx;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  float s = 0.f;
  for (int c = lane; c < d; c += kWave) s += relu_in ? fmaxf(xr[c], 0.f) : xr[c];
  const float mean = group_sum<kWave>(s) / static_cast<float>(d);
  float q = 0.f;
  for (int c = lane; c < d; c += kWave) {
    const float t = (relu_in ? fmaxf(xr[c], 0.f) : xr[c]) - mean;
    q += t * t;
  }
  const float rstd = rsqrtf(group_sum<kWave>(q) / static_cast<float>(d) + eps);
  for (int c = lane; c < d; c += kWave) {
    float o = ((relu_in ? fmaxf(xr[c], 0.f) : xr[c]) - mean) * rstd * gamma[c] + beta[c];
    if (p > 0.f) o *= keep_scale(seed, row * d + c, thr, inv_keep);
    y[row * ldy + c] = o;
  }
  if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}

// Backward of y = dropout(LN(relu?(x))).  Persistent grid-stride over rows; each lane keeps dgamma/dbeta
// partials for its 4 columns, reduced over the block's row groups through LDS, one partial row per block:
// part[blockIdx][0][c] = dgamma, part[blockIdx][1][c] = dbeta.
template <int LPR>
__global__ __launch_bounds__(kBlock) void ln_bwd_kernel(
    const float* __restrict__ gy, int64_t ldg, const float* __restrict__ x, int64_t ldx,
    const float* __restrict__ stats, const float* __restrict__ gamma, int relu_in, float p, uint64_t seed,
    float* __restrict__ gx, int64_t ldgx, float* __restrict__ part, int64_t n, int d, const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  constexpr int NS = kWave / LPR;
  constexpr int kGroups = kWavesPerBlock * NS;
  __shared__ float red[kGroups][2][LPR * 4];
  const int lane = lane_id();
  const int grp = (threadIdx.x >> 6) * NS + lane / LPR;
  const int li = lane % LPR;
  const int c0 = li * 4;
  const bool active = c0 < d;
  const float inv_d = 1.f / static_cast<float>(d);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  float4 g4 = make_float4(0, 0, 0, 0);
  if (active) g4 = *reinterpret_cast<const float4*>(gamma + c0);
  float4 dg = make_float4(0, 0, 0, 0), db = make_float4(0, 0, 0, 0);

  const int64_t rows_per_iter = static_cast<int64_t>(gridDim.x) * kGroups;
  // lanes of one LPR-lane group share `row`, and the shuffles below never leave the group, so groups of a
  // wave may run different trip counts
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * kGroups + grp; row < n; row += rows_per_iter) {
    const bool live = true;
    float4 xv = make_float4(0, 0, 0, 0), gv = make_float4(0, 0, 0, 0);
    float mean = 0.f, rstd = 0.f;
    if (live && active) {
      xv = *reinterpret_cast<const float4*>(x + row * ldx + c0);
      gv = *reinterpret_cast<const float4*>(gy + row * ldg + c0);
    }
    if (live) { mean = stats[row * 2]; rstd = stats[row * 2 + 1]; }
    float4 t = xv;
    if (relu_in) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
    float4 xh = make_float4((t.x - mean) * rstd, (t.y - mean) * rstd, (t.z - mean) * rstd, (t.w - mean) * rstd);
    if (!(live && active)) xh = make_float4(0, 0, 0, 0);
    if (p > 0.f && live && active) {
      const int64_t e = row * d + c0;
      float k0, k1, k2, k3;
      keep_scale2(seed, e, thr, inv_keep, k0, k1); keep_scale2(seed, e + 2, thr, inv_keep, k2, k3);
      gv.x *= k0; gv.y *= k1; gv.z *= k2; gv.w *= k3;
    }
    dg.x += gv.x * xh.x; dg.y += gv.y * xh.y; dg.z += gv.z * xh.z; dg.w += gv.w * xh.w;
    db.x += gv.x; db.y += gv.y; db.z += gv.z; db.w += gv.w;
    const float4 gh = make_float4(gv.x * g4.x, gv.y * g4.y, gv.z * g4.z, gv.w * g4.w);
    const float s1 = group_sum<LPR>(gh.x + gh.y + gh.z + gh.w) * inv_d;
    const float s2 = group_sum<LPR>(gh.x * xh.x + gh.y * xh.y + gh.z * xh.z + gh.w * xh.w) * inv_d;
    if (live && active) {
      float4 o = make_float4(rstd * (gh.x - s1 - xh.x * s2), rstd * (gh.y - s1 - xh.y * s2),
                             rstd * (gh.z - s1 - xh.z * s2), rstd * (gh.w - s1 - xh.w * s2));
      if (relu_in) {
        o.x = xv.x > 0.f ? o.x : 0.f; o.y = xv.y > 0.f ? o.y : 0.f;
        o.z = xv.z > 0.f ? o.z : 0.f; o.w = xv.w > 0.f ? o.w : 0.f;
      }
      if (gx != nullptr) *reinterpret_cast<float4*>(gx + row * ldgx + c0) = o;
    }
  }
  *reinterpret_cast<float4*>(&red[grp][0][c0]) = dg;
  *reinterpret_cast<float4*>(&red[grp][1][c0]) = db;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * LPR * 4; i += kBlock) {
    const int which = i / (LPR * 4), c = i % (LPR * 4);
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) s += red[g][which][c];
    if (c < d) part[(static_cast<int64_t>(blockIdx.x) * 2 + which) * d + c] = s;
  }
}

// generic width backward: one wave per row per iteration; each lane owns 8 column slots per sweep and accumulates
// dgamma/dbeta over the block's rows in registers; the block's waves are combined through LDS in a fixed order and
// written to the block's own partial row (no atomics: the result is run-to-run deterministic).
__global__ __launch_bounds__(kBlock) void ln_bwd_generic_kernel(
    const float* __restrict__ gy, int64_t ldg, const float* __restrict__ x, int64_t ldx,
    const float* __restrict__ stats, const float* __restrict__ gamma, int relu_in, float p, uint64_t seed,
    float* __restrict__ gx, int64_t ldgx, float* __restrict__ part, int64_t n, int d, const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  const int lane = lane_id();
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  __shared__ float red[kWavesPerBlock][2][8 * kWave];
  const int wave = threadIdx.x >> 6;
  // column c is always handled by lane c % 64 of some wave; accumulate per (wave, column-slot) over rows
  for (int cb = 0; cb < d; cb += kWave * 8) {       // up to 8 column slots per lane per sweep
    float dg[8], db[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { dg[k] = 0.f; db[k] = 0.f; }
    for (int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6); row < n; row += stride) {
      const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
      const float* xr = x + row * ldx;
      const float* gr = gy + row * ldg;
      if (cb == 0 && gx != nullptr) {   // first sweep also produces gx (needs the full-row sums); NULL: partials only
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < d; c += kWave) {
          const float t = relu_in ? fmaxf(xr[c], 0.f) : xr[c];
          const float xh = (t - mean) * rstd;
          const float gh = gr[c] * (p > 0.f ? keep_scale(seed, row * d + c, thr, inv_keep) : 1.f) * gamma[c];
          s1 += gh; s2 += gh * xh;
        }
        s1 = group_sum<kWave>(s1) / static_cast<float>(d);
        s2 = group_sum<kWave>(s2) / static_cast<float>(d);
        for (int c = lane; c < d; c += kWave) {
          const float t = relu_in ? fmaxf(xr[c], 0.f) : xr[c];
          const float xh = (t - mean) * rstd;
          const float gh = gr[c] * (p > 0.f ? keep_scale(seed, row * d + c, thr, inv_keep) : 1.f) * gamma[c];
          float o = rstd * (gh - s1 - xh * s2);
          if (relu_in && !(xr[c] > 0.f)) o = 0.f;
          gx[row * ldgx + c] = o;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = cb + k * kWave + lane;
        if (c < d) {
          const float t = relu_in ? fmaxf(xr[c], 0.f) : xr[c];
          const float g = gr[c] * (p > 0.f ? keep_scale(seed, row * d + c, thr, inv_keep) : 1.f);
          dg[k] += g * (t - mean) * rstd;
          db[k] += g;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[wave][0][k * kWave + lane] = dg[k]; red[wave][1][k * kWave + lane] = db[k]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * 8 * kWave; t += kBlock) {
      const int which = t / (8 * kWave), slot = t % (8 * kWave);
      const int c = cb + slot;
      if (c < d) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) sum += red[w][which][slot];
        part[(static_cast<int64_t>(blockIdx.x) * 2 + which) * d + c] = sum;
      }
    }
    __syncthreads();
  }
}

// y = dropout(relu(x)); backward gx = gy * inv_keep where y > 0 (y > 0 <=> kept and x > 0), else 0
__global__ __launch_bounds__(kBlock) void relu_dropout_fwd_kernel(const float* __restrict__ x, float p, uint64_t seed,
                                                                  float* __restrict__ y, int64_t n4,
                                                                  const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    if (p > 0.f) {
      const int64_t e = i * 4;
      float k0, k1, k2, k3;
      keep_scale2(seed, e, thr, inv_keep, k0, k1); keep_scale2(seed, e + 2, thr, inv_keep, k2, k3);
      v.x *= k0; v.y *= k1; v.z *= k2; v.w *= k3;
    }
    reinterpret_cast<float4*>(y)[i] = v;
  }
}

__global__ __launch_bounds__(kBlock) void relu_dropout_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                  float p, float* __restrict__ gx, int64_t n4) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const float4 g = reinterpret_cast<const float4*>(gy)[i];
    const float4 o = reinterpret_cast<const float4*>(y)[i];
    float4 r;
    r.x = o.x > 0.f ? g.x * inv_keep : 0.f; r.y = o.y > 0.f ? g.y * inv_keep : 0.f;
    r.z = o.z > 0.f ? g.z * inv_keep : 0.f; r.w = o.w > 0.f ? g.w * inv_keep : 0.f;
    reinterpret_cast<float4*>(gx)[i] = r;
  }
}

// scalar tail / unaligned variant (n elements)
__global__ void relu_dropout_fwd_scalar_kernel(const float* __restrict__ x, float p, uint64_t seed, float* __restrict__ y,
                                               int64_t n, int64_t offset, const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  const int64_t i = offset + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) {
    float v = fmaxf(x[i], 0.f);
    if (p > 0.f) v *= keep_scale(seed, i, thr, inv_keep);
    y[i] = v;
  }
}

__global__ void relu_dropout_bwd_scalar_kernel(const float* __restrict__ gy, const float* __restrict__ y, float p,
                                               float* __restrict__ gx, int64_t n, int64_t offset) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  const int64_t i = offset + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) gx[i] = y[i] > 0.f ? gy[i] * inv_keep : 0.f;
}

// ---- weight gradient: gW[o][i] = sum_r ga[r][o] * u[r][i],  gb[o] = sum_r ga[r][o] -----------------------
// Weight gradient.  Grid: x = 128x128 output macro-tile (o-block * tiles_i + i-block), y = split-K slice of the rows;
// K (= rows) is consumed 32 rows per stage through LDS (wgrad_x6_kernel below).
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kWgTile = 128;
constexpr int kWgRows = 32;

// Optional operand prologues (PRO = true): the two operands are recomputed on the fly from what the fused forward
// (csrc/fused_mlp.hip) keeps, instead of being read from materialised tensors:
//   A side: ga = gy * (y > 0 ? keep_out : 0)                       if y != nullptr   (relu/dropout epilogue)
//   B side: u  = dropout_in( LN_{stats,gamma,beta}( relu_in(x) ) )   from x and the saved row statistics
struct WgradPro {
  const float* y; int64_t ldy; float keep_out;
  const float* stats; const float* gamma; const float* beta; int has_ln; int relu_in; float p_in; uint64_t seed_in;
  const uint64_t* seed_base;
  const uint32_t* mask; int mask_nh;        // activation mask (replaces y; bf16x6 kernels only), O / 64
  int64_t pw_stride = 0, pb_stride = 0;     // floats between consecutive slices of part_w / part_b (0: O*I and O)
  int n_slices = 0;                         // > 0: the grid's y extent is rounded up to a multiple of 8 and remapped (wgrad_x6_kernel)
};

// out[s][c] = sum of part[p][c] over the s-th slab of kRedRows rows (p < P, c < M): one level of the tree that
// finishes every split-K / per-block partial scheme above.  2-D grid (column quads x row slabs), 8 row groups per
// workgroup with 8 independent loads each, combined through LDS.  Two levels reduce 512 x 16K partials (32 MB).
constexpr int kRedSplit = 8;                   // row groups per workgroup (256 threads = 32 column quads x 8 groups)
constexpr int kRedRows = 64;                   // rows per slab

// `slabs_here` > 1: one workgroup walks that many slabs itself (small reductions: one launch instead of two levels).
// (`bx`, `by`: the workgroup's column-quad block and first slab -- blockIdx of the plain kernel, table-relative in the batched one.)
template <bool OUT_BF16>
__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ part, int64_t P, int64_t M, float* __restrict__ out,
                                                     int64_t row_stride, int slabs_here, int64_t bx, int64_t by,
                                                     float4 (*red)[kBlock / kRedSplit], const float* __restrict__ acc_in = nullptr) {
  const int cq = threadIdx.x % (kBlock / kRedSplit), rg = threadIdx.x / (kBlock / kRedSplit);
  const int64_t c = (bx * (kBlock / kRedSplit) + cq) * 4;
  float4 acc = make_float4(0, 0, 0, 0);
  if (c < M) {
    for (int sl = 0; sl < slabs_here; ++sl) {
      const int64_t p0 = (by + sl) * kRedRows;
      float4 v[kRedRows / kRedSplit];
#pragma unroll
      for (int i = 0; i < kRedRows / kRedSplit; ++i) {
        const int64_t p = p0 + rg + static_cast<int64_t>(i) * kRedSplit;
        v[i] = p < P ? *reinterpret_cast<const float4*>(part + p * row_stride + c) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < kRedRows / kRedSplit; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
  }
  red[rg][cq] = acc;
  __syncthreads();
  if (rg == 0 && c < M) {
#pragma unroll
    for (int g = 1; g < kRedSplit; ++g) { const float4 t = red[g][cq]; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
    if (acc_in != nullptr) { const float4 a = *reinterpret_cast<const float4*>(acc_in + c); acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
    if constexpr (OUT_BF16)      // single-slab launches only: `out` is a bf16 vector of M elements
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + c) = make_uint2(cvt_pk_bf16(acc.x, acc.y), cvt_pk_bf16(acc.z, acc.w));
    else
      *reinterpret_cast<float4*>(out + by * M + c) = acc;
  }
}

// The TWO-LEVEL reduction of allset_reduce_partials (one launch that leaves a sum per 64-row slab, a second one that sums the
// <= 8 slab sums) done by ONE workgroup per column block, with exactly the tree's association -- per slab: eight row groups of
// eight sequential adds, the groups combined in order; then the slab sums added in order to +0 -- so the result is bit-identical
// to the two launches (ABI 14: what lets the batched entry take the large partial buffers of the [1M, 128] / [250k, 256] steps).
template <bool OUT_BF16>
__device__ __forceinline__ void reduce_partials_tree_body(const float* __restrict__ part, int64_t P, int64_t M, float* __restrict__ out,
                                                          int64_t row_stride, int slabs, int64_t bx, float4 (*red)[kBlock / kRedSplit],
                                                          const float* __restrict__ acc_in = nullptr) {
  const int cq = threadIdx.x % (kBlock / kRedSplit), rg = threadIdx.x / (kBlock / kRedSplit);
  const int64_t c = (bx * (kBlock / kRedSplit) + cq) * 4;
  float4 tot = make_float4(0, 0, 0, 0);
  for (int sl = 0; sl < slabs; ++sl) {
    float4 acc = make_float4(0, 0, 0, 0);
    if (c < M) {
      const int64_t p0 = static_cast<int64_t>(sl) * kRedRows;
      float4 v[kRedRows / kRedSplit];
#pragma unroll
      for (int i = 0; i < kRedRows / kRedSplit; ++i) {
        const int64_t p = p0 + rg + static_cast<int64_t>(i) * kRedSplit;
        v[i] = p < P ? *reinterpret_cast<const float4*>(part + p * row_stride + c) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < kRedRows / kRedSplit; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    red[rg][cq] = acc;
    __syncthreads();
    if (rg == 0 && c < M) {
#pragma unroll
      for (int g = 1; g < kRedSplit; ++g) { const float4 t = red[g][cq]; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
      tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w;
    }
    __syncthreads();
  }
  if (rg == 0 && c < M) {
    if (acc_in != nullptr) { const float4 a = *reinterpret_cast<const float4*>(acc_in + c); tot.x += a.x; tot.y += a.y; tot.z += a.z; tot.w += a.w; }
    if constexpr (OUT_BF16)
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + c) = make_uint2(cvt_pk_bf16(tot.x, tot.y), cvt_pk_bf16(tot.z, tot.w));
    else
      *reinterpret_cast<float4*>(out + c) = tot;
  }
}

template <bool OUT_BF16>
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const float* __restrict__ part, int64_t P, int64_t M,
                                                                float* __restrict__ out, int64_t row_stride, int slabs_here) {
  __shared__ float4 red[kRedSplit][kBlock / kRedSplit];
  reduce_partials_body<OUT_BF16>(part, P, M, out, row_stride, slabs_here, blockIdx.x, blockIdx.y, red);
}

// MANY small reductions in one launch (dataset-scale training steps are chains of ~5 us kernels, a third of them these): the
// table names up to kRedBatchMax partial buffers; every workgroup finds its buffer by a scan of the table's block offsets and then
// does exactly what the one-launch form of the kernel above does for it -- the sums are bit-identical to separate calls.
constexpr int kRedBatchMax = 48;
constexpr int kRedBatchMaxCounters = 64;
struct RedBatchTable {
  const float* part[kRedBatchMax];
  float* out[kRedBatchMax];
  int32_t P[kRedBatchMax], stride[kRedBatchMax], M[kRedBatchMax];
  int32_t first_block[kRedBatchMax + 1];
  const float* acc[kRedBatchMax];            // ABI 14: added to the sum of buffer k before it is written (an existing gradient; fp32 outputs), or NULL
  uint8_t out_bf16[kRedBatchMax];            // ABI 14: the sum of buffer k leaves as bf16 (rounded once), as allset_reduce_partials_ex writes it
  uint8_t tree[kRedBatchMax];                // ABI 14: buffer k is one the single call reduces in TWO launches (the tree body below)
  int32_t count;
  // counters advanced by one extra workgroup of the same launch (a training step's dropout-seed counter and its optimizer's step
  // counters: each is one more ~5 us launch in the step's dependent chain otherwise)
  int64_t* inc_i64;
  float* inc_f32[kRedBatchMaxCounters];
  int32_t n_inc_f32;
};
__global__ __launch_bounds__(kBlock) void reduce_partials_batched_kernel(RedBatchTable tb) {
  __shared__ float4 red[kRedSplit][kBlock / kRedSplit];
  const int b = blockIdx.x;
  if (b >= tb.first_block[tb.count]) {                       // the counters' workgroup
    if (static_cast<int>(threadIdx.x) < tb.n_inc_f32) tb.inc_f32[threadIdx.x][0] += 1.f;
    if (threadIdx.x == 0 && tb.inc_i64 != nullptr) tb.inc_i64[0] += 1;
    return;
  }
  int t = 0;
  while (t + 1 < tb.count && tb.first_block[t + 1] <= b) ++t;
  const int P = tb.P[t];
  const int slabs = (P + kRedRows - 1) / kRedRows;
  if (tb.tree[t]) {
    if (tb.out_bf16[t]) reduce_partials_tree_body<true>(tb.part[t], P, tb.M[t], tb.out[t], tb.stride[t], slabs, b - tb.first_block[t], red);
    else reduce_partials_tree_body<false>(tb.part[t], P, tb.M[t], tb.out[t], tb.stride[t], slabs, b - tb.first_block[t], red, tb.acc[t]);
  } else if (tb.out_bf16[t]) {
    reduce_partials_body<true>(tb.part[t], P, tb.M[t], tb.out[t], tb.stride[t], slabs, b - tb.first_block[t], 0, red);
  } else {
    reduce_partials_body<false>(tb.part[t], P, tb.M[t], tb.out[t], tb.stride[t], slabs, b - tb.first_block[t], 0, red, tb.acc[t]);
  }
}

static inline int ln_lpr(int64_t d) {
  int lpr = 8;
  while (lpr * 4 < d && lpr < 64) lpr <<= 1;
  return lpr;
}


// ---- the same weight gradient on the bf16 matrix pipe (bf16x6, see common.h) ------------------------------------------
// gW[o][i] = sum_r ga[r][o] * u[r][i]: the reduction runs over ROWS, so an MFMA operand fragment is 8 consecutive rows of
// one feature -- the transpose of how the tensors lie in memory.  The transpose happens on the way into LDS: a thread
// loads the same 4 columns of two ADJACENT rows, applies the operand prologue, splits each (row r, row r+1) value pair
// into three packed-bf16 dwords and writes them to plane[feature][row pair] (16 dwords = 32 rows per feature and stage,
// the 16-byte piece index XOR-swizzled so both the dword writes and the ds_read_b128 fragment reads are conflict-free).
// 8 waves per workgroup, wave tile 64 (o) x 32 (i) = 8 accumulators of 16x16; one v_mfma_f32_16x16x32_bf16 k-step per
// 32-row stage: 18 fragment reads feed 48 MFMAs.  Two LDS buffers: the next stage's global loads are in flight under
// this stage's MFMAs, and one wave's conversion overlaps the other wave's MFMAs on the same SIMD.
constexpr int kWx6Block = 512;
using bf16x8_t = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4_t = __attribute__((ext_vector_type(4))) float;
union WFrag { uint4 u; bf16x8_t v; };

__device__ __forceinline__ int wx6_swz(int feat) { return (((feat >> 2) & 3) >> 1) * 3; }

// MK: the source of the forward's "output > 0" test on ga (0 none, 1 its fp32 output y, 2 its 1-bit activation mask); LN: the
// LayerNorm-apply on u.  Template parameters since round 5: as run-time flags they put a branch around every optional load of the
// staging path, and a branch around a memory instruction makes hipcc count the loads of the longest path at every later wait.
template <bool PRO, int MK = 0, bool LN = false>
__global__ __launch_bounds__(kWx6Block) void wgrad_x6_kernel(
    const float* __restrict__ ga, int64_t lda, const float* __restrict__ u, int64_t ldu,
    float* __restrict__ part_w, float* __restrict__ part_b, int64_t n, int O, int I, int tiles_i,
    int64_t rows_per_slice, WgradPro pro) {
  __shared__ __attribute__((aligned(16))) uint32_t sP[2][2][3][kWgTile * 16];     // [buffer][A|B][plane][feature*16 + ..]
  // Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest).  With several 128 x 128 tiles per slice every
  // half-row of ga / u is read by tiles_i (tiles_o) workgroups: in launch order those sit on DIFFERENT XCDs and each read goes to
  // the fabric (2x the algorithmic bytes at 256 x 256, 4x at 512 x 512).  Remapped, XCD k owns ALL tiles of slices k, k + 8, ...:
  // the co-resident tiles of a slice walk the same rows at the same time and the re-reads are hits in that XCD's L2.
  int tile = blockIdx.x, slice = blockIdx.y;
  if (pro.n_slices > 0) {
    const int T = gridDim.x;
    const int64_t L = static_cast<int64_t>(blockIdx.y) * T + blockIdx.x;
    const int rem = static_cast<int>(L % (8 * T));
    slice = static_cast<int>(L / (8 * T)) * 8 + (rem & 7);
    tile = rem >> 3;
    if (slice >= pro.n_slices) return;
  }
  const int tile_o = tile / tiles_i, tile_i = tile % tiles_i;
  const int o_base = tile_o * kWgTile, i_base = tile_i * kWgTile;
  const int64_t r_begin = static_cast<int64_t>(slice) * rows_per_slice;
  const int64_t r_end = min(n, r_begin + rows_per_slice);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // staging map: row pair rp of the 32-row stage, 4 columns at s_col
  const int rp = lane & 15, s_col = (wave * 4 + (lane >> 4)) * 4;
  const bool a_ok = (o_base + s_col) < O, b_ok = (i_base + s_col) < I;           // O, I are multiples of 4
  const int a_col = a_ok ? o_base + s_col : 0, b_col = b_ok ? i_base + s_col : 0;
  const int w_off = s_col * 16 + 4 * ((rp >> 2) ^ wx6_swz(s_col)) + (rp & 3);     // + c*16 for column s_col + c
  float4 bsum = make_float4(0, 0, 0, 0);

  float4 g4 = make_float4(1, 1, 1, 1), be4 = make_float4(0, 0, 0, 0);
  float keep_in = 1.f;
  uint32_t thr_in = 0;
  if constexpr (PRO) {
    if (LN && b_ok) {
      g4 = *reinterpret_cast<const float4*>(pro.gamma + i_base + s_col);
      be4 = *reinterpret_cast<const float4*>(pro.beta + i_base + s_col);
    }
    keep_in = pro.p_in > 0.f ? 1.f / (1.f - pro.p_in) : 1.f;
    thr_in = drop_threshold(pro.p_in);
    pro.seed_in = resolve_seed(pro.seed_base, pro.seed_in);
  }
  constexpr bool has_mask = PRO && MK == 2;
  constexpr bool has_y = PRO && MK == 1;
  // mask word of (row r, columns a_col..+3): block (r/16, a_col/64), dword (r%16/4)*8 + (r%4)*2 + ((a_col%64)/32),
  // bits 8c + ((a_col%32)/4) for column a_col + c
  const int m_col = ((a_col / 64) * 32) + ((a_col % 64) / 32);
  const int m_bit = (a_col % 32) / 4;
  constexpr bool has_ln = PRO && LN;

  // Two register sets: the global loads run TWO stages ahead of the MFMAs (one stage of 32-48 KiB per CU in flight is
  // latency-bound at ~3 TB/s), the LDS conversion one stage ahead.
  struct Stage { float4 ra[2], rb[2], ry[2]; float2 rst[2]; uint32_t rm[2]; int64_t row0; };
  // issue only; unconditional loads on clamped rows / columns (no branches around memory instructions)
  auto load_stage = [&](Stage& sg, int64_t r0) {
    sg.row0 = r0 + 2 * rp;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t r = sg.row0 + h;
      r = r < r_end ? r : r_end - 1;
      sg.ra[h] = *reinterpret_cast<const float4*>(ga + r * lda + a_col);
      sg.rb[h] = *reinterpret_cast<const float4*>(u + r * ldu + b_col);
      if constexpr (PRO) {
        if (has_y) sg.ry[h] = *reinterpret_cast<const float4*>(pro.y + r * pro.ldy + a_col);
        if (has_mask) sg.rm[h] = pro.mask[(r >> 4) * (pro.mask_nh * 32) + m_col + ((r & 15) >> 2) * 8 + (r & 3) * 2];
        if (has_ln) sg.rst[h] = *reinterpret_cast<const float2*>(pro.stats + r * 2);
      }
    }
  };
  auto store_stage = [&](Stage& sg, int buf) {
    float4 va[2], vb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool in_range = (sg.row0 + h) < r_end;
      float4 a = sg.ra[h], t = sg.rb[h];
      if constexpr (PRO) {
        if (has_mask) {
          const uint32_t bits = sg.rm[h] >> m_bit;
          a.x = (bits & 1u) ? a.x * pro.keep_out : 0.f; a.y = (bits & 0x100u) ? a.y * pro.keep_out : 0.f;
          a.z = (bits & 0x10000u) ? a.z * pro.keep_out : 0.f; a.w = (bits & 0x1000000u) ? a.w * pro.keep_out : 0.f;
        } else if (has_y) {
          const float4 yv = sg.ry[h];
          a.x = yv.x > 0.f ? a.x * pro.keep_out : 0.f; a.y = yv.y > 0.f ? a.y * pro.keep_out : 0.f;
          a.z = yv.z > 0.f ? a.z * pro.keep_out : 0.f; a.w = yv.w > 0.f ? a.w * pro.keep_out : 0.f;
        }
        if (pro.relu_in) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
        if (has_ln) {
          const float2 st = sg.rst[h];
          t.x = fmaf((t.x - st.x) * st.y, g4.x, be4.x); t.y = fmaf((t.y - st.x) * st.y, g4.y, be4.y);
          t.z = fmaf((t.z - st.x) * st.y, g4.z, be4.z); t.w = fmaf((t.w - st.x) * st.y, g4.w, be4.w);
        }
        if (pro.p_in > 0.f) {
          float k0, k1, k2, k3;
          const int64_t e = (sg.row0 + h) * I + i_base + s_col;
          keep_scale2(pro.seed_in, e, thr_in, keep_in, k0, k1); keep_scale2(pro.seed_in, e + 2, thr_in, keep_in, k2, k3);
          t.x *= k0; t.y *= k1; t.z *= k2; t.w *= k3;
        }
      }
      if (!(in_range && a_ok)) a = make_float4(0, 0, 0, 0);
      if (!(in_range && b_ok)) t = make_float4(0, 0, 0, 0);
      va[h] = a; vb[h] = t;
      bsum.x += a.x; bsum.y += a.y; bsum.z += a.z; bsum.w += a.w;
    }
    uint32_t* pa = &sP[buf][0][0][w_off];
    uint32_t* pb = &sP[buf][1][0][w_off];
    constexpr int PL = kWgTile * 16;                 // dwords per plane
    uint32_t h0, m0, l0;
    split3_bf16(va[0].x, va[1].x, h0, m0, l0); pa[0] = h0; pa[PL] = m0; pa[2 * PL] = l0;
    split3_bf16(va[0].y, va[1].y, h0, m0, l0); pa[16] = h0; pa[PL + 16] = m0; pa[2 * PL + 16] = l0;
    split3_bf16(va[0].z, va[1].z, h0, m0, l0); pa[32] = h0; pa[PL + 32] = m0; pa[2 * PL + 32] = l0;
    split3_bf16(va[0].w, va[1].w, h0, m0, l0); pa[48] = h0; pa[PL + 48] = m0; pa[2 * PL + 48] = l0;
    split3_bf16(vb[0].x, vb[1].x, h0, m0, l0); pb[0] = h0; pb[PL] = m0; pb[2 * PL] = l0;
    split3_bf16(vb[0].y, vb[1].y, h0, m0, l0); pb[16] = h0; pb[PL + 16] = m0; pb[2 * PL + 16] = l0;
    split3_bf16(vb[0].z, vb[1].z, h0, m0, l0); pb[32] = h0; pb[PL + 32] = m0; pb[2 * PL + 32] = l0;
    split3_bf16(vb[0].w, vb[1].w, h0, m0, l0); pb[48] = h0; pb[PL + 48] = m0; pb[2 * PL + 48] = l0;
  };

  // MFMA side: wave tile 64 (o) x 32 (i); lane (j, g) reads piece g of feature row j of a 16-feature tile
  const int fj = lane & 15, fg = lane >> 4;
  const int ob = (wave >> 2) * 64, ib = (wave & 3) * 32;
  f32x4_t acc[4][2];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) { acc[ot][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[ot][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  int a_off[4], b_off[2];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) { const int f = ob + ot * 16 + fj; a_off[ot] = f * 16 + 4 * (fg ^ wx6_swz(f)); }
#pragma unroll
  for (int it = 0; it < 2; ++it) { const int f = ib + it * 16 + fj; b_off[it] = f * 16 + 4 * (fg ^ wx6_swz(f)); }
  bool o_live[4];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) o_live[ot] = (o_base + ob + ot * 16) < O;
  const bool i_live = (i_base + ib) < I;

  auto mfma_stage = [&](int buf) {
    if (i_live) {
      WFrag b[2][3];
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[it][pl].u = *reinterpret_cast<const uint4*>(&sP[buf][1][pl][b_off[it]]);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        if (!o_live[ot]) continue;
        WFrag a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[pl].u = *reinterpret_cast<const uint4*>(&sP[buf][0][pl][a_off[ot]]);
        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[0][0].v, acc[ot][0], 0, 0, 0);
        acc[ot][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[1][0].v, acc[ot][1], 0, 0, 0);
        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[0][2].v, acc[ot][0], 0, 0, 0);
        acc[ot][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[1][2].v, acc[ot][1], 0, 0, 0);
        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[0][1].v, acc[ot][0], 0, 0, 0);
        acc[ot][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[1][1].v, acc[ot][1], 0, 0, 0);
        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[0][0].v, acc[ot][0], 0, 0, 0);
        acc[ot][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[1][0].v, acc[ot][1], 0, 0, 0);
        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[0][1].v, acc[ot][0], 0, 0, 0);
        acc[ot][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[1][1].v, acc[ot][1], 0, 0, 0);
        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[0][0].v, acc[ot][0], 0, 0, 0);
        acc[ot][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[1][0].v, acc[ot][1], 0, 0, 0);
      }
    }
  };
  Stage s0, s1;
  if (r_begin < r_end) {                       // stage 0 -> LDS buffer 0; stages 1 and 2 on their way
    load_stage(s0, r_begin);
    load_stage(s1, r_begin + kWgRows);
    store_stage(s0, 0);
    load_stage(s0, r_begin + 2 * kWgRows);
  }
  __syncthreads();
  // stage k lives in LDS buffer k&1; register set s1 holds stage k+1, s0 stage k+2 (roles swap every iteration)
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * kWgRows) {
    mfma_stage(0);
    if (r0 + kWgRows < r_end) store_stage(s1, 1);
    load_stage(s1, r0 + 3 * kWgRows);
    __syncthreads();
    if (r0 + kWgRows < r_end) {
      mfma_stage(1);
      if (r0 + 2 * kWgRows < r_end) store_stage(s0, 0);
      load_stage(s0, r0 + 4 * kWgRows);
      __syncthreads();
    }
  }

  // epilogue: partial tile -> part_w[slice][O][I]; acc[ot][it][r] is (o = .. + 4*fg + r, i = .. + fj)
  float* pw = part_w + static_cast<int64_t>(slice) * (pro.pw_stride ? pro.pw_stride : static_cast<int64_t>(O) * I);
#pragma unroll
  for (int ot = 0; ot < 4; ++ot)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = i_base + ib + it * 16 + fj;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o_base + ob + ot * 16 + 4 * fg + r;
        if (o < O && i < I) pw[static_cast<int64_t>(o) * I + i] = acc[ot][it][r];
      }
    }
  // bias partial: the 16 row-pair lanes of a column quad fold by shuffles (only i-tile 0 writes)
  if (tile_i == 0 && part_b != nullptr) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      bsum.x += __shfl_xor(bsum.x, off); bsum.y += __shfl_xor(bsum.y, off);
      bsum.z += __shfl_xor(bsum.z, off); bsum.w += __shfl_xor(bsum.w, off);
    }
    if (rp == 0 && a_ok) *reinterpret_cast<float4*>(part_b + static_cast<int64_t>(slice) * (pro.pb_stride ? pro.pb_stride : O) + o_base + s_col) = bsum;
  }
}


// ---- LayerNorm with a fused sum in front and a relu behind: the PMA tail (reference layers.py:153-157) --------------------
//   y = dropout_p( relu_out?( LN_{gamma,beta}( x + colb + res ) ) )     colb [d] and res [n,d] optional
// serves `ln0(pooled + att_r)` (colb = the seed vector), `ln1(out + relu(rFF(out)))` (res) and the relu -> dropout that
// SetGNN puts behind every conv, each in ONE read-write pass instead of a torch add + LayerNorm (+ relu/dropout).
// Backward: gs = d loss / d (x + colb + res) -- the same tensor is the gradient of x and of res, its column sums the
// gradient of colb (third partial row); the relu mask is recomputed from the statistics (no y kept).
// VPL = 16-byte chunks per lane: 1 for d <= 256 (LPR lanes x 4 columns cover a row), 2 for 256 < d <= 512 (LPR = 64: lane li owns
// columns 4 li .. and 256 + 4 li ..; round 6, second session -- the reference's tuned AllSetTransformer runs use MLP_hidden 512).
template <int LPR, int VPL = 1>
__global__ __launch_bounds__(kBlock) void ln_res_fwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ colb, const float* __restrict__ res, int64_t ldr,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu_out, float p, uint64_t seed,
    float* __restrict__ y, int64_t ldy, float* __restrict__ stats, int64_t n, int d,
    const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  constexpr int NS = kWave / LPR;
  constexpr int RPG = VPL == 1 ? kLnRowsPerGroup : 2;              // rows in flight per lane group
  const int lane = lane_id();
  const int grp = (threadIdx.x >> 6) * NS + lane / LPR;
  const int li = lane % LPR;
  constexpr int kGroups = kWavesPerBlock * NS;
  const float inv_d = 1.f / static_cast<float>(d);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  int c0[VPL];
  bool active[VPL];
  float4 g4[VPL], b4[VPL], cb[VPL];
#pragma unroll
  for (int q = 0; q < VPL; ++q) {
    c0[q] = (li + LPR * q) * 4;
    active[q] = c0[q] < d;
    g4[q] = b4[q] = cb[q] = make_float4(0, 0, 0, 0);
    if (active[q]) {
      g4[q] = *reinterpret_cast<const float4*>(gamma + c0[q]);
      b4[q] = *reinterpret_cast<const float4*>(beta + c0[q]);
      if (colb) cb[q] = *reinterpret_cast<const float4*>(colb + c0[q]);
    }
  }
  const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * kGroups + grp) * RPG;
  float4 v[RPG][VPL], w[RPG][VPL];
#pragma unroll
  for (int r = 0; r < RPG; ++r) {                      // unconditional loads on clamped rows
    int64_t row = row0 + r;
    row = row < n ? row : n - 1;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      const int cc = active[q] ? c0[q] : 0;
      v[r][q] = *reinterpret_cast<const float4*>(x + row * ldx + cc);
      w[r][q] = res ? *reinterpret_cast<const float4*>(res + row * ldr + cc) : make_float4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < RPG; ++r) {
    const int64_t row = row0 + r;
    float4 t[VPL];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      t[q] = make_float4(v[r][q].x + w[r][q].x + cb[q].x, v[r][q].y + w[r][q].y + cb[q].y, v[r][q].z + w[r][q].z + cb[q].z,
                         v[r][q].w + w[r][q].w + cb[q].w);
      if (!active[q]) t[q] = make_float4(0, 0, 0, 0);
      s += t[q].x + t[q].y + t[q].z + t[q].w;
    }
    const float mean = group_sum<LPR>(s) * inv_d;
    float4 c[VPL];
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      c[q] = make_float4(t[q].x - mean, t[q].y - mean, t[q].z - mean, t[q].w - mean);
      if (!active[q]) c[q] = make_float4(0, 0, 0, 0);
      s2 += c[q].x * c[q].x + c[q].y * c[q].y + c[q].z * c[q].z + c[q].w * c[q].w;
    }
    const float var = group_sum<LPR>(s2) * inv_d;
    const float rstd = rsqrtf(var + eps);
    if (row < n) {
#pragma unroll
      for (int q = 0; q < VPL; ++q) {
        if (active[q]) {
          float4 o = make_float4(c[q].x * rstd * g4[q].x + b4[q].x, c[q].y * rstd * g4[q].y + b4[q].y, c[q].z * rstd * g4[q].z + b4[q].z,
                                 c[q].w * rstd * g4[q].w + b4[q].w);
          if (relu_out) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          if (p > 0.f) {
            const int64_t e = row * d + c0[q];
            float k0, k1, k2, k3;
            keep_scale2(seed, e, thr, inv_keep, k0, k1); keep_scale2(seed, e + 2, thr, inv_keep, k2, k3);
            o.x *= k0; o.y *= k1; o.z *= k2; o.w *= k3;
          }
          *reinterpret_cast<float4*>(y + row * ldy + c0[q]) = o;
        }
      }
      if (li == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    }
  }
}

// part[blockIdx][0|1|2][c] = dgamma, dbeta, dcolb (= column sums of gs)
template <int LPR, int VPL = 1>
__global__ __launch_bounds__(kBlock) void ln_res_bwd_kernel(
    const float* __restrict__ gy, int64_t ldg, const float* __restrict__ x, int64_t ldx, const float* __restrict__ colb,
    const float* __restrict__ res, int64_t ldr, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, int relu_out, float p, uint64_t seed, float* __restrict__ gs, int64_t ldgs,
    float* __restrict__ part, int64_t n, int d, const uint64_t* __restrict__ seed_base,
    const float* __restrict__ pma_m, const float* __restrict__ pma_l, float* __restrict__ pma_stats, int pma_heads) {
  seed = resolve_seed(seed_base, seed);
  constexpr int NS = kWave / LPR;
  constexpr int kGroups = kWavesPerBlock * NS;
  constexpr int W = LPR * 4 * VPL;                                  // columns covered
  __shared__ float red[kGroups][3][W];
  // optional epilogue for the PMA tail (x = the pooled output of allset_pma_fwd, gs = its gradient): the per-(row, head)
  // backward statistics {M = m + log(l + eps), delta = <x_head, gs_head>} that allset_pma_bwd_src gathers, written here
  // where both operands are in registers instead of by a separate pass over x and gs (allset_pma_bwd_stats)
  const int pma_g = pma_stats ? (d / pma_heads) / 4 : 1;           // lanes per head (a power of two <= LPR, checked by the host)
  const int lane = lane_id();
  const int grp = (threadIdx.x >> 6) * NS + lane / LPR;
  const int li = lane % LPR;
  const float inv_d = 1.f / static_cast<float>(d);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  int c0[VPL], cc[VPL];
  bool active[VPL];
  float4 g4[VPL], b4[VPL], cb[VPL], dg[VPL], db[VPL], dc[VPL];
#pragma unroll
  for (int q = 0; q < VPL; ++q) {
    c0[q] = (li + LPR * q) * 4;
    active[q] = c0[q] < d;
    cc[q] = active[q] ? c0[q] : 0;
    g4[q] = b4[q] = cb[q] = dg[q] = db[q] = dc[q] = make_float4(0, 0, 0, 0);
    if (active[q]) {
      g4[q] = *reinterpret_cast<const float4*>(gamma + c0[q]);
      b4[q] = *reinterpret_cast<const float4*>(beta + c0[q]);
      if (colb) cb[q] = *reinterpret_cast<const float4*>(colb + c0[q]);
    }
  }
  const int64_t rows_per_iter = static_cast<int64_t>(gridDim.x) * kGroups;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * kGroups + grp; row < n; row += rows_per_iter) {
    float4 xv[VPL], gv[VPL], xh[VPL], gh[VPL];
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      xv[q] = *reinterpret_cast<const float4*>(x + row * ldx + cc[q]);
      gv[q] = *reinterpret_cast<const float4*>(gy + row * ldg + cc[q]);
      const float4 rv = res ? *reinterpret_cast<const float4*>(res + row * ldr + cc[q]) : make_float4(0, 0, 0, 0);
      xh[q] = make_float4((xv[q].x + rv.x + cb[q].x - mean) * rstd, (xv[q].y + rv.y + cb[q].y - mean) * rstd,
                          (xv[q].z + rv.z + cb[q].z - mean) * rstd, (xv[q].w + rv.w + cb[q].w - mean) * rstd);
      if (!active[q]) { xh[q] = make_float4(0, 0, 0, 0); gv[q] = make_float4(0, 0, 0, 0); }
      if (p > 0.f) {
        const int64_t e = row * d + c0[q];
        float k0, k1, k2, k3;
        keep_scale2(seed, e, thr, inv_keep, k0, k1); keep_scale2(seed, e + 2, thr, inv_keep, k2, k3);
        gv[q].x *= k0; gv[q].y *= k1; gv[q].z *= k2; gv[q].w *= k3;
      }
      if (relu_out) {                                      // relu mask from the recomputed LayerNorm output
        if (!(fmaf(xh[q].x, g4[q].x, b4[q].x) > 0.f)) gv[q].x = 0.f;
        if (!(fmaf(xh[q].y, g4[q].y, b4[q].y) > 0.f)) gv[q].y = 0.f;
        if (!(fmaf(xh[q].z, g4[q].z, b4[q].z) > 0.f)) gv[q].z = 0.f;
        if (!(fmaf(xh[q].w, g4[q].w, b4[q].w) > 0.f)) gv[q].w = 0.f;
      }
      dg[q].x += gv[q].x * xh[q].x; dg[q].y += gv[q].y * xh[q].y; dg[q].z += gv[q].z * xh[q].z; dg[q].w += gv[q].w * xh[q].w;
      db[q].x += gv[q].x; db[q].y += gv[q].y; db[q].z += gv[q].z; db[q].w += gv[q].w;
      gh[q] = make_float4(gv[q].x * g4[q].x, gv[q].y * g4[q].y, gv[q].z * g4[q].z, gv[q].w * g4[q].w);
      t1 += gh[q].x + gh[q].y + gh[q].z + gh[q].w;
      t2 += gh[q].x * xh[q].x + gh[q].y * xh[q].y + gh[q].z * xh[q].z + gh[q].w * xh[q].w;
    }
    const float s1 = group_sum<LPR>(t1) * inv_d;
    const float s2 = group_sum<LPR>(t2) * inv_d;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      const float4 o = make_float4(rstd * (gh[q].x - s1 - xh[q].x * s2), rstd * (gh[q].y - s1 - xh[q].y * s2),
                                   rstd * (gh[q].z - s1 - xh[q].z * s2), rstd * (gh[q].w - s1 - xh[q].w * s2));
      if (active[q]) {
        dc[q].x += o.x; dc[q].y += o.y; dc[q].z += o.z; dc[q].w += o.w;
        *reinterpret_cast<float4*>(gs + row * ldgs + c0[q]) = o;
      }
      if (pma_stats != nullptr) {                           // (uniform branch; inactive lanes contribute 0 to the shuffles)
        float dot = active[q] ? (xv[q].x * o.x + xv[q].y * o.y + xv[q].z * o.z + xv[q].w * o.w) : 0.f;
        for (int off = 1; off < pma_g; off <<= 1) dot += __shfl_xor(dot, off);
        if (active[q] && (li % pma_g) == 0) {
          const int h = (li + LPR * q) / pma_g;
          const float lv = pma_l[row * pma_heads + h];
          // empty target: never gathered; exp(a - FLT_MAX) = 0 (same convention as pma_bwd_stats_kernel, csrc/pma.hip)
          const float M = lv > 0.f ? pma_m[row * pma_heads + h] + __logf(lv + 1e-16f) : 3.402823466e+38f;
          *reinterpret_cast<float2*>(pma_stats + (row * pma_heads + h) * 2) = make_float2(M, dot);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < VPL; ++q) {
    *reinterpret_cast<float4*>(&red[grp][0][(li + LPR * q) * 4]) = dg[q];
    *reinterpret_cast<float4*>(&red[grp][1][(li + LPR * q) * 4]) = db[q];
    *reinterpret_cast<float4*>(&red[grp][2][(li + LPR * q) * 4]) = dc[q];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * W; i += kBlock) {
    const int which = i / W, c = i % W;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) s += red[g][which][c];
    if (c < d) part[(static_cast<int64_t>(blockIdx.x) * 3 + which) * d + c] = s;
  }
}


// ---- weight gradient for bf16 activations (BASELINE configs[4] regime) ----------------------------------------------
// Same organisation as wgrad_x6_kernel (row-pair packed transposition into swizzled LDS planes, 64 x 32 wave tiles), but a
// bf16 x bf16 product is exact in the fp32 accumulator, so there is ONE plane per operand and one MFMA per tile and
// stage: the kernel is a pure stream over the two activation matrices (hipBLASLt tiles this [O x n] x [n x I] shape
// badly: 0.5-0.7 ms at n = 250k, d = 256 against ~0.05 ms of traffic).
template <int DUMMY>
__global__ __launch_bounds__(kWx6Block) void wgrad_bf16_kernel(
    const uint16_t* __restrict__ ga, int64_t lda, const uint16_t* __restrict__ u, int64_t ldu,
    float* __restrict__ part_w, float* __restrict__ part_b, int64_t n, int O, int I, int tiles_i, int64_t rows_per_slice,
    int64_t pw_stride, int64_t pb_stride) {
  __shared__ __attribute__((aligned(16))) uint32_t sP[2][2][kWgTile * 16];        // [buffer][A|B][feature*16 + ..]
  const int tile_o = blockIdx.x / tiles_i, tile_i = blockIdx.x % tiles_i;
  const int o_base = tile_o * kWgTile, i_base = tile_i * kWgTile;
  const int slice = blockIdx.y;
  const int64_t r_begin = static_cast<int64_t>(slice) * rows_per_slice;
  const int64_t r_end = min(n, r_begin + rows_per_slice);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rp = lane & 15, s_col = (wave * 4 + (lane >> 4)) * 4;
  const bool a_ok = (o_base + s_col) < O, b_ok = (i_base + s_col) < I;           // O, I are multiples of 4
  const int a_col = a_ok ? o_base + s_col : 0, b_col = b_ok ? i_base + s_col : 0;
  const int w_off = s_col * 16 + 4 * ((rp >> 2) ^ wx6_swz(s_col)) + (rp & 3);
  float4 bsum = make_float4(0, 0, 0, 0);

  struct Stage { uint2 ra[2], rb[2]; int64_t row0; };
  auto load_stage = [&](Stage& sg, int64_t r0) {            // unconditional loads on clamped rows / columns
    sg.row0 = r0 + 2 * rp;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t r = sg.row0 + h;
      r = r < r_end ? r : r_end - 1;
      sg.ra[h] = *reinterpret_cast<const uint2*>(ga + r * lda + a_col);
      sg.rb[h] = *reinterpret_cast<const uint2*>(u + r * ldu + b_col);
    }
  };
  auto store_stage = [&](Stage& sg, int buf) {
    uint2 a[2], b[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool in_range = (sg.row0 + h) < r_end;
      a[h] = (in_range && a_ok) ? sg.ra[h] : make_uint2(0u, 0u);
      b[h] = (in_range && b_ok) ? sg.rb[h] : make_uint2(0u, 0u);
      bsum.x += __uint_as_float(a[h].x << 16); bsum.y += __uint_as_float(a[h].x & 0xffff0000u);
      bsum.z += __uint_as_float(a[h].y << 16); bsum.w += __uint_as_float(a[h].y & 0xffff0000u);
    }
    uint32_t* pa = &sP[buf][0][w_off];
    uint32_t* pb = &sP[buf][1][w_off];
    // (row r, row r+1) of one column packed into a dword: low half = even row
    pa[0] = (a[0].x & 0xffffu) | (a[1].x << 16);  pa[16] = (a[0].x >> 16) | (a[1].x & 0xffff0000u);
    pa[32] = (a[0].y & 0xffffu) | (a[1].y << 16); pa[48] = (a[0].y >> 16) | (a[1].y & 0xffff0000u);
    pb[0] = (b[0].x & 0xffffu) | (b[1].x << 16);  pb[16] = (b[0].x >> 16) | (b[1].x & 0xffff0000u);
    pb[32] = (b[0].y & 0xffffu) | (b[1].y << 16); pb[48] = (b[0].y >> 16) | (b[1].y & 0xffff0000u);
  };

  const int fj = lane & 15, fg = lane >> 4;
  const int ob = (wave >> 2) * 64, ib = (wave & 3) * 32;
  f32x4_t acc[4][2];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) { acc[ot][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[ot][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  int a_off[4], b_off[2];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) { const int f = ob + ot * 16 + fj; a_off[ot] = f * 16 + 4 * (fg ^ wx6_swz(f)); }
#pragma unroll
  for (int it = 0; it < 2; ++it) { const int f = ib + it * 16 + fj; b_off[it] = f * 16 + 4 * (fg ^ wx6_swz(f)); }
  auto mfma_stage = [&](int buf) {
    WFrag b0, b1;
    b0.u = *reinterpret_cast<const uint4*>(&sP[buf][1][b_off[0]]);
    b1.u = *reinterpret_cast<const uint4*>(&sP[buf][1][b_off[1]]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
      WFrag a;
      a.u = *reinterpret_cast<const uint4*>(&sP[buf][0][a_off[ot]]);
      acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b0.v, acc[ot][0], 0, 0, 0);
      acc[ot][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b1.v, acc[ot][1], 0, 0, 0);
    }
  };

  Stage s0, s1;
  if (r_begin < r_end) {
    load_stage(s0, r_begin);
    load_stage(s1, r_begin + kWgRows);
    store_stage(s0, 0);
    load_stage(s0, r_begin + 2 * kWgRows);
  }
  __syncthreads();
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * kWgRows) {
    mfma_stage(0);
    if (r0 + kWgRows < r_end) store_stage(s1, 1);
    load_stage(s1, r0 + 3 * kWgRows);
    __syncthreads();
    if (r0 + kWgRows < r_end) {
      mfma_stage(1);
      if (r0 + 2 * kWgRows < r_end) store_stage(s0, 0);
      load_stage(s0, r0 + 4 * kWgRows);
      __syncthreads();
    }
  }

  float* pw = part_w + static_cast<int64_t>(slice) * pw_stride;
#pragma unroll
  for (int ot = 0; ot < 4; ++ot)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = i_base + ib + it * 16 + fj;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o_base + ob + ot * 16 + 4 * fg + r;
        if (o < O && i < I) pw[static_cast<int64_t>(o) * I + i] = acc[ot][it][r];
      }
    }
  if (tile_i == 0 && part_b != nullptr) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      bsum.x += __shfl_xor(bsum.x, off); bsum.y += __shfl_xor(bsum.y, off);
      bsum.z += __shfl_xor(bsum.z, off); bsum.w += __shfl_xor(bsum.w, off);
    }
    if (rp == 0 && a_ok) *reinterpret_cast<float4*>(part_b + static_cast<int64_t>(slice) * pb_stride + o_base + s_col) = bsum;
  }
}


// ---- the same weight gradient, full width per workgroup, transposed by the LDS read (round 2) ---------------------------------
// wgrad_bf16_kernel above tiles gW into 128 x 128 blocks, so at 256 x 256 (BASELINE configs[4]) four workgroups stream the same
// rows and every operand is read twice: 0.10 ms where one pass over the two [250k, 256] bf16 matrices is 0.05 ms.  Here one
// workgroup owns ALL of gW for its slice of rows: 8 waves x (O/4 x I/2) accumulators = 128 registers per lane at 256 x 256.
// A 32-row stage of both operands is copied into LDS exactly as it lies in memory (16-byte pieces, row-major, 32-byte chunks
// XOR-swizzled by the row) and the MFMA fragments -- 8 consecutive ROWS of one column -- come out of ds_read_b64_tr_b16
// (hardware 4 x 4 transpose, see fused_bwd.hip): no row-pair packing, no VALU in the staging path at all.
typedef short wv4s_t __attribute__((ext_vector_type(4)));
typedef __bf16 wv2bf_t __attribute__((ext_vector_type(2)));
union WTrFrag { uint4 u; bf16x8_t v; struct { wv4s_t lo, hi; } t; };

template <int PITCH>
__device__ __forceinline__ int wtr_off(int row, int cbyte) {
  constexpr int NCH = PITCH / 32;                                  // 32-byte chunks per row
  const int sw = ((row & 3) | (((row >> 3) & 1) << 2)) & (NCH - 1);
  return row * PITCH + ((((cbyte >> 5) ^ sw)) << 5) + (cbyte & 31);
}

// Round 6 -- MASKED: `ga` is the gradient BEFORE the relu mask and `bits` the forward's bit mask of that relu (fused_bf16.hip:
// bit 16 hb + j of word sq of a row is column 64 hb + 16 sq + j, N / 8 bytes per row); the mask is applied as the stage is copied
// into LDS, so the masked gradient never exists in memory.  AUX: four more rows of the gradient -- gWa[h][i] = sum_r g4[r][h] u[r][i],
// the weight gradient of PMA's folded logits (g4: fp32 [n, 4], rounded to bf16 here as the separate launch did) -- ride as a fifth,
// 16-column A tile (12 columns zero) on the two waves of the first O block; the partial row becomes [gW | gb | gWa (4 x I) | gba (4)].
template <int OTN, int ITN, bool MASKED, bool AUX>       // O = 64 OTN, I = 32 ITN; wave tile (16 OTN) x (16 ITN)
__global__ __launch_bounds__(kWx6Block) void wgrad_bf16_tr_kernel(
    const uint16_t* __restrict__ ga, int64_t lda, const uint8_t* __restrict__ bits, const float* __restrict__ g4,
    const uint16_t* __restrict__ u, int64_t ldu, float* __restrict__ part_w, float* __restrict__ part_b,
    float* __restrict__ part_x, int64_t n, int64_t rows_per_slice, int64_t pw_stride, int64_t pb_stride) {
  constexpr int O = 64 * OTN, I = 32 * ITN;
  constexpr int PA = O * 2, PB = I * 2;                            // row pitches (bytes)
  constexpr int SA = 32 * PA, SB = 32 * PB;                        // bytes per stage and operand
  constexpr int SX = AUX ? 32 * 32 : 0;                            // the auxiliary tile: 32 rows x 16 bf16
  constexpr int NPA = (32 * PA / 16 + kWx6Block - 1) / kWx6Block;  // 16-byte pieces per thread and stage
  constexpr int NPB = (32 * PB / 16 + kWx6Block - 1) / kWx6Block;
  __shared__ __attribute__((aligned(16))) uint8_t sS[2 * (SA + SB + SX)];
  const int slice = blockIdx.x;
  const int64_t r_begin = static_cast<int64_t>(slice) * rows_per_slice;
  const int64_t r_end = min(n, r_begin + rows_per_slice);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (AUX) {                                             // columns 4 .. 15 of the auxiliary tile stay zero
    for (int idx = tid; idx < 2 * SX / 4; idx += kWx6Block) {
      const int buf = idx / (SX / 4), w = idx % (SX / 4);
      reinterpret_cast<uint32_t*>(sS + buf * (SA + SB + SX) + SA + SB)[w] = 0u;
    }
  }

  struct Stage { uint4 a[NPA], b[NPB]; uint32_t am[MASKED ? NPA : 1]; float4 x; };
  auto load_stage = [&](Stage& sg, int64_t r0) {                 // unconditional loads on clamped rows; zeroed when stored
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
      const int p = tid + k * kWx6Block, row = p / (PA / 16), c16 = p % (PA / 16);
      int64_t r = r0 + row;
      r = r < r_end ? r : r_end - 1;
      if (32 * PA / 16 % kWx6Block == 0 || p < 32 * PA / 16) {
        sg.a[k] = *reinterpret_cast<const uint4*>(ga + r * lda + c16 * 8);
        if constexpr (MASKED)      // the piece's 8 columns 8 c16 ..: one byte of word sq = (c16 % 8) / 2, at bits 16 (c16 / 8) + 8 (c16 & 1)
          sg.am[k] = bits[r * (O / 8) + ((c16 & 7) >> 1) * (O / 32) + (c16 >> 3) * 2 + (c16 & 1)];
      }
    }
    if constexpr (AUX) {
      int64_t r = r0 + (tid & 31);
      r = r < r_end ? r : r_end - 1;
      if (tid < 32) sg.x = *reinterpret_cast<const float4*>(g4 + r * 4);
    }
#pragma unroll
    for (int k = 0; k < NPB; ++k) {
      const int p = tid + k * kWx6Block, row = p / (PB / 16), c16 = p % (PB / 16);
      int64_t r = r0 + row;
      r = r < r_end ? r : r_end - 1;
      if (32 * PB / 16 % kWx6Block == 0 || p < 32 * PB / 16) sg.b[k] = *reinterpret_cast<const uint4*>(u + r * ldu + c16 * 8);
    }
  };
  auto store_stage = [&](const Stage& sg, int64_t r0, int buf) {
    uint8_t* ba = sS + buf * (SA + SB + SX);
    uint8_t* bb = ba + SA;
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
      const int p = tid + k * kWx6Block, row = p / (PA / 16), c16 = p % (PA / 16);
      if (32 * PA / 16 % kWx6Block == 0 || p < 32 * PA / 16) {
        uint4 v = sg.a[k];
        if constexpr (MASKED) {
          // bit 2 i -> bit 0, bit 2 i + 1 -> bit 16 of (sp >> 2 i): the two halves' 0 / 1 factors of dword i (as in fused_bf16.hip)
          const uint32_t sp = sg.am[k] | (sg.am[k] << 15);
          auto keep = [&](uint32_t a, int i) {
            uint32_t r;
            asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"((sp >> (2 * i)) & 0x00010001u));
            return r;
          };
          v.x = keep(v.x, 0); v.y = keep(v.y, 1); v.z = keep(v.z, 2); v.w = keep(v.w, 3);
        }
        *reinterpret_cast<uint4*>(ba + wtr_off<PA>(row, c16 * 16)) = (r0 + row < r_end) ? v : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    if constexpr (AUX) {
      if (tid < 32) {
        const bool ok = r0 + tid < r_end;
        *reinterpret_cast<uint2*>(bb + SB + tid * 32) = ok ? make_uint2(cvt_pk_bf16(sg.x.x, sg.x.y), cvt_pk_bf16(sg.x.z, sg.x.w)) : make_uint2(0u, 0u);
      }
    }
#pragma unroll
    for (int k = 0; k < NPB; ++k) {
      const int p = tid + k * kWx6Block, row = p / (PB / 16), c16 = p % (PB / 16);
      if (32 * PB / 16 % kWx6Block == 0 || p < 32 * PB / 16)
        *reinterpret_cast<uint4*>(bb + wtr_off<PB>(row, c16 * 16)) = (r0 + row < r_end) ? sg.b[k] : make_uint4(0u, 0u, 0u, 0u);
    }
  };

  // fragment addresses: lane (i = lane & 15, g = lane >> 4) of a 16 x 16 x 32 operand supplies rows 8 g + (i >> 2) [+4 for the
  // second read], 4 columns at 4 (i & 3) of the tile's 16 columns (32 bytes = one swizzle chunk)
  const int fi = lane & 15, fg = lane >> 4;
  const int sw = (fi >> 2) | ((fg & 1) << 2);
  const int ob = (wave >> 1) * (16 * OTN), ib = (wave & 1) * (16 * ITN);     // this wave's tile origin
  const int rowoff_a = (8 * fg + (fi >> 2)) * PA + 8 * (fi & 3), rowoff_b = (8 * fg + (fi >> 2)) * PB + 8 * (fi & 3);
  f32x4_t acc[OTN][ITN];
#pragma unroll
  for (int ot = 0; ot < OTN; ++ot)
#pragma unroll
    for (int it = 0; it < ITN; ++it) acc[ot][it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float gbs[OTN];
#pragma unroll
  for (int ot = 0; ot < OTN; ++ot) gbs[ot] = 0.f;
  const wv2bf_t ones = __builtin_bit_cast(wv2bf_t, 0x3f803f80u);
  // the auxiliary tile's ITN column tiles of this wave's I half are shared out over the four waves of that half (wave >> 1):
  // ITN / 4 accumulator tiles each, so that no wave carries 32 more registers
  static_assert(!AUX || ITN % 4 == 0, "auxiliary rows: ITN must be a multiple of 4");
  constexpr int XT = AUX ? ITN / 4 : 1;
  const int wo = __builtin_amdgcn_readfirstlane(wave >> 1);        // wave-uniform
  f32x4_t accx[XT];
#pragma unroll
  for (int it = 0; it < XT; ++it) accx[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float gbx = 0.f;

  auto tr_frag = [&](const uint8_t* p, int half) {
    WTrFrag f;
    f.t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wv4s_t*)(p));
    f.t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wv4s_t*)(p + half));
    return f;
  };
  auto mfma_stage = [&](int buf) {
    const uint8_t* ba = sS + buf * (SA + SB + SX);
    const uint8_t* bb = ba + SA;
    WTrFrag a[OTN];
#pragma unroll
    for (int ot = 0; ot < OTN; ++ot) {
      const int chunk = (ob + ot * 16) / 16;
      a[ot] = tr_frag(ba + rowoff_a + (((chunk ^ sw) & (PA / 32 - 1)) << 5), 4 * PA);
    }
    WTrFrag ax;
    ax.u = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (AUX) {
      {
        ax = tr_frag(bb + SB + (8 * fg + (fi >> 2)) * 32 + 8 * (fi & 3), 4 * 32);
        if (wave == 0) {
          gbx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wv2bf_t, ax.u.x), ones, gbx, false);
          gbx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wv2bf_t, ax.u.y), ones, gbx, false);
          gbx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wv2bf_t, ax.u.z), ones, gbx, false);
          gbx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wv2bf_t, ax.u.w), ones, gbx, false);
        }
      }
    }
    if ((wave & 1) == 0 && part_b != nullptr) {
#pragma unroll
      for (int ot = 0; ot < OTN; ++ot) {
        gbs[ot] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wv2bf_t, a[ot].u.x), ones, gbs[ot], false);
        gbs[ot] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wv2bf_t, a[ot].u.y), ones, gbs[ot], false);
        gbs[ot] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wv2bf_t, a[ot].u.z), ones, gbs[ot], false);
        gbs[ot] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wv2bf_t, a[ot].u.w), ones, gbs[ot], false);
      }
    }
#pragma unroll
    for (int it = 0; it < ITN; ++it) {
      const int chunk = (ib + it * 16) / 16;
      const WTrFrag b = tr_frag(bb + rowoff_b + (((chunk ^ sw) & (PB / 32 - 1)) << 5), 4 * PB);
#pragma unroll
      for (int ot = 0; ot < OTN; ++ot) acc[ot][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ot].v, b.v, acc[ot][it], 0, 0, 0);
      if constexpr (AUX) {
        if (wo == it / XT) accx[it % XT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax.v, b.v, accx[it % XT], 0, 0, 0);
      }
    }
  };

  Stage s0, s1;
  if (r_begin < r_end) {                       // stage 0 -> LDS buffer 0; stages 1 and 2 on their way
    load_stage(s0, r_begin);
    load_stage(s1, r_begin + 32);
    store_stage(s0, r_begin, 0);
    load_stage(s0, r_begin + 64);
  }
  __syncthreads();
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 64) {
    mfma_stage(0);
    if (r0 + 32 < r_end) store_stage(s1, r0 + 32, 1);
    load_stage(s1, r0 + 96);
    __syncthreads();
    if (r0 + 32 < r_end) {
      mfma_stage(1);
      if (r0 + 64 < r_end) store_stage(s0, r0 + 64, 0);
      load_stage(s0, r0 + 128);
      __syncthreads();
    }
  }

  // partial tile -> part_w[slice][O][I]; acc[ot][it][r] is (o = ob + 16 ot + 4 fg + r, i = ib + 16 it + fi)
  float* pw = part_w + static_cast<int64_t>(slice) * pw_stride;
#pragma unroll
  for (int ot = 0; ot < OTN; ++ot)
#pragma unroll
    for (int it = 0; it < ITN; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(ob + ot * 16 + 4 * fg + r) * I + ib + it * 16 + fi] = acc[ot][it][r];
  if ((wave & 1) == 0 && part_b != nullptr) {  // lane (column fi, row group fg): fold the four row groups
#pragma unroll
    for (int ot = 0; ot < OTN; ++ot) {
      float v = gbs[ot];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 16) part_b[static_cast<int64_t>(slice) * pb_stride + ob + ot * 16 + fi] = v;
    }
  }
  if constexpr (AUX) {
    {                                          // accx[t][r] is (h = 4 fg + r, i = ib + 16 (wo XT + t) + fi): rows 0 .. 3 are real
      float* px = part_x + static_cast<int64_t>(slice) * pb_stride;
      if (fg == 0) {
#pragma unroll
        for (int t = 0; t < XT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) px[r * I + ib + (wo * XT + t) * 16 + fi] = accx[t][r];
      }
      if (wave == 0) {
        float v = gbx;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 4) px[4 * I + lane] = v;
      }
    }
  }
}

// ---- LayerNorm for bf16 activations (fp32 statistics and arithmetic, bf16 in / out; gamma, beta bf16) -------------------
// One lane = 8 consecutive columns (16 bytes), LPR = d / 8 lanes per row (d % 8 == 0, d <= 512), 64 / LPR rows per wave.
struct F8 { float v[8]; };
__device__ __forceinline__ F8 unpack8(uint4 q) {
  F8 r;
  r.v[0] = __uint_as_float(q.x << 16); r.v[1] = __uint_as_float(q.x & 0xffff0000u);
  r.v[2] = __uint_as_float(q.y << 16); r.v[3] = __uint_as_float(q.y & 0xffff0000u);
  r.v[4] = __uint_as_float(q.z << 16); r.v[5] = __uint_as_float(q.z & 0xffff0000u);
  r.v[6] = __uint_as_float(q.w << 16); r.v[7] = __uint_as_float(q.w & 0xffff0000u);
  return r;
}
__device__ __forceinline__ uint4 pack8(const F8& f) {
  return make_uint4(cvt_pk_bf16(f.v[0], f.v[1]), cvt_pk_bf16(f.v[2], f.v[3]), cvt_pk_bf16(f.v[4], f.v[5]), cvt_pk_bf16(f.v[6], f.v[7]));
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void ln_fwd_bf16_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
    float eps, int relu_in, float p, uint64_t seed, uint16_t* __restrict__ y, int64_t ldy,
    float* __restrict__ stats, int64_t n, int d, const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  constexpr int NS = kWave / LPR;
  const int lane = lane_id();
  const int grp = (threadIdx.x >> 6) * NS + lane / LPR;
  const int li = lane % LPR;
  const int c0 = li * 8;
  const bool active = c0 < d;
  constexpr int kGroups = kWavesPerBlock * NS;
  const float inv_d = 1.f / static_cast<float>(d);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  const int cc = active ? c0 : 0;
  const F8 g8 = unpack8(*reinterpret_cast<const uint4*>(gamma + cc)), b8 = unpack8(*reinterpret_cast<const uint4*>(beta + cc));
  const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * kGroups + grp) * kLnRowsPerGroup;
  uint4 raw[kLnRowsPerGroup];
#pragma unroll
  for (int r = 0; r < kLnRowsPerGroup; ++r) {
    int64_t row = row0 + r;
    row = row < n ? row : n - 1;
    raw[r] = *reinterpret_cast<const uint4*>(x + row * ldx + cc);
  }
#pragma unroll
  for (int r = 0; r < kLnRowsPerGroup; ++r) {
    const int64_t row = row0 + r;
    F8 t = unpack8(raw[r]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (relu_in) t.v[k] = fmaxf(t.v[k], 0.f); if (!active) t.v[k] = 0.f; s += t.v[k]; }
    const float mean = group_sum<LPR>(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { t.v[k] = active ? t.v[k] - mean : 0.f; q = fmaf(t.v[k], t.v[k], q); }
    const float rstd = rsqrtf(group_sum<LPR>(q) * inv_d + eps);
    if (row < n) {
      if (active) {
        F8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o.v[k] = fmaf(t.v[k] * rstd, g8.v[k], b8.v[k]);
        if (p > 0.f) {
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            float k0, k1;
            keep_scale2(seed, row * d + c0 + k, thr, inv_keep, k0, k1);
            o.v[k] *= k0; o.v[k + 1] *= k1;
          }
        }
        *reinterpret_cast<uint4*>(y + row * ldy + c0) = pack8(o);
      }
      if (li == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    }
  }
}

// part[blockIdx][0|1][c] = dgamma, dbeta (fp32)
template <int LPR>
__global__ __launch_bounds__(kBlock) void ln_bwd_bf16_kernel(
    const uint16_t* __restrict__ gy, int64_t ldg, const uint16_t* __restrict__ x, int64_t ldx,
    const float* __restrict__ stats, const uint16_t* __restrict__ gamma, int relu_in, float p, uint64_t seed,
    uint16_t* __restrict__ gx, int64_t ldgx, float* __restrict__ part, int64_t n, int d,
    const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  constexpr int NS = kWave / LPR;
  constexpr int kGroups = kWavesPerBlock * NS;
  __shared__ float red[kGroups][2][LPR * 8];
  const int lane = lane_id();
  const int grp = (threadIdx.x >> 6) * NS + lane / LPR;
  const int li = lane % LPR;
  const int c0 = li * 8;
  const bool active = c0 < d;
  const int cc = active ? c0 : 0;
  const float inv_d = 1.f / static_cast<float>(d);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  const F8 g8 = unpack8(*reinterpret_cast<const uint4*>(gamma + cc));
  F8 dg, db;
#pragma unroll
  for (int k = 0; k < 8; ++k) { dg.v[k] = 0.f; db.v[k] = 0.f; }
  const int64_t rows_per_iter = static_cast<int64_t>(gridDim.x) * kGroups;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * kGroups + grp; row < n; row += rows_per_iter) {
    const F8 xv = unpack8(*reinterpret_cast<const uint4*>(x + row * ldx + cc));
    F8 gv = unpack8(*reinterpret_cast<const uint4*>(gy + row * ldg + cc));
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    F8 xh, gh;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      if (p > 0.f) {
        float k0, k1;
        keep_scale2(seed, row * d + c0 + k, thr, inv_keep, k0, k1);
        gv.v[k] *= k0; gv.v[k + 1] *= k1;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float t = relu_in ? fmaxf(xv.v[k], 0.f) : xv.v[k];
      xh.v[k] = active ? (t - mean) * rstd : 0.f;
      if (!active) gv.v[k] = 0.f;
      dg.v[k] = fmaf(gv.v[k], xh.v[k], dg.v[k]);
      db.v[k] += gv.v[k];
      gh.v[k] = gv.v[k] * g8.v[k];
      s1 += gh.v[k];
      s2 = fmaf(gh.v[k], xh.v[k], s2);
    }
    s1 = group_sum<LPR>(s1) * inv_d;
    s2 = group_sum<LPR>(s2) * inv_d;
    if (active && gx != nullptr) {
      F8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o.v[k] = rstd * (gh.v[k] - s1 - xh.v[k] * s2);
        if (relu_in && !(xv.v[k] > 0.f)) o.v[k] = 0.f;
      }
      *reinterpret_cast<uint4*>(gx + row * ldgx + c0) = pack8(o);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[grp][0][c0 + k] = dg.v[k]; red[grp][1][c0 + k] = db.v[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * LPR * 8; i += kBlock) {
    const int which = i / (LPR * 8), c = i % (LPR * 8);
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) s += red[g][which][c];
    if (c < d) part[(static_cast<int64_t>(blockIdx.x) * 2 + which) * d + c] = s;
  }
}


// ---- ln_res_* for bf16 activations: y = dropout_p(relu_out(LN(x + colb + res))), bf16 in / out, fp32 arithmetic ------------
// HAS_RES / DROP (and PMA in the backward) are compile-time: as run-time switches -- uniform as they are -- they put a branch around a
// load and around the dropout hashes in every row of the loop (round 6, the lesson of fused_bf16.hip's epilogue)
template <int LPR, bool HAS_RES, bool DROP>
__global__ __launch_bounds__(kBlock) void ln_res_fwd_bf16_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ colb, const uint16_t* __restrict__ res,
    int64_t ldr, const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta, float eps, int relu_out, float p,
    uint64_t seed, uint16_t* __restrict__ y, int64_t ldy, float* __restrict__ stats, int64_t n, int d,
    const uint64_t* __restrict__ seed_base) {
  seed = resolve_seed(seed_base, seed);
  constexpr int NS = kWave / LPR;
  const int lane = lane_id();
  const int grp = (threadIdx.x >> 6) * NS + lane / LPR;
  const int li = lane % LPR;
  const int c0 = li * 8;
  const bool active = c0 < d;
  const int cc = active ? c0 : 0;
  constexpr int kGroups = kWavesPerBlock * NS;
  const float inv_d = 1.f / static_cast<float>(d);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  const F8 g8 = unpack8(*reinterpret_cast<const uint4*>(gamma + cc)), b8 = unpack8(*reinterpret_cast<const uint4*>(beta + cc));
  F8 cb;
#pragma unroll
  for (int k = 0; k < 8; ++k) cb.v[k] = 0.f;
  if (colb) cb = unpack8(*reinterpret_cast<const uint4*>(colb + cc));
  const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * kGroups + grp) * kLnRowsPerGroup;
  uint4 rx[kLnRowsPerGroup], rr[kLnRowsPerGroup];
#pragma unroll
  for (int r = 0; r < kLnRowsPerGroup; ++r) {
    int64_t row = row0 + r;
    row = row < n ? row : n - 1;
    rx[r] = *reinterpret_cast<const uint4*>(x + row * ldx + cc);
    if constexpr (HAS_RES) rr[r] = *reinterpret_cast<const uint4*>(res + row * ldr + cc);
    else rr[r] = make_uint4(0u, 0u, 0u, 0u);
  }
  const uint32_t relu_sel = relu_out ? 0xffffffffu : 0u;          // (bit-select instead of a branch per element; a no-relu NaN stays a NaN)
#pragma unroll
  for (int r = 0; r < kLnRowsPerGroup; ++r) {
    const int64_t row = row0 + r;
    F8 t = unpack8(rx[r]);
    const F8 w = unpack8(rr[r]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { t.v[k] = active ? t.v[k] + w.v[k] + cb.v[k] : 0.f; s += t.v[k]; }
    const float mean = group_sum<LPR>(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { t.v[k] = active ? t.v[k] - mean : 0.f; q = fmaf(t.v[k], t.v[k], q); }
    const float rstd = rsqrtf(group_sum<LPR>(q) * inv_d + eps);
    if (row < n) {
      if (active) {
        F8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = fmaf(t.v[k] * rstd, g8.v[k], b8.v[k]);
          o.v[k] = __uint_as_float((__float_as_uint(fmaxf(v, 0.f)) & relu_sel) | (__float_as_uint(v) & ~relu_sel));
        }
        if constexpr (DROP) {
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            float k0, k1;
            keep_scale2(seed, row * d + c0 + k, thr, inv_keep, k0, k1);
            o.v[k] *= k0; o.v[k + 1] *= k1;
          }
        }
        *reinterpret_cast<uint4*>(y + row * ldy + c0) = pack8(o);
      }
      if (li == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    }
  }
}

// part[blockIdx][0|1|2][c] = dgamma, dbeta, dcolb (fp32)
template <int LPR, bool HAS_RES, bool DROP, bool PMA>
__global__ __launch_bounds__(kBlock) void ln_res_bwd_bf16_kernel(
    const uint16_t* __restrict__ gy, int64_t ldg, const uint16_t* __restrict__ x, int64_t ldx,
    const uint16_t* __restrict__ colb, const uint16_t* __restrict__ res, int64_t ldr, const float* __restrict__ stats,
    const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta, int relu_out, float p, uint64_t seed,
    uint16_t* __restrict__ gs, int64_t ldgs, float* __restrict__ part, int64_t n, int d,
    const uint64_t* __restrict__ seed_base, const float* __restrict__ pma_m, const float* __restrict__ pma_l,
    float* __restrict__ pma_stats, int pma_heads) {
  seed = resolve_seed(seed_base, seed);
  constexpr int NS = kWave / LPR;
  constexpr int kGroups = kWavesPerBlock * NS;
  const int pma_g = PMA ? (d / pma_heads) / 8 : 1;                  // lanes per head (a power of two, checked by the host)
  const float relu_floor = relu_out ? 0.f : -__builtin_inff();
  __shared__ float red[kGroups][3][LPR * 8];
  const int lane = lane_id();
  const int grp = (threadIdx.x >> 6) * NS + lane / LPR;
  const int li = lane % LPR;
  const int c0 = li * 8;
  const bool active = c0 < d;
  const int cc = active ? c0 : 0;
  const float inv_d = 1.f / static_cast<float>(d);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t thr = drop_threshold(p);
  const F8 g8 = unpack8(*reinterpret_cast<const uint4*>(gamma + cc)), b8 = unpack8(*reinterpret_cast<const uint4*>(beta + cc));
  F8 cb, dg, db, dc;
#pragma unroll
  for (int k = 0; k < 8; ++k) { cb.v[k] = 0.f; dg.v[k] = 0.f; db.v[k] = 0.f; dc.v[k] = 0.f; }
  if (colb) cb = unpack8(*reinterpret_cast<const uint4*>(colb + cc));
  const int64_t rows_per_iter = static_cast<int64_t>(gridDim.x) * kGroups;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * kGroups + grp; row < n; row += rows_per_iter) {
    const F8 xv = unpack8(*reinterpret_cast<const uint4*>(x + row * ldx + cc));
    F8 gv = unpack8(*reinterpret_cast<const uint4*>(gy + row * ldg + cc));
    uint4 rraw = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (HAS_RES) rraw = *reinterpret_cast<const uint4*>(res + row * ldr + cc);
    const F8 rv = unpack8(rraw);
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    if constexpr (DROP) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        float k0, k1;
        keep_scale2(seed, row * d + c0 + k, thr, inv_keep, k0, k1);
        gv.v[k] *= k0; gv.v[k + 1] *= k1;
      }
    }
    F8 xh, gh;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      xh.v[k] = active ? (xv.v[k] + rv.v[k] + cb.v[k] - mean) * rstd : 0.f;
      if (!active) gv.v[k] = 0.f;
      if (!(fmaf(xh.v[k], g8.v[k], b8.v[k]) > relu_floor)) gv.v[k] = 0.f;     // (no relu: the floor is -inf; a NaN output drops its gradient either way)
      dg.v[k] = fmaf(gv.v[k], xh.v[k], dg.v[k]);
      db.v[k] += gv.v[k];
      gh.v[k] = gv.v[k] * g8.v[k];
      s1 += gh.v[k];
      s2 = fmaf(gh.v[k], xh.v[k], s2);
    }
    s1 = group_sum<LPR>(s1) * inv_d;
    s2 = group_sum<LPR>(s2) * inv_d;
    float dot = 0.f;
    if (active) {
      F8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) { o.v[k] = rstd * (gh.v[k] - s1 - xh.v[k] * s2); dc.v[k] += o.v[k]; }
      const uint4 packed = pack8(o);
      *reinterpret_cast<uint4*>(gs + row * ldgs + c0) = packed;
      if constexpr (PMA) {                                // <x, gs> of this lane's 8 columns, on the values as stored
        const F8 r = unpack8(packed);
#pragma unroll
        for (int k = 0; k < 8; ++k) dot = fmaf(xv.v[k], r.v[k], dot);
      }
    }
    if constexpr (PMA) {                                  // (inactive lanes contribute 0 to the shuffles)
      for (int off = 1; off < pma_g; off <<= 1) dot += __shfl_xor(dot, off);
      if (active && (li % pma_g) == 0) {
        const int h = li / pma_g;
        const float lv = pma_l[row * pma_heads + h];
        // empty target: never gathered; exp(a - FLT_MAX) = 0 (same convention as pma_bwd_stats_kernel, csrc/pma.hip)
        const float M = lv > 0.f ? pma_m[row * pma_heads + h] + __logf(lv + 1e-16f) : 3.402823466e+38f;
        *reinterpret_cast<float2*>(pma_stats + (row * pma_heads + h) * 2) = make_float2(M, dot);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[grp][0][c0 + k] = dg.v[k]; red[grp][1][c0 + k] = db.v[k]; red[grp][2][c0 + k] = dc.v[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * LPR * 8; i += kBlock) {
    const int which = i / (LPR * 8), c = i % (LPR * 8);
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) s += red[g][which][c];
    if (c < d) part[(static_cast<int64_t>(blockIdx.x) * 3 + which) * d + c] = s;
  }
}

}  // namespace allset

using namespace allset;


extern "C" int allset_ln_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                             int relu_in, float p, uint64_t seed, float* y, int64_t ldy, float* stats,
                             int64_t n, int64_t d, const uint64_t* seed_base, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1 && d < INT32_MAX, "ln_fwd: bad size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "ln_fwd: dropout p must be in [0,1)");
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(x && gamma && beta && y && stats, "ln_fwd: null pointer");
  ALLSET_REQUIRE(ldx >= d && ldy >= d, "ln_fwd: leading dimension smaller than d");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool fast = d <= 256 && d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y) &&
                    aligned16(gamma) && aligned16(beta);
  const int di = static_cast<int>(d);
  if (fast) {
    const int lpr = ln_lpr(d);
    const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr) * kLnRowsPerGroup;
    const unsigned grid = static_cast<unsigned>((n + rows_per_block - 1) / rows_per_block);
    switch (lpr) {
      case 8:  ln_fwd_kernel<8><<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base); break;
      case 16: ln_fwd_kernel<16><<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base); break;
      case 32: ln_fwd_kernel<32><<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base); break;
      default: ln_fwd_kernel<64><<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base); break;
    }
  } else {
    const unsigned grid = static_cast<unsigned>((n + kWavesPerBlock - 1) / kWavesPerBlock);
    if (d <= 8 * kWave) ln_fwd_rows_kernel<8><<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base);
    else if (d <= 16 * kWave) ln_fwd_rows_kernel<16><<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base);
    else if (d <= 24 * kWave) ln_fwd_rows_kernel<24><<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base);
    else if (d <= 32 * kWave) ln_fwd_rows_kernel<32><<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base);
    else ln_fwd_generic_kernel<<<grid, kBlock, 0, st>>>(x, ldx, gamma, beta, eps, relu_in, p, seed, y, ldy, stats, n, di, seed_base);
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_ln_bwd_partials(int64_t n, int64_t d, int64_t* n_partials) {
  clear_error();
  ALLSET_REQUIRE(n_partials != nullptr && n >= 0 && d >= 1, "ln_bwd_partials: bad argument");
  const bool fast = d <= 256 && d % 4 == 0;
  if (!fast) {      // generic kernel: one partial row per block, 4 rows per block-iteration, at most 512 blocks
    const int64_t want = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    *n_partials = want < 1 ? 1 : (want > 512 ? 512 : want);
    return ALLSET_OK;
  }
  const int lpr = ln_lpr(d);
  const int64_t groups = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr);
  const int64_t want = (n + groups - 1) / groups;
  *n_partials = want < 1 ? 1 : (want > 2048 ? 2048 : want);
  return ALLSET_OK;
}

extern "C" int allset_ln_bwd(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* stats,
                             const float* gamma, int relu_in, float p, uint64_t seed, float* gx, int64_t ldgx,
                             float* partials, int64_t n_partials, int64_t n, int64_t d, const uint64_t* seed_base,
                             void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1 && d < INT32_MAX, "ln_bwd: bad size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "ln_bwd: dropout p must be in [0,1)");
  ALLSET_REQUIRE(partials != nullptr && n_partials >= 1, "ln_bwd: partials buffer required");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool shape_fast = d <= 256 && d % 4 == 0;
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(partials, 0, static_cast<size_t>(n_partials) * 2 * d * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && x && stats && gamma, "ln_bwd: null pointer");         // gx may be NULL: parameter partials only
  ALLSET_REQUIRE(ldg >= d && ldx >= d && (gx == nullptr || ldgx >= d), "ln_bwd: leading dimension smaller than d");
  const bool fast = shape_fast && ldg % 4 == 0 && ldx % 4 == 0 && (gx == nullptr || (ldgx % 4 == 0 && aligned16(gx))) &&
                    aligned16(gy) && aligned16(x) && aligned16(gamma);
  const int di = static_cast<int>(d);
  if (fast) {
    const unsigned grid = static_cast<unsigned>(n_partials);
    switch (ln_lpr(d)) {
      case 8:  ln_bwd_kernel<8><<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, stats, gamma, relu_in, p, seed, gx, ldgx, partials, n, di, seed_base); break;
      case 16: ln_bwd_kernel<16><<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, stats, gamma, relu_in, p, seed, gx, ldgx, partials, n, di, seed_base); break;
      case 32: ln_bwd_kernel<32><<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, stats, gamma, relu_in, p, seed, gx, ldgx, partials, n, di, seed_base); break;
      default: ln_bwd_kernel<64><<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, stats, gamma, relu_in, p, seed, gx, ldgx, partials, n, di, seed_base); break;
    }
  } else {
    // generic path: block b owns partial row b; rows past the grid (if the caller sized for the fast path) are zeroed
    const int64_t want = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t cap = n_partials < 512 ? n_partials : 512;
    const unsigned grid = static_cast<unsigned>(want > cap ? cap : want);
    if (n_partials > grid)
      ALLSET_HIP_CHECK(hipMemsetAsync(partials + static_cast<size_t>(grid) * 2 * d, 0,
                                      static_cast<size_t>(n_partials - grid) * 2 * d * sizeof(float), st));
    if (d <= 8 * kWave) ln_bwd_rows_kernel<8><<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, stats, gamma, relu_in, p, seed, gx, ldgx, partials, n, di, seed_base);
    else if (d <= 16 * kWave) ln_bwd_rows_kernel<16><<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, stats, gamma, relu_in, p, seed, gx, ldgx, partials, n, di, seed_base);
    else if (d <= 24 * kWave) ln_bwd_rows_kernel<24><<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, stats, gamma, relu_in, p, seed, gx, ldgx, partials, n, di, seed_base);
    else ln_bwd_generic_kernel<<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, stats, gamma, relu_in, p, seed, gx, ldgx, partials, n, di, seed_base);
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_relu_dropout_fwd(const float* x, float p, uint64_t seed, float* y, int64_t numel,
                                       const uint64_t* seed_base, void* stream) {
  clear_error();
  ALLSET_REQUIRE(numel >= 0, "relu_dropout_fwd: negative size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "relu_dropout_fwd: dropout p must be in [0,1)");
  if (numel == 0) return ALLSET_OK;
  ALLSET_REQUIRE(x && y, "relu_dropout_fwd: null pointer");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t done = 0;
  if (aligned16(x) && aligned16(y) && numel >= 4) {
    const int64_t n4 = numel / 4;
    const int64_t want = (n4 + kBlock - 1) / kBlock;
    relu_dropout_fwd_kernel<<<static_cast<unsigned>(want > 8192 ? 8192 : want), kBlock, 0, st>>>(x, p, seed, y, n4, seed_base);
    done = n4 * 4;
  }
  if (done < numel) {
    const int64_t rest = numel - done;
    relu_dropout_fwd_scalar_kernel<<<static_cast<unsigned>((rest + kBlock - 1) / kBlock), kBlock, 0, st>>>(x, p, seed, y, numel, done, seed_base);
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_relu_dropout_bwd(const float* gy, const float* y, float p, float* gx, int64_t numel, void* stream) {
  clear_error();
  ALLSET_REQUIRE(numel >= 0, "relu_dropout_bwd: negative size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "relu_dropout_bwd: dropout p must be in [0,1)");
  if (numel == 0) return ALLSET_OK;
  ALLSET_REQUIRE(gy && y && gx, "relu_dropout_bwd: null pointer");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t done = 0;
  if (aligned16(gy) && aligned16(y) && aligned16(gx) && numel >= 4) {
    const int64_t n4 = numel / 4;
    const int64_t want = (n4 + kBlock - 1) / kBlock;
    relu_dropout_bwd_kernel<<<static_cast<unsigned>(want > 8192 ? 8192 : want), kBlock, 0, st>>>(gy, y, p, gx, n4);
    done = n4 * 4;
  }
  if (done < numel) {
    const int64_t rest = numel - done;
    relu_dropout_bwd_scalar_kernel<<<static_cast<unsigned>((rest + kBlock - 1) / kBlock), kBlock, 0, st>>>(gy, y, p, gx, numel, done);
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

// the launch grid of wgrad_x6_kernel: several tiles per slice -> y rounded up to a multiple of 8 and the XCD-aware remap switched on
static inline dim3 wgrad_grid(int tiles, int64_t n_slices, WgradPro& pro) {
  if (tiles == 1) { pro.n_slices = 0; return dim3(1u, static_cast<unsigned>(n_slices)); }
  pro.n_slices = static_cast<int>(n_slices);
  return dim3(static_cast<unsigned>(tiles), static_cast<unsigned>((n_slices + 7) / 8 * 8));
}

extern "C" int allset_wgrad_slices(int64_t n, int64_t O, int64_t I, int64_t* n_slices) {
  clear_error();
  ALLSET_REQUIRE(n_slices != nullptr && n >= 0 && O >= 1 && I >= 1, "wgrad_slices: bad argument");
  const int64_t tiles = ((O + kWgTile - 1) / kWgTile) * ((I + kWgTile - 1) / kWgTile);
  // aim at ~512 workgroups in total (two 64-KiB-LDS workgroups per CU), at least 256 rows (8 stages) per slice; fewer slices also
  // means fewer partial tiles for the caller to sum.  Where that leaves most of the chip idle (dataset scale: 3.3k rows of a 256 x 256
  // weight are 52 workgroups) slices go down to 128 rows (round 6: the launch is one workgroup's chain of stages; measured on the
  // tuned-width steps, -5 % at 256 x 256 -- at 512 x 512, already 208 workgroups, twice the partial tiles cost +4 %, so not there)
  int64_t s = 512 / tiles;
  int64_t max_by_rows = (n + 255) / 256;
  if ((s < max_by_rows ? s : max_by_rows) * tiles < 128) max_by_rows = (n + 127) / 128;
  if (s > max_by_rows) s = max_by_rows;
  if (s < 1) s = 1;
  *n_slices = s;
  return ALLSET_OK;
}

extern "C" int allset_wgrad(const float* ga, int64_t lda, const float* u, int64_t ldu, float* part_w, float* part_b,
                            int64_t n_slices, int64_t n, int64_t O, int64_t I, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && O >= 1 && I >= 1 && O < INT32_MAX && I < INT32_MAX, "wgrad: bad size");
  ALLSET_REQUIRE(n_slices >= 1 && n_slices < 65536, "wgrad: bad slice count");
  ALLSET_REQUIRE(part_w != nullptr, "wgrad: null partial buffer");
  ALLSET_REQUIRE(n == 0 || (ga && u), "wgrad: null input");
  if (O % 4 != 0 || I % 4 != 0 || lda % 4 != 0 || ldu % 4 != 0 || !aligned16(ga) || !aligned16(u)) {
    set_error("wgrad: needs out/in features and leading dimensions that are multiples of 4 and 16-byte aligned inputs");
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(lda >= O && ldu >= I, "wgrad: leading dimension smaller than the feature width");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles_o = static_cast<int>((O + kWgTile - 1) / kWgTile), tiles_i = static_cast<int>((I + kWgTile - 1) / kWgTile);
  int64_t rows_per_slice = (n + n_slices - 1) / n_slices;
  rows_per_slice = (rows_per_slice + kWgRows - 1) / kWgRows * kWgRows;
  if (rows_per_slice < kWgRows) rows_per_slice = kWgRows;
  WgradPro pro{};
  const dim3 grid = wgrad_grid(tiles_o * tiles_i, n_slices, pro);
  wgrad_x6_kernel<false><<<grid, kWx6Block, 0, st>>>(ga, lda, u, ldu, part_w, part_b, n, static_cast<int>(O),
                                                     static_cast<int>(I), tiles_i, rows_per_slice, pro);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static int wgrad_fused_impl(const float* gy, int64_t ldg, const float* y, int64_t ldy, float p_out,
                            const float* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                            int relu_in, float p_in, uint64_t seed_in, float* part_w, float* part_b, int64_t pw_stride,
                            int64_t pb_stride, int64_t n_slices, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base,
                            const uint32_t* mask, void* stream);

extern "C" int allset_wgrad_fused(const float* gy, int64_t ldg, const float* y, int64_t ldy, float p_out,
                                  const float* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                                  int relu_in, float p_in, uint64_t seed_in, float* part_w, float* part_b,
                                  int64_t n_slices, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base,
                                  const uint32_t* mask, void* stream) {
  clear_error();
  return wgrad_fused_impl(gy, ldg, y, ldy, p_out, x, ldx, stats, gamma, beta, relu_in, p_in, seed_in, part_w, part_b, 0, 0, n_slices, n,
                          O, I, seed_base, mask, stream);
}

extern "C" int allset_wgrad_fused_ex(const float* gy, int64_t ldg, const float* y, int64_t ldy, float p_out,
                                     const float* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                                     int relu_in, float p_in, uint64_t seed_in, float* part, int64_t part_stride, int want_bias,
                                     int64_t n_slices, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base,
                                     const uint32_t* mask, void* stream) {
  clear_error();
  ALLSET_REQUIRE(part != nullptr && part_stride >= O * I + (want_bias ? O : 0) && part_stride % 4 == 0 && aligned16(part),
                 "wgrad_fused_ex: part must be 16-byte aligned rows of at least O*I (+O) floats, stride a multiple of 4");
  return wgrad_fused_impl(gy, ldg, y, ldy, p_out, x, ldx, stats, gamma, beta, relu_in, p_in, seed_in, part,
                          want_bias ? part + O * I : nullptr, part_stride, part_stride, n_slices, n, O, I, seed_base, mask, stream);
}

static int wgrad_fused_impl(const float* gy, int64_t ldg, const float* y, int64_t ldy, float p_out,
                            const float* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                            int relu_in, float p_in, uint64_t seed_in, float* part_w, float* part_b, int64_t pw_stride,
                            int64_t pb_stride, int64_t n_slices, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base,
                            const uint32_t* mask, void* stream) {
  ALLSET_REQUIRE(n >= 0 && O >= 1 && I >= 1 && O < INT32_MAX && I < INT32_MAX, "wgrad_fused: bad size");
  if (mask != nullptr && O % 64 != 0) {
    set_error("wgrad_fused: the activation mask needs out features % 64 == 0");
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(n_slices >= 1 && n_slices < 65536, "wgrad_fused: bad slice count");
  ALLSET_REQUIRE(part_w != nullptr, "wgrad_fused: null partial buffer");
  ALLSET_REQUIRE(n == 0 || (gy && x), "wgrad_fused: null input");
  ALLSET_REQUIRE(p_in >= 0.f && p_in < 1.f && p_out >= 0.f && p_out < 1.f, "wgrad_fused: dropout p must be in [0,1)");
  ALLSET_REQUIRE((stats == nullptr) == (gamma == nullptr) && (gamma == nullptr) == (beta == nullptr),
                 "wgrad_fused: stats, gamma and beta must come together");
  if (O % 4 != 0 || I % 4 != 0 || ldg % 4 != 0 || ldx % 4 != 0 || !aligned16(gy) || !aligned16(x) ||
      (y != nullptr && (ldy % 4 != 0 || !aligned16(y))) || (gamma != nullptr && (!aligned16(gamma) || !aligned16(beta)))) {
    set_error("wgrad_fused: needs feature widths / leading dimensions that are multiples of 4 and 16-byte aligned inputs");
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(ldg >= O && ldx >= I && (y == nullptr || ldy >= O), "wgrad_fused: leading dimension too small");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles_o = static_cast<int>((O + kWgTile - 1) / kWgTile), tiles_i = static_cast<int>((I + kWgTile - 1) / kWgTile);
  int64_t rows_per_slice = (n + n_slices - 1) / n_slices;
  rows_per_slice = (rows_per_slice + kWgRows - 1) / kWgRows * kWgRows;
  if (rows_per_slice < kWgRows) rows_per_slice = kWgRows;
  WgradPro pro;
  pro.y = y; pro.ldy = ldy; pro.keep_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  pro.stats = stats; pro.gamma = gamma; pro.beta = beta; pro.has_ln = stats != nullptr;
  pro.relu_in = relu_in; pro.p_in = p_in; pro.seed_in = seed_in; pro.seed_base = seed_base;
  pro.mask = mask; pro.mask_nh = static_cast<int>(O / 64);
  pro.pw_stride = pw_stride; pro.pb_stride = pb_stride;
  const dim3 grid = wgrad_grid(tiles_o * tiles_i, n_slices, pro);
  // no operand prologue at all (allset_wgrad through the one-buffer entry): the plain instantiation, 12 % faster
  const bool plain = y == nullptr && mask == nullptr && stats == nullptr && !relu_in && p_in == 0.f && p_out == 0.f;
  if (plain)
    wgrad_x6_kernel<false><<<grid, kWx6Block, 0, st>>>(gy, ldg, x, ldx, part_w, part_b, n, static_cast<int>(O),
                                                       static_cast<int>(I), tiles_i, rows_per_slice, pro);
  else {
#define ALLSET_WX6(MKV, LNV) wgrad_x6_kernel<true, MKV, LNV><<<grid, kWx6Block, 0, st>>>(gy, ldg, x, ldx, part_w, part_b, n, static_cast<int>(O), \
                                                                                      static_cast<int>(I), tiles_i, rows_per_slice, pro)
    const int mk = mask != nullptr ? 2 : (y != nullptr ? 1 : 0);
    const bool ln = stats != nullptr;
    if (mk == 2) { if (ln) ALLSET_WX6(2, true); else ALLSET_WX6(2, false); }
    else if (mk == 1) { if (ln) ALLSET_WX6(1, true); else ALLSET_WX6(1, false); }
    else { if (ln) ALLSET_WX6(0, true); else ALLSET_WX6(0, false); }
#undef ALLSET_WX6
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static inline bool reduce_one_launch(int64_t P, int64_t M) {
  const int64_t slabs = (P + kRedRows - 1) / kRedRows;
  return slabs == 1 || (slabs <= 8 && P * M <= (int64_t{1} << 21));
}

static int reduce_partials_impl(const float* part, int64_t P, int64_t row_stride, int64_t M, void* out, int out_bf16, float* scratch,
                                void* stream);

extern "C" int allset_reduce_partials(const float* part, int64_t P, int64_t M, float* out, float* scratch, void* stream) {
  clear_error();
  return reduce_partials_impl(part, P, M, M, out, 0, scratch, stream);
}

extern "C" int allset_reduce_partials_ex(const float* part, int64_t P, int64_t row_stride, int64_t M, void* out, int out_dtype,
                                         float* scratch, void* stream) {
  clear_error();
  ALLSET_REQUIRE(out_dtype == ALLSET_F32 || out_dtype == ALLSET_BF16, "reduce_partials_ex: out_dtype must be ALLSET_F32 or ALLSET_BF16");
  ALLSET_REQUIRE(row_stride >= M && row_stride % 4 == 0, "reduce_partials_ex: row_stride must be >= M and a multiple of 4");
  return reduce_partials_impl(part, P, row_stride, M, out, out_dtype == ALLSET_BF16, scratch, stream);
}

static int reduce_partials_impl(const float* part, int64_t P, int64_t row_stride, int64_t M, void* out_v, int out_bf16, float* scratch,
                                void* stream) {
  float* out = static_cast<float*>(out_v);
  ALLSET_REQUIRE(P >= 1 && M >= 1, "reduce_partials: bad size");
  ALLSET_REQUIRE(part && out, "reduce_partials: null pointer");
  if (M % 4 != 0 || !aligned16(part) || (reinterpret_cast<uintptr_t>(out) & (out_bf16 ? 7u : 15u)) || (scratch && !aligned16(scratch))) {
    set_error("reduce_partials: M must be a multiple of 4 and the buffers 16-byte aligned");
    return ALLSET_ERR_UNSUPPORTED;
  }
  const int64_t slabs = (P + kRedRows - 1) / kRedRows;
  ALLSET_REQUIRE(slabs == 1 || scratch != nullptr, "reduce_partials: P > %d needs a scratch buffer of ceil(P/%d)*M floats", kRedRows, kRedRows);   // (not touched by the one-launch path)
  ALLSET_REQUIRE(slabs <= kRedRows, "reduce_partials: at most %d partial rows", kRedRows * kRedRows);
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t quads = M / 4, per_block = kBlock / kRedSplit;
  const unsigned gx = static_cast<unsigned>((quads + per_block - 1) / per_block);
  // small reductions (dataset-scale steps are launch-bound: 16 of a Cora step's 63 kernels were these): one launch, each
  // workgroup walks all slabs; large ones keep the two-level tree (1024 x 16.5k partials want every CU)
  const bool one_launch = reduce_one_launch(P, M);
  if (one_launch) {
    const int sh = static_cast<int>(slabs);
    if (out_bf16) reduce_partials_kernel<true><<<dim3(gx, 1), kBlock, 0, st>>>(part, P, M, out, row_stride, sh);
    else reduce_partials_kernel<false><<<dim3(gx, 1), kBlock, 0, st>>>(part, P, M, out, row_stride, sh);
  } else {
    reduce_partials_kernel<false><<<dim3(gx, static_cast<unsigned>(slabs)), kBlock, 0, st>>>(part, P, M, scratch, row_stride, 1);
    if (out_bf16) reduce_partials_kernel<true><<<dim3(gx, 1), kBlock, 0, st>>>(scratch, slabs, M, out, M, 1);
    else reduce_partials_kernel<false><<<dim3(gx, 1), kBlock, 0, st>>>(scratch, slabs, M, out, M, 1);
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_reduce_partials_batch_max(void) { return kRedBatchMax; }

// 1 when the batched entry takes a (P, M) reduction: the ones allset_reduce_partials finishes in ONE launch and (ABI 14) the
// two-launch ones of at most 8 slabs of 64 partial rows, which one workgroup per column block walks with the tree's association.
extern "C" int allset_reduce_partials_batchable(int64_t P, int64_t M) {
  return (P >= 1 && P <= 8 * kRedRows && M >= 4 && M % 4 == 0 && M < INT32_MAX && P * M < (int64_t{1} << 31)) ? 1 : 0;
}

static int reduce_partials_batched_impl(const float* const* parts, const int64_t* P, const int64_t* row_stride, const int64_t* M,
                                        void* const* outs, const float* const* accs, const int32_t* out_dtypes, int64_t count,
                                        int64_t* inc_i64, float* const* inc_f32, int64_t n_inc_f32, void* stream) {
  ALLSET_REQUIRE(count >= 0 && count <= kRedBatchMax, "reduce_partials_batched: at most %d buffers per call", kRedBatchMax);
  ALLSET_REQUIRE(n_inc_f32 >= 0 && n_inc_f32 <= kRedBatchMaxCounters && (n_inc_f32 == 0 || inc_f32 != nullptr),
                 "reduce_partials_batched: at most %d float counters per call", kRedBatchMaxCounters);
  const bool bump = inc_i64 != nullptr || n_inc_f32 > 0;
  if (count == 0 && !bump) return ALLSET_OK;
  ALLSET_REQUIRE(count == 0 || (parts && P && row_stride && M && outs), "reduce_partials_batched: null pointer");
  RedBatchTable tb;
  int64_t blocks = 0;
  const int64_t per_block = kBlock / kRedSplit;
  for (int64_t k = 0; k < count; ++k) {
    ALLSET_REQUIRE(parts[k] && outs[k], "reduce_partials_batched: null buffer %lld", static_cast<long long>(k));
    ALLSET_REQUIRE(allset_reduce_partials_batchable(P[k], M[k]), "reduce_partials_batched: buffer %lld is not batchable (see allset_reduce_partials_batchable)", static_cast<long long>(k));
    const int odt = out_dtypes ? (out_dtypes[k] & 0xff) : ALLSET_F32;
    const bool force_tree = out_dtypes != nullptr && (out_dtypes[k] & ALLSET_REDUCE_AS_TREE) != 0;
    ALLSET_REQUIRE((out_dtypes == nullptr || (out_dtypes[k] & ~(0xff | ALLSET_REDUCE_AS_TREE)) == 0) && (odt == ALLSET_F32 || odt == ALLSET_BF16), "reduce_partials_batched: buffer %lld: out dtype must be ALLSET_F32 or ALLSET_BF16", static_cast<long long>(k));
    ALLSET_REQUIRE(row_stride[k] >= M[k] && row_stride[k] % 4 == 0 && row_stride[k] < INT32_MAX && aligned16(parts[k]) &&
                   (reinterpret_cast<uintptr_t>(outs[k]) & (odt == ALLSET_BF16 ? 7u : 15u)) == 0,
                   "reduce_partials_batched: buffer %lld: row_stride >= M, a multiple of 4; 16-byte aligned pointers (8 for a bf16 output)", static_cast<long long>(k));
    tb.part[k] = parts[k]; tb.out[k] = static_cast<float*>(outs[k]);
    tb.out_bf16[k] = odt == ALLSET_BF16 ? 1 : 0;
    tb.acc[k] = accs ? accs[k] : nullptr;
    ALLSET_REQUIRE(tb.acc[k] == nullptr || (odt == ALLSET_F32 && aligned16(tb.acc[k])),
                   "reduce_partials_batched: buffer %lld: an accumulated-into gradient needs an fp32 output and 16-byte alignment", static_cast<long long>(k));
    tb.tree[k] = (force_tree || !reduce_one_launch(P[k], M[k])) ? 1 : 0;
    tb.P[k] = static_cast<int32_t>(P[k]); tb.stride[k] = static_cast<int32_t>(row_stride[k]); tb.M[k] = static_cast<int32_t>(M[k]);
    tb.first_block[k] = static_cast<int32_t>(blocks);
    blocks += (M[k] / 4 + per_block - 1) / per_block;
  }
  tb.first_block[count] = static_cast<int32_t>(blocks);
  tb.count = static_cast<int32_t>(count);
  tb.inc_i64 = inc_i64;
  tb.n_inc_f32 = static_cast<int32_t>(n_inc_f32);
  for (int64_t k = 0; k < n_inc_f32; ++k) {
    ALLSET_REQUIRE(inc_f32[k] != nullptr, "reduce_partials_batched: null counter %lld", static_cast<long long>(k));
    tb.inc_f32[k] = inc_f32[k];
  }
  reduce_partials_batched_kernel<<<static_cast<unsigned>(blocks + (bump ? 1 : 0)), kBlock, 0, static_cast<hipStream_t>(stream)>>>(tb);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_reduce_partials_batched(const float* const* parts, const int64_t* P, const int64_t* row_stride, const int64_t* M,
                                              float* const* outs, int64_t count, void* stream) {
  clear_error();
  return reduce_partials_batched_impl(parts, P, row_stride, M, reinterpret_cast<void* const*>(outs), nullptr, nullptr, count, nullptr, nullptr, 0, stream);
}

extern "C" int allset_reduce_partials_batched_ex(const float* const* parts, const int64_t* P, const int64_t* row_stride, const int64_t* M,
                                                 float* const* outs, int64_t count, int64_t* inc_i64, float* const* inc_f32,
                                                 int64_t n_inc_f32, void* stream) {
  clear_error();
  return reduce_partials_batched_impl(parts, P, row_stride, M, reinterpret_cast<void* const*>(outs), nullptr, nullptr, count, inc_i64, inc_f32, n_inc_f32, stream);
}

// 1 when allset_reduce_partials(_ex) sums a (P, M) buffer as the two-launch tree (whose association differs from the one-launch form's)
extern "C" int allset_reduce_partials_is_tree(int64_t P, int64_t M) { return (P >= 1 && M >= 1 && !reduce_one_launch(P, M)) ? 1 : 0; }

// ABI 14: the same with an output type per buffer (out_dtypes[k] = ALLSET_F32 or ALLSET_BF16: outs[k] is then a bf16 vector of M[k]
// elements, each sum rounded once as allset_reduce_partials_ex rounds it; NULL = all fp32) and an optional accumuland per buffer
// (accs[k]: f32[M[k]] added to the sum before it is written -- a gradient that already exists; NULL array or NULL entry = none).
extern "C" int allset_reduce_partials_batched_ex2(const float* const* parts, const int64_t* P, const int64_t* row_stride, const int64_t* M,
                                                  void* const* outs, const float* const* accs, const int32_t* out_dtypes, int64_t count,
                                                  int64_t* inc_i64, float* const* inc_f32, int64_t n_inc_f32, void* stream) {
  clear_error();
  return reduce_partials_batched_impl(parts, P, row_stride, M, outs, accs, out_dtypes, count, inc_i64, inc_f32, n_inc_f32, stream);
}

extern "C" int allset_reduce_partials_batch_max_counters(void) { return kRedBatchMaxCounters; }

extern "C" int allset_ln_res_supported(int64_t d) { return (d >= 4 && d <= 512 && d % 4 == 0) ? 1 : 0; }

extern "C" int allset_ln_res_fwd(const float* x, int64_t ldx, const float* colb, const float* res, int64_t ldr,
                                 const float* gamma, const float* beta, float eps, int relu_out, float p, uint64_t seed,
                                 float* y, int64_t ldy, float* stats, int64_t n, int64_t d, const uint64_t* seed_base,
                                 void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1, "ln_res_fwd: bad size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "ln_res_fwd: dropout p must be in [0,1)");
  if (!allset_ln_res_supported(d)) { set_error("ln_res_fwd: width %lld not built (d %% 4 == 0, d <= 512)", static_cast<long long>(d)); return ALLSET_ERR_UNSUPPORTED; }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(x && gamma && beta && y && stats, "ln_res_fwd: null pointer");
  ALLSET_REQUIRE(ldx >= d && ldy >= d && (res == nullptr || ldr >= d), "ln_res_fwd: leading dimension smaller than d");
  ALLSET_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) &&
                 (colb == nullptr || aligned16(colb)) && (res == nullptr || (ldr % 4 == 0 && aligned16(res))),
                 "ln_res_fwd: rows and parameter vectors must be 16-byte aligned");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int di = static_cast<int>(d), lpr = ln_lpr(d);
  const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr) * (d > 256 ? 2 : kLnRowsPerGroup);
  const unsigned grid = static_cast<unsigned>((n + rows_per_block - 1) / rows_per_block);
#define ALLSET_LNRES_FWD(L, V) ln_res_fwd_kernel<L, V><<<grid, kBlock, 0, st>>>(x, ldx, colb, res, ldr, gamma, beta, eps, relu_out, p, seed, y, ldy, stats, n, di, seed_base)
  if (d > 256) ALLSET_LNRES_FWD(64, 2);             // two 16-byte chunks per lane: 256 < d <= 512
  else switch (lpr) {
    case 8: ALLSET_LNRES_FWD(8, 1); break;
    case 16: ALLSET_LNRES_FWD(16, 1); break;
    case 32: ALLSET_LNRES_FWD(32, 1); break;
    default: ALLSET_LNRES_FWD(64, 1); break;
  }
#undef ALLSET_LNRES_FWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_ln_res_bwd_partials(int64_t n, int64_t d, int64_t* n_partials) {
  clear_error();
  ALLSET_REQUIRE(n_partials != nullptr && n >= 0 && allset_ln_res_supported(d), "ln_res_bwd_partials: bad argument");
  const int lpr = ln_lpr(d);
  const int64_t groups = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr);
  const int64_t want = (n + groups - 1) / groups;
  // persistent grid: 78 VGPRs = 6 waves per SIMD = 6 workgroups of 4 waves per CU -> 1536 resident workgroups; with 2048 the last
  // quarter runs as a second, third-full round (0.315 -> see profiles/r02_ln_res_bench.txt)
  const int64_t cap = 1536;
  int64_t np = want < 1 ? 1 : (want > cap ? cap : want);
  // a grid that one launch fills anyway (<= 1536 workgroups) keeps at most 512 partial rows: what the batched reduction of a
  // backward pass takes (allset_reduce_partials_batchable) -- at dataset scale these four reductions are launches of their own otherwise
  if (np > 512 && want <= cap) np = 512;
  *n_partials = np;
  return ALLSET_OK;
}

static int ln_res_bwd_impl(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* colb,
                           const float* res, int64_t ldr, const float* stats, const float* gamma, const float* beta,
                           int relu_out, float p, uint64_t seed, float* gs, int64_t ldgs, float* partials,
                           int64_t n_partials, int64_t n, int64_t d, const uint64_t* seed_base, const float* pma_m,
                           const float* pma_l, float* pma_stats, int64_t pma_heads, void* stream);

extern "C" int allset_ln_res_bwd(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* colb,
                                 const float* res, int64_t ldr, const float* stats, const float* gamma, const float* beta,
                                 int relu_out, float p, uint64_t seed, float* gs, int64_t ldgs, float* partials,
                                 int64_t n_partials, int64_t n, int64_t d, const uint64_t* seed_base, void* stream) {
  return ln_res_bwd_impl(gy, ldg, x, ldx, colb, res, ldr, stats, gamma, beta, relu_out, p, seed, gs, ldgs, partials, n_partials, n, d,
                         seed_base, nullptr, nullptr, nullptr, 1, stream);
}

extern "C" int allset_ln_res_bwd_pma_supported(int64_t d, int64_t heads) {
  if (!allset_ln_res_supported(d) || heads < 1 || d % heads != 0 || (d / heads) % 4 != 0) return 0;
  const int64_t g = (d / heads) / 4;                  // lanes per head: a power of two, inside one 64-lane chunk row
  return ((g & (g - 1)) == 0 && g <= 64) ? 1 : 0;
}

extern "C" int allset_ln_res_bwd_pma(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* colb,
                                     const float* stats, const float* gamma, const float* beta, float* gs, int64_t ldgs,
                                     float* partials, int64_t n_partials, int64_t n, int64_t d, const float* pma_m,
                                     const float* pma_l, float* pma_stats, int64_t heads, void* stream) {
  clear_error();
  if (!allset_ln_res_bwd_pma_supported(d, heads)) {
    set_error("ln_res_bwd_pma: d=%lld heads=%lld not built (channels per head must be 4 x a power of two)",
              static_cast<long long>(d), static_cast<long long>(heads));
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(n == 0 || (pma_m && pma_l && pma_stats), "ln_res_bwd_pma: null statistics pointer");
  ALLSET_REQUIRE(pma_stats == nullptr || (reinterpret_cast<uintptr_t>(pma_stats) & 7u) == 0, "ln_res_bwd_pma: stats must be 8-byte aligned");
  return ln_res_bwd_impl(gy, ldg, x, ldx, colb, nullptr, 0, stats, gamma, beta, 0, 0.f, 0, gs, ldgs, partials, n_partials, n, d, nullptr,
                         pma_m, pma_l, pma_stats, heads, stream);
}

static int ln_res_bwd_impl(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* colb,
                           const float* res, int64_t ldr, const float* stats, const float* gamma, const float* beta,
                           int relu_out, float p, uint64_t seed, float* gs, int64_t ldgs, float* partials,
                           int64_t n_partials, int64_t n, int64_t d, const uint64_t* seed_base, const float* pma_m,
                           const float* pma_l, float* pma_stats, int64_t pma_heads, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1, "ln_res_bwd: bad size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "ln_res_bwd: dropout p must be in [0,1)");
  if (!allset_ln_res_supported(d)) { set_error("ln_res_bwd: width %lld not built", static_cast<long long>(d)); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(partials != nullptr && n_partials >= 1, "ln_res_bwd: partials buffer required");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(partials, 0, static_cast<size_t>(n_partials) * 3 * d * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && x && stats && gamma && beta && gs, "ln_res_bwd: null pointer");
  ALLSET_REQUIRE(ldg >= d && ldx >= d && ldgs >= d && (res == nullptr || ldr >= d), "ln_res_bwd: leading dimension smaller than d");
  ALLSET_REQUIRE(ldg % 4 == 0 && ldx % 4 == 0 && ldgs % 4 == 0 && aligned16(gy) && aligned16(x) && aligned16(gs) &&
                 aligned16(gamma) && aligned16(beta) && (colb == nullptr || aligned16(colb)) &&
                 (res == nullptr || (ldr % 4 == 0 && aligned16(res))), "ln_res_bwd: rows and parameter vectors must be 16-byte aligned");
  const int di = static_cast<int>(d);
  const unsigned grid = static_cast<unsigned>(n_partials);
#define ALLSET_LNRES_BWD(L, V) ln_res_bwd_kernel<L, V><<<grid, kBlock, 0, st>>>(gy, ldg, x, ldx, colb, res, ldr, stats, gamma, beta, relu_out, p, seed, gs, ldgs, partials, n, di, seed_base, pma_m, pma_l, pma_stats, static_cast<int>(pma_heads))
  if (d > 256) ALLSET_LNRES_BWD(64, 2);
  else switch (ln_lpr(d)) {
    case 8: ALLSET_LNRES_BWD(8, 1); break;
    case 16: ALLSET_LNRES_BWD(16, 1); break;
    case 32: ALLSET_LNRES_BWD(32, 1); break;
    default: ALLSET_LNRES_BWD(64, 1); break;
  }
#undef ALLSET_LNRES_BWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static inline bool wgrad_bf16_full_width(int64_t O, int64_t I) {
  return (O == 64 || O == 128 || O == 256) && (I == 64 || I == 128 || I == 256);
}

extern "C" int allset_wgrad_bf16_slices(int64_t n, int64_t O, int64_t I, int64_t* n_slices) {
  clear_error();
  ALLSET_REQUIRE(n_slices != nullptr && n >= 0 && O >= 1 && I >= 1, "wgrad_bf16_slices: bad argument");
  if (!wgrad_bf16_full_width(O, I)) return allset_wgrad_slices(n, O, I, n_slices);
  // one workgroup owns the whole gW of its rows: two 64-KiB-LDS workgroups per CU at most, at least 256 rows per slice
  int64_t s = (O == 256 && I == 256) ? 256 : 512;
  const int64_t max_by_rows = (n + 255) / 256;
  if (s > max_by_rows) s = max_by_rows;
  if (s < 1) s = 1;
  *n_slices = s;
  return ALLSET_OK;
}

static int wgrad_bf16_impl(const void* ga, int64_t lda, const void* bits, const float* g4, const void* u, int64_t ldu, float* part_w,
                           float* part_b, float* part_x, int64_t pw_stride, int64_t pb_stride, int64_t n_slices, int64_t n, int64_t O,
                           int64_t I, void* stream);

extern "C" int allset_wgrad_bf16(const void* ga, int64_t lda, const void* u, int64_t ldu, float* part_w, float* part_b,
                                 int64_t n_slices, int64_t n, int64_t O, int64_t I, void* stream) {
  clear_error();
  return wgrad_bf16_impl(ga, lda, nullptr, nullptr, u, ldu, part_w, part_b, nullptr, O * I, O, n_slices, n, O, I, stream);
}

// 1 when allset_wgrad_bf16_ex2 takes `bits` / `g4` at these widths (the full-width kernel; g4 needs O = I = 256 or 128-wide O)
extern "C" int allset_wgrad_bf16_ex2_supported(int64_t O, int64_t I, int has_bits, int has_aux) {
  if (!((O == 128 || O == 256) && (I == 128 || I == 256))) return 0;
  (void)has_bits;
  return (!has_aux || (O == 256 && I == 256)) ? 1 : 0;
}

// allset_wgrad_bf16_ex with the relu BIT mask of `ga` applied on the way (bits: allset_linear_bf16_fwd_mask's output; may be
// NULL) and, with g4 (fp32 [n, 4], may be NULL), the weight gradient of four auxiliary output columns in the same pass: the
// partial row is [gW (O x I) | gb (O, if want_bias) | gWa (4 x I) | gba (4)] -- the last two only with g4.
extern "C" int allset_wgrad_bf16_ex2(const void* ga, int64_t lda, const void* bits, const float* g4, const void* u, int64_t ldu,
                                     float* part, int64_t part_stride, int want_bias, int64_t n_slices, int64_t n, int64_t O,
                                     int64_t I, void* stream) {
  clear_error();
  const int64_t need = O * I + (want_bias ? O : 0) + (g4 ? 4 * I + 4 : 0);
  ALLSET_REQUIRE(part != nullptr && part_stride >= need && part_stride % 4 == 0 && aligned16(part),
                 "wgrad_bf16_ex2: part must be 16-byte aligned rows of at least O*I (+O) (+4I+4) floats, stride a multiple of 4");
  if (!allset_wgrad_bf16_ex2_supported(O, I, bits != nullptr, g4 != nullptr) || lda % 8 != 0 || ldu % 8 != 0 || !aligned16(ga) || !aligned16(u)) {
    set_error("wgrad_bf16_ex2: O, I in {128, 256} (g4: 256 x 256), 16-byte aligned rows with leading dimensions that are multiples of 8");
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(g4 == nullptr || aligned16(g4), "wgrad_bf16_ex2: g4 must be 16-byte aligned");
  float* pb = want_bias ? part + O * I : nullptr;
  float* px = g4 ? part + O * I + (want_bias ? O : 0) : nullptr;
  return wgrad_bf16_impl(ga, lda, bits, g4, u, ldu, part, pb, px, part_stride, part_stride, n_slices, n, O, I, stream);
}

extern "C" int allset_wgrad_bf16_ex(const void* ga, int64_t lda, const void* u, int64_t ldu, float* part, int64_t part_stride,
                                    int want_bias, int64_t n_slices, int64_t n, int64_t O, int64_t I, void* stream) {
  clear_error();
  ALLSET_REQUIRE(part != nullptr && part_stride >= O * I + (want_bias ? O : 0) && part_stride % 4 == 0 && aligned16(part),
                 "wgrad_bf16_ex: part must be 16-byte aligned rows of at least O*I (+O) floats, stride a multiple of 4");
  return wgrad_bf16_impl(ga, lda, nullptr, nullptr, u, ldu, part, want_bias ? part + O * I : nullptr, nullptr, part_stride, part_stride, n_slices, n, O, I, stream);
}

static int wgrad_bf16_impl(const void* ga, int64_t lda, const void* bits, const float* g4, const void* u, int64_t ldu, float* part_w,
                           float* part_b, float* part_x, int64_t pw_stride, int64_t pb_stride, int64_t n_slices, int64_t n, int64_t O,
                           int64_t I, void* stream) {
  ALLSET_REQUIRE(n >= 0 && O >= 1 && I >= 1 && O < INT32_MAX && I < INT32_MAX, "wgrad_bf16: bad size");
  ALLSET_REQUIRE(n_slices >= 1 && n_slices < 65536, "wgrad_bf16: bad slice count");
  ALLSET_REQUIRE(part_w != nullptr, "wgrad_bf16: null partial buffer");
  ALLSET_REQUIRE(n == 0 || (ga && u), "wgrad_bf16: null input");
  if (O % 4 != 0 || I % 4 != 0 || lda % 4 != 0 || ldu % 4 != 0 || (reinterpret_cast<uintptr_t>(ga) & 7u) || (reinterpret_cast<uintptr_t>(u) & 7u)) {
    set_error("wgrad_bf16: needs feature widths / leading dimensions that are multiples of 4 and 8-byte aligned inputs");
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(lda >= O && ldu >= I, "wgrad_bf16: leading dimension smaller than the feature width");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (wgrad_bf16_full_width(O, I) && lda % 8 == 0 && ldu % 8 == 0 && aligned16(ga) && aligned16(u)) {
    // the full-width kernel: one read of each operand (the tiled one below re-reads them per 128 x 128 tile)
    int64_t rps = (n + n_slices - 1) / n_slices;
    rps = (rps + 31) / 32 * 32;
    if (rps < 32) rps = 32;
    const unsigned grid1 = static_cast<unsigned>(n_slices);
    const uint16_t* a16 = static_cast<const uint16_t*>(ga);
    const uint16_t* u16 = static_cast<const uint16_t*>(u);
    const uint8_t* b8 = static_cast<const uint8_t*>(bits);
#define ALLSET_WGTR_X(OTN, ITN, MK, AX) wgrad_bf16_tr_kernel<OTN, ITN, MK, AX><<<grid1, kWx6Block, 0, st>>>(a16, lda, b8, g4, u16, ldu, part_w, part_b, part_x, n, rps, pw_stride, pb_stride)
#define ALLSET_WGTR(OTN, ITN) ALLSET_WGTR_X(OTN, ITN, false, false)
    const int otn = static_cast<int>(O / 64), itn = static_cast<int>(I / 32);
    if (b8 != nullptr || g4 != nullptr) {
      ALLSET_REQUIRE((otn == 4 || otn == 2) && (itn == 8 || itn == 4) && (g4 == nullptr || (otn == 4 && itn == 8)), "wgrad_bf16: bits / g4 at unsupported widths");
      if (g4 != nullptr) { if (b8) ALLSET_WGTR_X(4, 8, true, true); else ALLSET_WGTR_X(4, 8, false, true); }
      else if (otn == 4 && itn == 8) ALLSET_WGTR_X(4, 8, true, false);
      else if (otn == 4 && itn == 4) ALLSET_WGTR_X(4, 4, true, false);
      else if (otn == 2 && itn == 8) ALLSET_WGTR_X(2, 8, true, false);
      else ALLSET_WGTR_X(2, 4, true, false);
    }
    else if (otn == 4 && itn == 8) ALLSET_WGTR(4, 8);
    else if (otn == 4 && itn == 4) ALLSET_WGTR(4, 4);
    else if (otn == 4 && itn == 2) ALLSET_WGTR(4, 2);
    else if (otn == 2 && itn == 8) ALLSET_WGTR(2, 8);
    else if (otn == 2 && itn == 4) ALLSET_WGTR(2, 4);
    else if (otn == 2 && itn == 2) ALLSET_WGTR(2, 2);
    else if (otn == 1 && itn == 8) ALLSET_WGTR(1, 8);
    else if (otn == 1 && itn == 4) ALLSET_WGTR(1, 4);
    else ALLSET_WGTR(1, 2);
#undef ALLSET_WGTR
#undef ALLSET_WGTR_X
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(bits == nullptr && g4 == nullptr, "wgrad_bf16: bits / g4 need the full-width kernel (16-byte aligned rows)");
  const int tiles_o = static_cast<int>((O + kWgTile - 1) / kWgTile), tiles_i = static_cast<int>((I + kWgTile - 1) / kWgTile);
  int64_t rows_per_slice = (n + n_slices - 1) / n_slices;
  rows_per_slice = (rows_per_slice + kWgRows - 1) / kWgRows * kWgRows;
  if (rows_per_slice < kWgRows) rows_per_slice = kWgRows;
  const dim3 grid(static_cast<unsigned>(tiles_o * tiles_i), static_cast<unsigned>(n_slices));
  wgrad_bf16_kernel<0><<<grid, kWx6Block, 0, st>>>(static_cast<const uint16_t*>(ga), lda, static_cast<const uint16_t*>(u), ldu,
                                                   part_w, part_b, n, static_cast<int>(O), static_cast<int>(I), tiles_i, rows_per_slice,
                                                   pw_stride, pb_stride);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static inline int ln_bf16_lpr(int64_t d) { return d <= 64 ? 8 : (d <= 128 ? 16 : (d <= 256 ? 32 : 64)); }

extern "C" int allset_ln_bf16_supported(int64_t d) { return (d >= 8 && d <= 512 && d % 8 == 0) ? 1 : 0; }

extern "C" int allset_ln_fwd_bf16(const void* x, int64_t ldx, const void* gamma, const void* beta, float eps, int relu_in,
                                  float p, uint64_t seed, void* y, int64_t ldy, float* stats, int64_t n, int64_t d,
                                  const uint64_t* seed_base, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1, "ln_fwd_bf16: bad size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "ln_fwd_bf16: dropout p must be in [0,1)");
  if (!allset_ln_bf16_supported(d)) { set_error("ln_fwd_bf16: width %lld not built (d %% 8 == 0, d <= 512)", static_cast<long long>(d)); return ALLSET_ERR_UNSUPPORTED; }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(x && gamma && beta && y && stats, "ln_fwd_bf16: null pointer");
  ALLSET_REQUIRE(ldx >= d && ldy >= d && ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta),
                 "ln_fwd_bf16: rows and parameter vectors must be 16-byte aligned");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int di = static_cast<int>(d), lpr = ln_bf16_lpr(d);
  const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr) * kLnRowsPerGroup;
  const unsigned grid = static_cast<unsigned>((n + rows_per_block - 1) / rows_per_block);
  const uint16_t *xx = static_cast<const uint16_t*>(x), *gg = static_cast<const uint16_t*>(gamma), *bb = static_cast<const uint16_t*>(beta);
  uint16_t* yy = static_cast<uint16_t*>(y);
#define ALLSET_LNB_FWD(L) ln_fwd_bf16_kernel<L><<<grid, kBlock, 0, st>>>(xx, ldx, gg, bb, eps, relu_in, p, seed, yy, ldy, stats, n, di, seed_base)
  switch (lpr) { case 8: ALLSET_LNB_FWD(8); break; case 16: ALLSET_LNB_FWD(16); break; case 32: ALLSET_LNB_FWD(32); break; default: ALLSET_LNB_FWD(64); break; }
#undef ALLSET_LNB_FWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_ln_bwd_bf16_partials(int64_t n, int64_t d, int64_t* n_partials) {
  clear_error();
  ALLSET_REQUIRE(n_partials != nullptr && n >= 0 && allset_ln_bf16_supported(d), "ln_bwd_bf16_partials: bad argument");
  const int64_t groups = static_cast<int64_t>(kWavesPerBlock) * (kWave / ln_bf16_lpr(d));
  const int64_t want = (n + groups - 1) / groups;
  *n_partials = want < 1 ? 1 : (want > 2048 ? 2048 : want);     // (1024 = the resident count at 117 VGPRs measured the same at [250k, 256])
  return ALLSET_OK;
}

extern "C" int allset_ln_bwd_bf16(const void* gy, int64_t ldg, const void* x, int64_t ldx, const float* stats,
                                  const void* gamma, int relu_in, float p, uint64_t seed, void* gx, int64_t ldgx,
                                  float* partials, int64_t n_partials, int64_t n, int64_t d, const uint64_t* seed_base,
                                  void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1, "ln_bwd_bf16: bad size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "ln_bwd_bf16: dropout p must be in [0,1)");
  if (!allset_ln_bf16_supported(d)) { set_error("ln_bwd_bf16: width %lld not built", static_cast<long long>(d)); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(partials != nullptr && n_partials >= 1, "ln_bwd_bf16: partials buffer required");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(partials, 0, static_cast<size_t>(n_partials) * 2 * d * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && x && stats && gamma, "ln_bwd_bf16: null pointer");
  ALLSET_REQUIRE(ldg >= d && ldx >= d && ldg % 8 == 0 && ldx % 8 == 0 && aligned16(gy) && aligned16(x) && aligned16(gamma) &&
                 (gx == nullptr || (ldgx >= d && ldgx % 8 == 0 && aligned16(gx))), "ln_bwd_bf16: rows must be 16-byte aligned");
  const int di = static_cast<int>(d);
  const unsigned grid = static_cast<unsigned>(n_partials);
  const uint16_t *gg = static_cast<const uint16_t*>(gy), *xx = static_cast<const uint16_t*>(x), *gm = static_cast<const uint16_t*>(gamma);
  uint16_t* go = static_cast<uint16_t*>(gx);
#define ALLSET_LNB_BWD(L) ln_bwd_bf16_kernel<L><<<grid, kBlock, 0, st>>>(gg, ldg, xx, ldx, stats, gm, relu_in, p, seed, go, ldgx, partials, n, di, seed_base)
  switch (ln_bf16_lpr(d)) { case 8: ALLSET_LNB_BWD(8); break; case 16: ALLSET_LNB_BWD(16); break; case 32: ALLSET_LNB_BWD(32); break; default: ALLSET_LNB_BWD(64); break; }
#undef ALLSET_LNB_BWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_ln_res_fwd_bf16(const void* x, int64_t ldx, const void* colb, const void* res, int64_t ldr,
                                      const void* gamma, const void* beta, float eps, int relu_out, float p, uint64_t seed,
                                      void* y, int64_t ldy, float* stats, int64_t n, int64_t d, const uint64_t* seed_base,
                                      void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1, "ln_res_fwd_bf16: bad size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "ln_res_fwd_bf16: dropout p must be in [0,1)");
  if (!allset_ln_bf16_supported(d)) { set_error("ln_res_fwd_bf16: width %lld not built (d %% 8 == 0, d <= 512)", static_cast<long long>(d)); return ALLSET_ERR_UNSUPPORTED; }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(x && gamma && beta && y && stats, "ln_res_fwd_bf16: null pointer");
  ALLSET_REQUIRE(ldx >= d && ldy >= d && ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) &&
                 aligned16(beta) && (colb == nullptr || aligned16(colb)) && (res == nullptr || (ldr >= d && ldr % 8 == 0 && aligned16(res))),
                 "ln_res_fwd_bf16: rows and parameter vectors must be 16-byte aligned");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int di = static_cast<int>(d), lpr = ln_bf16_lpr(d);
  const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr) * kLnRowsPerGroup;
  const unsigned grid = static_cast<unsigned>((n + rows_per_block - 1) / rows_per_block);
  typedef const uint16_t* CP;
#define ALLSET_LNRB_FWD2(L, R, D) ln_res_fwd_bf16_kernel<L, R, D><<<grid, kBlock, 0, st>>>((CP)x, ldx, (CP)colb, (CP)res, ldr, (CP)gamma, (CP)beta, eps, relu_out, p, seed, (uint16_t*)y, ldy, stats, n, di, seed_base)
#define ALLSET_LNRB_FWD(L)                                                                        \
  do {                                                                                            \
    if (res != nullptr) { if (p > 0.f) ALLSET_LNRB_FWD2(L, true, true); else ALLSET_LNRB_FWD2(L, true, false); } \
    else { if (p > 0.f) ALLSET_LNRB_FWD2(L, false, true); else ALLSET_LNRB_FWD2(L, false, false); }              \
  } while (0)
  switch (lpr) { case 8: ALLSET_LNRB_FWD(8); break; case 16: ALLSET_LNRB_FWD(16); break; case 32: ALLSET_LNRB_FWD(32); break; default: ALLSET_LNRB_FWD(64); break; }
#undef ALLSET_LNRB_FWD
#undef ALLSET_LNRB_FWD2
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static int ln_res_bwd_bf16_impl(const void* gy, int64_t ldg, const void* x, int64_t ldx, const void* colb,
                                const void* res, int64_t ldr, const float* stats, const void* gamma, const void* beta,
                                int relu_out, float p, uint64_t seed, void* gs, int64_t ldgs, float* partials,
                                int64_t n_partials, int64_t n, int64_t d, const uint64_t* seed_base, const float* pma_m,
                                const float* pma_l, float* pma_stats, int64_t pma_heads, void* stream);

extern "C" int allset_ln_res_bwd_bf16(const void* gy, int64_t ldg, const void* x, int64_t ldx, const void* colb,
                                      const void* res, int64_t ldr, const float* stats, const void* gamma, const void* beta,
                                      int relu_out, float p, uint64_t seed, void* gs, int64_t ldgs, float* partials,
                                      int64_t n_partials, int64_t n, int64_t d, const uint64_t* seed_base, void* stream) {
  clear_error();
  return ln_res_bwd_bf16_impl(gy, ldg, x, ldx, colb, res, ldr, stats, gamma, beta, relu_out, p, seed, gs, ldgs, partials, n_partials,
                              n, d, seed_base, nullptr, nullptr, nullptr, 1, stream);
}

extern "C" int allset_ln_res_bwd_pma_bf16_supported(int64_t d, int64_t heads) {
  if (!allset_ln_bf16_supported(d) || heads < 1 || d % heads != 0 || (d / heads) % 8 != 0) return 0;
  const int64_t g = (d / heads) / 8;
  return (g & (g - 1)) == 0 ? 1 : 0;
}

extern "C" int allset_ln_res_bwd_pma_bf16(const void* gy, int64_t ldg, const void* x, int64_t ldx, const void* colb,
                                          const float* stats, const void* gamma, const void* beta, void* gs, int64_t ldgs,
                                          float* partials, int64_t n_partials, int64_t n, int64_t d, const float* pma_m,
                                          const float* pma_l, float* pma_stats, int64_t heads, void* stream) {
  clear_error();
  if (!allset_ln_res_bwd_pma_bf16_supported(d, heads)) {
    set_error("ln_res_bwd_pma_bf16: d=%lld heads=%lld not built (channels per head must be 8 x a power of two)",
              static_cast<long long>(d), static_cast<long long>(heads));
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(n == 0 || (pma_m && pma_l && pma_stats), "ln_res_bwd_pma_bf16: null statistics pointer");
  ALLSET_REQUIRE(pma_stats == nullptr || (reinterpret_cast<uintptr_t>(pma_stats) & 7u) == 0, "ln_res_bwd_pma_bf16: stats must be 8-byte aligned");
  return ln_res_bwd_bf16_impl(gy, ldg, x, ldx, colb, nullptr, 0, stats, gamma, beta, 0, 0.f, 0, gs, ldgs, partials, n_partials, n, d,
                              nullptr, pma_m, pma_l, pma_stats, heads, stream);
}

static int ln_res_bwd_bf16_impl(const void* gy, int64_t ldg, const void* x, int64_t ldx, const void* colb,
                                const void* res, int64_t ldr, const float* stats, const void* gamma, const void* beta,
                                int relu_out, float p, uint64_t seed, void* gs, int64_t ldgs, float* partials,
                                int64_t n_partials, int64_t n, int64_t d, const uint64_t* seed_base, const float* pma_m,
                                const float* pma_l, float* pma_stats, int64_t pma_heads, void* stream) {
  ALLSET_REQUIRE(n >= 0 && d >= 1, "ln_res_bwd_bf16: bad size");
  ALLSET_REQUIRE(p >= 0.f && p < 1.f, "ln_res_bwd_bf16: dropout p must be in [0,1)");
  if (!allset_ln_bf16_supported(d)) { set_error("ln_res_bwd_bf16: width %lld not built", static_cast<long long>(d)); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(partials != nullptr && n_partials >= 1, "ln_res_bwd_bf16: partials buffer required");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    ALLSET_HIP_CHECK(hipMemsetAsync(partials, 0, static_cast<size_t>(n_partials) * 3 * d * sizeof(float), st));
    return ALLSET_OK;
  }
  ALLSET_REQUIRE(gy && x && stats && gamma && beta && gs, "ln_res_bwd_bf16: null pointer");
  ALLSET_REQUIRE(ldg >= d && ldx >= d && ldgs >= d && ldg % 8 == 0 && ldx % 8 == 0 && ldgs % 8 == 0 && aligned16(gy) && aligned16(x) &&
                 aligned16(gs) && aligned16(gamma) && aligned16(beta) && (colb == nullptr || aligned16(colb)) &&
                 (res == nullptr || (ldr >= d && ldr % 8 == 0 && aligned16(res))), "ln_res_bwd_bf16: rows must be 16-byte aligned");
  const int di = static_cast<int>(d);
  const unsigned grid = static_cast<unsigned>(n_partials);
  typedef const uint16_t* CP;
#define ALLSET_LNRB_BWD2(L, R, D, P) ln_res_bwd_bf16_kernel<L, R, D, P><<<grid, kBlock, 0, st>>>((CP)gy, ldg, (CP)x, ldx, (CP)colb, (CP)res, ldr, stats, (CP)gamma, (CP)beta, relu_out, p, seed, (uint16_t*)gs, ldgs, partials, n, di, seed_base, pma_m, pma_l, pma_stats, static_cast<int>(pma_heads))
#define ALLSET_LNRB_BWD(L)                                                                        \
  do {                                                                                            \
    const bool r_ = res != nullptr, d_ = p > 0.f;                                                 \
    if (pma_stats != nullptr) {                                                                   \
      if (r_) { if (d_) ALLSET_LNRB_BWD2(L, true, true, true); else ALLSET_LNRB_BWD2(L, true, false, true); }    \
      else { if (d_) ALLSET_LNRB_BWD2(L, false, true, true); else ALLSET_LNRB_BWD2(L, false, false, true); }     \
    } else {                                                                                      \
      if (r_) { if (d_) ALLSET_LNRB_BWD2(L, true, true, false); else ALLSET_LNRB_BWD2(L, true, false, false); }  \
      else { if (d_) ALLSET_LNRB_BWD2(L, false, true, false); else ALLSET_LNRB_BWD2(L, false, false, false); }   \
    }                                                                                             \
  } while (0)
  switch (ln_bf16_lpr(d)) { case 8: ALLSET_LNRB_BWD(8); break; case 16: ALLSET_LNRB_BWD(16); break; case 32: ALLSET_LNRB_BWD(32); break; default: ALLSET_LNRB_BWD(64); break; }
#undef ALLSET_LNRB_BWD
#undef ALLSET_LNRB_BWD2
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
