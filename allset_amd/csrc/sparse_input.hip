// The first Linear of a model on SPARSE raw features (bag-of-words rows: 18 of 1433 entries on Cora, 32 of 3703 on Citeseer; the
// reference's loaders hand them over as a dense matrix, models.py:473-476 multiplies all of it).  Same layer as input_linear.hip --
// [dropout ->] LayerNorm -> Linear on an input without gradient -- with the two GEMMs replaced by sums over the non-zeros:
//
//   x_hat[r, j] = rstd_r (v_rj - mean_r),  v = dropout(x)             (a row's zeros all share one value: -rstd_r mean_r)
//   y[r, :]  = rstd_r ( sum_{j in nnz(r)} v_rj W'[:, j]  -  mean_r s ) + b'      W' = W * gamma,  s = sum_j W'[:, j],  b' = b + W beta
//   M[:, j]  = gy^T x_hat[:, j] = sum_{r in nnz(j)} rstd_r v_rj gy[r, :]  -  u,  u = sum_r rstd_r mean_r gy[r, :]
//
// and M unfolds into the four parameter gradients exactly as in input_linear.hip (with s_gy = sum_r gy[r, :] for its ones column).
// Row statistics are exact two-pass sums over the non-zeros plus the closed-form contribution of the zeros.  The dropout is the
// library's counter hash of (seed, r * d + j): the same positions the dense kernels would drop.
//
//   fold_t_kernel          W'^T [d + 2, O]: rows j < d = W[:, j] * gamma_j (transposed through LDS), row d = s, row d + 1 = b'
//   sparse_ln_fwd_kernel   one lane group per row: statistics, gather of W'^T rows, y; keeps w = rstd * v per non-zero, rstd * mean per row
//   sparse_ln_bwd_kernel   one lane group per FEATURE (CSC): gather of gy rows -> M[:, j]; extra workgroups: partial sums of s_gy and u
//   (unfold: allset_unfold_ln_linear_ex in input_linear.hip)
#include "common.h"

namespace allset {

constexpr int kSpSlices = 16;        // row slices of the (s_gy, u) partial sums
constexpr int kSpInFlight = 16;      // gathers a lane group keeps in flight (plain-Linear kernels; a row / feature of ~30 non-zeros is then
                                     // two dependent round trips instead of four: these launches ARE their latency chains)

__global__ __launch_bounds__(kBlock) void fold_t_kernel(const float* __restrict__ W, int64_t ldw, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ b, int O, int d,
                                                        float* __restrict__ WT) {
  __shared__ float tile[64][65];
  __shared__ float red[2][kBlock];
  const int t = threadIdx.x;
  const int n_tiles = (d + 63) / 64;
  if (static_cast<int>(blockIdx.x) < n_tiles) {               // transpose 64 input columns x O outputs, 64 outputs at a time
    const int j0 = blockIdx.x * 64, jl = t & 63, kq = t >> 6;
    const int j = j0 + jl;
    const float g = j < d ? gamma[j] : 0.f;
    for (int k0 = 0; k0 < O; k0 += 64) {
      __syncthreads();
      for (int k = kq; k < 64; k += 4)
        tile[jl][k] = (j < d && k0 + k < O) ? W[static_cast<int64_t>(k0 + k) * ldw + j] * g : 0.f;
      __syncthreads();
      for (int r = kq; r < 64; r += 4)                         // row j0 + r of W'^T, columns k0 + jl
        if (j0 + r < d && k0 + jl < O) WT[static_cast<int64_t>(j0 + r) * O + k0 + jl] = tile[r][jl];
    }
    return;
  }
  const int k = blockIdx.x - n_tiles;                          // one workgroup per output: s[k], b'[k] (fixed-order sums)
  const float* w = W + static_cast<int64_t>(k) * ldw;
  float as = 0.f, ab = 0.f;
  for (int j = t; j < d; j += kBlock) {
    const float wv = w[j];
    as = fmaf(wv, gamma[j], as);
    ab = fmaf(wv, beta[j], ab);
  }
  red[0][t] = as; red[1][t] = ab;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; }
    __syncthreads();
  }
  if (t == 0) {
    WT[static_cast<int64_t>(d) * O + k] = red[0][0];
    WT[static_cast<int64_t>(d + 1) * O + k] = red[1][0] + (b ? b[k] : 0.f);
  }
}

template <int LPR>
__device__ __forceinline__ float slot_sum(float v) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// LPR lanes (O = 4 * LPR outputs) per row, 64 / LPR rows per wave.
template <int LPR>
__global__ __launch_bounds__(kBlock) void sparse_ln_fwd_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                               const float* __restrict__ val, int64_t n, int d,
                                                               const float* __restrict__ WT, float eps, float p_pre, uint64_t seed,
                                                               const uint64_t* __restrict__ seed_base, float* __restrict__ y,
                                                               int64_t ldy, float* __restrict__ w_out, float* __restrict__ rm) {
  constexpr int O = 4 * LPR, NS = kWave / LPR;
  seed = resolve_seed(seed_base, seed);
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR, lane0 = slot * LPR;
  const int64_t r = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)) * NS + slot;
  const bool live = r < n;
  const int p0 = live ? rowptr[r] : 0, p1 = live ? rowptr[r + 1] : 0;
  const float inv_keep = p_pre > 0.f ? 1.f / (1.f - p_pre) : 1.f;
  const uint32_t thr = drop_threshold(p_pre);
  auto value = [&](int p, int c) -> float {
    float v = val[p];
    if (p_pre > 0.f) v *= keep_scale(seed, r * d + c, thr, inv_keep);
    return v;
  };
  // (trip counts differ between the slots of a wave but are uniform inside a slot, and every shuffle below stays inside its slot)
  const int len = p1 - p0;
  float s1 = 0.f;
  for (int b0 = 0; b0 < len; b0 += LPR) {
    const int p = p0 + b0 + li;
    if (b0 + li < len) s1 += value(p, col[p]);
  }
  const float mean = slot_sum<LPR>(s1) / static_cast<float>(d);
  float s2 = 0.f;
  for (int b0 = 0; b0 < len; b0 += LPR) {
    const int p = p0 + b0 + li;
    if (b0 + li < len) { const float c = value(p, col[p]) - mean; s2 = fmaf(c, c, s2); }
  }
  s2 = slot_sum<LPR>(s2) + static_cast<float>(d - len) * mean * mean;
  const float rstd = rsqrtf(s2 / static_cast<float>(d) + eps);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = 0; b0 < len; b0 += LPR) {
    const int p = p0 + b0 + li;
    int c = 0;
    float v = 0.f;
    if (b0 + li < len) {
      c = col[p];
      v = value(p, c);
      w_out[p] = rstd * v;
    }
    const int nb = min(LPR, len - b0);
    for (int j = 0; j < nb; j += 8) {                     // eight gathers in flight (past the batch's end: v = 0, row 0 of W'^T)
      float4 wv[8];
      float vj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int src = lane0 + ((j + u) & (LPR - 1));
        const int cj = (j + u < nb) ? __shfl(c, src) : 0;
        vj[u] = (j + u < nb) ? __shfl(v, src) : 0.f;
        wv[u] = *reinterpret_cast<const float4*>(WT + static_cast<int64_t>(cj) * O + 4 * li);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc.x = fmaf(vj[u], wv[u].x, acc.x); acc.y = fmaf(vj[u], wv[u].y, acc.y);
        acc.z = fmaf(vj[u], wv[u].z, acc.z); acc.w = fmaf(vj[u], wv[u].w, acc.w);
      }
    }
  }
  if (!live) return;
  const float4 s = *reinterpret_cast<const float4*>(WT + static_cast<int64_t>(d) * O + 4 * li);
  const float4 bp = *reinterpret_cast<const float4*>(WT + static_cast<int64_t>(d + 1) * O + 4 * li);
  float4 o;
  o.x = fmaf(rstd, acc.x - mean * s.x, bp.x); o.y = fmaf(rstd, acc.y - mean * s.y, bp.y);
  o.z = fmaf(rstd, acc.z - mean * s.z, bp.z); o.w = fmaf(rstd, acc.w - mean * s.w, bp.w);
  *reinterpret_cast<float4*>(y + r * ldy + 4 * li) = o;
  if (li == 0) rm[r] = rstd * mean;
}

// Workgroups [0, feat_blocks): LPR lanes per feature j (CSC row), M[k, j] = sum_p w[posT[p]] gy[rowT[p], k] (k = 4 li .. 4 li + 3).
// Workgroups [feat_blocks, feat_blocks + kSpSlices): su_part[slice][0][k] = sum_r gy[r, k], [1][k] = sum_r rm[r] gy[r, k] over the slice.
template <int LPR>
__global__ __launch_bounds__(kBlock) void sparse_ln_bwd_kernel(const int32_t* __restrict__ colptr, const int32_t* __restrict__ rowT,
                                                               const int32_t* __restrict__ posT, const float* __restrict__ w,
                                                               const float* __restrict__ rm, const float* __restrict__ gy,
                                                               int64_t ldg, int64_t n, int d, float* __restrict__ M, int64_t ldm,
                                                               float* __restrict__ su_part, int feat_blocks) {
  constexpr int O = 4 * LPR, NS = kWave / LPR;
  if (static_cast<int>(blockIdx.x) >= feat_blocks) {
    __shared__ float4 red[2][kBlock];
    const int slice = blockIdx.x - feat_blocks;
    const int t = threadIdx.x, q = t % LPR, g = t / LPR;
    constexpr int G = kBlock / LPR;
    const int64_t rows = (n + kSpSlices - 1) / kSpSlices;
    const int64_t r0 = slice * rows, r1 = min(r0 + rows, n);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), u = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int64_t r = r0 + g; r < r1; r += G) {
      const float4 v = *reinterpret_cast<const float4*>(gy + r * ldg + 4 * q);
      const float m = rm[r];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      u.x = fmaf(m, v.x, u.x); u.y = fmaf(m, v.y, u.y); u.z = fmaf(m, v.z, u.z); u.w = fmaf(m, v.w, u.w);
    }
    red[0][t] = s; red[1][t] = u;
    __syncthreads();
    if (g == 0) {
      for (int gg = 1; gg < G; ++gg) {
        const float4 a = red[0][gg * LPR + q], c = red[1][gg * LPR + q];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        u.x += c.x; u.y += c.y; u.z += c.z; u.w += c.w;
      }
      float* out = su_part + static_cast<int64_t>(slice) * 2 * O;
      *reinterpret_cast<float4*>(out + 4 * q) = s;
      *reinterpret_cast<float4*>(out + O + 4 * q) = u;
    }
    return;
  }
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR, lane0 = slot * LPR;
  const int j = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * NS + slot;
  const bool live = j < d;
  const int p0 = live ? colptr[j] : 0, p1 = live ? colptr[j + 1] : 0;
  const int len = p1 - p0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = 0; b0 < len; b0 += LPR) {
    const int p = p0 + b0 + li;
    int rr = 0;
    float wv = 0.f;
    if (b0 + li < len) { rr = rowT[p]; wv = w[posT[p]]; }
    const int nb = min(LPR, len - b0);
    for (int i = 0; i < nb; i += 8) {                     // eight gathers in flight (past the batch's end: w = 0, row 0 of gy)
      float4 g4[8];
      float wi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int src = lane0 + ((i + u) & (LPR - 1));
        const int ri = (i + u < nb) ? __shfl(rr, src) : 0;
        wi[u] = (i + u < nb) ? __shfl(wv, src) : 0.f;
        g4[u] = *reinterpret_cast<const float4*>(gy + static_cast<int64_t>(ri) * ldg + 4 * li);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc.x = fmaf(wi[u], g4[u].x, acc.x); acc.y = fmaf(wi[u], g4[u].y, acc.y);
        acc.z = fmaf(wi[u], g4[u].z, acc.z); acc.w = fmaf(wi[u], g4[u].w, acc.w);
      }
    }
  }
  if (!live) return;
  M[static_cast<int64_t>(4 * li) * ldm + j] = acc.x;
  M[static_cast<int64_t>(4 * li + 1) * ldm + j] = acc.y;
  M[static_cast<int64_t>(4 * li + 2) * ldm + j] = acc.z;
  M[static_cast<int64_t>(4 * li + 3) * ldm + j] = acc.w;
}

// ---- the same idea for a PLAIN Linear on sparse raw features: PMA's value projection and its folded logits on the first conv of an
// AllSetTransformer (reference models.py:473 + layers.py:126-131: dropout(x) through lin_V and lin_K on Citeseer's 3703-wide rows;
// round 6: these two Linears were 147 us of library GEMMs and 18 us of dropout in a 440-us graphed training step).
//   [y | a][r, :] = sum_{j in nnz(r)} v_rj [W1; W2]^T[j, :] + [b1 | b2],  v = dropout(x)
//   gW[:, j] = sum_{r in nnz(j)} v_rj [gy | ga][r, :],  gb = sum_r [gy | ga][r, :]
// W1 [O1, d] (the projection), W2 [O2 <= 4, d] (the folded logit rows); one lane owns 4 consecutive outputs of the stacked row.
__global__ __launch_bounds__(kBlock) void sparse_wt_kernel(const float* __restrict__ W1, int64_t ld1, int O1, const float* __restrict__ W2,
                                                           int64_t ld2, int O2, const float* __restrict__ b1, const float* __restrict__ b2,
                                                           int d, float* __restrict__ WT, int pitch) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x;
  const int n_tiles = (d + 63) / 64;
  if (static_cast<int>(blockIdx.x) == n_tiles) {                // row d: the stacked bias
    if (blockIdx.y == 0)
      for (int k = t; k < pitch; k += kBlock)
        WT[static_cast<int64_t>(d) * pitch + k] = k < O1 ? (b1 ? b1[k] : 0.f) : (k < O1 + O2 ? (b2 ? b2[k - O1] : 0.f) : 0.f);
    return;
  }
  const int j0 = blockIdx.x * 64, jl = t & 63, kq = t >> 6;     // one 64 x 64 tile per workgroup (grid y: the 64-output blocks)
  const int j = j0 + jl;
  {
    const int k0 = blockIdx.y * 64;
    for (int k = kq; k < 64; k += 4) {
      const int o = k0 + k;
      float v = 0.f;
      if (j < d) {
        if (o < O1) v = W1[static_cast<int64_t>(o) * ld1 + j];
        else if (o < O1 + O2) v = W2[static_cast<int64_t>(o - O1) * ld2 + j];
      }
      tile[jl][k] = v;
    }
    __syncthreads();
    for (int r = kq; r < 64; r += 4)
      if (j0 + r < d && k0 + jl < pitch) WT[static_cast<int64_t>(j0 + r) * pitch + k0 + jl] = tile[r][jl];
  }
}

// LPR lanes per row, 64 / LPR rows per wave; a lane owns the 16-byte chunks li + LPR v (v < VPL) of the stacked output row -- the first
// O1 / 4 chunks are the projection's, the next ceil(O2 / 4) the auxiliary rows' (y2: [n, 4 ceil(O2 / 4)]); chunks past the row idle.
// (LPR, VPL) = (32, 1) for O1 = 64, (64, 1) for 128, (64, 2) for 256, (64, 3) for 512.
template <int LPR, int VPL>
__global__ __launch_bounds__(kBlock) void sparse_lin_fwd_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const float* __restrict__ val, int64_t n, int d,
                                                                const float* __restrict__ WT, int pitch, int O1, int O2, float p_pre,
                                                                uint64_t seed, const uint64_t* __restrict__ seed_base,
                                                                float* __restrict__ y, int64_t ldy, float* __restrict__ y2,
                                                                float* __restrict__ w_out) {
  constexpr int NS = kWave / LPR;
  constexpr int IF = VPL == 1 ? kSpInFlight : 8;                   // non-zeros in flight (x VPL gathers each)
  seed = resolve_seed(seed_base, seed);
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR, lane0 = slot * LPR;
  const int64_t r = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)) * NS + slot;
  const bool live = r < n;
  const int nch = pitch / 4, mch = O1 / 4, ld2 = pitch - O1;
  const int p0 = live ? rowptr[r] : 0, p1 = live ? rowptr[r + 1] : 0;
  const float inv_keep = p_pre > 0.f ? 1.f / (1.f - p_pre) : 1.f;
  const uint32_t thr = drop_threshold(p_pre);
  const int len = p1 - p0;
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = 0; b0 < len; b0 += LPR) {
    const int p = p0 + b0 + li;
    int c = 0;
    float xv = 0.f;
    if (b0 + li < len) {
      c = col[p];
      xv = val[p];
      if (p_pre > 0.f) xv *= keep_scale(seed, r * d + c, thr, inv_keep);
      w_out[p] = xv;
    }
    const int nb = min(LPR, len - b0);
    for (int j = 0; j < nb; j += IF) {                    // IF non-zeros in flight (past the batch's end: value 0, row 0 of the weight)
      float4 wv[IF][VPL];
      float vj[IF];
#pragma unroll
      for (int u = 0; u < IF; ++u) {
        const int src = lane0 + ((j + u) & (LPR - 1));
        const int cj = (j + u < nb) ? __shfl(c, src) : 0;
        vj[u] = (j + u < nb) ? __shfl(xv, src) : 0.f;
        const float* wr = WT + static_cast<int64_t>(cj) * pitch;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
          wv[u][v] = (li + LPR * v < nch) ? *reinterpret_cast<const float4*>(wr + 4 * (li + LPR * v)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < IF; ++u)
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          acc[v].x = fmaf(vj[u], wv[u][v].x, acc[v].x); acc[v].y = fmaf(vj[u], wv[u][v].y, acc[v].y);
          acc[v].z = fmaf(vj[u], wv[u][v].z, acc[v].z); acc[v].w = fmaf(vj[u], wv[u][v].w, acc[v].w);
        }
    }
  }
  if (!live) return;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int ch = li + LPR * v;
    if (ch >= nch) continue;
    const float4 bp = *reinterpret_cast<const float4*>(WT + static_cast<int64_t>(d) * pitch + 4 * ch);
    const float4 o = make_float4(acc[v].x + bp.x, acc[v].y + bp.y, acc[v].z + bp.z, acc[v].w + bp.w);
    if (ch < mch) *reinterpret_cast<float4*>(y + r * ldy + 4 * ch) = o;
    else *reinterpret_cast<float4*>(y2 + r * ld2 + 4 * (ch - mch)) = o;
  }
}

// Workgroups [0, feat_blocks): LPR lanes per feature j (CSC row), gW[k, j] = sum_p w[posT[p]] g[rowT[p], k] with g = [gy | g2].
// A workgroup owns BLK / LPR CONSECUTIVE features: the sums meet in an LDS tile [output][feature] and leave as runs of consecutive
// floats along j (a lane writing its four outputs itself puts 4 bytes into each of four rows 4 d bytes apart: one partial 64-byte
// sector per float).  BLK = 1024 threads (16 features at LPR = 64) for one chunk per lane, 512 for two or three (register budget).
// Workgroups [feat_blocks, feat_blocks + kSpSlices): sb_part[slice][k] = sum_r g[r, k] over the slice (the stacked bias gradient).
constexpr int kSpMaxPitch = 520;
template <int LPR, int VPL>
__global__ __launch_bounds__(VPL == 1 ? 1024 : 512) void sparse_lin_bwd_kernel(
    const int32_t* __restrict__ colptr, const int32_t* __restrict__ rowT, const int32_t* __restrict__ posT, const float* __restrict__ w,
    const float* __restrict__ gy, int64_t ldg, const float* __restrict__ g2, int64_t n, int d, int pitch, int O1, int O2,
    float* __restrict__ gW1, int64_t ldw1, float* __restrict__ gW2, int64_t ldw2, float* __restrict__ sb_part, int feat_blocks,
    unsigned* __restrict__ ticket, float* __restrict__ sb_total) {
  constexpr int BLK = VPL == 1 ? 1024 : 512;
  constexpr int NS = kWave / LPR, F = BLK / LPR;                   // features per wave / per workgroup
  constexpr int IF = VPL == 1 ? kSpInFlight : 8;
  constexpr int TILE4 = (VPL == 1 ? 136 : kSpMaxPitch) * F / 4;    // the [pitch][F] tile, in float4s
  __shared__ float4 lds4[TILE4 > BLK ? TILE4 : BLK];               // the bias partials' exchange, or the tile
  const int nch = pitch / 4, mch = O1 / 4, ld2 = pitch - O1;
  if (static_cast<int>(blockIdx.x) >= feat_blocks) {
    float4* red = lds4;
    const int slice = blockIdx.x - feat_blocks;
    const int t = threadIdx.x, q = t % LPR, g = t / LPR;
    constexpr int G = BLK / LPR;
    const int64_t rows = (n + kSpSlices - 1) / kSpSlices;
    const int64_t r0 = slice * rows, r1 = min(r0 + rows, n);
    for (int v = 0; v < VPL; ++v) {
      const int ch = q + LPR * v;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ch < nch) {
#pragma unroll 4
        for (int64_t r = r0 + g; r < r1; r += G) {
          const float4 x4 = ch < mch ? *reinterpret_cast<const float4*>(gy + r * ldg + 4 * ch)
                                     : *reinterpret_cast<const float4*>(g2 + r * ld2 + 4 * (ch - mch));
          s.x += x4.x; s.y += x4.y; s.z += x4.z; s.w += x4.w;
        }
      }
      __syncthreads();
      red[t] = s;
      __syncthreads();
      if (g == 0 && ch < nch) {
        for (int gg = 1; gg < G; ++gg) {
          const float4 a = red[gg * LPR + q];
          s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        *reinterpret_cast<float4*>(sb_part + static_cast<int64_t>(slice) * pitch + 4 * ch) = s;
      }
    }
    if (ticket != nullptr) {
      // the LAST slice workgroup to arrive adds the slices in index order (the same sum whichever workgroup that is) and re-arms the
      // ticket: the bias gradients leave this launch finished (csrc/loss.hip has the same construct)
      __shared__ int s_last;
      __syncthreads();
      if (t == 0) {
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == static_cast<unsigned>(kSpSlices - 1);
      }
      __syncthreads();
      if (s_last) {
        __threadfence();
        for (int k = t; k < pitch; k += BLK) {
          float acc = 0.f;
          for (int sl = 0; sl < kSpSlices; ++sl) acc += __hip_atomic_load(sb_part + static_cast<int64_t>(sl) * pitch + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sb_total[k] = acc;
        }
        if (t == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  float* tile = reinterpret_cast<float*>(lds4);
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR, lane0 = slot * LPR;
  const int fl = (threadIdx.x >> 6) * NS + slot;                   // this lane group's feature inside the workgroup
  const int j0 = blockIdx.x * F;
  const int j = j0 + fl;
  const bool live = j < d;
  const int p0 = live ? colptr[j] : 0, p1 = live ? colptr[j + 1] : 0;
  const int len = p1 - p0;
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = 0; b0 < len; b0 += LPR) {
    const int p = p0 + b0 + li;
    int rr = 0;
    float wv = 0.f;
    if (b0 + li < len) { rr = rowT[p]; wv = w[posT[p]]; }
    const int nb = min(LPR, len - b0);
    for (int i = 0; i < nb; i += IF) {                    // IF non-zeros in flight (past the batch's end: w = 0, row 0)
      float4 g4[IF][VPL];
      float wi[IF];
#pragma unroll
      for (int u = 0; u < IF; ++u) {
        const int src = lane0 + ((i + u) & (LPR - 1));
        const int ri = (i + u < nb) ? __shfl(rr, src) : 0;
        wi[u] = (i + u < nb) ? __shfl(wv, src) : 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int ch = li + LPR * v;
          g4[u][v] = ch < mch ? *reinterpret_cast<const float4*>(gy + static_cast<int64_t>(ri) * ldg + 4 * ch)
                              : (ch < nch ? *reinterpret_cast<const float4*>(g2 + static_cast<int64_t>(ri) * ld2 + 4 * (ch - mch))
                                          : make_float4(0.f, 0.f, 0.f, 0.f));
        }
      }
#pragma unroll
      for (int u = 0; u < IF; ++u)
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          acc[v].x = fmaf(wi[u], g4[u][v].x, acc[v].x); acc[v].y = fmaf(wi[u], g4[u][v].y, acc[v].y);
          acc[v].z = fmaf(wi[u], g4[u][v].z, acc[v].z); acc[v].w = fmaf(wi[u], g4[u][v].w, acc[v].w);
        }
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int ch = li + LPR * v;
    if (ch < nch) {
      tile[(4 * ch) * F + fl] = acc[v].x; tile[(4 * ch + 1) * F + fl] = acc[v].y;
      tile[(4 * ch + 2) * F + fl] = acc[v].z; tile[(4 * ch + 3) * F + fl] = acc[v].w;
    }
  }
  __syncthreads();
  const int rows_out = O1 + O2;
  for (int idx = threadIdx.x; idx < rows_out * F; idx += BLK) {
    const int o = idx / F, f = idx % F;
    if (j0 + f < d) {
      const float x1 = tile[o * F + f];
      if (o < O1) gW1[static_cast<int64_t>(o) * ldw1 + j0 + f] = x1;
      else gW2[static_cast<int64_t>(o - O1) * ldw2 + j0 + f] = x1;
    }
  }
}

}  // namespace allset

using namespace allset;

extern "C" int allset_sparse_ln_linear_supported(int64_t O) { return (O == 64 || O == 128 || O == 256) ? 1 : 0; }
extern "C" int allset_sparse_ln_linear_slices(void) { return kSpSlices; }

extern "C" int allset_fold_ln_linear_t(const float* W, int64_t ldw, const float* gamma, const float* beta, const float* b, int64_t O,
                                       int64_t d, float* WT, void* stream) {
  clear_error();
  ALLSET_REQUIRE(O >= 1 && O < (1 << 20) && d >= 1 && d < INT32_MAX - 64, "fold_ln_linear_t: bad size");
  ALLSET_REQUIRE(W && gamma && beta && WT, "fold_ln_linear_t: null pointer");
  ALLSET_REQUIRE(ldw >= d, "fold_ln_linear_t: leading dimension smaller than d");
  const unsigned grid = static_cast<unsigned>((d + 63) / 64 + O);
  fold_t_kernel<<<grid, kBlock, 0, static_cast<hipStream_t>(stream)>>>(W, ldw, gamma, beta, b, static_cast<int>(O), static_cast<int>(d), WT);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_sparse_ln_linear_fwd(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n, int64_t d,
                                           const float* WT, int64_t O, float eps, float p_pre, uint64_t seed, const uint64_t* seed_base,
                                           float* y, int64_t ldy, float* w_out, float* rm, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1 && d < INT32_MAX, "sparse_ln_linear_fwd: bad size");
  ALLSET_REQUIRE(p_pre >= 0.f && p_pre < 1.f, "sparse_ln_linear_fwd: dropout p must be in [0,1)");
  if (!allset_sparse_ln_linear_supported(O)) {
    set_error("sparse_ln_linear_fwd: O must be 64, 128 or 256 (got %lld)", static_cast<long long>(O));
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(rowptr && WT && y && rm, "sparse_ln_linear_fwd: null pointer");
  ALLSET_REQUIRE(ldy >= O && ldy % 4 == 0 && aligned16(y) && aligned16(WT), "sparse_ln_linear_fwd: y rows / WT must be 16-byte aligned");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int lpr = static_cast<int>(O / 4);
  const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr);
  const unsigned grid = static_cast<unsigned>((n + rows_per_block - 1) / rows_per_block);
  const int di = static_cast<int>(d);
  switch (lpr) {
    case 16: sparse_ln_fwd_kernel<16><<<grid, kBlock, 0, st>>>(rowptr, col, val, n, di, WT, eps, p_pre, seed, seed_base, y, ldy, w_out, rm); break;
    case 32: sparse_ln_fwd_kernel<32><<<grid, kBlock, 0, st>>>(rowptr, col, val, n, di, WT, eps, p_pre, seed, seed_base, y, ldy, w_out, rm); break;
    default: sparse_ln_fwd_kernel<64><<<grid, kBlock, 0, st>>>(rowptr, col, val, n, di, WT, eps, p_pre, seed, seed_base, y, ldy, w_out, rm); break;
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_sparse_ln_linear_bwd(const int32_t* colptr, const int32_t* rowT, const int32_t* posT, const float* w,
                                           const float* rm, const float* gy, int64_t ldg, int64_t n, int64_t d, int64_t O, float* M,
                                           int64_t ldm, float* su_part, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1 && d < INT32_MAX, "sparse_ln_linear_bwd: bad size");
  if (!allset_sparse_ln_linear_supported(O)) {
    set_error("sparse_ln_linear_bwd: O must be 64, 128 or 256 (got %lld)", static_cast<long long>(O));
    return ALLSET_ERR_UNSUPPORTED;
  }
  ALLSET_REQUIRE(colptr && M && su_part && (n == 0 || (rm && gy)), "sparse_ln_linear_bwd: null pointer");
  ALLSET_REQUIRE(ldm >= d && (n == 0 || (ldg >= O && ldg % 4 == 0 && aligned16(gy))) && aligned16(su_part),
                 "sparse_ln_linear_bwd: ldm >= d; gy rows and su_part 16-byte aligned");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int lpr = static_cast<int>(O / 4);
  const int64_t feats_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr);
  const int feat_blocks = static_cast<int>((d + feats_per_block - 1) / feats_per_block);
  const unsigned grid = static_cast<unsigned>(feat_blocks + kSpSlices);
  const int di = static_cast<int>(d);
  switch (lpr) {
    case 16: sparse_ln_bwd_kernel<16><<<grid, kBlock, 0, st>>>(colptr, rowT, posT, w, rm, gy, ldg, n, di, M, ldm, su_part, feat_blocks); break;
    case 32: sparse_ln_bwd_kernel<32><<<grid, kBlock, 0, st>>>(colptr, rowT, posT, w, rm, gy, ldg, n, di, M, ldm, su_part, feat_blocks); break;
    default: sparse_ln_bwd_kernel<64><<<grid, kBlock, 0, st>>>(colptr, rowT, posT, w, rm, gy, ldg, n, di, M, ldm, su_part, feat_blocks); break;
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

// ---- ABI 14: the plain Linear (+ up to 4 auxiliary output rows) on sparse raw features -------------------------------------------------
// Built for O1 in {64, 128, 256, 512} and O2 <= 8 (PMA's value projection with its folded logit rows: the tuned configurations of
// the reference's run_AllSetTransformer.sh use MLP_hidden 64 ... 512 and 1 ... 8 heads).  pitch = allset_sparse_linear_pitch(O1, O2).
extern "C" int allset_sparse_linear_supported(int64_t O1, int64_t O2) {
  return ((O1 == 64 || O1 == 128 || O1 == 256 || O1 == 512) && O2 >= 0 && O2 <= 8) ? 1 : 0;
}
extern "C" int64_t allset_sparse_linear_pitch(int64_t O1, int64_t O2) { return allset_sparse_linear_supported(O1, O2) ? O1 + 4 * ((O2 + 3) / 4) : 0; }

extern "C" int allset_sparse_linear_wt(const float* W1, int64_t ld1, int64_t O1, const float* W2, int64_t ld2, int64_t O2, const float* b1,
                                       const float* b2, int64_t d, float* WT, void* stream) {
  clear_error();
  if (!allset_sparse_linear_supported(O1, O2)) { set_error("sparse_linear_wt: O1 must be 64, 128, 256 or 512 and O2 <= 8"); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(d >= 1 && d < INT32_MAX - 64, "sparse_linear_wt: bad size");
  ALLSET_REQUIRE(W1 && WT && (O2 == 0 || W2) && ld1 >= d && (O2 == 0 || ld2 >= d), "sparse_linear_wt: null pointer or leading dimension smaller than d");
  const int pitch = static_cast<int>(allset_sparse_linear_pitch(O1, O2));
  const unsigned grid = static_cast<unsigned>((d + 63) / 64 + 1);
  sparse_wt_kernel<<<dim3(grid, static_cast<unsigned>((pitch + 63) / 64)), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      W1, ld1, static_cast<int>(O1), W2, ld2, static_cast<int>(O2), b1, b2, static_cast<int>(d), WT, pitch);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

// y [n, O1] (ldy), y2 [n, 4 ceil(O2 / 4)] (the auxiliary columns, padded to whole 16-byte chunks), w_out [nnz] = the values after the dropout (kept for the backward)
extern "C" int allset_sparse_linear_fwd(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n, int64_t d, const float* WT,
                                        int64_t O1, int64_t O2, float p_pre, uint64_t seed, const uint64_t* seed_base, float* y, int64_t ldy,
                                        float* y2, float* w_out, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1 && d < INT32_MAX, "sparse_linear_fwd: bad size");
  ALLSET_REQUIRE(p_pre >= 0.f && p_pre < 1.f, "sparse_linear_fwd: dropout p must be in [0,1)");
  if (!allset_sparse_linear_supported(O1, O2)) { set_error("sparse_linear_fwd: O1 must be 64, 128, 256 or 512 and O2 <= 8"); return ALLSET_ERR_UNSUPPORTED; }
  if (n == 0) return ALLSET_OK;
  ALLSET_REQUIRE(rowptr && WT && y && w_out && (O2 == 0 || y2), "sparse_linear_fwd: null pointer");
  ALLSET_REQUIRE(ldy >= O1 && ldy % 4 == 0 && aligned16(y) && aligned16(WT) && (O2 == 0 || aligned16(y2)),
                 "sparse_linear_fwd: y rows, y2 and WT must be 16-byte aligned");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int pitch = static_cast<int>(allset_sparse_linear_pitch(O1, O2));
  const int lpr = O1 == 64 ? 32 : 64;
  const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr);
  const unsigned grid = static_cast<unsigned>((n + rows_per_block - 1) / rows_per_block);
  const int di = static_cast<int>(d), o1 = static_cast<int>(O1), o2 = static_cast<int>(O2);
#define ALLSET_SPL_FWD(L, V) sparse_lin_fwd_kernel<L, V><<<grid, kBlock, 0, st>>>(rowptr, col, val, n, di, WT, pitch, o1, o2, p_pre, seed, seed_base, y, ldy, y2, w_out)
  if (O1 == 64) ALLSET_SPL_FWD(32, 1);
  else if (O1 == 128) ALLSET_SPL_FWD(64, 1);
  else if (O1 == 256) ALLSET_SPL_FWD(64, 2);
  else ALLSET_SPL_FWD(64, 3);
#undef ALLSET_SPL_FWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

// gW1 [O1, d] (ldw1), gW2 [O2, d] (ldw2), sb_part [allset_sparse_ln_linear_slices()][pitch]: the caller sums the slices into [gb1 | gb2]
extern "C" int allset_sparse_linear_bwd(const int32_t* colptr, const int32_t* rowT, const int32_t* posT, const float* w, const float* gy,
                                        int64_t ldg, const float* g2, int64_t n, int64_t d, int64_t O1, int64_t O2, float* gW1, int64_t ldw1,
                                        float* gW2, int64_t ldw2, float* sb_part, uint32_t* ticket, float* sb_total, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && d >= 1 && d < INT32_MAX, "sparse_linear_bwd: bad size");
  ALLSET_REQUIRE((ticket == nullptr) == (sb_total == nullptr), "sparse_linear_bwd: ticket and sb_total go together");
  if (!allset_sparse_linear_supported(O1, O2)) { set_error("sparse_linear_bwd: O1 must be 64, 128, 256 or 512 and O2 <= 8"); return ALLSET_ERR_UNSUPPORTED; }
  ALLSET_REQUIRE(colptr && gW1 && sb_part && (O2 == 0 || gW2) && (n == 0 || (gy && w && (O2 == 0 || g2))), "sparse_linear_bwd: null pointer");
  ALLSET_REQUIRE(ldw1 >= d && (O2 == 0 || ldw2 >= d) && (n == 0 || (ldg >= O1 && ldg % 4 == 0 && aligned16(gy) && (O2 == 0 || aligned16(g2)))) &&
                 aligned16(sb_part), "sparse_linear_bwd: leading dimensions; gy rows, g2 and sb_part 16-byte aligned");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int pitch = static_cast<int>(allset_sparse_linear_pitch(O1, O2));
  const int lpr = O1 == 64 ? 32 : 64;
  const int blk = O1 <= 128 ? 1024 : 512;
  const int64_t feats_per_block = blk / lpr;
  const int feat_blocks = static_cast<int>((d + feats_per_block - 1) / feats_per_block);
  const unsigned grid = static_cast<unsigned>(feat_blocks + kSpSlices);
  const int di = static_cast<int>(d), o1 = static_cast<int>(O1), o2 = static_cast<int>(O2);
#define ALLSET_SPL_BWD(L, V) sparse_lin_bwd_kernel<L, V><<<grid, blk, 0, st>>>(colptr, rowT, posT, w, gy, ldg, g2, n, di, pitch, o1, o2, gW1, ldw1, gW2, ldw2, sb_part, feat_blocks, ticket, sb_total)
  if (O1 == 64) ALLSET_SPL_BWD(32, 1);
  else if (O1 == 128) ALLSET_SPL_BWD(64, 1);
  else if (O1 == 256) ALLSET_SPL_BWD(64, 2);
  else ALLSET_SPL_BWD(64, 3);
#undef ALLSET_SPL_BWD
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
