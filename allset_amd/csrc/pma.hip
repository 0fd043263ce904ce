// PMA (multi-head softmax-attention pooling with a source-only logit) for gfx950.
// Replaces, per direction, the ~10 tensor ops of reference layers.py:145,168-194 (two index_select
// lifts, leaky_relu, the 6-temporary segment softmax, the [nnz,H,C] product and the scatter-add)
// with one gather pass forward and ONE gather pass backward:
//
//   forward  (target-major CSR):  per target row, each lane owns 16 B of one head's channels and
//            runs its own online softmax over the incidences (running max m, running sum l,
//            rescaled accumulator) -- the logit alpha[src,h] is a 4-byte load that rides along
//            with the 16-byte V gather, so no cross-lane traffic is needed until the NS slots of
//            the wave are merged at the end.  Saves m,l per (target, head).
//   backward (source-major CSR):  p_j is recomputed from (alpha[s,h], m[t,h], l[t,h]); because p_j
//            is identical in all lanes of a head, sum_j p_j*(<V_s,gO_t> - delta_t) commutes with the
//            cross-lane dot product, so lanes accumulate partial dots over the whole row and the
//            per-head reduction happens once per ROW, not once per incidence.  No [nnz,H] or
//            [nnz,H,C] temporary exists anywhere (math: SURVEY.md A.5).
//
// Lane layout is the one of segreduce.hip: LPR lanes x VEC f32 cover a feature row, NS = 64/LPR
// incidences are gathered per wave-wide load.  A lane's VEC channels never straddle a head (C % VEC == 0).
#include <float.h>

#include "common.h"

namespace allset {

constexpr int kPmaUnroll = 8;
constexpr int kMaxHeads = 256;   // per-wave LDS accumulators for the per-head scalars of the backward
constexpr float kSoftmaxEps = 1e-16f;   // torch_geometric.utils.softmax denominator guard [external]

// Sum `val` over the lanes [grp_start, grp_end) of this lane's LPR-lane row group; valid in lane grp_start.
template <int LPR>
__device__ __forceinline__ float head_group_reduce(float val, int li, int grp_end) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) {
    const float o = __shfl_down(val, off);
    if (li + off < grp_end) val += o;
  }
  return val;
}

template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void pma_fwd_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ alpha, int64_t lda,
    const T* __restrict__ V, int64_t ldv, float slope, T* __restrict__ out, int64_t ldo,
    float* __restrict__ m_out, float* __restrict__ l_out, int n_t, int H, int C, const int32_t* __restrict__ row_order) {
  constexpr int NS = kWave / LPR;
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int slot_row = static_cast<int>(blk) * kWavesPerBlock + (threadIdx.x >> 6);
  if (slot_row >= n_t) return;
  const int row = row_order ? row_order[slot_row] : slot_row;      // long rows first (see segreduce_kernel)
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int start = rowptr[row], end = rowptr[row + 1];
  const int d = H * C;

  for (int cb = 0; cb < d; cb += LPR * VEC) {
    const int c0 = cb + li * VEC;
    const bool active = c0 < d;
    const int h = active ? c0 / C : 0;
    float m = -FLT_MAX, l = 0.f;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;

    for (int base = start; base < end; base += kWave) {
      const int n = min(kWave, end - base);
      const int my_col = (lane < n) ? col[base + lane] : 0;
      for (int j = 0; j < n; j += NS * kPmaUnroll) {
        Raw<T, VEC> raw[kPmaUnroll];
        float a[kPmaUnroll];
        bool ok[kPmaUnroll];
#pragma unroll
        for (int u = 0; u < kPmaUnroll; ++u) {
          const int jj = j + u * NS + slot;
          ok[u] = (jj < n) && active;
          const int src = __shfl(my_col, jj & (kWave - 1));
          if (ok[u]) {
            a[u] = alpha[static_cast<int64_t>(src) * lda + h];
            raw[u] = load_raw<T, VEC>(V + static_cast<int64_t>(src) * ldv + c0);
          }
        }
#pragma unroll
        for (int u = 0; u < kPmaUnroll; ++u) {
          if (ok[u]) {
            const FVec<VEC> vu = unpack<T, VEC>(raw[u]);
            const float av = leaky_relu(a[u], slope);
            const float m_new = fmaxf(m, av);
            const float sc = __expf(m - m_new);       // 0 on the first incidence (m = -FLT_MAX)
            const float pe = __expf(av - m_new);
            l = fmaf(l, sc, pe);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(acc[k], sc, pe * vu.v[k]);
            m = m_new;
          }
        }
      }
    }

    // merge the NS slots' (m, l, acc) triples
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
      const float mo = __shfl_xor(m, off);
      const float lo = __shfl_xor(l, off);
      const float m_new = fmaxf(m, mo);
      const float s1 = __expf(m - m_new), s2 = __expf(mo - m_new);   // both -FLT_MAX -> exp(0) * (l = 0)
      l = l * s1 + lo * s2;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float ao = __shfl_xor(acc[k], off);
        acc[k] = acc[k] * s1 + ao * s2;
      }
      m = m_new;
    }

    if (slot == 0 && active) {
      const float inv = l > 0.f ? 1.f / (l + kSoftmaxEps) : 0.f;   // empty row -> 0
      FVec<VEC> r;
#pragma unroll
      for (int k = 0; k < VEC; ++k) r.v[k] = acc[k] * inv;
      store_vec<T, VEC>(out + static_cast<int64_t>(row) * ldo + c0, r);
      if (c0 % C == 0) {
        m_out[static_cast<int64_t>(row) * H + h] = l > 0.f ? m : 0.f;
        l_out[static_cast<int64_t>(row) * H + h] = l;
      }
    }
  }
}


// ---- short-row variant of pma_fwd (see segreduce_flat_kernel): each LPR-lane slot owns kPmaFlatRows consecutive
// target rows and walks their incidences as one stream, restarting its online-softmax state at every row end.
// Needed where rows are short by construction: E->V under hyperedge sharding (each rank holds ~deg/P incidences of a
// vertex) and graphs with self-loop hyperedges.  Whole row in one column chunk (d <= LPR*VEC).
constexpr int kPmaFlatRows = 7;

template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void pma_fwd_flat_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ alpha, int64_t lda,
    const T* __restrict__ V, int64_t ldv, float slope, T* __restrict__ out, int64_t ldo,
    float* __restrict__ m_out, float* __restrict__ l_out, int n_t, int H, int C, const int32_t* __restrict__ row_ids) {
  // row_ids (optional): the CSR handed in is a COMPACTED one -- row i of it is row row_ids[i] of the outputs (the short rows of a
  // skewed incidence; the long ones go to the one-wave-per-row kernel with their own list: ops.CSR.split)
  constexpr int NS = kWave / LPR;
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int lane0 = slot * LPR;
  const int64_t slot_global = (static_cast<int64_t>(blk) * kWavesPerBlock + (threadIdx.x >> 6)) * NS + slot;
  const int64_t r_begin64 = slot_global * kPmaFlatRows;
  if (r_begin64 - static_cast<int64_t>(slot) * kPmaFlatRows >= n_t) return;
  const int r_begin = static_cast<int>(min(r_begin64, static_cast<int64_t>(n_t)));
  const int r_end = min(r_begin + kPmaFlatRows, n_t);
  const int d = H * C;
  const int c0 = li * VEC;
  const bool active = c0 < d;
  const int h = active ? c0 / C : 0;
  const bool head_leader = active && (c0 % C == 0);
  const int rp = (li <= r_end - r_begin) ? rowptr[r_begin + li] : 0;
  const int rid = (li < r_end - r_begin) ? (row_ids ? row_ids[r_begin + li] : r_begin + li) : 0;      // output row of slot row li
  const int q0 = __shfl(rp, lane0);
  const int q_end = __shfl(rp, lane0 + (r_end - r_begin));

  int cur_row = r_begin;
  int cur_end = (r_begin < r_end) ? __shfl(rp, lane0 + 1) : q0;
  float m = -FLT_MAX, l = 0.f;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;

  auto flush = [&]() {
    const int orow = __shfl(rid, lane0 + min(cur_row - r_begin, LPR - 1));
    if (active) {
      const float inv = l > 0.f ? 1.f / (l + kSoftmaxEps) : 0.f;
      FVec<VEC> r;
#pragma unroll
      for (int k = 0; k < VEC; ++k) { r.v[k] = acc[k] * inv; acc[k] = 0.f; }
      store_vec<T, VEC>(out + static_cast<int64_t>(orow) * ldo + c0, r);
      if (head_leader) {
        m_out[static_cast<int64_t>(orow) * H + h] = l > 0.f ? m : 0.f;
        l_out[static_cast<int64_t>(orow) * H + h] = l;
      }
    }
    m = -FLT_MAX; l = 0.f;
    ++cur_row;
    cur_end = __shfl(rp, lane0 + min(cur_row - r_begin + 1, LPR - 1));
  };

  for (int base = q0; base < q_end; base += LPR) {
    const int n = min(LPR, q_end - base);
    const int my_col = (li < n) ? col[base + li] : 0;
    for (int j = 0; j < n; j += kPmaUnroll) {
      Raw<T, VEC> raw[kPmaUnroll];
      float a[kPmaUnroll];
#pragma unroll
      for (int u = 0; u < kPmaUnroll; ++u) {
        const int jj = j + u;
        const int src = __shfl(my_col, lane0 + (jj & (LPR - 1)));
        a[u] = 0.f;
        raw[u] = zero_raw<T, VEC>();
        if (jj < n && active) {
          a[u] = alpha[static_cast<int64_t>(src) * lda + h];
          raw[u] = load_raw<T, VEC>(V + static_cast<int64_t>(src) * ldv + c0);
        }
      }
#pragma unroll
      for (int u = 0; u < kPmaUnroll; ++u) {
        const int pos = base + j + u;
        if (j + u < n) {
          while (pos >= cur_end) flush();
          const FVec<VEC> vu = unpack<T, VEC>(raw[u]);
          const float av = leaky_relu(a[u], slope);
          const float m_new = fmaxf(m, av);
          const float sc = __expf(m - m_new);
          const float pe = __expf(av - m_new);
          l = fmaf(l, sc, pe);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = fmaf(acc[k], sc, pe * vu.v[k]);
          m = m_new;
        }
      }
    }
  }
  while (cur_row < r_end) flush();
}

// p[j,h] in CSR order, for return_attention_weights
__global__ __launch_bounds__(kBlock) void pma_attention_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ alpha,
    const float* __restrict__ m, const float* __restrict__ l, float slope, float* __restrict__ p, int n_t, int H) {
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int row = static_cast<int>(blk) * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= n_t) return;
  const int lane = lane_id();
  const int start = rowptr[row], end = rowptr[row + 1];
  const int64_t items = static_cast<int64_t>(end - start) * H;
  for (int64_t k = lane; k < items; k += kWave) {
    const int inc = static_cast<int>(k / H), h = static_cast<int>(k % H);
    const float a = leaky_relu(alpha[static_cast<int64_t>(col[start + inc]) * H + h], slope);
    const int64_t th = static_cast<int64_t>(row) * H + h;
    p[(static_cast<int64_t>(start) + inc) * H + h] = __expf(a - m[th]) / (l[th] + kSoftmaxEps);
  }
}

// stats[t,h] = {M, delta}:  M = m + log(l + eps)  (so that p = exp(a - M)),  delta = <out[t,h,:], gout[t,h,:]>
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void pma_bwd_stats_kernel(
    const T* __restrict__ out, int64_t ldo, const T* __restrict__ gout, int64_t ldg,
    const float* __restrict__ m, const float* __restrict__ l, float* __restrict__ stats, int64_t lds, int n_t, int H, int C) {
  __shared__ float red[kWavesPerBlock][kMaxHeads];
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int wave = threadIdx.x >> 6;
  const int row = static_cast<int>(blk) * kWavesPerBlock + wave;
  if (row >= n_t) return;
  const int lane = lane_id();
  const int li = lane % LPR, slot = lane / LPR;
  const int d = H * C, G = C / VEC;
  for (int h = lane; h < H; h += kWave) red[wave][h] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  for (int cb = 0; cb < d; cb += LPR * VEC) {
    const int c0 = cb + li * VEC;
    const bool active = (c0 < d) && slot == 0;
    const int h = (c0 < d) ? c0 / C : 0;
    const int q = (c0 < d) ? (c0 % C) / VEC : 0;
    float part = 0.f;
    if (active) {
      const FVec<VEC> o = load_vec<T, VEC>(out + static_cast<int64_t>(row) * ldo + c0);
      const FVec<VEC> g = load_vec<T, VEC>(gout + static_cast<int64_t>(row) * ldg + c0);
#pragma unroll
      for (int k = 0; k < VEC; ++k) part = fmaf(o.v[k], g.v[k], part);
    }
    const int n_act = min(LPR, (d - cb) / VEC);
    const int grp_end = min(li - q + G, n_act);
    part = head_group_reduce<LPR>(part, li, grp_end);
    if (active && (q == 0 || li == 0)) atomicAdd(&red[wave][h], part);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  for (int h = lane; h < H; h += kWave) {
    const int64_t th = static_cast<int64_t>(row) * H + h;
    const float lv = l[th];
    float2 s;
    s.x = lv > 0.f ? m[th] + __logf(lv + kSoftmaxEps) : FLT_MAX;   // empty target: never gathered; exp(a - FLT_MAX) = 0
    s.y = red[wave][h];
    *reinterpret_cast<float2*>(stats + static_cast<int64_t>(row) * lds + h * 2) = s;
  }
}


// Single-chunk variant of pma_bwd_stats (H*C <= LPR*VEC): every LPR-lane slot takes its own rows (4 in flight), the
// head sums stay inside the slot (no LDS), so a d = 128 fp32 row uses 32 lanes instead of idling half the wave.
constexpr int kStatsRows = 4;

template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void pma_bwd_stats_flat_kernel(
    const T* __restrict__ out, int64_t ldo, const T* __restrict__ gout, int64_t ldg,
    const float* __restrict__ m, const float* __restrict__ l, float* __restrict__ stats, int64_t lds, int n_t, int H, int C) {
  constexpr int NS = kWave / LPR;
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int d = H * C, G = C / VEC;
  const int c0 = li * VEC;
  const bool active = c0 < d;
  const int h = active ? c0 / C : 0;
  const int q = active ? (c0 % C) / VEC : 0;
  const int n_act = min(LPR, d / VEC);
  const int grp_end = min(li - q + G, n_act);
  const int64_t row0 = ((static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)) * NS + slot) * kStatsRows;
  Raw<T, VEC> o[kStatsRows], g[kStatsRows];
#pragma unroll
  for (int r = 0; r < kStatsRows; ++r) {
    o[r] = zero_raw<T, VEC>();
    g[r] = zero_raw<T, VEC>();
    if (active && row0 + r < n_t) {
      o[r] = load_raw<T, VEC>(out + (row0 + r) * ldo + c0);
      g[r] = load_raw<T, VEC>(gout + (row0 + r) * ldg + c0);
    }
  }
#pragma unroll
  for (int r = 0; r < kStatsRows; ++r) {
    const FVec<VEC> ov = unpack<T, VEC>(o[r]), gv = unpack<T, VEC>(g[r]);
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) part = fmaf(ov.v[k], gv.v[k], part);
    part = head_group_reduce<LPR>(part, li, grp_end);
    if (active && q == 0 && row0 + r < n_t) {
      const int64_t th = (row0 + r) * H + h;
      const float lv = l[th];
      float2 st;
      st.x = lv > 0.f ? m[th] + __logf(lv + kSoftmaxEps) : FLT_MAX;
      st.y = part;
      *reinterpret_cast<float2*>(stats + (row0 + r) * lds + h * 2) = st;
    }
  }
}

template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void pma_bwd_src_kernel(
    const int32_t* __restrict__ rowptrT, const int32_t* __restrict__ colT, const float* __restrict__ alpha,
    const T* __restrict__ V, int64_t ldv, const T* __restrict__ gout, int64_t ldg,
    const float* __restrict__ stats, int64_t lds, float slope, T* __restrict__ gV, int64_t ldgv,
    float* __restrict__ galpha, int n_s, int H, int C, const int32_t* __restrict__ row_order) {
  constexpr int NS = kWave / LPR;
  __shared__ float red[kWavesPerBlock][kMaxHeads];
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int wave = threadIdx.x >> 6;
  const int slot_row = static_cast<int>(blk) * kWavesPerBlock + wave;
  if (slot_row >= n_s) return;
  const int row = row_order ? row_order[slot_row] : slot_row;      // long rows first (see segreduce_kernel)
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int start = rowptrT[row], end = rowptrT[row + 1];
  const int d = H * C, G = C / VEC;
  for (int h = lane; h < H; h += kWave) red[wave][h] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  for (int cb = 0; cb < d; cb += LPR * VEC) {
    const int c0 = cb + li * VEC;
    const bool active = c0 < d;
    const int h = active ? c0 / C : 0;
    const int q = active ? (c0 % C) / VEC : 0;
    FVec<VEC> vown;
    float a_s = 0.f;
    if (active) {
      vown = load_vec<T, VEC>(V + static_cast<int64_t>(row) * ldv + c0);
      a_s = leaky_relu(alpha[static_cast<int64_t>(row) * H + h], slope);
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) vown.v[k] = 0.f;
    }
    float gv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) gv[k] = 0.f;
    float S = 0.f, D = 0.f;

    for (int base = start; base < end; base += kWave) {
      const int n = min(kWave, end - base);
      const int my_col = (lane < n) ? colT[base + lane] : 0;
      for (int j = 0; j < n; j += NS * kPmaUnroll) {
        Raw<T, VEC> g[kPmaUnroll];
        float2 st[kPmaUnroll];
        bool ok[kPmaUnroll];
#pragma unroll
        for (int u = 0; u < kPmaUnroll; ++u) {
          const int jj = j + u * NS + slot;
          ok[u] = (jj < n) && active;
          const int t = __shfl(my_col, jj & (kWave - 1));
          if (ok[u]) {
            g[u] = load_raw<T, VEC>(gout + static_cast<int64_t>(t) * ldg + c0);
            st[u] = *reinterpret_cast<const float2*>(stats + static_cast<int64_t>(t) * lds + h * 2);
          }
        }
#pragma unroll
        for (int u = 0; u < kPmaUnroll; ++u) {
          if (ok[u]) {
            const float p = __expf(a_s - st[u].x);
            const FVec<VEC> gu = unpack<T, VEC>(g[u]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) gv[k] = fmaf(p, gu.v[k], gv[k]);
            D = fmaf(p, st[u].y, D);
          }
        }
      }
    }
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) gv[k] += __shfl_xor(gv[k], off);
      D += __shfl_xor(D, off);
    }
    // sum_j p_j <V_s, gO_j> = <V_s, sum_j p_j gO_j> = <V_s, gV_s>: no per-incidence dot product is needed
#pragma unroll
    for (int k = 0; k < VEC; ++k) S = fmaf(vown.v[k], gv[k], S);
    if (slot == 0 && active) {
      FVec<VEC> r;
#pragma unroll
      for (int k = 0; k < VEC; ++k) r.v[k] = gv[k];
      store_vec<T, VEC>(gV + static_cast<int64_t>(row) * ldgv + c0, r);
    }
    // per-head: sum_lanes S  -  D (D is identical in every lane of the head; subtract it once)
    const int n_act = min(LPR, (d - cb) / VEC);
    const int grp_end = min(li - q + G, n_act);
    float part = (slot == 0 && active) ? S : 0.f;
    part = head_group_reduce<LPR>(part, li, grp_end);
    if (slot == 0 && active && (q == 0 || li == 0)) atomicAdd(&red[wave][h], part - (q == 0 ? D : 0.f));
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  for (int h = lane; h < H; h += kWave) {
    const float av = alpha[static_cast<int64_t>(row) * H + h];
    galpha[static_cast<int64_t>(row) * H + h] = (av > 0.f ? 1.f : slope) * red[wave][h];
  }
}


// ---- short-row variant of pma_bwd_src: each slot owns kPmaFlatRows consecutive SOURCE rows (transposed CSR) and walks
// their incidences as one stream.  The rows' own logits and V rows are requested up front (static shift registers keep
// the indexing compile-time), so a row end costs VEC FMAs, one head-group reduction and two stores.
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void pma_bwd_src_flat_kernel(
    const int32_t* __restrict__ rowptrT, const int32_t* __restrict__ colT, const float* __restrict__ alpha,
    const T* __restrict__ V, int64_t ldv, const T* __restrict__ gout, int64_t ldg,
    const float* __restrict__ stats, int64_t lds, float slope, T* __restrict__ gV, int64_t ldgv,
    float* __restrict__ galpha, int n_s, int H, int C, const int32_t* __restrict__ row_ids) {
  // row_ids (optional): compacted transposed CSR -- its row i is source row row_ids[i] (see pma_fwd_flat_kernel)
  constexpr int NS = kWave / LPR;
  constexpr int R = kPmaFlatRows;
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int lane0 = slot * LPR;
  const int64_t slot_global = (static_cast<int64_t>(blk) * kWavesPerBlock + (threadIdx.x >> 6)) * NS + slot;
  const int64_t r_begin64 = slot_global * R;
  if (r_begin64 - static_cast<int64_t>(slot) * R >= n_s) return;
  const int r_begin = static_cast<int>(min(r_begin64, static_cast<int64_t>(n_s)));
  const int r_end = min(r_begin + R, n_s);
  const int d = H * C, G = C / VEC;
  const int c0 = li * VEC;
  const bool active = c0 < d;
  const int h = active ? c0 / C : 0;
  const int q = active ? (c0 % C) / VEC : 0;
  const int n_act = min(LPR, d / VEC);
  const int grp_end = min(li - q + G, n_act);
  const int rp = (li <= r_end - r_begin) ? rowptrT[r_begin + li] : 0;
  const int rid = (li < r_end - r_begin) ? (row_ids ? row_ids[r_begin + li] : r_begin + li) : 0;      // source row of slot row li
  const int q0 = __shfl(rp, lane0);
  const int q_end = __shfl(rp, lane0 + (r_end - r_begin));
  // own-row data of the slot's rows, requested now, consumed at the row ends
  float a_raw[R];
  Raw<T, VEC> v_raw[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    a_raw[i] = 0.f;
    v_raw[i] = zero_raw<T, VEC>();
    const int srow = __shfl(rid, lane0 + i);
    if (active && r_begin + i < r_end) {
      a_raw[i] = alpha[static_cast<int64_t>(srow) * H + h];
      v_raw[i] = load_raw<T, VEC>(V + static_cast<int64_t>(srow) * ldv + c0);
    }
  }
  int cur_row = r_begin;
  int cur_end = (r_begin < r_end) ? __shfl(rp, lane0 + 1) : q0;
  float a_s = leaky_relu(a_raw[0], slope);
  float gv[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) gv[k] = 0.f;
  float D = 0.f;

  auto flush = [&]() {
    const FVec<VEC> vown = unpack<T, VEC>(v_raw[0]);
    const int orow = __shfl(rid, lane0 + min(cur_row - r_begin, LPR - 1));
    float S = 0.f;
    if (active) {
      FVec<VEC> r;
#pragma unroll
      for (int k = 0; k < VEC; ++k) { r.v[k] = gv[k]; S = fmaf(vown.v[k], gv[k], S); gv[k] = 0.f; }
      store_vec<T, VEC>(gV + static_cast<int64_t>(orow) * ldgv + c0, r);
    }
    S = head_group_reduce<LPR>(S, li, grp_end);
    if (active && q == 0)
      galpha[static_cast<int64_t>(orow) * H + h] = (a_raw[0] > 0.f ? 1.f : slope) * (S - D);
    D = 0.f;
#pragma unroll
    for (int i = 0; i + 1 < R; ++i) { a_raw[i] = a_raw[i + 1]; v_raw[i] = v_raw[i + 1]; }
    a_s = leaky_relu(a_raw[0], slope);
    ++cur_row;
    cur_end = __shfl(rp, lane0 + min(cur_row - r_begin + 1, LPR - 1));
  };

  for (int base = q0; base < q_end; base += LPR) {
    const int n = min(LPR, q_end - base);
    const int my_col = (li < n) ? colT[base + li] : 0;
    for (int j = 0; j < n; j += kPmaUnroll) {
      Raw<T, VEC> g[kPmaUnroll];
      float2 st[kPmaUnroll];
#pragma unroll
      for (int u = 0; u < kPmaUnroll; ++u) {
        const int jj = j + u;
        const int t = __shfl(my_col, lane0 + (jj & (LPR - 1)));
        g[u] = zero_raw<T, VEC>();
        st[u] = make_float2(0.f, 0.f);
        if (jj < n && active) {
          g[u] = load_raw<T, VEC>(gout + static_cast<int64_t>(t) * ldg + c0);
          st[u] = *reinterpret_cast<const float2*>(stats + static_cast<int64_t>(t) * lds + h * 2);
        }
      }
#pragma unroll
      for (int u = 0; u < kPmaUnroll; ++u) {
        const int pos = base + j + u;
        if (j + u < n) {
          while (pos >= cur_end) flush();
          const float p = __expf(a_s - st[u].x);
          const FVec<VEC> gu = unpack<T, VEC>(g[u]);
#pragma unroll
          for (int k = 0; k < VEC; ++k) gv[k] = fmaf(p, gu.v[k], gv[k]);
          D = fmaf(p, st[u].y, D);
        }
      }
    }
  }
  while (cur_row < r_end) flush();
}

static inline unsigned row_grid(int64_t rows) { return static_cast<unsigned>((rows + kWavesPerBlock - 1) / kWavesPerBlock); }

static inline int pick_lpr(int64_t d, int vec) {
  const int64_t need = (d + vec - 1) / vec;
  int lpr = 8;
  while (lpr < need && lpr < 64) lpr <<= 1;
  return lpr;
}

#define ALLSET_PMA_DISPATCH_T(KERNEL, T, WIDE, GRID, ST, ...)                                \
  do {                                                                                       \
    if (wide_ok) {                                                                           \
      switch (pick_lpr(d, WIDE)) {                                                           \
        case 8:  KERNEL<T, WIDE, 8><<<GRID, kBlock, 0, ST>>>(__VA_ARGS__); break;            \
        case 16: KERNEL<T, WIDE, 16><<<GRID, kBlock, 0, ST>>>(__VA_ARGS__); break;           \
        case 32: KERNEL<T, WIDE, 32><<<GRID, kBlock, 0, ST>>>(__VA_ARGS__); break;           \
        default: KERNEL<T, WIDE, 64><<<GRID, kBlock, 0, ST>>>(__VA_ARGS__); break;           \
      }                                                                                      \
    } else {                                                                                 \
      KERNEL<T, 1, 64><<<GRID, kBlock, 0, ST>>>(__VA_ARGS__);                                \
    }                                                                                        \
  } while (0)

static int check_pma_dims(const char* who, int64_t n_a, int64_t n_b, int64_t H, int64_t C) {
  ALLSET_REQUIRE(n_a >= 0 && n_b >= 0, "%s: negative size", who);
  ALLSET_REQUIRE(H >= 1 && C >= 1, "%s: heads/channels must be >= 1", who);
  ALLSET_REQUIRE(n_a < INT32_MAX && n_b < INT32_MAX && H * C < INT32_MAX, "%s: size exceeds int32", who);
  if (H > kMaxHeads) {
    set_error("%s: heads=%lld exceeds the built maximum %d", who, static_cast<long long>(H), kMaxHeads);
    return ALLSET_ERR_UNSUPPORTED;
  }
  return ALLSET_OK;
}


// ---- layout change around the column-sharded layer's all-to-all (allset_amd/dist.py) ---------------------------------------
// rows x (P blocks of q 16-byte packets)  <->  P blocks x rows x q packets.  to_blocks: dst[(j*rows + r)*q + t] = src[r*lds + j*q + t]
// (pack: the per-destination column blocks become contiguous); else dst[r*ldd + j*q + t] = src[(j*rows + r)*q + t] (unpack).
// Threads enumerate the ROW-MAJOR side in order, so a wavefront reads (writes) whole rows and touches, per block, the
// consecutive rows of that block -- adjacent in memory -- and every cache line is completed by one wavefront.
__global__ __launch_bounds__(kBlock) void block_transpose_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                                 int64_t rows, int P, int q, int64_t ld, int to_blocks) {
  const int per_row = P * q;
  const int64_t total = rows * per_row;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t r = idx / per_row;
    const int c = static_cast<int>(idx - r * per_row);
    const int j = c / q, t = c - j * q;
    const int64_t rm = r * ld + c, bl = (static_cast<int64_t>(j) * rows + r) * q + t;
    if (to_blocks) dst[bl] = src[rm]; else dst[rm] = src[bl];
  }
}

// ---- cross-shard merge of partial PMA results (multi-GPU E->V) ----------------------------------
// packed[r] = [ out_loc[r, h, :] * w[r, h]  for all h | w[r, 0..H-1] ],  w = l_loc > 0 ? l_loc * exp(m_loc - m_glob) : 0
// -- the per-rank numerators and denominators relative to the GLOBAL row maximum, laid out as one row so that a single
// sum-reduce-scatter merges the ranks.  One pass instead of torch's multiply + concatenate over [N*n, d].
__global__ __launch_bounds__(kBlock) void pma_merge_pack_kernel(
    const float* __restrict__ o, int64_t ldo, const float* __restrict__ m_loc, const float* __restrict__ l_loc,
    const float* __restrict__ m_g, float* __restrict__ packed, int64_t ldp, int64_t n, int H, int C, int vec) {
  const int d = H * C;
  const int per_row = vec ? d / 4 + (H + 3) / 4 : d + H;          // work items per row
  const int64_t total = n * per_row;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t r = idx / per_row;
    const int j = static_cast<int>(idx % per_row);
    auto weight = [&](int h) {
      const float l = l_loc[r * H + h];
      return l > 0.f ? l * __expf(m_loc[r * H + h] - m_g[r * H + h]) : 0.f;
    };
    if (vec) {
      if (j < d / 4) {                                             // C % 4 == 0: the four columns share a head
        const float w = weight((4 * j) / C);
        float4 v = *reinterpret_cast<const float4*>(o + r * ldo + 4 * j);
        v.x *= w; v.y *= w; v.z *= w; v.w *= w;
        *reinterpret_cast<float4*>(packed + r * ldp + 4 * j) = v;
      } else {
        const int h0 = 4 * (j - d / 4);
        for (int h = h0; h < h0 + 4 && h < H; ++h) packed[r * ldp + d + h] = weight(h);
      }
    } else {
      packed[r * ldp + j] = j < d ? o[r * ldo + j] * weight(j / C) : weight(j - d);
    }
  }
}

}  // namespace allset

using namespace allset;

static int pma_fwd_impl(int dtype, int variant, int64_t nnz_hint, const int32_t* row_order, const int32_t* rowptr,
                        const int32_t* col, const float* alpha, int64_t lda, const void* V, int64_t ldv, float slope, void* out,
                        int64_t ldo, float* m, float* l, int64_t n_t, int64_t n_s, int64_t H, int64_t C, void* stream);

extern "C" int allset_pma_fwd(int dtype, const int32_t* rowptr, const int32_t* col, const float* alpha,
                              const void* V, int64_t ldv, float slope, void* out, int64_t ldo, float* m, float* l,
                              int64_t n_t, int64_t n_s, int64_t H, int64_t C, void* stream) {
  return pma_fwd_impl(dtype, 0, -1, nullptr, rowptr, col, alpha, H, V, ldv, slope, out, ldo, m, l, n_t, n_s, H, C, stream);
}

extern "C" int allset_pma_fwd_ex(int dtype, int variant, int64_t nnz, const int32_t* row_order, const int32_t* rowptr,
                                 const int32_t* col, const float* alpha, const void* V, int64_t ldv, float slope,
                                 void* out, int64_t ldo, float* m, float* l, int64_t n_t, int64_t n_s, int64_t H,
                                 int64_t C, void* stream) {
  return pma_fwd_impl(dtype, variant, nnz, row_order, rowptr, col, alpha, H, V, ldv, slope, out, ldo, m, l, n_t, n_s, H, C, stream);
}

extern "C" int allset_pma_fwd_ld(int dtype, int variant, int64_t nnz, const int32_t* row_order, const int32_t* rowptr,
                                 const int32_t* col, const float* alpha, int64_t lda, const void* V, int64_t ldv, float slope,
                                 void* out, int64_t ldo, float* m, float* l, int64_t n_t, int64_t n_s, int64_t H,
                                 int64_t C, void* stream) {
  return pma_fwd_impl(dtype, variant, nnz, row_order, rowptr, col, alpha, lda, V, ldv, slope, out, ldo, m, l, n_t, n_s, H, C, stream);
}

static int pma_fwd_impl(int dtype, int variant, int64_t nnz_hint, const int32_t* row_order, const int32_t* rowptr,
                        const int32_t* col, const float* alpha, int64_t lda, const void* V, int64_t ldv, float slope, void* out,
                        int64_t ldo, float* m, float* l, int64_t n_t, int64_t n_s, int64_t H, int64_t C, void* stream) {
  clear_error();
  ALLSET_REQUIRE(lda >= H, "pma_fwd: logit leading dimension smaller than H");
  ALLSET_REQUIRE(variant >= 0 && variant <= 2, "pma_fwd: bad variant %d", variant);
  int rc = check_pma_dims("pma_fwd", n_t, n_s, H, C);
  if (rc != ALLSET_OK) return rc;
  ALLSET_REQUIRE(dtype == ALLSET_F32 || dtype == ALLSET_BF16, "pma_fwd: bad dtype %d", dtype);
  if (n_t == 0) return ALLSET_OK;
  const int64_t d = H * C;
  ALLSET_REQUIRE(rowptr && out && m && l, "pma_fwd: null rowptr/out/m/l");
  ALLSET_REQUIRE(n_s == 0 || (alpha && V && (col || nnz_hint == 0)), "pma_fwd: null col/alpha/V with n_s > 0");
  ALLSET_REQUIRE(ldv >= d && ldo >= d, "pma_fwd: leading dimension smaller than H*C");
  const int wide = dtype == ALLSET_F32 ? 4 : 8;
  const bool wide_ok = (C % wide == 0) && (ldv % wide == 0) && (ldo % wide == 0) && aligned16(V) && aligned16(out);
  const hipStream_t st = static_cast<hipStream_t>(stream);
  // variant: 0 auto (short-row kernel when nnz / n_t < 6), 1 one wave per row, 2 short-row kernel
  const bool flat_ok = wide_ok && d <= 64 * wide;
  if (variant == 2 && !flat_ok) {
    set_error("pma_fwd: the short-row variant needs 16-byte aligned rows, C %% %d == 0 and H*C <= %d", wide, 64 * wide);
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (flat_ok && (variant == 2 || (variant == 0 && nnz_hint >= 0 && static_cast<double>(nnz_hint) < 6.0 * static_cast<double>(n_t)))) {
    const int lpr = pick_lpr(d, wide);
    const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr) * kPmaFlatRows;
    const unsigned fgrid = static_cast<unsigned>((n_t + rows_per_block - 1) / rows_per_block);
#define ALLSET_PMA_FLAT(T, WIDE, LPRV)                                                                              \
  pma_fwd_flat_kernel<T, WIDE, LPRV><<<fgrid, kBlock, 0, st>>>(rowptr, col, alpha, lda, static_cast<const T*>(V), ldv, slope, \
                                                               static_cast<T*>(out), ldo, m, l, static_cast<int>(n_t),   \
                                                               static_cast<int>(H), static_cast<int>(C), variant == 2 ? row_order : nullptr)
    if (dtype == ALLSET_F32) {
      switch (lpr) { case 8: ALLSET_PMA_FLAT(float, 4, 8); break; case 16: ALLSET_PMA_FLAT(float, 4, 16); break;
                     case 32: ALLSET_PMA_FLAT(float, 4, 32); break; default: ALLSET_PMA_FLAT(float, 4, 64); break; }
    } else {
      switch (lpr) { case 8: ALLSET_PMA_FLAT(bf16_t, 8, 8); break; case 16: ALLSET_PMA_FLAT(bf16_t, 8, 16); break;
                     case 32: ALLSET_PMA_FLAT(bf16_t, 8, 32); break; default: ALLSET_PMA_FLAT(bf16_t, 8, 64); break; }
    }
#undef ALLSET_PMA_FLAT
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
  if (dtype == ALLSET_F32)
    ALLSET_PMA_DISPATCH_T(pma_fwd_kernel, float, 4, row_grid(n_t), st, rowptr, col, alpha, lda, static_cast<const float*>(V), ldv, slope,
                          static_cast<float*>(out), ldo, m, l, static_cast<int>(n_t), static_cast<int>(H), static_cast<int>(C), row_order);
  else
    ALLSET_PMA_DISPATCH_T(pma_fwd_kernel, bf16_t, 8, row_grid(n_t), st, rowptr, col, alpha, lda, static_cast<const bf16_t*>(V), ldv, slope,
                          static_cast<bf16_t*>(out), ldo, m, l, static_cast<int>(n_t), static_cast<int>(H), static_cast<int>(C), row_order);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_pma_attention(const int32_t* rowptr, const int32_t* col, const float* alpha, const float* m,
                                    const float* l, float slope, float* p, int64_t n_t, int64_t H, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n_t >= 0 && H >= 1 && n_t < INT32_MAX && H < INT32_MAX, "pma_attention: bad size");
  if (n_t == 0) return ALLSET_OK;
  ALLSET_REQUIRE(rowptr && col && alpha && m && l && p, "pma_attention: null pointer");
  pma_attention_kernel<<<row_grid(n_t), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      rowptr, col, alpha, m, l, slope, p, static_cast<int>(n_t), static_cast<int>(H));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static int pma_bwd_stats_impl(int dtype, const void* out, int64_t ldo, const void* gout, int64_t ldg, const float* m,
                              const float* l, float* stats, int64_t lds, int64_t n_t, int64_t H, int64_t C, void* stream);

extern "C" int allset_pma_bwd_stats(int dtype, const void* out, int64_t ldo, const void* gout, int64_t ldg,
                                    const float* m, const float* l, float* stats, int64_t n_t, int64_t H, int64_t C,
                                    void* stream) {
  return pma_bwd_stats_impl(dtype, out, ldo, gout, ldg, m, l, stats, 2 * H, n_t, H, C, stream);
}

extern "C" int allset_pma_bwd_stats_ld(int dtype, const void* out, int64_t ldo, const void* gout, int64_t ldg,
                                       const float* m, const float* l, float* stats, int64_t lds, int64_t n_t, int64_t H,
                                       int64_t C, void* stream) {
  return pma_bwd_stats_impl(dtype, out, ldo, gout, ldg, m, l, stats, lds, n_t, H, C, stream);
}

static int pma_bwd_stats_impl(int dtype, const void* out, int64_t ldo, const void* gout, int64_t ldg, const float* m,
                              const float* l, float* stats, int64_t lds, int64_t n_t, int64_t H, int64_t C, void* stream) {
  clear_error();
  ALLSET_REQUIRE(lds >= 2 * H && lds % 2 == 0, "pma_bwd_stats: stats leading dimension must be even and >= 2H");
  int rc = check_pma_dims("pma_bwd_stats", n_t, 0, H, C);
  if (rc != ALLSET_OK) return rc;
  ALLSET_REQUIRE(dtype == ALLSET_F32 || dtype == ALLSET_BF16, "pma_bwd_stats: bad dtype %d", dtype);
  if (n_t == 0) return ALLSET_OK;
  const int64_t d = H * C;
  ALLSET_REQUIRE(out && gout && m && l && stats, "pma_bwd_stats: null pointer");
  ALLSET_REQUIRE(ldo >= d && ldg >= d, "pma_bwd_stats: leading dimension smaller than H*C");
  ALLSET_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 7u) == 0, "pma_bwd_stats: stats must be 8-byte aligned");
  const int wide = dtype == ALLSET_F32 ? 4 : 8;
  const bool wide_ok = (C % wide == 0) && (ldo % wide == 0) && (ldg % wide == 0) && aligned16(out) && aligned16(gout);
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (wide_ok && d <= 64 * wide) {        // the whole row in one chunk: slot-per-row kernel, no LDS
    const int lpr = pick_lpr(d, wide);
    const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr) * kStatsRows;
    const unsigned fgrid = static_cast<unsigned>((n_t + rows_per_block - 1) / rows_per_block);
#define ALLSET_PMA_STATS(T, WIDE, LPRV)                                                                               \
  pma_bwd_stats_flat_kernel<T, WIDE, LPRV><<<fgrid, kBlock, 0, st>>>(static_cast<const T*>(out), ldo,                   \
                                                                     static_cast<const T*>(gout), ldg, m, l, stats, lds, \
                                                                     static_cast<int>(n_t), static_cast<int>(H),        \
                                                                     static_cast<int>(C))
    if (dtype == ALLSET_F32) {
      switch (lpr) { case 8: ALLSET_PMA_STATS(float, 4, 8); break; case 16: ALLSET_PMA_STATS(float, 4, 16); break;
                     case 32: ALLSET_PMA_STATS(float, 4, 32); break; default: ALLSET_PMA_STATS(float, 4, 64); break; }
    } else {
      switch (lpr) { case 8: ALLSET_PMA_STATS(bf16_t, 8, 8); break; case 16: ALLSET_PMA_STATS(bf16_t, 8, 16); break;
                     case 32: ALLSET_PMA_STATS(bf16_t, 8, 32); break; default: ALLSET_PMA_STATS(bf16_t, 8, 64); break; }
    }
#undef ALLSET_PMA_STATS
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
  if (dtype == ALLSET_F32)
    ALLSET_PMA_DISPATCH_T(pma_bwd_stats_kernel, float, 4, row_grid(n_t), st, static_cast<const float*>(out), ldo,
                          static_cast<const float*>(gout), ldg, m, l, stats, lds, static_cast<int>(n_t), static_cast<int>(H),
                          static_cast<int>(C));
  else
    ALLSET_PMA_DISPATCH_T(pma_bwd_stats_kernel, bf16_t, 8, row_grid(n_t), st, static_cast<const bf16_t*>(out), ldo,
                          static_cast<const bf16_t*>(gout), ldg, m, l, stats, lds, static_cast<int>(n_t), static_cast<int>(H),
                          static_cast<int>(C));
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

static int pma_bwd_src_impl(int dtype, int variant, int64_t nnz_hint, const int32_t* row_order, const int32_t* rowptrT,
                            const int32_t* colT, const float* alpha, const void* V, int64_t ldv, const void* gout,
                            int64_t ldg, const float* stats, int64_t lds, float slope, void* gV, int64_t ldgv, float* galpha,
                            int64_t n_s, int64_t n_t, int64_t H, int64_t C, void* stream);

extern "C" int allset_pma_bwd_src(int dtype, const int32_t* rowptrT, const int32_t* colT, const float* alpha,
                                  const void* V, int64_t ldv, const void* gout, int64_t ldg, const float* stats,
                                  float slope, void* gV, int64_t ldgv, float* galpha, int64_t n_s, int64_t n_t,
                                  int64_t H, int64_t C, void* stream) {
  return pma_bwd_src_impl(dtype, 0, -1, nullptr, rowptrT, colT, alpha, V, ldv, gout, ldg, stats, 2 * H, slope, gV, ldgv, galpha, n_s, n_t, H, C, stream);
}

extern "C" int allset_pma_bwd_src_ex(int dtype, int variant, int64_t nnz, const int32_t* row_order,
                                     const int32_t* rowptrT, const int32_t* colT, const float* alpha, const void* V,
                                     int64_t ldv, const void* gout, int64_t ldg, const float* stats, float slope,
                                     void* gV, int64_t ldgv, float* galpha, int64_t n_s, int64_t n_t, int64_t H,
                                     int64_t C, void* stream) {
  return pma_bwd_src_impl(dtype, variant, nnz, row_order, rowptrT, colT, alpha, V, ldv, gout, ldg, stats, 2 * H, slope, gV, ldgv, galpha, n_s, n_t, H, C, stream);
}

extern "C" int allset_pma_bwd_src_ld(int dtype, int variant, int64_t nnz, const int32_t* row_order,
                                     const int32_t* rowptrT, const int32_t* colT, const float* alpha, const void* V,
                                     int64_t ldv, const void* gout, int64_t ldg, const float* stats, int64_t lds, float slope,
                                     void* gV, int64_t ldgv, float* galpha, int64_t n_s, int64_t n_t, int64_t H,
                                     int64_t C, void* stream) {
  return pma_bwd_src_impl(dtype, variant, nnz, row_order, rowptrT, colT, alpha, V, ldv, gout, ldg, stats, lds, slope, gV, ldgv, galpha, n_s, n_t, H, C, stream);
}

static int pma_bwd_src_impl(int dtype, int variant, int64_t nnz_hint, const int32_t* row_order, const int32_t* rowptrT,
                            const int32_t* colT, const float* alpha, const void* V, int64_t ldv, const void* gout,
                            int64_t ldg, const float* stats, int64_t lds, float slope, void* gV, int64_t ldgv, float* galpha,
                            int64_t n_s, int64_t n_t, int64_t H, int64_t C, void* stream) {
  clear_error();
  ALLSET_REQUIRE(variant >= 0 && variant <= 2, "pma_bwd_src: bad variant %d", variant);
  ALLSET_REQUIRE(lds >= 2 * H && lds % 2 == 0, "pma_bwd_src: stats leading dimension must be even and >= 2H");
  int rc = check_pma_dims("pma_bwd_src", n_s, n_t, H, C);
  if (rc != ALLSET_OK) return rc;
  ALLSET_REQUIRE(dtype == ALLSET_F32 || dtype == ALLSET_BF16, "pma_bwd_src: bad dtype %d", dtype);
  if (n_s == 0) return ALLSET_OK;
  const int64_t d = H * C;
  ALLSET_REQUIRE(rowptrT && alpha && V && gV && galpha, "pma_bwd_src: null pointer");
  ALLSET_REQUIRE(n_t == 0 || (gout && stats && (colT || nnz_hint == 0)), "pma_bwd_src: null colT/gout/stats with n_t > 0");
  ALLSET_REQUIRE(ldv >= d && ldg >= d && ldgv >= d, "pma_bwd_src: leading dimension smaller than H*C");
  ALLSET_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 7u) == 0, "pma_bwd_src: stats must be 8-byte aligned");
  const int wide = dtype == ALLSET_F32 ? 4 : 8;
  const bool wide_ok = (C % wide == 0) && (ldv % wide == 0) && (ldg % wide == 0) && (ldgv % wide == 0) && aligned16(V) &&
                       aligned16(gout) && aligned16(gV);
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool flat_ok = wide_ok && d <= 64 * wide;
  if (variant == 2 && !flat_ok) {
    set_error("pma_bwd_src: the short-row variant needs 16-byte aligned rows, C %% %d == 0 and H*C <= %d", wide, 64 * wide);
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (flat_ok && (variant == 2 || (variant == 0 && nnz_hint >= 0 && static_cast<double>(nnz_hint) < 6.0 * static_cast<double>(n_s)))) {
    const int lpr = pick_lpr(d, wide);
    const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * (kWave / lpr) * kPmaFlatRows;
    const unsigned fgrid = static_cast<unsigned>((n_s + rows_per_block - 1) / rows_per_block);
#define ALLSET_PMA_FLATB(T, WIDE, LPRV)                                                                                  \
  pma_bwd_src_flat_kernel<T, WIDE, LPRV><<<fgrid, kBlock, 0, st>>>(rowptrT, colT, alpha, static_cast<const T*>(V), ldv,   \
                                                                   static_cast<const T*>(gout), ldg, stats, lds, slope,    \
                                                                   static_cast<T*>(gV), ldgv, galpha, static_cast<int>(n_s), \
                                                                   static_cast<int>(H), static_cast<int>(C), variant == 2 ? row_order : nullptr)
    if (dtype == ALLSET_F32) {
      switch (lpr) { case 8: ALLSET_PMA_FLATB(float, 4, 8); break; case 16: ALLSET_PMA_FLATB(float, 4, 16); break;
                     case 32: ALLSET_PMA_FLATB(float, 4, 32); break; default: ALLSET_PMA_FLATB(float, 4, 64); break; }
    } else {
      switch (lpr) { case 8: ALLSET_PMA_FLATB(bf16_t, 8, 8); break; case 16: ALLSET_PMA_FLATB(bf16_t, 8, 16); break;
                     case 32: ALLSET_PMA_FLATB(bf16_t, 8, 32); break; default: ALLSET_PMA_FLATB(bf16_t, 8, 64); break; }
    }
#undef ALLSET_PMA_FLATB
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
  if (dtype == ALLSET_F32)
    ALLSET_PMA_DISPATCH_T(pma_bwd_src_kernel, float, 4, row_grid(n_s), st, rowptrT, colT, alpha, static_cast<const float*>(V), ldv,
                          static_cast<const float*>(gout), ldg, stats, lds, slope, static_cast<float*>(gV), ldgv, galpha,
                          static_cast<int>(n_s), static_cast<int>(H), static_cast<int>(C), row_order);
  else
    ALLSET_PMA_DISPATCH_T(pma_bwd_src_kernel, bf16_t, 8, row_grid(n_s), st, rowptrT, colT, alpha, static_cast<const bf16_t*>(V), ldv,
                          static_cast<const bf16_t*>(gout), ldg, stats, lds, slope, static_cast<bf16_t*>(gV), ldgv, galpha,
                          static_cast<int>(n_s), static_cast<int>(H), static_cast<int>(C), row_order);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_pma_merge_pack(const float* out_loc, int64_t ldo, const float* m_loc, const float* l_loc,
                                     const float* m_glob, float* packed, int64_t ldp, int64_t n, int64_t H, int64_t C,
                                     void* stream) {
  clear_error();
  int rc = check_pma_dims("pma_merge_pack", n, 0, H, C);
  if (rc != ALLSET_OK) return rc;
  if (n == 0) return ALLSET_OK;
  const int64_t d = H * C;
  ALLSET_REQUIRE(out_loc && m_loc && l_loc && m_glob && packed, "pma_merge_pack: null pointer");
  ALLSET_REQUIRE(ldo >= d && ldp >= d + H, "pma_merge_pack: leading dimension too small");
  const int vec = (C % 4 == 0 && ldo % 4 == 0 && ldp % 4 == 0 && aligned16(out_loc) && aligned16(packed)) ? 1 : 0;
  const int64_t per_row = vec ? d / 4 + (H + 3) / 4 : d + H;
  const int64_t want = (n * per_row + kBlock - 1) / kBlock;
  const unsigned grid = static_cast<unsigned>(want > 65536 ? 65536 : want);
  pma_merge_pack_kernel<<<grid, kBlock, 0, static_cast<hipStream_t>(stream)>>>(out_loc, ldo, m_loc, l_loc, m_glob, packed, ldp,
                                                                              n, static_cast<int>(H), static_cast<int>(C), vec);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_block_transpose(const void* src, void* dst, int64_t rows, int64_t P, int64_t block_bytes, int64_t ld_bytes,
                                      int to_blocks, void* stream) {
  clear_error();
  ALLSET_REQUIRE(rows >= 0 && P >= 1 && block_bytes >= 16 && block_bytes % 16 == 0, "block_transpose: blocks must be whole 16-byte packets");
  ALLSET_REQUIRE(ld_bytes >= P * block_bytes && ld_bytes % 16 == 0, "block_transpose: bad leading dimension");
  ALLSET_REQUIRE(P * (block_bytes / 16) < (1ll << 30), "block_transpose: row too wide");
  if (rows == 0) return ALLSET_OK;
  ALLSET_REQUIRE(src && dst && aligned16(src) && aligned16(dst), "block_transpose: pointers must be 16-byte aligned");
  const int q = static_cast<int>(block_bytes / 16);
  const int64_t want = (rows * P * q + kBlock - 1) / kBlock;
  const unsigned grid = static_cast<unsigned>(want > 65536 ? 65536 : want);
  block_transpose_kernel<<<grid, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const uint4*>(src), static_cast<uint4*>(dst), rows, static_cast<int>(P), q, ld_bytes / 16, to_blocks);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
