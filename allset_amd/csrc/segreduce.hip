// Deep Sets aggregation kernels for gfx950:  out[t,:] = reduce_j w[j] * x[col[j],:]  over CSR rows.
// Replaces the index_select -> mul -> torch_scatter.scatter triple of reference layers.py:633-656
// (three [nnz,d] temporaries + atomics) with ONE gather pass: rows are never materialised per
// incidence, the segment is reduced in registers and written once, coalesced, without atomics.
//
// Mapping to the machine (CDNA4, wave = 64):
//   * one wavefront per CSR row, 4 rows per 256-thread workgroup;
//   * a feature row of d f32 is covered by LPR lanes x 16 B (VEC = 4); with d = 128 that is 32
//     lanes = one 512-B row per half-wave, so a single global_load_dwordx4 gathers NS = 64/LPR
//     different source rows at once (two for d = 128) -- every lane always moves 16 B;
//   * the up-to-64 column ids of the row are fetched with ONE coalesced load (lane j holds id j)
//     and broadcast with ds_bpermute, so the U*NS row gathers of an unrolled step are independent
//     loads issued back-to-back (U = 8: 8 KiB in flight per wave, > 1 MiB per CU at full occupancy);
//   * the NS partial sums are combined with log2(NS) xor-shuffles; slot 0 stores the row.
// HBM-bound by construction: algorithmic bytes per launch are nnz*(4*d + 4) + (n_t+1)*4 + n_t*4*d
// (SURVEY.md section 8(d3)); VALU work is one FMA per gathered element.
#include "common.h"

namespace allset {

enum { kModeSum = 0, kModeExt = 1 };
constexpr int kUnroll = 8;
// gather batches of the one-wave-per-row kernel when a row takes the whole wave (NS = 1: d = 256 f32 / 512 bf16 and wider);
// tools/segreduce_ablation.py rebuilds with other values
#ifndef ALLSET_SEG_UNROLL_ONE_SLOT
#define ALLSET_SEG_UNROLL_ONE_SLOT 8
#endif
#ifndef ALLSET_SEG_MAX_LPR
#define ALLSET_SEG_MAX_LPR 64
#endif
constexpr int kBwdUnroll = 4;      // segmax_bwd: incidences per slot in flight (two 16-byte loads each)

template <typename T, int VEC, int LPR, int MODE, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void segreduce_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
    const T* __restrict__ x, int64_t ldx, T* __restrict__ out, int64_t ldo,
    int32_t* __restrict__ argext, int n_t, int d, int mean, float sign, const int32_t* __restrict__ row_order) {
  constexpr int NS = kWave / LPR;
  constexpr int U = NS == 1 ? ALLSET_SEG_UNROLL_ONE_SLOT : kUnroll;
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int slot_row = static_cast<int>(blk) * kWavesPerBlock + (threadIdx.x >> 6);
  if (slot_row >= n_t) return;  // whole wave exits together
  // optional processing order (skewed degree distributions): the long rows of each XCD's range come first, so that
  // a 4096-member row, which keeps one wave busy for ~0.3 ms, starts at t = 0 instead of setting the kernel's tail
  const int row = row_order ? row_order[slot_row] : slot_row;
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int start = rowptr[row], end = rowptr[row + 1];

  for (int cb = 0; cb < d; cb += LPR * VEC) {
    const int c0 = cb + li * VEC;
    const bool active = c0 < d;
    float acc[VEC];
    int32_t arg[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { acc[k] = 0.f; arg[k] = -1; }

    for (int base = start; base < end; base += kWave) {
      const int n = min(kWave, end - base);
      int my_col = 0;
      float my_w = 0.f;
      if (lane < n) {
        my_col = col[base + lane];
        if constexpr (WEIGHTED) my_w = w[base + lane];
      }
      for (int j = 0; j < n; j += NS * U) {
        Raw<T, VEC> raw[U];        // kept packed while in flight (16 B per lane and load)
        float ww[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jj = j + u * NS + slot;
          ok[u] = (jj < n) && active;
          const int src = __shfl(my_col, jj & (kWave - 1));
          if constexpr (WEIGHTED) ww[u] = __shfl(my_w, jj & (kWave - 1)); else ww[u] = 1.f;
          if (ok[u]) raw[u] = load_raw<T, VEC>(x + static_cast<int64_t>(src) * ldx + c0);
          else raw[u] = zero_raw<T, VEC>();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const FVec<VEC> v = unpack<T, VEC>(raw[u]);
          if constexpr (MODE == kModeSum) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(ww[u], v.v[k], acc[k]);   // ok==false -> v == 0
          } else {
            if (ok[u]) {
              const int pos = base + j + u * NS + slot;
#pragma unroll
              for (int k = 0; k < VEC; ++k) {
                const float val = sign * (ww[u] * v.v[k]);
                if (arg[k] < 0 || val > acc[k]) { acc[k] = val; arg[k] = pos; }
              }
            }
          }
        }
      }
    }

    // combine the NS slots (each reduced a disjoint subset of the row's incidences)
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float o = __shfl_xor(acc[k], off);
        if constexpr (MODE == kModeSum) {
          acc[k] += o;
        } else {
          const int oa = __shfl_xor(arg[k], off);
          if (oa >= 0 && (arg[k] < 0 || o > acc[k] || (o == acc[k] && oa < arg[k]))) { acc[k] = o; arg[k] = oa; }
        }
      }
    }

    if (slot == 0 && active) {
      FVec<VEC> r;
      if constexpr (MODE == kModeSum) {
        const float scale = mean ? 1.f / static_cast<float>(max(end - start, 1)) : 1.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) r.v[k] = acc[k] * scale;
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) r.v[k] = arg[k] >= 0 ? sign * acc[k] : 0.f;   // empty row -> 0
        if (argext != nullptr) store_vec_i32<VEC>(argext + static_cast<int64_t>(row) * d + c0, arg);
      }
      store_vec<T, VEC>(out + static_cast<int64_t>(row) * ldo + c0, r);
    }
  }
}


// ---- short-row variant (sum / mean, single column chunk): several consecutive CSR rows per LPR-lane group ----------
// One wave per row spends a wave launch and three dependent round trips (rowptr -> col -> gather) on every row; at
// degree <= 4 that overhead, not bandwidth, sets the time (profiles: 4M rows take ~1.6 ms whether they hold 1 or 4
// incidences).  Here each LPR-lane group ("slot") owns kFlatRows consecutive rows and walks their incidences as ONE
// stream: the rows' rowptr entries arrive in one load (lane i holds rowptr[r0+i]), the column ids of consecutive rows
// are contiguous in the CSR and arrive LPR at a time, the gathers of a batch are in flight together regardless of row
// boundaries, and a row is flushed (one coalesced store) whenever the stream crosses its end.  Slots never combine.
constexpr int kFlatMinRows = 16384;   // AUTO: below this many target rows the one-group-per-row kernel (see allset_segreduce_fwd)
constexpr int kFlatRows = 7;       // rows per slot; kFlatRows + 1 rowptr entries must fit in the smallest slot (8 lanes)

template <typename T, int VEC, int LPR, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void segreduce_flat_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
    const T* __restrict__ x, int64_t ldx, T* __restrict__ out, int64_t ldo, int n_t, int d, int mean) {
  constexpr int NS = kWave / LPR;
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int lane0 = slot * LPR;                                   // first lane of this slot
  const int64_t slot_global = (static_cast<int64_t>(blk) * kWavesPerBlock + (threadIdx.x >> 6)) * NS + slot;
  const int64_t r_begin64 = slot_global * kFlatRows;
  if (r_begin64 - static_cast<int64_t>(slot) * kFlatRows >= n_t) return;      // whole wave beyond the last row
  const int r_begin = static_cast<int>(min(r_begin64, static_cast<int64_t>(n_t)));
  const int r_end = min(r_begin + kFlatRows, n_t);
  const int c0 = li * VEC;
  const bool active = c0 < d;
  // lane i of the slot holds rowptr[r_begin + i], i = 0 .. r_end - r_begin
  const int rp = (li <= r_end - r_begin) ? rowptr[r_begin + li] : 0;
  const int q0 = __shfl(rp, lane0);
  const int q_end = __shfl(rp, lane0 + (r_end - r_begin));

  int cur_row = r_begin;
  int cur_start = q0;
  int cur_end = (r_begin < r_end) ? __shfl(rp, lane0 + 1) : q0;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;

  auto flush = [&]() {       // store the finished row and step to the next one (slot-uniform control flow)
    if (active) {
      const float scale = mean ? 1.f / static_cast<float>(max(cur_end - cur_start, 1)) : 1.f;
      FVec<VEC> r;
#pragma unroll
      for (int k = 0; k < VEC; ++k) { r.v[k] = acc[k] * scale; acc[k] = 0.f; }
      store_vec<T, VEC>(out + static_cast<int64_t>(cur_row) * ldo + c0, r);
    }
    ++cur_row;
    cur_start = cur_end;
    cur_end = __shfl(rp, lane0 + min(cur_row - r_begin + 1, LPR - 1));
  };

  for (int base = q0; base < q_end; base += LPR) {
    const int n = min(LPR, q_end - base);
    int my_col = 0;
    float my_w = 0.f;
    if (li < n) {
      my_col = col[base + li];
      if constexpr (WEIGHTED) my_w = w[base + li];
    }
    for (int j = 0; j < n; j += kUnroll) {
      Raw<T, VEC> raw[kUnroll];
      float ww[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int jj = j + u;
        const int src = __shfl(my_col, lane0 + (jj & (LPR - 1)));
        if constexpr (WEIGHTED) ww[u] = __shfl(my_w, lane0 + (jj & (LPR - 1))); else ww[u] = 1.f;
        if (jj < n && active) raw[u] = load_raw<T, VEC>(x + static_cast<int64_t>(src) * ldx + c0);
        else raw[u] = zero_raw<T, VEC>();
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int pos = base + j + u;
        if (j + u < n) {
          while (pos >= cur_end) flush();                          // also steps over empty rows
          const FVec<VEC> v = unpack<T, VEC>(raw[u]);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = fmaf(ww[u], v.v[k], acc[k]);
        }
      }
    }
  }
  while (cur_row < r_end) flush();                                 // last row and trailing empty rows
}

// gx[s,c] = sum_{j in T-row s} [argext[colT[j],c] == posT[j]] * wT[j] * gout[colT[j],c]
template <int VEC, int LPR, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void segmax_bwd_kernel(
    const int32_t* __restrict__ rowptrT, const int32_t* __restrict__ colT, const int32_t* __restrict__ posT,
    const float* __restrict__ wT, const int32_t* __restrict__ argext, const float* __restrict__ gout,
    int64_t ldg, float* __restrict__ gx, int64_t ldx, int n_s, int d) {
  constexpr int NS = kWave / LPR;
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int row = static_cast<int>(blk) * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= n_s) return;
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int start = rowptrT[row], end = rowptrT[row + 1];

  for (int cb = 0; cb < d; cb += LPR * VEC) {
    const int c0 = cb + li * VEC;
    const bool active = c0 < d;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int base = start; base < end; base += kWave) {
      const int n = min(kWave, end - base);
      int my_col = 0, my_pos = -2;
      float my_w = 1.f;
      if (lane < n) {
        my_col = colT[base + lane];
        my_pos = posT[base + lane];
        if constexpr (WEIGHTED) my_w = wT[base + lane];
      }
      // kBwdUnroll incidences per slot in flight: each is two independent 16-byte loads (the gradient row and the arg row of
      // the target), issued back to back before any of them is consumed
      for (int j = 0; j < n; j += NS * kBwdUnroll) {
        FVec<VEC> g[kBwdUnroll];
        int32_t a[kBwdUnroll][VEC];
        int pos[kBwdUnroll];
        float ww[kBwdUnroll];
#pragma unroll
        for (int u = 0; u < kBwdUnroll; ++u) {
          const int jj = j + u * NS + slot;
          const bool ok = (jj < n) && active;
          const int t = __shfl(my_col, jj & (kWave - 1));             // (shuffles run with every lane active: before the select)
          const int ps = __shfl(my_pos, jj & (kWave - 1));
          pos[u] = ok ? ps : -2;                                      // -2 never equals an arg entry (>= -1)
          ww[u] = __shfl(my_w, jj & (kWave - 1));
          if (ok) {
            g[u] = load_vec<float, VEC>(gout + static_cast<int64_t>(t) * ldg + c0);
            load_vec_i32<VEC>(argext + static_cast<int64_t>(t) * d + c0, a[u]);
          } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { g[u].v[k] = 0.f; a[u][k] = -1; }
          }
        }
#pragma unroll
        for (int u = 0; u < kBwdUnroll; ++u)
#pragma unroll
          for (int k = 0; k < VEC; ++k)
            if (a[u][k] == pos[u]) acc[k] = fmaf(ww[u], g[u].v[k], acc[k]);
      }
    }
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1)
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off);
    if (slot == 0 && active) {
      FVec<VEC> r;
#pragma unroll
      for (int k = 0; k < VEC; ++k) r.v[k] = acc[k];
      store_vec<float, VEC>(gx + static_cast<int64_t>(row) * ldx + c0, r);
    }
  }
}

// gw[j] = scale(t) * sum_c m(j,c) * x[col[j],c] * gout[t,c]   (one wave per CSR row t; LearnMask path)
template <int MODE>
__global__ __launch_bounds__(kBlock) void sddmm_rowdot_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ x,
    int64_t ldx, const float* __restrict__ gout, int64_t ldg, const int32_t* __restrict__ argext,
    float* __restrict__ gw, int n_t, int d, int mean) {
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int row = static_cast<int>(blk) * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= n_t) return;
  const int lane = lane_id();
  const int start = rowptr[row], end = rowptr[row + 1];
  const float scale = mean ? 1.f / static_cast<float>(max(end - start, 1)) : 1.f;
  const float* g = gout + static_cast<int64_t>(row) * ldg;
  const int32_t* ap = argext ? argext + static_cast<int64_t>(row) * d : nullptr;
  for (int j = start; j < end; ++j) {
    const float* xr = x + static_cast<int64_t>(col[j]) * ldx;
    float part = 0.f;
    for (int c = lane; c < d; c += kWave) {
      bool take = true;
      if constexpr (MODE == kModeExt) take = (ap[c] == j);
      if (take) part = fmaf(xr[c], g[c], part);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) gw[j] = part * scale;
  }
}

// The same product on the segreduce skeleton (round 3): the target's gradient row lives in registers (16 bytes per lane,
// LPR lanes per row), the <= 64 column ids of a batch arrive in one coalesced load and are broadcast by ds_bpermute, the
// kUnroll * NS source-row gathers of a step are independent 16-byte loads issued back to back, every gathered row is
// reduced against the register row by four FMAs + a DPP slot sum, and the step's kUnroll * NS results leave in ONE store
// of contiguous floats (lane li < kUnroll of slot s holds incidence j + li * NS + s).  The scalar kernel above -- a serial
// loop over incidences with 4-byte loads, a 6-step shuffle reduction and a lane-0 store per incidence, nothing in flight --
// ran at 0.3 of the gather roofline (3.25 ms per pass at 1M rows x 16 x 128 where segreduce moves the same rows in 1.19).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int LPR>
__device__ __forceinline__ float slot_sum(float v) {      // sum over the LPR lanes of a slot, result in every lane of it
  v = dpp_add<0xB1>(v);                                   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                                   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);                                  // row_half_mirror: the other quad of the 8
  if constexpr (LPR >= 16) v = dpp_add<0x140>(v);         // row_mirror: the other half of the 16
  if constexpr (LPR >= 32) v += __shfl_xor(v, 16);
  if constexpr (LPR >= 64) v += __shfl_xor(v, 32);
  return v;
}

template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void sddmm_rowdot_vec_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ x,
    int64_t ldx, const float* __restrict__ gout, int64_t ldg, const int32_t* __restrict__ argext,
    float* __restrict__ gw, int n_t, int d, int mean) {
  constexpr int VEC = 4, NS = kWave / LPR;
  static_assert(LPR >= kUnroll, "the packed store needs a lane per unrolled incidence");
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int row = static_cast<int>(blk) * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= n_t) return;
  const int lane = lane_id();
  const int slot = lane / LPR, li = lane % LPR;
  const int start = rowptr[row], end = rowptr[row + 1];
  const float scale = mean ? 1.f / static_cast<float>(max(end - start, 1)) : 1.f;
  const int c0 = li * VEC;
  const bool active = c0 < d;
  FVec<VEC> g;
  int32_t ap[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) { g.v[k] = 0.f; ap[k] = -1; }
  if (active) {
    g = load_vec<float, VEC>(gout + static_cast<int64_t>(row) * ldg + c0);
    if constexpr (MODE == kModeExt) load_vec_i32<VEC>(argext + static_cast<int64_t>(row) * d + c0, ap);
  }
  for (int base = start; base < end; base += kWave) {
    const int n = min(kWave, end - base);
    const int my_col = lane < n ? col[base + lane] : 0;
    for (int j = 0; j < n; j += NS * kUnroll) {
      Raw<float, VEC> raw[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int jj = j + u * NS + slot;
        const int src = __shfl(my_col, jj & (kWave - 1));
        if (jj < n && active) raw[u] = load_raw<float, VEC>(x + static_cast<int64_t>(src) * ldx + c0);
        else raw[u] = zero_raw<float, VEC>();
      }
      float res = 0.f;
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const FVec<VEC> v = unpack<float, VEC>(raw[u]);
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          if constexpr (MODE == kModeExt) part = (ap[k] == base + j + u * NS + slot) ? fmaf(v.v[k], g.v[k], part) : part;
          else part = fmaf(v.v[k], g.v[k], part);
        }
        part = slot_sum<LPR>(part);
        res = (li == u) ? part : res;
      }
      const int jo = j + li * NS + slot;                 // lanes li < kUnroll: kUnroll * NS consecutive incidences
      if (li < kUnroll && jo < n) gw[base + jo] = res * scale;
    }
  }
}

// ---- host-side dispatch ------------------------------------------------------------------------

static inline unsigned row_grid(int64_t rows) { return static_cast<unsigned>((rows + kWavesPerBlock - 1) / kWavesPerBlock); }

// lanes-per-row for the 16-byte path: smallest power of two >= d/vec, in [8, 64]
static inline int pick_lpr(int64_t d, int vec = 4, int max_lpr = 64) {
  const int64_t need = (d + vec - 1) / vec;
  int lpr = 8;
  while (lpr < need && lpr < max_lpr) lpr <<= 1;
  return lpr;
}

template <typename T, int VEC, int LPR>
static void launch_segreduce(int mode_ext, bool weighted, unsigned grid, hipStream_t st,
                             const int32_t* rowptr, const int32_t* col, const float* w, const T* x,
                             int64_t ldx, T* out, int64_t ldo, int32_t* argext, int n_t, int d, int mean,
                             float sign, const int32_t* row_order) {
  if (!mode_ext) {
    if (weighted) segreduce_kernel<T, VEC, LPR, kModeSum, true><<<grid, kBlock, 0, st>>>(rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order);
    else          segreduce_kernel<T, VEC, LPR, kModeSum, false><<<grid, kBlock, 0, st>>>(rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order);
  } else {
    if (weighted) segreduce_kernel<T, VEC, LPR, kModeExt, true><<<grid, kBlock, 0, st>>>(rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order);
    else          segreduce_kernel<T, VEC, LPR, kModeExt, false><<<grid, kBlock, 0, st>>>(rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order);
  }
}

template <typename T, int VEC, int LPR>
static void launch_flat(bool weighted, hipStream_t st, const int32_t* rowptr, const int32_t* col, const float* w,
                        const T* x, int64_t ldx, T* out, int64_t ldo, int n_t, int d, int mean) {
  constexpr int NS = kWave / LPR;
  const int64_t rows_per_block = static_cast<int64_t>(kWavesPerBlock) * NS * kFlatRows;
  const unsigned grid = static_cast<unsigned>((n_t + rows_per_block - 1) / rows_per_block);
  if (weighted) segreduce_flat_kernel<T, VEC, LPR, true><<<grid, kBlock, 0, st>>>(rowptr, col, w, x, ldx, out, ldo, n_t, d, mean);
  else          segreduce_flat_kernel<T, VEC, LPR, false><<<grid, kBlock, 0, st>>>(rowptr, col, w, x, ldx, out, ldo, n_t, d, mean);
}

// short-row path: sum/mean, 16-byte packets, the whole row in one column chunk
template <typename T, int WIDE>
static void dispatch_flat(bool weighted, hipStream_t st, const int32_t* rowptr, const int32_t* col, const float* w,
                          const T* x, int64_t ldx, T* out, int64_t ldo, int n_t, int d, int mean) {
  switch (pick_lpr(d, WIDE)) {
    case 8:  launch_flat<T, WIDE, 8>(weighted, st, rowptr, col, w, x, ldx, out, ldo, n_t, d, mean); break;
    case 16: launch_flat<T, WIDE, 16>(weighted, st, rowptr, col, w, x, ldx, out, ldo, n_t, d, mean); break;
    case 32: launch_flat<T, WIDE, 32>(weighted, st, rowptr, col, w, x, ldx, out, ldo, n_t, d, mean); break;
    default: launch_flat<T, WIDE, 64>(weighted, st, rowptr, col, w, x, ldx, out, ldo, n_t, d, mean); break;
  }
}

template <typename T, int WIDE>
static void dispatch_segreduce(bool wide_ok, int mode_ext, bool weighted, unsigned grid, hipStream_t st,
                               const int32_t* rowptr, const int32_t* col, const float* w, const T* x, int64_t ldx,
                               T* out, int64_t ldo, int32_t* argext, int n_t, int d, int mean, float sign,
                               const int32_t* row_order) {
  if (wide_ok) {
    switch (pick_lpr(d, WIDE, ALLSET_SEG_MAX_LPR)) {
      case 8:  launch_segreduce<T, WIDE, 8>(mode_ext, weighted, grid, st, rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order); break;
      case 16: launch_segreduce<T, WIDE, 16>(mode_ext, weighted, grid, st, rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order); break;
      case 32: launch_segreduce<T, WIDE, 32>(mode_ext, weighted, grid, st, rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order); break;
      default: launch_segreduce<T, WIDE, 64>(mode_ext, weighted, grid, st, rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order); break;
    }
  } else {
    launch_segreduce<T, 1, 64>(mode_ext, weighted, grid, st, rowptr, col, w, x, ldx, out, ldo, argext, n_t, d, mean, sign, row_order);
  }
}

template <int VEC, int LPR>
static void launch_segmax_bwd(bool weighted, unsigned grid, hipStream_t st, const int32_t* rowptrT,
                              const int32_t* colT, const int32_t* posT, const float* wT, const int32_t* argext,
                              const float* gout, int64_t ldg, float* gx, int64_t ldx, int n_s, int d) {
  if (weighted) segmax_bwd_kernel<VEC, LPR, true><<<grid, kBlock, 0, st>>>(rowptrT, colT, posT, wT, argext, gout, ldg, gx, ldx, n_s, d);
  else          segmax_bwd_kernel<VEC, LPR, false><<<grid, kBlock, 0, st>>>(rowptrT, colT, posT, wT, argext, gout, ldg, gx, ldx, n_s, d);
}

}  // namespace allset

using namespace allset;

// mean degree below which the short-row kernel is used (AUTO); see allset_segreduce_fwd_ex
constexpr double kFlatMaxMeanDegree = 6.0;

static int segreduce_impl(int reduce, int dtype, int variant, int64_t nnz_hint, const int32_t* row_order,
                          const int32_t* rowptr, const int32_t* col,
                          const float* w, const void* x, int64_t ldx, void* out, int64_t ldo,
                          int32_t* argext, int64_t n_t, int64_t n_s, int64_t d, void* stream);

extern "C" int allset_segreduce_fwd(int reduce, int dtype, const int32_t* rowptr, const int32_t* col,
                                    const float* w, const void* x, int64_t ldx, void* out, int64_t ldo,
                                    int32_t* argext, int64_t n_t, int64_t n_s, int64_t d, void* stream) {
  return segreduce_impl(reduce, dtype, 0, -1, nullptr, rowptr, col, w, x, ldx, out, ldo, argext, n_t, n_s, d, stream);
}

extern "C" int allset_segreduce_fwd_ex(int reduce, int dtype, int variant, int64_t nnz, const int32_t* row_order,
                                       const int32_t* rowptr, const int32_t* col, const float* w, const void* x,
                                       int64_t ldx, void* out, int64_t ldo, int32_t* argext, int64_t n_t, int64_t n_s,
                                       int64_t d, void* stream) {
  return segreduce_impl(reduce, dtype, variant, nnz, row_order, rowptr, col, w, x, ldx, out, ldo, argext, n_t, n_s, d, stream);
}

static int segreduce_impl(int reduce, int dtype, int variant, int64_t nnz_hint, const int32_t* row_order,
                          const int32_t* rowptr, const int32_t* col,
                          const float* w, const void* x, int64_t ldx, void* out, int64_t ldo,
                          int32_t* argext, int64_t n_t, int64_t n_s, int64_t d, void* stream) {
  clear_error();
  ALLSET_REQUIRE(variant >= 0 && variant <= 2, "segreduce_fwd: bad variant %d", variant);
  ALLSET_REQUIRE(reduce >= ALLSET_SUM && reduce <= ALLSET_MIN, "segreduce_fwd: bad reduce %d", reduce);
  ALLSET_REQUIRE(n_t >= 0 && n_s >= 0 && d >= 0, "segreduce_fwd: negative size");
  ALLSET_REQUIRE(n_t < INT32_MAX && n_s < INT32_MAX && d < INT32_MAX, "segreduce_fwd: size exceeds int32");
  ALLSET_REQUIRE(dtype == ALLSET_F32 || dtype == ALLSET_BF16, "segreduce_fwd: bad dtype %d", dtype);
  if (n_t == 0 || d == 0) return ALLSET_OK;
  ALLSET_REQUIRE(rowptr && out, "segreduce_fwd: null rowptr/out");
  ALLSET_REQUIRE(ldx >= d && ldo >= d, "segreduce_fwd: leading dimension smaller than d");
  // col/x may only be null when there is nothing to gather; without the caller's nnz that cannot be known (no sync
  // here), so require them whenever a source table is declared -- unless nnz == 0 was stated (the _ex entry).
  ALLSET_REQUIRE(n_s == 0 || nnz_hint == 0 || (col && x), "segreduce_fwd: null col/x with n_s > 0");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool ext = (reduce == ALLSET_MAX || reduce == ALLSET_MIN);
  const float sign = (reduce == ALLSET_MIN) ? -1.f : 1.f;
  const int mean = (reduce == ALLSET_MEAN);
  const int wide = dtype == ALLSET_F32 ? 4 : 8;                       // elements in a 16-byte packet
  const bool wide_ok = (d % wide == 0) && (ldx % wide == 0) && (ldo % wide == 0) && aligned16(x) && aligned16(out) &&
                       (argext == nullptr || aligned16(argext));
  const unsigned grid = row_grid(n_t);
  const int nt = static_cast<int>(n_t), di = static_cast<int>(d);
  // variant: 0 = AUTO (short-row kernel when the caller's nnz says the mean degree is small), 1 = one wave per row,
  // 2 = short-row kernel.  The short-row kernel needs sum/mean, 16-byte packets and d <= 64 packets.
  const bool flat_ok = !ext && wide_ok && d <= 64 * wide;
  // (AUTO at dataset scale -- every row gets its own lane group on a machine this size anyway: the one-group-per-row kernel is one
  //  dependent round trip shorter than a slot walking seven rows as a stream, and the step there is a chain of such latencies)
  const bool use_flat = flat_ok && (variant == 2 || (variant == 0 && nnz_hint >= 0 && n_t > kFlatMinRows &&
                                                     static_cast<double>(nnz_hint) < kFlatMaxMeanDegree * static_cast<double>(n_t)));
  if (variant == 2 && !flat_ok) {
    set_error("segreduce_fwd: the short-row variant needs sum/mean, 16-byte aligned rows and d <= %d", 64 * wide);
    return ALLSET_ERR_UNSUPPORTED;
  }
  if (use_flat) {
    if (dtype == ALLSET_F32)
      dispatch_flat<float, 4>(w != nullptr, st, rowptr, col, w, static_cast<const float*>(x), ldx, static_cast<float*>(out), ldo, nt, di, mean);
    else
      dispatch_flat<bf16_t, 8>(w != nullptr, st, rowptr, col, w, static_cast<const bf16_t*>(x), ldx, static_cast<bf16_t*>(out), ldo, nt, di, mean);
    ALLSET_LAUNCH_CHECK();
    return ALLSET_OK;
  }
  if (dtype == ALLSET_F32)
    dispatch_segreduce<float, 4>(wide_ok, ext, w != nullptr, grid, st, rowptr, col, w, static_cast<const float*>(x), ldx,
                                 static_cast<float*>(out), ldo, argext, nt, di, mean, sign, row_order);
  else
    dispatch_segreduce<bf16_t, 8>(wide_ok, ext, w != nullptr, grid, st, rowptr, col, w, static_cast<const bf16_t*>(x), ldx,
                                  static_cast<bf16_t*>(out), ldo, argext, nt, di, mean, sign, row_order);
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_segmax_bwd(const int32_t* rowptrT, const int32_t* colT, const int32_t* posT,
                                 const float* wT, const int32_t* argext, const float* gout, int64_t ldg,
                                 float* gx, int64_t ldx, int64_t n_s, int64_t n_t, int64_t d, void* stream) {
  clear_error();
  ALLSET_REQUIRE(n_t >= 0 && n_s >= 0 && d >= 0, "segmax_bwd: negative size");
  ALLSET_REQUIRE(n_t < INT32_MAX && n_s < INT32_MAX && d < INT32_MAX, "segmax_bwd: size exceeds int32");
  if (n_s == 0 || d == 0) return ALLSET_OK;
  ALLSET_REQUIRE(rowptrT && gx, "segmax_bwd: null rowptrT/gx");
  ALLSET_REQUIRE(n_t == 0 || (colT && posT && argext && gout), "segmax_bwd: null input with n_t > 0");
  ALLSET_REQUIRE(ldg >= d && ldx >= d, "segmax_bwd: leading dimension smaller than d");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec4 = (d % 4 == 0) && (ldg % 4 == 0) && (ldx % 4 == 0) && aligned16(gout) && aligned16(gx);
  const unsigned grid = row_grid(n_s);
  const int ns = static_cast<int>(n_s), di = static_cast<int>(d);
  if (vec4) {
    switch (pick_lpr(d)) {
      case 8:  launch_segmax_bwd<4, 8>(wT != nullptr, grid, st, rowptrT, colT, posT, wT, argext, gout, ldg, gx, ldx, ns, di); break;
      case 16: launch_segmax_bwd<4, 16>(wT != nullptr, grid, st, rowptrT, colT, posT, wT, argext, gout, ldg, gx, ldx, ns, di); break;
      case 32: launch_segmax_bwd<4, 32>(wT != nullptr, grid, st, rowptrT, colT, posT, wT, argext, gout, ldg, gx, ldx, ns, di); break;
      default: launch_segmax_bwd<4, 64>(wT != nullptr, grid, st, rowptrT, colT, posT, wT, argext, gout, ldg, gx, ldx, ns, di); break;
    }
  } else {
    launch_segmax_bwd<1, 64>(wT != nullptr, grid, st, rowptrT, colT, posT, wT, argext, gout, ldg, gx, ldx, ns, di);
  }
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}

extern "C" int allset_sddmm_rowdot(int reduce, const int32_t* rowptr, const int32_t* col, const float* x,
                                   int64_t ldx, const float* gout, int64_t ldg, const int32_t* argext,
                                   float* gw, int64_t n_t, int64_t n_s, int64_t d, void* stream) {
  clear_error();
  ALLSET_REQUIRE(reduce >= ALLSET_SUM && reduce <= ALLSET_MIN, "sddmm_rowdot: bad reduce %d", reduce);
  ALLSET_REQUIRE(n_t >= 0 && n_s >= 0 && d >= 0, "sddmm_rowdot: negative size");
  ALLSET_REQUIRE(n_t < INT32_MAX && n_s < INT32_MAX && d < INT32_MAX, "sddmm_rowdot: size exceeds int32");
  if (n_t == 0 || n_s == 0) return ALLSET_OK;
  ALLSET_REQUIRE(rowptr && col && x && gout && gw, "sddmm_rowdot: null pointer");
  ALLSET_REQUIRE(ldx >= d && ldg >= d, "sddmm_rowdot: leading dimension smaller than d");
  const bool ext = (reduce == ALLSET_MAX || reduce == ALLSET_MIN);
  ALLSET_REQUIRE(!ext || argext, "sddmm_rowdot: MAX/MIN need argext");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = row_grid(n_t);
  const int mean = (reduce == ALLSET_MEAN);
  const int nt = static_cast<int>(n_t), di = static_cast<int>(d);
  // 16-byte path: one column block (d <= 256), aligned rows; anything else takes the scalar kernel
  const bool vec4 = (d % 4 == 0) && d <= 256 && (ldx % 4 == 0) && (ldg % 4 == 0) && aligned16(x) && aligned16(gout) &&
                    (!ext || aligned16(argext));
#define ALLSET_SDDMM(LPR_)                                                                                                   \
  do {                                                                                                                       \
    if (ext) sddmm_rowdot_vec_kernel<LPR_, kModeExt><<<grid, kBlock, 0, st>>>(rowptr, col, x, ldx, gout, ldg, argext, gw, nt, di, mean); \
    else     sddmm_rowdot_vec_kernel<LPR_, kModeSum><<<grid, kBlock, 0, st>>>(rowptr, col, x, ldx, gout, ldg, nullptr, gw, nt, di, mean); \
  } while (0)
  if (vec4) {
    switch (pick_lpr(d)) {
      case 8:  ALLSET_SDDMM(8); break;
      case 16: ALLSET_SDDMM(16); break;
      case 32: ALLSET_SDDMM(32); break;
      default: ALLSET_SDDMM(64); break;
    }
  } else if (ext) {
    sddmm_rowdot_kernel<kModeExt><<<grid, kBlock, 0, st>>>(rowptr, col, x, ldx, gout, ldg, argext, gw, nt, di, mean);
  } else {
    sddmm_rowdot_kernel<kModeSum><<<grid, kBlock, 0, st>>>(rowptr, col, x, ldx, gout, ldg, nullptr, gw, nt, di, mean);
  }
#undef ALLSET_SDDMM
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
