// Shared helpers for the gfx950 kernels behind include/allset_hip.h.  CDNA4 only: wave = 64 lanes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/allset_hip_ext.h"

namespace allset {

constexpr int kWave = 64;
constexpr int kBlock = 256;                 // 4 waves per workgroup, one CSR row per wave
constexpr int kWavesPerBlock = kBlock / kWave;

// ---- error plumbing (thread-local message, integer status, nothing throws) -----------------
void set_error(const char* fmt, ...);
void clear_error();

#define ALLSET_REQUIRE(cond, ...)                      \
  do {                                                 \
    if (!(cond)) {                                     \
      allset::set_error(__VA_ARGS__);                  \
      return ALLSET_ERR_INVALID_ARGUMENT;              \
    }                                                  \
  } while (0)

#define ALLSET_HIP_CHECK(expr)                                                          \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      allset::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ALLSET_ERR_HIP;                                                            \
    }                                                                                   \
  } while (0)

#define ALLSET_LAUNCH_CHECK()                                                           \
  do {                                                                                  \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) {                                                             \
      allset::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return ALLSET_ERR_HIP;                                                            \
    }                                                                                   \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers ---------------------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// The hardware dispatcher places workgroup b on XCD b % 8 (MI355X_MICROARCH.md "Workgroup dispatch").
// Remap so that each XCD walks a CONTIGUOUS range of row-blocks: neighbouring CSR rows (which on
// real hypergraphs share members) then hit the same 4 MiB L2.  Bijective for any grid size; a
// different placement changes speed only.
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned b, unsigned nb) {
  const unsigned xcd = b & 7u, idx = b >> 3;
  const unsigned q = nb >> 3, r = nb & 7u;
  const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// Storage types: float, or bf16 carried as raw 16-bit words (fp32 accumulation everywhere).
struct bf16_t { uint16_t bits; };

__device__ __forceinline__ float bf16_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {            // round to nearest even
  const uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;       // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// ---- fp32 on the bf16 matrix pipe ("bf16x6") --------------------------------------------------------------------
// gfx950's fp32 MFMA runs at 1/16 of its bf16 MFMA (157 TF vs 2.5 PF dense).  An fp32 value splits EXACTLY into three
// bf16 values x = h + m + l (round-to-nearest at each step: 8 + 8 + 8 significant bits cover fp32's 24), and
//   x * w = hh' + hm' + mh' + hl' + lh' + mm'  + (ml' + lm' + ll')
// where every kept product of two bf16 is exact in the fp32 accumulator and the dropped group is <= 2^-23 |x w|
// (one fp32 rounding's worth).  Six bf16 MFMAs replace one fp32 MFMA: 16/6 = 2.7x the matrix throughput at fp32-level
// accuracy, which is what turns the fused Linear kernels from matrix-pipe-bound into HBM-bound.
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {   // {bf16(hi), bf16(lo)} packed, RNE
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// x0, x1 -> packed fp16 planes {hi half: x1, lo half: x0}: h = RN16(x), l = RN16(x - h) (x - h is exact in fp32; fp16 denormals are
// produced and the f16 MFMA honours them) -- "fp16x3": (x s)(w t) = h h' + h l' + l h' + (l l' <= 2^-22, dropped); the operands must have
// been scaled into fp16's window by a power of two (csrc/fused_bwd6.hip has the scheme and its error model)
__device__ __forceinline__ void split2_f16c(float x0, float x1, uint32_t& ph, uint32_t& pl) {
  float r0, r1;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(x0), "v"(x1));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(ph), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(ph), "v"(x1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pl) : "v"(r0), "v"(r1));
}

__device__ __forceinline__ void split3_bf16(float x0, float x1, uint32_t& ph, uint32_t& pm, uint32_t& pl) {
  ph = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(ph << 16), r1 = x1 - __uint_as_float(ph & 0xffff0000u);
  pm = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(pm << 16), s1 = r1 - __uint_as_float(pm & 0xffff0000u);
  pl = cvt_pk_bf16(s0, s1);                                             // exact: s has <= 8 significant bits
}

// VEC consecutive elements, unpacked to float.
template <int VEC>
struct FVec {
  float v[VEC];
};

// A lane's packet exactly as loaded (kept packed while the gather is in flight: 16 B = 4 VGPRs for both
// f32 x 4 and bf16 x 8), unpacked to floats only when it is consumed.
template <typename T, int VEC>
struct Raw;

template <>
struct Raw<float, 4> { float4 r; };
template <>
struct Raw<float, 1> { float r; };
template <>
struct Raw<bf16_t, 8> { uint4 r; };
template <>
struct Raw<bf16_t, 1> { uint16_t r; };

template <typename T, int VEC>
__device__ __forceinline__ Raw<T, VEC> load_raw(const T* __restrict__ p) {
  Raw<T, VEC> t;
  if constexpr (sizeof(T) == 4 && VEC == 4) t.r = *reinterpret_cast<const float4*>(p);
  else if constexpr (sizeof(T) == 4 && VEC == 1) t.r = *reinterpret_cast<const float*>(p);
  else if constexpr (sizeof(T) == 2 && VEC == 8) t.r = *reinterpret_cast<const uint4*>(p);
  else t.r = *reinterpret_cast<const uint16_t*>(p);
  return t;
}

template <typename T, int VEC>
__device__ __forceinline__ Raw<T, VEC> zero_raw() {
  Raw<T, VEC> t;
  if constexpr (sizeof(T) == 4 && VEC == 4) t.r = make_float4(0.f, 0.f, 0.f, 0.f);
  else if constexpr (sizeof(T) == 4 && VEC == 1) t.r = 0.f;
  else if constexpr (sizeof(T) == 2 && VEC == 8) t.r = make_uint4(0u, 0u, 0u, 0u);
  else t.r = 0;
  return t;
}

template <typename T, int VEC>
__device__ __forceinline__ FVec<VEC> unpack(const Raw<T, VEC>& t) {
  FVec<VEC> f;
  if constexpr (sizeof(T) == 4 && VEC == 4) { f.v[0] = t.r.x; f.v[1] = t.r.y; f.v[2] = t.r.z; f.v[3] = t.r.w; }
  else if constexpr (sizeof(T) == 4 && VEC == 1) { f.v[0] = t.r; }
  else if constexpr (sizeof(T) == 2 && VEC == 8) {
    const uint32_t w[4] = {t.r.x, t.r.y, t.r.z, t.r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { f.v[2 * k] = __uint_as_float(w[k] << 16); f.v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
  } else { f.v[0] = bf16_to_f32(t.r); }
  return f;
}

template <typename T, int VEC>
__device__ __forceinline__ FVec<VEC> load_vec(const T* __restrict__ p) { return unpack<T, VEC>(load_raw<T, VEC>(p)); }

template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* __restrict__ p, const FVec<VEC>& f) {
  if constexpr (sizeof(T) == 4 && VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
  else if constexpr (sizeof(T) == 4 && VEC == 1) *reinterpret_cast<float*>(p) = f.v[0];
  else if constexpr (sizeof(T) == 2 && VEC == 8) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = f32_to_bf16(f.v[2 * k]) | (f32_to_bf16(f.v[2 * k + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  } else *reinterpret_cast<uint16_t*>(p) = static_cast<uint16_t>(f32_to_bf16(f.v[0]));
}

template <int VEC>
__device__ __forceinline__ void store_vec_i32(int32_t* __restrict__ p, const int32_t (&a)[VEC]) {
  if constexpr (VEC == 8) {
    *reinterpret_cast<int4*>(p) = make_int4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<int4*>(p + 4) = make_int4(a[4], a[5], a[6], a[7]);
  } else if constexpr (VEC == 4) {
    *reinterpret_cast<int4*>(p) = make_int4(a[0], a[1], a[2], a[3]);
  } else {
    *p = a[0];
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec_i32(const int32_t* __restrict__ p, int32_t (&a)[VEC]) {
  if constexpr (VEC == 4) {
    const int4 t = *reinterpret_cast<const int4*>(p);
    a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) a[k] = p[k];
  }
}

__device__ __forceinline__ float leaky_relu(float x, float slope) { return x > 0.f ? x : x * slope; }

// ---- dropout mask shared by every dense-tail kernel (forward kernels apply it, backward kernels regenerate it).
// Counter-based: one 32-bit hash per PAIR (16 bits per element) or per QUAD (8 bits per element) of consecutive elements -- see
// drop_threshold below for which.
__device__ __forceinline__ uint32_t pair_hash(uint64_t seed, int64_t pair) {
  // A Weyl-scrambled counter keyed by the 64-bit seed, then one xorshift-multiply-xorshift round.  32-bit integer multiplies run
  // at a quarter of the vector rate on gfx950 and a dropout-bearing kernel hashes once per element pair, so the count matters:
  // round 1 used four (two in the key, two in the finaliser) and the hashing cost as many cycles as the kernel's MFMAs; this
  // form uses two plus a full-rate 24-bit one for the upper half of the counter (non-zero only past 2^32 pairs).  Checked on
  // 4M consecutive pairs and four seeds against the old form: keep rates at p = 0.5 / 0.2, correlation between the two 16-bit
  // halves, between neighbouring pairs, between rows and between neighbouring seeds, and the balance of all 32 bits are the
  // same to within sampling noise.
  const uint32_t lo = static_cast<uint32_t>(pair), hi = static_cast<uint32_t>(static_cast<uint64_t>(pair) >> 32);
  uint32_t x = (lo ^ static_cast<uint32_t>(seed)) * 0x9E3779B1U + __umul24(hi, 0x5EBCA7U) + static_cast<uint32_t>(seed >> 32);
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
  return x;
}
// Seeds: entry points take a host-side 64-bit seed by value and, optionally, `seed_base`, a DEVICE pointer to a 64-bit
// counter.  With a counter the effective seed is counter * golden-ratio + seed, read at kernel start -- so a captured
// hipGraph draws fresh masks on every replay (the host value is baked into the graph, the counter is not).
__device__ __forceinline__ uint64_t resolve_seed(const uint64_t* base, uint64_t salt) {
  return base ? (*base) * 0x9E3779B97F4A7C15ULL + salt : salt;
}
// Two resolutions, chosen by p alone (so that every kernel agrees without being told):
//   p * 256 an integer (0.5, 0.25, 0.125, ... -- the reference scripts' dropout 0.5 among them): 8 bits per element, ONE hash per
//   FOUR consecutive elements (counter = idx >> 2, field = idx & 3), keep iff u8 >= p * 256 -- exact, and half the hashes of
//   the 16-bit form below (a dropout-bearing fused kernel spends 10-17 % of its vector cycles hashing);
//   any other p: 16 bits per element, one hash per PAIR (counter = idx >> 1), keep iff u16 >= p * 65536.
// drop_threshold() returns the threshold with the resolution in bit 16 (kDrop8).
constexpr uint32_t kDrop8 = 0x10000u;
__device__ __forceinline__ uint32_t drop_threshold(float p) {
#ifndef ALLSET_ABL_DROP16          // (ablation builds: the 16-bit form for every p)
  const float t8 = p * 256.0f;
  if (t8 == floorf(t8)) return kDrop8 | static_cast<uint32_t>(t8);
#endif
  return static_cast<uint32_t>(p * 65536.0f);
}
__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t idx, uint32_t thr, float inv_keep) {
  if (thr & kDrop8) {
    const uint32_t h = pair_hash(seed, idx >> 2);
    return ((h >> (8 * static_cast<uint32_t>(idx & 3))) & 0xffu) >= (thr & 0xffu) ? inv_keep : 0.f;
  }
  const uint32_t h = pair_hash(seed, idx >> 1);
  const uint32_t u = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return u >= thr ? inv_keep : 0.f;
}
// two consecutive elements starting at an EVEN index: one hash (in the 8-bit form the pair (idx, idx + 1) and the pair
// (idx + 2, idx + 3) of an aligned quad evaluate the SAME hash: two calls on one float4 cost one hash after CSE)
__device__ __forceinline__ void keep_scale2(uint64_t seed, int64_t idx_even, uint32_t thr, float inv_keep, float& s0, float& s1) {
  if (thr & kDrop8) {
    const uint32_t h = pair_hash(seed, idx_even >> 2);
    const uint32_t sh = (static_cast<uint32_t>(idx_even) & 2u) << 3, t8 = thr & 0xffu;
    s0 = ((h >> sh) & 0xffu) >= t8 ? inv_keep : 0.f;
    s1 = ((h >> (sh + 8u)) & 0xffu) >= t8 ? inv_keep : 0.f;
    return;
  }
  const uint32_t h = pair_hash(seed, idx_even >> 1);
  s0 = (h & 0xffffu) >= thr ? inv_keep : 0.f;
  s1 = (h >> 16) >= thr ? inv_keep : 0.f;
}
// four consecutive elements starting at a multiple of FOUR: one hash (8-bit form) or two (16-bit form)
__device__ __forceinline__ float4 keep_scale4(uint64_t seed, int64_t idx4, uint32_t thr, float inv_keep) {
  float4 k;
  if (thr & kDrop8) {
    const uint32_t h = pair_hash(seed, idx4 >> 2), t8 = thr & 0xffu;
    k.x = (h & 0xffu) >= t8 ? inv_keep : 0.f; k.y = ((h >> 8) & 0xffu) >= t8 ? inv_keep : 0.f;
    k.z = ((h >> 16) & 0xffu) >= t8 ? inv_keep : 0.f; k.w = (h >> 24) >= t8 ? inv_keep : 0.f;
    return k;
  }
  keep_scale2(seed, idx4, thr, inv_keep, k.x, k.y);
  keep_scale2(seed, idx4 + 2, thr, inv_keep, k.z, k.w);
  return k;
}

#endif  // __HIPCC__

}  // namespace allset
