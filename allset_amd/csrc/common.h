// Shared helpers for the gfx950 kernels behind include/allset_hip.h.  CDNA4 only: wave = 64 lanes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/allset_hip.h"

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

// VEC consecutive elements as float, 4*VEC (f32) or 2*VEC (bf16) bytes, one global_load per call.
template <int VEC>
struct FVec {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ FVec<VEC> load_vec(const float* __restrict__ p) {
  FVec<VEC> r;
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    r.v[0] = t.x; r.v[1] = t.y;
  } else {
    static_assert(VEC == 1, "VEC must be 1, 2 or 4 for f32");
    r.v[0] = *p;
  }
  return r;
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const FVec<VEC>& r) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
  } else {
    *p = r.v[0];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec_i32(int32_t* __restrict__ p, const int32_t (&a)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<int4*>(p) = make_int4(a[0], a[1], a[2], a[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<int2*>(p) = make_int2(a[0], a[1]);
  } else {
    *p = a[0];
  }
}

__device__ __forceinline__ float leaky_relu(float x, float slope) { return x > 0.f ? x : x * slope; }

#endif  // __HIPCC__

}  // namespace allset
