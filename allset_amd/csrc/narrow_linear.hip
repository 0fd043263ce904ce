// Backward of a Linear with a NARROW output: the classifier head `Linear(hidden -> num_classes)` (reference models.py:449-456: the
// last layer of `self.classifier = MLP(..., out_channels=num_classes)`; 7 classes on Cora, 6 on Citeseer).  The library's GEMMs
// for [7, n] x [n, 64] run on one workgroup (22 us at n = 2708, the second-longest kernel of a graphed Cora-shaped step), and the
// three pieces (input gradient, weight gradient, bias gradient) are three launches.  Here: ONE kernel, every row tile read once.
//
//   gx[r, :] = gy[r, :] W                        (N <= 16 terms per element)
//   gW[k, j] = sum_r gy[r, k] x[r, j]            per-workgroup partial sums -> part[slice][k * K + j]
//   gb[k]    = sum_r gy[r, k]                                                   part[slice][N * K + k]
//
// A workgroup stages TR = 32 rows of x and gy in LDS (W stays there for the whole kernel); thread t owns the weight-gradient
// elements t, t + 256, ... (consecutive threads -> consecutive input columns: conflict-free LDS reads, gy broadcast) and
// accumulates them in registers over its tiles; the partial sums are reduced by the caller (allset_reduce_partials or the batched
// form) in a fixed order -- no atomics.  HBM-bound streaming of x (n * K * 4 B read + the same written for gx).
#include "common.h"

namespace allset {

constexpr int kNlRows = 32;          // rows per tile
constexpr int kNlMaxN = 16;
constexpr int kNlMaxK = 256;          // LDS: W 16 KB + x tile 32 KB + gy tile 2 KB at the maxima
constexpr int kNlMaxSlices = 256;

template <int EPT>
__global__ __launch_bounds__(kBlock) void linear_narrow_bwd_kernel(const float* __restrict__ gy, int64_t ldg, const float* __restrict__ x,
                                                                   int64_t ldx, const float* __restrict__ W, int64_t n, int N, int K,
                                                                   float* __restrict__ gx, int64_t ldgx, float* __restrict__ part,
                                                                   int64_t part_stride) {
  extern __shared__ float lds[];
  float* w_s = lds;                          // [N][K]
  float* x_s = w_s + N * K;                  // [TR][K]
  float* g_s = x_s + kNlRows * K;            // [TR][kNlMaxN]
  const int t = threadIdx.x;
  const int NK = N * K, KQ = K >> 2;
  for (int e = t * 4; e < NK; e += kBlock * 4) *reinterpret_cast<float4*>(w_s + e) = *reinterpret_cast<const float4*>(W + e);
  float acc[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) acc[i] = 0.f;
  float accb = 0.f;
  const int64_t tiles = (n + kNlRows - 1) / kNlRows;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t r0 = tile * kNlRows;
    const int rows = static_cast<int>(min<int64_t>(kNlRows, n - r0));
    __syncthreads();                         // (previous tile's readers are done; first trip: W is in place)
    for (int q = t; q < kNlRows * KQ; q += kBlock) {
      const int r = q / KQ, c = (q - r * KQ) * 4;
      *reinterpret_cast<float4*>(x_s + r * K + c) =
          r < rows ? *reinterpret_cast<const float4*>(x + (r0 + r) * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int q = t; q < kNlRows * kNlMaxN; q += kBlock) {
      const int r = q / kNlMaxN, k = q % kNlMaxN;
      g_s[q] = (r < rows && k < N) ? gy[(r0 + r) * ldg + k] : 0.f;
    }
    __syncthreads();
    if (gx) {
      for (int q = t; q < rows * KQ; q += kBlock) {
        const int r = q / KQ, c = (q - r * KQ) * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < N; ++k) {
          const float g = g_s[r * kNlMaxN + k];
          const float4 w = *reinterpret_cast<const float4*>(w_s + k * K + c);
          o.x = fmaf(g, w.x, o.x); o.y = fmaf(g, w.y, o.y); o.z = fmaf(g, w.z, o.z); o.w = fmaf(g, w.w, o.w);
        }
        *reinterpret_cast<float4*>(gx + (r0 + r) * ldgx + c) = o;
      }
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = t + i * kBlock;
      if (e < NK) {
        const int k = e / K, j = e - k * K;
        float a = acc[i];
#pragma unroll 8
        for (int r = 0; r < kNlRows; ++r) a = fmaf(g_s[r * kNlMaxN + k], x_s[r * K + j], a);
        acc[i] = a;
      }
    }
    if (t < N) {
      float a = accb;
      for (int r = 0; r < kNlRows; ++r) a += g_s[r * kNlMaxN + t];
      accb = a;
    }
  }
  float* out = part + static_cast<int64_t>(blockIdx.x) * part_stride;
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = t + i * kBlock;
    if (e < NK) out[e] = acc[i];
  }
  if (t < N) out[NK + t] = accb;
}

}  // namespace allset

using namespace allset;

extern "C" int allset_linear_narrow_supported(int64_t N, int64_t K) {
  return (N >= 1 && N <= kNlMaxN && K >= 4 && K <= kNlMaxK && K % 4 == 0) ? 1 : 0;
}

extern "C" int allset_linear_narrow_slices(int64_t n, int64_t* n_slices) {
  clear_error();
  ALLSET_REQUIRE(n >= 0 && n_slices != nullptr, "linear_narrow_slices: bad argument");
  const int64_t tiles = (n + kNlRows - 1) / kNlRows;
  *n_slices = tiles < 1 ? 1 : (tiles < kNlMaxSlices ? tiles : kNlMaxSlices);
  return ALLSET_OK;
}

extern "C" int allset_linear_narrow_bwd(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* W, int64_t n, int64_t N,
                                        int64_t K, float* gx, int64_t ldgx, float* part, int64_t part_stride, int64_t n_slices,
                                        void* stream) {
  clear_error();
  ALLSET_REQUIRE(n >= 0, "linear_narrow_bwd: bad size");
  if (!allset_linear_narrow_supported(N, K)) {
    set_error("linear_narrow_bwd: N <= %d outputs, K <= %d inputs, K %% 4 == 0 (got N = %lld, K = %lld)", kNlMaxN, kNlMaxK,
              static_cast<long long>(N), static_cast<long long>(K));
    return ALLSET_ERR_UNSUPPORTED;
  }
  int64_t want = 0;
  allset_linear_narrow_slices(n, &want);
  ALLSET_REQUIRE(part != nullptr && n_slices == want && part_stride >= N * K + N,
                 "linear_narrow_bwd: part must hold allset_linear_narrow_slices(n) rows of part_stride >= N*K + N floats");
  ALLSET_REQUIRE(W != nullptr && aligned16(W), "linear_narrow_bwd: W null or not 16-byte aligned (dense [N, K])");
  ALLSET_REQUIRE(n == 0 || (gy && x), "linear_narrow_bwd: null pointer");
  ALLSET_REQUIRE(ldg >= N && ldx >= K && (gx == nullptr || ldgx >= K), "linear_narrow_bwd: leading dimension too small");
  ALLSET_REQUIRE(n == 0 || (ldx % 4 == 0 && aligned16(x) && (gx == nullptr || (ldgx % 4 == 0 && aligned16(gx)))),
                 "linear_narrow_bwd: x / gx rows must be 16-byte aligned");
  const size_t lds = static_cast<size_t>(N * K + kNlRows * K + kNlRows * kNlMaxN) * sizeof(float);
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = static_cast<unsigned>(n_slices);
  const int64_t ept = (N * K + kBlock - 1) / kBlock;
#define ALLSET_NL(E)                                                                                                                   \
  linear_narrow_bwd_kernel<E><<<grid, kBlock, lds, st>>>(gy, ldg, x, ldx, W, n, static_cast<int>(N), static_cast<int>(K), gx, ldgx, part, \
                                                         part_stride)
  if (ept <= 2) ALLSET_NL(2);
  else if (ept <= 4) ALLSET_NL(4);
  else if (ept <= 8) ALLSET_NL(8);
  else ALLSET_NL(16);
#undef ALLSET_NL
  ALLSET_LAUNCH_CHECK();
  return ALLSET_OK;
}
