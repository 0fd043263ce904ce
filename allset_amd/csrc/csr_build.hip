// Device-side construction of the incidence CSR (rowptr / col / perm) from the reference's
// [2, nnz] int64 edge list.  Done ONCE per hypergraph; it replaces the per-forward
// `edge_index[1] -= cidx`, `torch.stack` of the reversed index (reference models.py:453-456) and the
// `index.max()+1` host syncs (layers.py:174,656).
//
// Stable LSD radix sort (rocPRIM via hipCUB) of (row key, original position) pairs: equal keys keep
// the caller's order, so the order of incidences inside a CSR row equals their order in edge_index.
// rowptr comes from a binary search of the sorted keys (no atomics, deterministic).
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace allset {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline int key_bits(int64_t n_rows) {
  int bits = 1;
  while (bits < 32 && (int64_t{1} << bits) < n_rows) ++bits;
  return bits;
}

__global__ void csr_prepare_kernel(const int64_t* __restrict__ row_ids, int64_t row_base, int64_t nnz,
                                   int32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < nnz) {
    keys[i] = static_cast<int32_t>(row_ids[i] - row_base);
    vals[i] = static_cast<int32_t>(i);
  }
}

__global__ void csr_finish_kernel(const int64_t* __restrict__ col_ids, int64_t col_base,
                                  const int32_t* __restrict__ perm, int64_t nnz, int32_t* __restrict__ col) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < nnz) col[i] = static_cast<int32_t>(col_ids[perm[i]] - col_base);
}

// rowptr[r] = number of sorted keys < r  (r = 0 .. n_rows)
__global__ void csr_rowptr_kernel(const int32_t* __restrict__ sorted_keys, int64_t nnz, int64_t n_rows,
                                  int32_t* __restrict__ rowptr) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r > n_rows) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<int64_t>(sorted_keys[mid]) < r) lo = mid + 1; else hi = mid;
  }
  rowptr[r] = static_cast<int32_t>(lo);
}

struct CsrWorkspace {
  size_t keys_in, keys_out, vals_in, cub, total, cub_bytes;
};

static int plan_workspace(int64_t nnz, int64_t n_rows, CsrWorkspace* ws) {
  size_t cub_bytes = 0;
  if (nnz > 0) {
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(
        nullptr, cub_bytes, static_cast<const int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
        static_cast<const int32_t*>(nullptr), static_cast<int32_t*>(nullptr), static_cast<int>(nnz), 0,
        key_bits(n_rows), static_cast<hipStream_t>(nullptr));
    if (e != hipSuccess) {
      set_error("csr_build: hipcub temp-size query failed: %s", hipGetErrorString(e));
      return ALLSET_ERR_HIP;
    }
  }
  const size_t arr = align_up(static_cast<size_t>(nnz) * sizeof(int32_t), 256);
  ws->keys_in = 0;
  ws->keys_out = arr;
  ws->vals_in = 2 * arr;
  ws->cub = 3 * arr;
  ws->cub_bytes = cub_bytes;
  ws->total = 3 * arr + align_up(cub_bytes, 256);
  return ALLSET_OK;
}

}  // namespace allset

using namespace allset;

extern "C" int allset_csr_build_workspace_bytes(int64_t nnz, int64_t n_rows, size_t* bytes) {
  clear_error();
  ALLSET_REQUIRE(bytes != nullptr, "csr_build_workspace_bytes: null output");
  ALLSET_REQUIRE(nnz >= 0 && n_rows >= 0, "csr_build_workspace_bytes: negative size");
  ALLSET_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX, "csr_build_workspace_bytes: size exceeds int32");
  CsrWorkspace ws;
  const int rc = plan_workspace(nnz, n_rows, &ws);
  if (rc != ALLSET_OK) return rc;
  *bytes = ws.total;
  return ALLSET_OK;
}

extern "C" int allset_csr_build(const int64_t* row_ids, const int64_t* col_ids, int64_t nnz, int64_t row_base,
                                int64_t col_base, int64_t n_rows, int32_t* rowptr, int32_t* col, int32_t* perm,
                                void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  ALLSET_REQUIRE(nnz >= 0 && n_rows >= 0, "csr_build: negative size");
  ALLSET_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX, "csr_build: size exceeds int32");
  ALLSET_REQUIRE(rowptr != nullptr, "csr_build: null rowptr");
  ALLSET_REQUIRE(nnz == 0 || (row_ids && col_ids && col && perm), "csr_build: null pointer with nnz > 0");
  const hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int kT = 256;
  if (nnz > 0) {
    CsrWorkspace ws;
    const int rc = plan_workspace(nnz, n_rows, &ws);
    if (rc != ALLSET_OK) return rc;
    if (workspace == nullptr || workspace_bytes < ws.total) {
      set_error("csr_build: workspace of %zu bytes given, %zu needed", workspace_bytes, ws.total);
      return ALLSET_ERR_WORKSPACE;
    }
    ALLSET_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "csr_build: workspace must be 256-byte aligned");
    char* base = static_cast<char*>(workspace);
    int32_t* keys_in = reinterpret_cast<int32_t*>(base + ws.keys_in);
    int32_t* keys_out = reinterpret_cast<int32_t*>(base + ws.keys_out);
    int32_t* vals_in = reinterpret_cast<int32_t*>(base + ws.vals_in);
    void* cub_tmp = base + ws.cub;
    size_t cub_bytes = ws.cub_bytes;
    const unsigned g = static_cast<unsigned>((nnz + kT - 1) / kT);
    csr_prepare_kernel<<<g, kT, 0, st>>>(row_ids, row_base, nnz, keys_in, vals_in);
    ALLSET_LAUNCH_CHECK();
    ALLSET_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys_in, keys_out, vals_in, perm,
                                                        static_cast<int>(nnz), 0, key_bits(n_rows), st));
    csr_finish_kernel<<<g, kT, 0, st>>>(col_ids, col_base, perm, nnz, col);
    ALLSET_LAUNCH_CHECK();
    const unsigned gr = static_cast<unsigned>((n_rows + 1 + kT - 1) / kT);
    csr_rowptr_kernel<<<gr, kT, 0, st>>>(keys_out, nnz, n_rows, rowptr);
    ALLSET_LAUNCH_CHECK();
  } else {
    ALLSET_HIP_CHECK(hipMemsetAsync(rowptr, 0, static_cast<size_t>(n_rows + 1) * sizeof(int32_t), st));
  }
  return ALLSET_OK;
}
