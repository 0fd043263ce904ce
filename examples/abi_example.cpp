// The C ABI used without any Python or torch: device memory from hipMalloc, the caller's own stream, status codes.
// Builds the CSR pair of a small random hypergraph (allset_csr_build), runs the V->E sum aggregation
// (allset_segreduce_fwd, the replacement of layers.py:633-656) and the PMA pooling (allset_pma_fwd, layers.py:145,168-194)
// and checks both against plain loops on the host.
//   hipcc --offload-arch=gfx950 -Iinclude examples/abi_example.cpp -Lallset_amd -lallset_hip -Wl,-rpath,$PWD/allset_amd -o abi_example
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "allset_hip.h"

#define HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 2; } } while (0)
#define ABI_OK(e) do { int _r = (e); if (_r != ALLSET_OK) { std::printf("ABI error %d: %s (line %d)\n", _r, allset_last_error(), __LINE__); return 3; } } while (0)

template <typename T> static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main() {
  // the core surface only (include/allset_hip.h): its own, frozen version
  if (allset_core_version() != ALLSET_CORE_ABI_VERSION) { std::printf("core ABI version mismatch\n"); return 1; }
  const int64_t n_v = 1000, n_e = 300, nnz = 6000, d = 64, H = 4, C = d / H;
  std::srand(7);
  std::vector<int64_t> vid(nnz), eid(nnz);
  for (int64_t i = 0; i < nnz; ++i) { vid[i] = std::rand() % n_v; eid[i] = std::rand() % n_e; }
  std::vector<float> x(n_v * d), alpha(n_v * H);
  for (auto& v : x) v = (std::rand() % 2001 - 1000) / 1000.f;
  for (auto& v : alpha) v = (std::rand() % 2001 - 1000) / 500.f;

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  int64_t *d_vid = to_device(vid), *d_eid = to_device(eid);
  float *d_x = to_device(x), *d_alpha = to_device(alpha);
  int32_t *rowptr, *col, *perm;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&rowptr), (n_e + 1) * 4));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&col), nnz * 4));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&perm), nnz * 4));
  size_t ws_bytes = 0;
  ABI_OK(allset_csr_build_workspace_bytes(nnz, n_e, &ws_bytes));
  void* ws;
  HIP_OK(hipMalloc(&ws, ws_bytes ? ws_bytes : 1));
  // hyperedge-major CSR: rows = hyperedge ids (targets of V->E), cols = member vertices
  ABI_OK(allset_csr_build(d_eid, d_vid, nnz, 0, 0, n_e, rowptr, col, perm, ws, ws_bytes, stream));

  float *out, *pout, *m, *l;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&out), n_e * d * 4));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&pout), n_e * d * 4));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&m), n_e * H * 4));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&l), n_e * H * 4));
  ABI_OK(allset_segreduce_fwd(ALLSET_SUM, ALLSET_F32, rowptr, col, nullptr, d_x, d, out, d, nullptr, n_e, n_v, d, stream));
  ABI_OK(allset_pma_fwd(ALLSET_F32, rowptr, col, d_alpha, d_x, d, 0.2f, pout, d, m, l, n_e, n_v, H, C, stream));
  HIP_OK(hipStreamSynchronize(stream));

  std::vector<float> h_out(n_e * d), h_p(n_e * d);
  HIP_OK(hipMemcpy(h_out.data(), out, h_out.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_p.data(), pout, h_p.size() * 4, hipMemcpyDeviceToHost));

  // host references: scatter-add, and leaky_relu -> segment softmax -> weighted sum per head
  std::vector<double> ref(n_e * d, 0.0), refp(n_e * d, 0.0), mx(n_e * H, -1e30), den(n_e * H, 0.0);
  auto lrelu = [](double a) { return a > 0 ? a : 0.2 * a; };
  for (int64_t i = 0; i < nnz; ++i)
    for (int64_t c = 0; c < d; ++c) ref[eid[i] * d + c] += x[vid[i] * d + c];
  for (int64_t i = 0; i < nnz; ++i)
    for (int64_t h = 0; h < H; ++h) mx[eid[i] * H + h] = std::fmax(mx[eid[i] * H + h], lrelu(alpha[vid[i] * H + h]));
  for (int64_t i = 0; i < nnz; ++i)
    for (int64_t h = 0; h < H; ++h) den[eid[i] * H + h] += std::exp(lrelu(alpha[vid[i] * H + h]) - mx[eid[i] * H + h]);
  for (int64_t i = 0; i < nnz; ++i)
    for (int64_t h = 0; h < H; ++h) {
      const double p = std::exp(lrelu(alpha[vid[i] * H + h]) - mx[eid[i] * H + h]) / den[eid[i] * H + h];
      for (int64_t c = 0; c < C; ++c) refp[eid[i] * d + h * C + c] += p * x[vid[i] * d + h * C + c];
    }
  double e1 = 0, e2 = 0;
  for (size_t i = 0; i < ref.size(); ++i) { e1 = std::fmax(e1, std::fabs(ref[i] - h_out[i])); e2 = std::fmax(e2, std::fabs(refp[i] - h_p[i])); }
  std::printf("segreduce max|err| = %.3g   pma_fwd max|err| = %.3g\n", e1, e2);
  // error path: a negative size must come back as a status + message, never an exception
  const int rc = allset_segreduce_fwd(ALLSET_SUM, ALLSET_F32, rowptr, col, nullptr, d_x, d, out, d, nullptr, -1, n_v, d, stream);
  std::printf("negative size -> status %d (%s)\n", rc, allset_last_error());
  const bool ok = e1 < 1e-4 && e2 < 1e-4 && rc == ALLSET_ERR_INVALID_ARGUMENT;
  std::printf(ok ? "OK\n" : "FAILED\n");
  return ok ? 0 : 4;
}
