// VERDICT r5 item 5 -- "aggregation -> f_dec fused through LDS tiles": the experiment that decides it BEFORE a fused kernel is built.
//
// A fused kernel hosts the gather role and the dense role in ONE launch, so both get the same register allocation and share the CU's
// 32 wave slots and 160 KB of LDS.  The split-role forward (csrc/fused_fwd2.hip, fp16x3) needs 12 waves of 112 registers and 65-83 KB;
// the one-pass backward (csrc/fused_bwd6.hip) 12 waves of 168 registers and 130 KB.  At 112 registers a SIMD holds 4 waves, at 168
// registers 3: the fused forward leaves 16 - 12 = 4 gather waves per CU, the fused backward 12 - 12 = 0 (it would have to shrink its
// vector role to make room).  The stand-alone gather kernel runs 32 waves per CU.  What does the gather RATE do when its waves per
// CU drop?  This program answers with the production kernel itself: segreduce_kernel<float, 4, 32, sum, unweighted> (the bench's
// dominant kernel, included from the library source) on BASELINE configs[2]'s shape, with unused dynamic LDS limiting the number of
// resident workgroups per CU -- 8, 7, 6, 5, 4, 3, 2, 1 workgroups = 32 ... 4 waves per CU (argument: d = 128 or 256).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/micro/segreduce_occupancy.hip -o tools/micro/segreduce_occupancy.bin
//   tools/micro/segreduce_occupancy.bin            (prints one line per occupancy; profiles/r06_agg_dec_fusion.txt)
#include "../../allset_amd/csrc/segreduce.hip"

#include <vector>

namespace allset {
thread_local char g_err[8] = {0};
void set_error(const char*, ...) {}
void clear_error() {}
}  // namespace allset

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

template <int LPR>
static int run(int d);

int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 128;
  return d == 256 ? run<64>(256) : run<32>(128);
}

template <int LPR>
static int run(int d) {
  const int n_s = 1000000, n_t = 1000000, deg = 16;
  const int64_t nnz = static_cast<int64_t>(n_t) * deg;
  std::vector<int32_t> rowptr(n_t + 1), col(nnz);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (int r = 0; r <= n_t; ++r) rowptr[r] = r * deg;
  for (int64_t i = 0; i < nnz; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; col[i] = static_cast<int32_t>((s >> 33) % n_s); }
  std::vector<float> x(static_cast<size_t>(n_s) * d);
  for (size_t i = 0; i < x.size(); ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; x[i] = static_cast<float>((s >> 40) & 0xffff) * (1.f / 65536.f) - 0.5f; }
  int32_t *d_rp, *d_col; float *d_x, *d_out;
  CK(hipMalloc(&d_rp, rowptr.size() * 4)); CK(hipMalloc(&d_col, col.size() * 4));
  CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMalloc(&d_out, static_cast<size_t>(n_t) * d * 4));
  CK(hipMemcpy(d_rp, rowptr.data(), rowptr.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col, col.data(), col.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  auto kern = allset::segreduce_kernel<float, 4, LPR, allset::kModeSum, false>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const double algo = static_cast<double>(nnz) * (4.0 * d + 4) + (n_t + 1) * 4.0 + static_cast<double>(n_t) * 4 * d;
  const unsigned grid = (n_t + allset::kWavesPerBlock - 1) / allset::kWavesPerBlock;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("segreduce_kernel<float,4,%d,sum> on |V| = |E| = 1M, 16 members, d = %d (algorithmic %.3e B per launch); unused dynamic LDS limits the resident workgroups\n", LPR, d, algo);
  printf("%-14s %-14s %-10s %-12s %-8s\n", "dyn LDS (KB)", "waves per CU", "ms", "TB/s algo", "of 8 TB/s");
  const int lds_kb[] = {0, 22, 26, 32, 40, 50, 80, 160};
  const int blocks[] = {8, 7, 6, 5, 4, 3, 2, 1};
  double first_sum = 0.0;
  for (int v = 0; v < 8; ++v) {
    const size_t dyn = static_cast<size_t>(lds_kb[v]) * 1024 - (lds_kb[v] == 160 ? 64 : 0);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), dyn, 0, d_rp, d_col, nullptr, d_x, d, d_out, d, nullptr, n_t, d, 0, 1.f, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int iters = 10;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), dyn, 0, d_rp, d_col, nullptr, d_x, d, d_out, d, nullptr, n_t, d, 0, 1.f, nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    std::vector<float> probe(d);
    CK(hipMemcpy(probe.data(), d_out + static_cast<size_t>(12345) * d, d * 4, hipMemcpyDeviceToHost));
    double sum = 0.0;
    for (float f : probe) sum += f;
    if (v == 0) first_sum = sum;
    printf("%-14d %-14d %-10.3f %-12.2f %-8.3f %s\n", lds_kb[v], blocks[v] * 4, ms, algo / (ms * 1e-3) / 1e12, algo / (ms * 1e-3) / 8e12,
           sum == first_sum ? "" : "(row 12345 differs!)");
  }
  return 0;
}
