// Phase timing of the blocked Cholesky kernels (s_memtime stamps per wave).  Build on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I pl-viwo_amd/csrc -I include \
//         tools/ubench/bchol_time.hip -o tools/ubench/bchol_time.bin
#ifndef NO_STAMPS
#define PLV_BCHOL_TIMING 1
#endif
#define PLV_BCHOL_NO_LAUNCHERS 1
#include "../../pl-viwo_amd/csrc/blocked_chol.hip"
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
namespace plv { void set_last_error(const char*, ...) {} }

int main() {
  const int k = 104, nc = k + 1, m = 400;   // (the bench's update at workload C: 16 clones x 6 + 8 intrinsics)
  std::mt19937 rng(3);
  std::normal_distribution<double> nd;
  std::vector<double> A((size_t)m * nc), G((size_t)nc * nc, 0.0);
  for (auto& v : A) v = nd(rng);
  for (int i = 0; i < nc; ++i)
    for (int j = 0; j < nc; ++j) {
      double s = 0;
      for (int q = 0; q < m; ++q) s += A[(size_t)i * m + q] * A[(size_t)j * m + q];
      G[(size_t)j * nc + i] = s;
    }
  double *dG, *dR, *dz;
  long long* dst;
  CK(hipMalloc(&dG, G.size() * 8));
  CK(hipMalloc(&dR, (size_t)k * k * 8));
  CK(hipMalloc(&dz, k * 8));
  CK(hipMalloc(&dst, 16 * 64 * 8));
  CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dst, 0, 16 * 64 * 8));
#ifndef NO_STAMPS
  CK(hipMemcpyToSymbol(HIP_SYMBOL(plv::g_bchol_stamps), &dst, sizeof(dst)));
#endif
  auto report = [&](const char *what, int nwaves) {
    std::vector<long long> st2(16 * 64);
    if (hipMemcpy(st2.data(), dst, st2.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    // the critical path panel by panel: the diagonal wave's chain (stamp 1+5p -> 43+p of the wave with the SHORTEST such interval
    // that is positive: strips start early and end late), then everything up to the next panel's start on the next diagonal wave
    long long t00 = st2[0];
    for (int w = 1; w < nwaves; ++w) t00 = std::min(t00, st2[w * 64]);
    long long end = 0;
    for (int w = 0; w < nwaves; ++w) end = std::max(end, st2[w * 64 + 50]);
    printf("%s: %lld ticks from the first load to the end;", what, end - t00);
    for (int p = 0; p < 7; ++p) {
      long long chain = 1ll << 60, start = 0;
      for (int w = 0; w < nwaves; ++w) {
        const long long a = st2[w * 64 + 1 + 5 * p], b = st2[w * 64 + 43 + p];
        if (a > 0 && b > a && b - a < chain) chain = b - a, start = a;
      }
      printf(" p%d chain %lld", p, chain < (1ll << 60) ? chain : -1);
      (void)start;
    }
    printf("\n");
  };
  {  // the EKF solve: S = G[:k, :k] (positive definite), [Mt ; res] = 119 + 1 border rows -> 8 workgroups, stamps of workgroup 0
    const int r = k, n = 121;
    double *dMt, *dres, *dW;
    int *dflag;
    CK(hipMalloc(&dMt, (size_t)n * r * 8));
    CK(hipMalloc(&dres, r * 8));
    CK(hipMalloc(&dW, (size_t)(n + 1) * r * 8));
    CK(hipMalloc(&dflag, 16));
    std::vector<double> Mt((size_t)n * r), res(r);
    for (auto &v : Mt) v = nd(rng);
    for (auto &v : res) v = nd(rng);
    CK(hipMemcpy(dMt, Mt.data(), Mt.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dres, res.data(), res.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dflag, 0, 16));
    CK(hipMemset(dst, 0, 16 * 64 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(plv::bchol_ekf_kernel<7>, dim3((n + 1 + 15) / 16), dim3(64 * 8), 0, 0, dG, nc, r, dMt, r, n, dres, dW, r, dflag, (const int *)nullptr, plv::WhitenC1{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, -1});
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("bchol_ekf_kernel<7>, r = %d, n = %d: %.1f us by events\n", r, n, ms * 1e3);
    {  // 200 launches back to back: (total / 200) = kernel + launch gap, the figure to compare variants on (build with -DNO_STAMPS)
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < 200; ++it)
        hipLaunchKernelGGL(plv::bchol_ekf_kernel<7>, dim3((n + 1 + 15) / 16), dim3(64 * 8), 0, 0, dG, nc, r, dMt, r, n, dres, dW, r, dflag, (const int *)nullptr, plv::WhitenC1{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, -1});
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("bchol_ekf_kernel<7> x 200 back to back: %.2f us per launch\n", ms * 1e3 / 200);
    }
    std::vector<long long> s2(16 * 64);
    CK(hipMemcpy(s2.data(), dst, s2.size() * 8, hipMemcpyDeviceToHost));
    for (int w = 0; w < 8; ++w) {
      long long *s = &s2[w * 64];
      printf("ekf wave %d:", w);
      for (int p = 0; p < 7; ++p)
        printf(" p%d: @%lld chain %lld (+%lld) bar %lld trail %lld |", p, s[1 + 5 * p] - s2[0], s[43 + p] - s[1 + 5 * p], s[2 + 5 * p] - s[43 + p],
               s[3 + 5 * p] - s[2 + 5 * p], s[4 + 5 * p] - s[3 + 5 * p]);
      printf(" end@%lld\n", s[50] - s2[0]);
    }
    report("EKF solve (r = 104, 122 border rows)", 8);
    {  // fingerprints of the results: a restructured hand-over must leave every bit where it was
      std::vector<double> Wh((size_t)(n + 1) * r);
      CK(hipMemcpy(Wh.data(), dW, Wh.size() * 8, hipMemcpyDeviceToHost));
      auto fp = [](const std::vector<double> &v) {
        unsigned long long h = 1469598103934665603ull;
        for (double x : v) {
          unsigned long long b;
          memcpy(&b, &x, 8);
          h = (h ^ b) * 1099511628211ull;
        }
        return h;
      };
      printf("fingerprint: W %016llx\n", fp(Wh));
    }
  }
  return 0;
}
