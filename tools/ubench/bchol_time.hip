// Phase timing of the blocked Cholesky kernels (s_memtime stamps per wave).  Build on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I pl-viwo_amd/csrc -I include \
//         tools/ubench/bchol_time.hip -o tools/ubench/bchol_time.bin
#define PLV_BCHOL_TIMING 1
#define PLV_BCHOL_NO_LAUNCHERS 1
#include "../../pl-viwo_amd/csrc/blocked_chol.hip"
#include <cstdio>
#include <algorithm>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
namespace plv { void set_last_error(const char*, ...) {} }

int main() {
  const int k = 98, nc = k + 1, m = 400;
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
  CK(hipMemcpyToSymbol(HIP_SYMBOL(plv::g_bchol_stamps), &dst, sizeof(dst)));
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL(plv::bchol_compress_kernel<7>, dim3(1), dim3(64 * 8), 0, 0, dG, nc, dR, k, dz, (const int *)nullptr, (int *)nullptr);
    CK(hipDeviceSynchronize());
  }
  std::vector<long long> st(16 * 64);
  CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
  long long t0 = st[0];
  for (int w = 0; w < 8; ++w) {
    printf("wave %d:", w);
    long long* s = &st[w * 64];
    printf(" load@%lld |", s[0] - t0);
    for (int p = 0; p < 7; ++p)
      printf(" p%d: @%lld chain %lld (+%lld to the barrier) bar %lld trail %lld |", p, s[1 + 5 * p] - t0, s[43 + p] - s[1 + 5 * p],
             s[2 + 5 * p] - s[43 + p], s[3 + 5 * p] - s[2 + 5 * p], s[4 + 5 * p] - s[3 + 5 * p]);
    printf(" end@%lld | last chain: steps0-7 %lld steps8-15 %lld\n", s[50] - t0, s[41] - s[40], s[42] - s[41]);
  }
  {  // the factorisation's length, and the sum over panels of the diagonal wave's chain (the wave with the longest chain of each panel)
    long long chain = 0;
    for (int p = 0; p < 7; ++p) {
      long long best = 0;
      for (int w = 0; w < 7; ++w) best = std::max(best, st[w * 64 + 2 + 5 * p] - st[w * 64 + 1 + 5 * p]);
      chain += best;
    }
    printf("total %lld ticks, diagonal chains %lld (%.0f per pivot)\n", st[50] - t0, chain, chain / 112.0);
  }
  return 0;
}
