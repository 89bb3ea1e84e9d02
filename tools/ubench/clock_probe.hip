// In-kernel clock and the issue cadence of a lone wave under two launch patterns: back to back, and one ~80 us launch per 0.7 ms
// (the camera step's duty cycle).  clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o clock_probe.bin clock_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void __launch_bounds__(256) probe(unsigned long long *out, float *sink, int iters) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;  // a dependent chain: v_mul, v_add per step
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x] = c1 - c0;
    out[4 * blockIdx.x + 1] = r1 - r0;
  }
  if (x == 12345.678f) sink[0] = x;
}
int main() {
  const int B = 340, iters = 20000;
  unsigned long long *d;
  float *sink;
  hipMalloc(&d, B * 4 * sizeof(unsigned long long));
  hipMalloc(&sink, 4);
  std::vector<unsigned long long> h(B * 4);
  auto run = [&](const char *name, int n, int gap_us) {
    std::vector<double> clk, cyc;
    for (int i = 0; i < n; ++i) {
      hipLaunchKernelGGL(probe, dim3(B), dim3(256), 0, 0, d, sink, iters);
      hipDeviceSynchronize();
      if (gap_us) {
        const auto t = std::chrono::steady_clock::now() + std::chrono::microseconds(gap_us);
        while (std::chrono::steady_clock::now() < t) {
        }
      }
      if (i < n / 2) continue;
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> c, k;
      for (int b = 0; b < B; ++b) c.push_back((double)h[4 * b] / (double)h[4 * b + 1] * 100.0), k.push_back((double)h[4 * b] / (2.0 * iters));
      std::sort(c.begin(), c.end()), std::sort(k.begin(), k.end());
      clk.push_back(c[B / 2]), cyc.push_back(k[B / 2]);
    }
    std::sort(clk.begin(), clk.end()), std::sort(cyc.begin(), cyc.end());
    printf("%-28s in-kernel clock %7.1f MHz (min %7.1f max %7.1f), %5.2f cycles per dependent VALU instruction\n", name, clk[clk.size() / 2], clk.front(),
           clk.back(), cyc[cyc.size() / 2]);
  };
  run("back to back", 400, 0);
  run("one launch per 0.7 ms", 400, 600);
  run("one launch per 3 ms", 200, 3000);
  run("back to back again", 400, 0);
  return 0;
}
