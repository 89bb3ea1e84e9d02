// Does the clock the latency-bound kernels run at depend on how busy the device looks?  One workgroup spins for N seconds (in 100 ms
// launches) while another process runs bench.py; compare its ms_per_step with and without.   usage: ./keepwarm <seconds> [workgroups]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long ticks, int *out) {
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < ticks) t = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (int)(t - t0);
}
int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
  const int groups = argc > 2 ? atoi(argv[2]) : 1;
  int *d;
  hipMalloc(&d, 64);
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipLaunchKernelGGL(spin, dim3(groups), dim3(64), 0, 0, 10000000LL, d);  // 100 ms at 100 MHz
    hipDeviceSynchronize();
  }
  return 0;
}
