// Do two HIP streams overlap on this part?  A latency-bound kernel (64 one-wave workgroups spinning ~50 us) is launched on one stream,
// then on two streams at once; the host times both cases.    hipcc --offload-arch=gfx950 -O2 -o streams streams.hip && ./streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int *out) {
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < cycles) t = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (int)(t - t0);
}
int main() {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  int *d;
  hipMalloc(&d, 64);
  const long long cyc = 5000;  // wall_clock64 ticks at 100 MHz: 50 us
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    auto t0 = now();
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, cyc, d);
    hipStreamSynchronize(a);
    auto t1 = now();
    for (int i = 0; i < 4; ++i) {
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, cyc, d);
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, cyc, d + 8);
    }
    hipStreamSynchronize(a);
    hipStreamSynchronize(b);
    auto t2 = now();
    printf("4 kernels on one stream: %.1f us;  4 + 4 on two streams: %.1f us\n", us(t0, t1), us(t1, t2));
  }
  return 0;
}
