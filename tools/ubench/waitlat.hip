// waitlat.hip — how long after a kernel's last instruction does the host know?  Three ways of waiting for the same ~30 us kernel:
// hipStreamSynchronize, hipEventSynchronize on an event recorded behind it, and spinning on a word in pinned host memory that the
// kernel's last thread writes (system-scope release).   build: hipcc -O2 --offload-arch=gfx950 waitlat.hip -o waitlat
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <immintrin.h>

__global__ void work(long long ticks, volatile unsigned *flag, unsigned val, double *sink) {
  const long long t0 = wall_clock64();
  double x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) x = x * 1.0000001 + 1e-9;
  if (x == 12345.678) *sink = x;
  if (flag && threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store((unsigned *)flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  unsigned *flag;
  hipHostMalloc((void **)&flag, 64, hipHostMallocDefault);
  double *sink;
  hipMalloc((void **)&sink, 8);
  *flag = 0;
  const long long ticks = 3000;  // wall_clock64 runs at 100 MHz: 30 us
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int mode = 0; mode < 5; ++mode) {
    std::vector<double> t;
    for (int it = 0; it < 300; ++it) {
      const unsigned val = (unsigned)(mode * 1000 + it + 1);
      auto a = now();
      hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, ticks, mode == 2 ? flag : nullptr, val, sink);
      if (mode == 0) {
        hipStreamSynchronize(s);
      } else if (mode == 1) {
        hipEventRecord(ev, s);
        hipEventSynchronize(ev);
      } else if (mode == 2) {
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != val) _mm_pause();
      } else if (mode == 3) {
        hipEventRecord(ev, s);
        while (hipEventQuery(ev) == hipErrorNotReady) _mm_pause();
      } else {
        while (hipStreamQuery(s) == hipErrorNotReady) _mm_pause();
      }
      auto b = now();
      if (it >= 20) t.push_back(us(a, b));
      if (mode == 2) hipStreamSynchronize(s);
    }
    std::sort(t.begin(), t.end());
    printf("%-28s launch + 30 us kernel + wait: p50 %.1f us  p10 %.1f  p90 %.1f\n",
           mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "hipEventSynchronize" : mode == 2 ? "spin on pinned word" : mode == 3 ? "spin on hipEventQuery" : "spin on hipStreamQuery", t[t.size() / 2], t[t.size() / 10],
           t[t.size() * 9 / 10]);
  }
  return 0;
}
