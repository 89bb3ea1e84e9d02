// h2dlat.hip — what does the upload in front of a kernel cost?  A 30 us kernel that needs `bytes` of host-produced input, three ways:
//   copy    hipMemcpyAsync (pinned -> device) in front of the kernel, kernel reads the device copy
//   direct  no copy command: the kernel reads the pinned host block itself (zero-copy over PCIe), then works
//   none    the kernel alone (reference)
// and the same with an event wait on a second stream in front (cross-stream dependency).   build: hipcc -O2 --offload-arch=gfx950 h2dlat.hip -o h2dlat
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void work(const double *in, int n, long long ticks, double *sink) {
  const long long t0 = wall_clock64();
  double x = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x += in[i];
  while (wall_clock64() - t0 < ticks) x = x * 1.0000001 + 1e-9;
  if (x == 12345.678) *sink = x;
}
__global__ void tiny(double *sink) {
  if (threadIdx.x == 999) *sink = 1.0;
}

int main() {
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t ev, fork, join;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  hipEventCreateWithFlags(&fork, hipEventDisableTiming);
  hipEventCreateWithFlags(&join, hipEventDisableTiming);
  double *sink, *dbuf, *hbuf;
  hipMalloc((void **)&sink, 8);
  const size_t sizes[] = {64 << 10};
  hipMalloc((void **)&dbuf, 256 << 10);
  hipHostMalloc((void **)&hbuf, 256 << 10, hipHostMallocDefault);
  memset(hbuf, 0, 256 << 10);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (size_t bytes : sizes)
    for (int mode = 0; mode < 10; ++mode) {
      std::vector<double> t, tl;
      for (int it = 0; it < 300; ++it) {
        hbuf[it & 1023] = it;
        auto a = now();
        const int n = (int)(bytes / 8);
        if (mode == 0) {
          hipMemcpyAsync(dbuf, hbuf, bytes, hipMemcpyHostToDevice, s);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, n, 3000LL, sink);
        } else if (mode == 1) {
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, hbuf, n, 3000LL, sink);
        } else if (mode == 2) {
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
        } else if (mode == 3) {  // fork a side kernel and join it before the main kernel
          hipEventRecord(fork, s);
          hipStreamWaitEvent(s2, fork, 0);
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, sink);
          hipEventRecord(join, s2);
          hipStreamWaitEvent(s, join, 0);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
        } else if (mode == 4) {  // two dependent kernels on one stream (the cost of a kernel boundary)
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, sink);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
        } else if (mode == 5) {  // the side kernel is long done when the main stream reaches the join: kernel, [join], kernel
          hipEventRecord(fork, s);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
          hipStreamWaitEvent(s2, fork, 0);
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, sink);
          hipEventRecord(join, s2);
          hipStreamWaitEvent(s, join, 0);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
        } else if (mode == 6) {  // reference for the above: kernel, kernel
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
        } else if (mode == 7) {  // the library's shape: upload, fork, kernel, side kernel, join, kernel
          hipMemcpyAsync(dbuf, hbuf, bytes, hipMemcpyHostToDevice, s);
          hipEventRecord(fork, s);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, n, 3000LL, sink);
          hipStreamWaitEvent(s2, fork, 0);
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, sink);
          hipEventRecord(join, s2);
          hipStreamWaitEvent(s, join, 0);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
        } else if (mode == 8) {  // reference: upload, kernel, kernel
          hipMemcpyAsync(dbuf, hbuf, bytes, hipMemcpyHostToDevice, s);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, n, 3000LL, sink);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
        } else {  // fork taken BEFORE the upload: upload, kernel || side kernel, join, kernel
          hipEventRecord(fork, s);
          hipMemcpyAsync(dbuf, hbuf, bytes, hipMemcpyHostToDevice, s);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, n, 3000LL, sink);
          hipStreamWaitEvent(s2, fork, 0);
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, sink);
          hipEventRecord(join, s2);
          hipStreamWaitEvent(s, join, 0);
          hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, dbuf, 0, 3000LL, sink);
        }
        auto l = now();
        hipEventRecord(ev, s);
        hipEventSynchronize(ev);
        auto b = now();
        if (it >= 20) t.push_back(us(a, b)), tl.push_back(us(a, l));
      }
      std::sort(t.begin(), t.end());
      std::sort(tl.begin(), tl.end());
      const char *names[] = {"copy command + kernel", "kernel reads pinned host", "kernel alone", "fork/join side kernel + kernel", "tiny kernel + kernel", "kernel, side kernel joined, kernel", "kernel, kernel", "upload, fork, kernel, side, join, kernel", "upload, kernel, kernel", "fork, upload, kernel, side, join, kernel"};
      printf("%4zu KB  %-32s p50 %.1f us  p10 %.1f  p90 %.1f   (host enqueue p50 %.1f us)\n", bytes >> 10, names[mode], t[t.size() / 2], t[t.size() / 10],
             t[t.size() * 9 / 10], tl[tl.size() / 2]);
    }
  return 0;
}
