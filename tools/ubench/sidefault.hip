// sidefault.hip — does the first launch on a side stream after the host has waited on the main stream cost page faults?
// Per iteration: [main: kernel, host waits] [side: kernel] [main: kernel] [side: kernel] ... reports minor faults and system time of
// the calling thread around every launch call.   build: hipcc -O2 --offload-arch=gfx950 sidefault.hip -o sidefault
#include <hip/hip_runtime.h>
#include <sys/resource.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

__global__ void work(long long ticks, double *sink) {
  const long long t0 = wall_clock64();
  double x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) x = x * 1.0000001 + 1e-9;
  if (x == 12345.678) *sink = x;
}
static long flt() {
  struct rusage ru;
  getrusage(RUSAGE_THREAD, &ru);
  return ru.ru_minflt;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  double *sink;
  hipMalloc((void **)&sink, 8);
  double acc_f[4] = {0, 0, 0, 0}, acc_t[4] = {0, 0, 0, 0};
  const int iters = 300;
  for (int it = 0; it < iters + 20; ++it) {
    // "frame": main work + host wait
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, 3000LL, sink);
    if (variant == 1) {
      hipStreamSynchronize(s);
    } else {
      hipEventRecord(ev, s);
      hipEventSynchronize(ev);
    }
    if (variant == 2) usleep(200);
    for (int q = 0; q < 4; ++q) {
      hipStream_t st = (q & 1) ? s : s2;  // side, main, side, main
      const long f0 = flt();
      const double t0 = now_us();
      hipLaunchKernelGGL(work, dim3(8), dim3(256), 0, st, 2000LL, sink);
      const double t1 = now_us();
      const long f1 = flt();
      if (it >= 20) acc_f[q] += f1 - f0, acc_t[q] += t1 - t0;
      if (q == 1) {  // wait for the main stream again in the middle, like the frame's second wait
        hipEventRecord(ev, s);
        hipEventSynchronize(ev);
      }
    }
    hipStreamSynchronize(s);
    hipStreamSynchronize(s2);
  }
  const char *names[4] = {"side stream, 1st use", "main stream", "side stream, 2nd use", "main stream"};
  for (int q = 0; q < 4; ++q) printf("variant %d  %-22s launch call: %.2f faults  %.1f us\n", variant, names[q], acc_f[q] / iters, acc_t[q] / iters);
  return 0;
}
