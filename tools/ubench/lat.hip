// Latency micro-benchmarks that calibrate the design of the latency-bound update kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void k_barrier(int steps, double* out, long long* clk) {
  __shared__ double s[1024];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  long long r0 = __builtin_amdgcn_s_memrealtime();
  double acc = 0;
  for (int i = 0; i < steps; ++i) {
    acc += s[(threadIdx.x + i) & 1023];
    __syncthreads();
    s[threadIdx.x] = acc;
    __syncthreads();
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

__device__ __forceinline__ double shfl_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return v + __hiloint2double(hi2, lo2);
}
// full wave sum via DPP: row_shr 1,2,4,8 then row_bcast15, row_bcast31; result valid in lane 63
__device__ __forceinline__ double dpp_sum(double v) {
  v = dpp_add<0x111>(v);  // row_shr:1
  v = dpp_add<0x112>(v);  // row_shr:2
  v = dpp_add<0x114>(v);  // row_shr:4
  v = dpp_add<0x118>(v);  // row_shr:8
  v = dpp_add<0x142>(v);  // row_bcast:15
  v = dpp_add<0x143>(v);  // row_bcast:31
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__global__ void k_shfl(int steps, double* out, long long* clk, int mode) {
  double v = threadIdx.x * 1e-3 + 1.0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < steps; ++i) {
    double s = mode == 0 ? shfl_sum(v) : dpp_sum(v);
    v = v * 0.5 + s * 1e-3;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = v;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_math(int steps, double* out, long long* clk, int mode) {
  double v = threadIdx.x * 1e-3 + 1.5;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < steps; ++i) {
    if (mode == 0) v = sqrt(v + 2.0);
    else if (mode == 1) v = 1.0 / (v + 0.5);
    else v = v * 1.0000001 + 0.5;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = v;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_empty() {}

int main() {
  double* out; long long* clk;
  CK(hipMalloc(&out, 1024 * 8)); CK(hipMalloc(&clk, 64));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  long long h[2]; float ms;
  const int steps = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k_barrier, 1, threads, 0, 0, steps, out, clk); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    }
    CK(hipEventElapsedTime(&ms, a, b)); CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    printf("barrier x2 + LDS rw, %4d threads: %.1f ns/step, %.0f cycles/step, clock %.2f GHz\n", threads, ms * 1e6 / steps,
           (double)h[0] / steps, (double)h[0] / ((double)h[1] * 10.0));
  }
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(a)); hipLaunchKernelGGL(k_shfl, 1, 64, 0, 0, steps, out, clk, mode); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); }
    CK(hipEventElapsedTime(&ms, a, b)); CK(hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost));
    double o; CK(hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost));
    printf("wave f64 sum (%s): %.1f ns, %.0f cycles  (check %.6f)\n", mode ? "dpp" : "shfl_xor", ms * 1e6 / steps, (double)h[0] / steps, o);
  }
  const char* names[3] = {"sqrt f64", "div f64", "fma f64"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(a)); hipLaunchKernelGGL(k_math, 1, 64, 0, 0, steps, out, clk, mode); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); }
    CK(hipEventElapsedTime(&ms, a, b)); CK(hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost));
    printf("dependent %s: %.0f cycles\n", names[mode], (double)h[0] / steps);
  }
  // launch overhead: 100 empty kernels back to back
  for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(a)); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, 1, 64, 0, 0); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); }
  CK(hipEventElapsedTime(&ms, a, b));
  printf("empty kernel back-to-back: %.2f us each\n", ms * 1e3 / 100);
  return 0;
}
