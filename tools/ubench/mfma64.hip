// fp64 MFMA issue / latency on gfx950: dependent chain vs independent accumulators, 1 wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
template <int NACC>
__global__ void k(int steps, double* out, long long* clk) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double r = 0;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x + blockIdx.x * blockDim.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
// dependent chain through a VALU op on the result (like the elimination step)
__global__ void kdep(int steps, double* out, long long* clk) {
  d4 acc = {1, 1, 1, 1};
  double a = threadIdx.x * 1e-3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    double x = acc[0] * 1e-3;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a, acc, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void kfma(int steps, double* out, long long* clk) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  double a = 1.0000001;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fma(v[i], a, 1e-9);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double r = 0;
  for (int i = 0; i < 8; ++i) r += v[i];
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void kfma_dep(int steps, double* out, long long* clk) {
  double v = threadIdx.x;
  double a = 1.0000001;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v = fma(v, a, 1e-9);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = v;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
  double* out; long long* clk; long long h;
  CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&clk, 64));
  const int steps = 2000;
#define RUN(NAME, KERN, THREADS, PER)                                                   \
  for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(KERN, dim3(1), dim3(THREADS), 0, 0, steps, out, clk); CK(hipDeviceSynchronize()); } \
  CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost)); printf("%-44s %8.1f ticks per op\n", NAME, (double)h / steps / (PER));
  RUN("mfma f64 16x16x4 dependent, 1 wave", k<1>, 64, 1)
  RUN("mfma f64 16x16x4 2 independent acc, 1 wave", k<2>, 64, 2)
  RUN("mfma f64 16x16x4 4 independent acc, 1 wave", k<4>, 64, 4)
  RUN("mfma f64 4 indep acc, 4 waves (1/SIMD)", k<4>, 256, 4)
  RUN("mfma f64 4 indep acc, 8 waves (2/SIMD)", k<4>, 512, 4)
  RUN("mfma f64 dep through VALU mul", kdep, 64, 1)
  RUN("v_fma_f64 8 independent, 1 wave", kfma, 64, 8)
  RUN("v_fma_f64 dependent, 1 wave", kfma_dep, 64, 8)
  RUN("v_fma_f64 8 independent, 8 waves", kfma, 512, 8)
  return 0;
}
