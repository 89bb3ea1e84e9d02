// What does one step of a FOLLOWING strip cost when the step's data is already published (blocked_chol.hpp: strip_chain)?
// One wave, 16 steps per repetition, everything in LDS before the clock starts.  Variants (bits):
//   1  the step's {row, multiplier} pair is read from LDS (else it sits in registers)
//   2  the flag word is read and tested as well (the loop of strip_chain, never taken here)
//   4  the captured row (cap) is kept as strip_chain keeps it
//   8  the row operand comes from the PUBLISHED data of the previous step instead of from the accumulator (no read-back of the MFMA
//      result on the chain: the product the MFMA subtracts is formed from data that was there before it was issued)
//  16  the next step's pair is requested before this step's MFMA is issued (software pipelining, no flag)
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench/strip_step.hip -o tools/ubench/strip_step.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define PLV_LDS __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ T lds_vload(const T *p) { return *(const volatile PLV_LDS T *)(p); }
struct Lds {
  double Ts[16][64][2];
  int flag;
};
template <int V>
__global__ void k(int reps, const double *in, double *out, long long *clk) {
  __shared__ Lds lds;
  const int lane = threadIdx.x & 63, lq = lane >> 4;
  d4 W0;
  for (int q = 0; q < 4; ++q) W0[q] = in[lane * 4 + q];
  for (int j = 0; j < 16; ++j) {
    lds.Ts[j][lane][0] = 1e-3 * in[(lane * 4 + j) & 255];
    lds.Ts[j][lane][1] = ((lane >> 4) == (j & 3)) ? -0.01 : 0.0;
  }
  if (lane == 0) lds.flag = 1 << 20;
  __syncthreads();
  double sum = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    d4 W = W0, cap = {0, 0, 0, 0};
    d2 dn = {0, 0};
    if (V & 16) dn = lds_vload(reinterpret_cast<const d2 *>(&lds.Ts[0][lane][0]));
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int kk = jj & 3, rq = jj >> 2;
      d2 d;
      if (V & 16) {
        d = dn;
        if (jj < 15) dn = lds_vload(reinterpret_cast<const d2 *>(&lds.Ts[jj + 1][lane][0]));
      } else if (V & 1) {
        if (V & 2) {
          for (;;) {
            const int f = lds_vload(&lds.flag);
            d = lds_vload(reinterpret_cast<const d2 *>(&lds.Ts[jj][lane][0]));
            if (__builtin_amdgcn_readfirstlane(f) >= 16 * r + jj + 1 - (1 << 19)) break;
            __builtin_amdgcn_s_sleep(1);
          }
        } else {
          d = lds_vload(reinterpret_cast<const d2 *>(&lds.Ts[jj][lane][0]));
        }
      } else {
        d = d2{1e-3 * W0[kk], (lq == kk) ? -0.01 : 0.0};
      }
      const double brow = (V & 8) ? d[0] * 1.5 : W[rq];
      if (V & 4) cap[rq] = (lq == kk) ? brow : cap[rq];
      W = __builtin_amdgcn_mfma_f64_16x16x4f64(d[0], brow * d[1], W, 0, 0, 0);
    }
    sum += W[0] + W[1] + W[2] + W[3] + cap[0] + cap[3];
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = sum;
  if (lane == 0) clk[0] = t1 - t0;
}
int main() {
  double *in, *out;
  long long *clk, h;
  CK(hipMalloc(&in, 256 * 8));
  CK(hipMalloc(&out, 64 * 8));
  CK(hipMalloc(&clk, 64));
  double hin[256];
  for (int i = 0; i < 256; ++i) hin[i] = 1.0 / (1 + (i % 17)) + 0.01 * (i % 5);
  CK(hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice));
  const int reps = 200;
#define RUN(V, what)                                                                                 \
  for (int it = 0; it < 2; ++it) {                                                                   \
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, reps, in, out, clk);                           \
    CK(hipDeviceSynchronize());                                                                      \
  }                                                                                                  \
  CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));                                                  \
  printf("variant %2d  %-78s %7.1f ticks per step\n", V, what, (double)h / reps / 16);
  RUN(0, "registers only: MFMA -> accumulator read-back -> multiply -> MFMA")
  RUN(4, "+ captured row")
  RUN(1, "+ the pair from LDS")
  RUN(5, "+ the pair from LDS + captured row")
  RUN(7, "strip_chain's step (flag + pair from LDS, captured row)")
  RUN(20, "pair requested one step ahead + captured row")
  RUN(8, "no read-back of the accumulator (row operand from published data), registers only")
  RUN(9, "no read-back, pair from LDS")
  RUN(24, "no read-back, pair requested one step ahead")
  return 0;
}
