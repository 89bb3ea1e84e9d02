// Where do the ~400 ticks per elimination step of diag_factor go?  One wave, 16 steps, variants.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__device__ __forceinline__ double readlane_f64(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}
struct Lds { double Ts[16][64]; double Bs[16][64]; double piv[16]; };
// V bit0: B mfma, bit1: LDS dumps, bit2: lookahead pivot (else read after the MFMA), bit3: full-precision rcp (else raw v_rcp)
template <int V>
__global__ void k(int reps, const double* in, double* out, long long* clk) {
  __shared__ Lds lds;
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  d4 T0;
  for (int q = 0; q < 4; ++q) T0[q] = in[lane * 4 + q];
  double mask01[4];
  for (int q = 0; q < 4; ++q) mask01[q] = (lq == q) ? 1.0 : 0.0;
  double sum = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    d4 T = T0, B;
    for (int q = 0; q < 4; ++q) B[q] = (lq + 4 * q == li) ? 1.0 : 0.0;
    double pv = readlane_f64(T[0], 0);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int kk = jj & 3, rq = jj >> 2;
      const bool live = pv > 0.0;
      const double rc = (V & 8) ? rcp_nr(pv) : __builtin_amdgcn_rcp(pv);
      const double ninv = live ? -rc : 0.0;
      const double nm = ninv * mask01[kk];
      const double trow = T[rq], brow = B[rq];
      if (V & 2) { lds.Ts[jj][lane] = trow; lds.Bs[jj][lane] = brow; lds.piv[jj] = pv; }
      if ((V & 4) && jj < 15) {
        const double an = readlane_f64(T[rq], 16 * kk + jj + 1);
        const double tn = readlane_f64(T[(jj + 1) >> 2], 16 * ((jj + 1) & 3) + jj + 1);
        pv = fma(an * ninv, an, tn);
      }
      T = __builtin_amdgcn_mfma_f64_16x16x4f64(trow * nm, trow, T, 0, 0, 0);
      if (V & 1) B = __builtin_amdgcn_mfma_f64_16x16x4f64(trow, brow * nm, B, 0, 0, 0);
      if (!(V & 4) && jj < 15) pv = readlane_f64(T[(jj + 1) >> 2], 16 * ((jj + 1) & 3) + jj + 1);
    }
    sum += T[0] + T[3] + B[1] + B[2];
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = sum + lds.Ts[3][lane] + lds.piv[2];
  if (lane == 0) clk[0] = t1 - t0;
}
int main() {
  double *in, *out; long long* clk; long long h;
  CK(hipMalloc(&in, 256 * 8)); CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&clk, 64));
  double hin[256];
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) { int row = (l >> 4) + 4 * q, col = l & 15; hin[l * 4 + q] = (row == col ? 20.0 : 0.0) + 1.0 / (1 + row + col); }
  CK(hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice));
  const int reps = 200;
#define RUN(V) for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, reps, in, out, clk); CK(hipDeviceSynchronize()); } \
  CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost)); printf("variant %2d (B-mfma %d, lds %d, lookahead %d, newton %d): %7.1f ticks per step\n", V, V & 1, (V >> 1) & 1, (V >> 2) & 1, (V >> 3) & 1, (double)h / reps / 16);
  RUN(0) RUN(1) RUN(2) RUN(4) RUN(8) RUN(12) RUN(13) RUN(15) RUN(11) RUN(9)
  return 0;
}
