// mfma_tile.hpp — one wave, one 16x16 fp64 tile:  acc(i,j) += sum_k a(i,k) * b(k,j)  on
// v_mfma_f64_16x16x4_f64.  Operand lane map: lane l feeds A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; result lane map: col = l&15, row = (l>>4) + 4*reg.
//
// Every kernel of the update path is latency-bound at the reference's sizes (k ~ 100), so what
// matters is how many operand loads are in flight, not MFMA issue rate.  This helper keeps two
// register buffers of CH k-steps each: the loads of chunk c+1 are issued before the MFMAs of chunk
// c.  Accessors are only called with 0 <= k < K (the index is clamped, the A operand is zeroed
// beyond K) and must be branch-free themselves (clamp, load, select) so that the compiler does not
// fence each load with its own s_waitcnt.  All 64 lanes must call this together.
#pragma once
#include <hip/hip_runtime.h>

namespace plv {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH, class FA, class FB>
__device__ __forceinline__ d4 mfma_tile_f64_pipe(FA a, FB b, int K, d4 acc) {
  if (K <= 0) return acc;
  const int lane = threadIdx.x & 63;
  const int ij = lane & 15, kq = lane >> 4;
  double a0[CH], b0[CH], a1[CH], b1[CH];
  const int step = 4 * CH;
#define PLV_TILE_LOAD(AV, BV, K0)                  \
  _Pragma("unroll") for (int u = 0; u < CH; ++u) { \
    const int kk = (K0) + 4 * u + kq;              \
    const int kc = min(kk, K - 1);                 \
    const double x = a(ij, kc);                    \
    AV[u] = kk < K ? x : 0.0;                      \
    BV[u] = b(kc, ij);                             \
  }
#define PLV_TILE_MMA(AV, BV) \
  _Pragma("unroll") for (int u = 0; u < CH; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(AV[u], BV[u], acc, 0, 0, 0);
  PLV_TILE_LOAD(a0, b0, 0)
  for (int k0 = 0; k0 < K; k0 += 2 * step) {
    const bool more1 = k0 + step < K;  // uniform
    if (more1) { PLV_TILE_LOAD(a1, b1, k0 + step) }
    PLV_TILE_MMA(a0, b0)
    if (more1) {
      if (k0 + 2 * step < K) { PLV_TILE_LOAD(a0, b0, k0 + 2 * step) }
      PLV_TILE_MMA(a1, b1)
    }
  }
#undef PLV_TILE_LOAD
#undef PLV_TILE_MMA
  return acc;
}

}  // namespace plv
