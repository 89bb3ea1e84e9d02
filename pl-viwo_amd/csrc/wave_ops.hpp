// wave_ops.hpp — wave64 cross-lane reductions on DPP (no LDS traffic).  Measured on MI355X
// (tools/ubench/lat.hip): an fp64 wave sum costs ~220 cycles with DPP against ~600 with
// __shfl_xor (ds_bpermute).
#pragma once
#include <hip/hip_runtime.h>

namespace plv {

// v + (v moved by DPP control CTRL), invalid source lanes contribute 0
template <int CTRL> __device__ __forceinline__ int dpp_mov_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(dpp_mov_i32<CTRL>(hi), dpp_mov_i32<CTRL>(lo));
}
template <int CTRL> __device__ __forceinline__ long long dpp_mov_i64(long long v) {
  int lo = (int)(v & 0xffffffffLL), hi = (int)(v >> 32);
  return ((long long)dpp_mov_i32<CTRL>(hi) << 32) | (unsigned)dpp_mov_i32<CTRL>(lo);
}

// DPP controls: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143,
// quad_perm[1,0,3,2] = 0xB1, quad_perm[2,3,0,1] = 0x4E
#define PLV_WAVE_REDUCE_BODY(MOV)                      \
  v += MOV<0x111>(v);                                  \
  v += MOV<0x112>(v);                                  \
  v += MOV<0x114>(v);                                  \
  v += MOV<0x118>(v);                                  \
  v += MOV<0x142>(v);                                  \
  v += MOV<0x143>(v);

// Sum over the 64 lanes, result broadcast to every lane (all lanes must be active).
__device__ __forceinline__ double wave_sum_f64(double v) {
  PLV_WAVE_REDUCE_BODY(dpp_mov_f64)
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
  PLV_WAVE_REDUCE_BODY(dpp_mov_i64)
  int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
  return ((long long)hi << 32) | (unsigned)lo;
}
// Exact 64-bit sum of 32-bit lane values whose sums over any 16-lane row fit in int32: the four in-row steps run on 32-bit
// registers (one DPP move + one add each), only the two cross-row steps carry 64 bits.
__device__ __forceinline__ long long wave_sum_i32_rows(int v) {
  v += dpp_mov_i32<0x111>(v);
  v += dpp_mov_i32<0x112>(v);
  v += dpp_mov_i32<0x114>(v);
  v += dpp_mov_i32<0x118>(v);
  long long w = v;  // (lane 15 of every row holds the row's sum)
  w += dpp_mov_i64<0x142>(w);
  w += dpp_mov_i64<0x143>(w);
  int lo = __builtin_amdgcn_readlane((int)(w & 0xffffffffLL), 63), hi = __builtin_amdgcn_readlane((int)(w >> 32), 63);
  return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
  PLV_WAVE_REDUCE_BODY(dpp_mov_i32)
  return __builtin_amdgcn_readlane(v, 63);
}
// the same sum, left where the reduction ends: valid in lane 63 only (no broadcast: for a caller whose lane 63 stores it)
__device__ __forceinline__ int wave_sum_i32_lane63(int v) {
  PLV_WAVE_REDUCE_BODY(dpp_mov_i32)
  return v;
}
// Sum over each aligned group of 4 lanes, result in all 4 lanes.
__device__ __forceinline__ double quad_sum_f64(double v) {
  v += dpp_mov_f64<0xB1>(v);
  v += dpp_mov_f64<0x4E>(v);
  return v;
}

}  // namespace plv
