// jacobian_kernels.hip — K10: per-observation linearisation of the MSCKF point measurement (fp64).
//
//   jacobian_kernel   CamHelper::get_feature_jacobian_full            REF: PL/update/cam/CamHelper.cpp:58-267
//                     State::get_interpolated_jacobian (FEJ polynomial) REF: PL/state/State.cpp:833-973
//                     State::get_interpolated_pose_poly (residual pose) REF: PL/state/State.cpp:979-1021
//                     CamRadtan::distort_f / compute_distort_jacobian  REF: OV/cam/CamRadtan.h:127-198
//
// One THREAD per observation: the work per observation is a few thousand scalar fp64 operations on
// 3x3 blocks (two polynomial interpolations on so(3) x R^3, projection, distortion Jacobians, 2x2
// whitening) and nothing is shared between observations except read-only clone poses, so the
// natural GPU shape is "one lane = one (feature, observation)", ~1000 lanes per frame.  Rows are
// written straight into the [Hf | Hx | res] batch layout that nullspace_kernel consumes; the batch
// is zero-filled by a memset node in front of the launch.
#include <hip/hip_runtime.h>
// Phase stamps of the projected Jacobian launches (measurement aid, PLV_KNOB_KERNEL_STAMPS: the launcher hangs a buffer on the pointer
// and prints mean / max cycles per phase when the library unloads).  Off: one scalar load and a branch per stamp.
namespace plv {
#define JAC_NSTAMP 32
__device__ long long *g_jac_stamps = nullptr;  // [workgroups][JAC_NSTAMP]
__device__ __forceinline__ void jac_stamp(int id) {
  long long *s = g_jac_stamps;
  if (s && threadIdx.x == 0) s[blockIdx.x * JAC_NSTAMP + id] = (long long)__builtin_amdgcn_s_memtime();
}
}  // namespace plv
#define GATE_STAMP(id) jac_stamp(id)
#include "gate_core.hpp"
#include "jacobian_kernels.hpp"
#include "nullspace_core.hpp"

namespace plv {

struct V3 {
  double v[3];
  __device__ double &operator[](int i) { return v[i]; }
  __device__ double operator[](int i) const { return v[i]; }
};
struct M3 {
  double m[9];
  __device__ double &operator()(int r, int c) { return m[3 * r + c]; }
  __device__ double operator()(int r, int c) const { return m[3 * r + c]; }
};
__device__ __forceinline__ M3 eye3() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
__device__ __forceinline__ M3 mm(const M3 &a, const M3 &b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
  return c;
}
__device__ __forceinline__ M3 tp(const M3 &a) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c(i, j) = a(j, i);
  return c;
}
__device__ __forceinline__ V3 mv(const M3 &a, const V3 &x) {
  return V3{{a(0, 0) * x[0] + a(0, 1) * x[1] + a(0, 2) * x[2], a(1, 0) * x[0] + a(1, 1) * x[1] + a(1, 2) * x[2],
             a(2, 0) * x[0] + a(2, 1) * x[1] + a(2, 2) * x[2]}};
}
__device__ __forceinline__ M3 ms(const M3 &a, double s) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] * s;
  return c;
}
__device__ __forceinline__ M3 ma(const M3 &a, const M3 &b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] + b.m[i];
  return c;
}
__device__ __forceinline__ V3 vsub(const V3 &a, const V3 &b) { return V3{{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
__device__ __forceinline__ V3 vadd(const V3 &a, const V3 &b) { return V3{{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
__device__ __forceinline__ V3 vsc(const V3 &a, double s) { return V3{{a[0] * s, a[1] * s, a[2] * s}}; }
__device__ __forceinline__ double vnorm(const V3 &a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
__device__ __forceinline__ M3 skew3(const V3 &w) { return M3{{0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}}; }
__device__ __forceinline__ M3 inv3(const M3 &a) {
  double c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1), c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2),
         c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  double det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02;
  double id = 1.0 / det;
  M3 r;
  r(0, 0) = c00 * id;
  r(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id;
  r(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
  r(1, 0) = c01 * id;
  r(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id;
  r(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
  r(2, 0) = c02 * id;
  r(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id;
  r(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
  return r;
}
__device__ M3 exp_so3(const V3 &w) {  // REF: quat_ops.h:231-251
  const M3 wx = skew3(w);
  const double theta = vnorm(w);
  double A, B;
  if (theta < 1e-7) {
    A = 1;
    B = 0.5;
  } else {
    A = sin(theta) / theta;
    B = (1 - cos(theta)) / (theta * theta);
  }
  if (theta == 0) return eye3();
  return ma(ma(eye3(), ms(wx, A)), ms(mm(wx, wx), B));
}
__device__ V3 log_so3(const M3 &R) {  // REF: quat_ops.h:273-313
  const double R11 = R(0, 0), R12 = R(0, 1), R13 = R(0, 2), R21 = R(1, 0), R22 = R(1, 1), R23 = R(1, 2), R31 = R(2, 0),
               R32 = R(2, 1), R33 = R(2, 2);
  const double trc = R11 + R22 + R33;
  const double PI = 3.14159265358979323846;
  if (trc + 1.0 < 1e-10) {
    if (fabs(R33 + 1.0) > 1e-5) return vsc(V3{{R13, R23, 1.0 + R33}}, PI / sqrt(2.0 + 2.0 * R33));
    if (fabs(R22 + 1.0) > 1e-5) return vsc(V3{{R12, 1.0 + R22, R32}}, PI / sqrt(2.0 + 2.0 * R22));
    return vsc(V3{{1.0 + R11, R21, R31}}, PI / sqrt(2.0 + 2.0 * R11));
  }
  double magnitude;
  const double tr_3 = trc - 3.0;
  if (tr_3 < -1e-7) {
    const double theta = acos((trc - 1.0) / 2.0);
    magnitude = theta / (2.0 * sin(theta));
  } else {
    magnitude = 0.5 - tr_3 / 12.0;
  }
  return vsc(V3{{R32 - R23, R13 - R31, R21 - R12}}, magnitude);
}
__device__ M3 Jl_so3(const V3 &w) {  // REF: quat_ops.h:515-526
  const double theta = vnorm(w);
  if (theta < 1e-6) return eye3();
  const V3 a = vsc(w, 1.0 / theta);
  M3 aat;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) aat(i, j) = a[i] * a[j];
  return ma(ma(ms(eye3(), sin(theta) / theta), ms(aat, 1 - sin(theta) / theta)), ms(skew3(a), (1 - cos(theta)) / theta));
}
// exp_so3(w) and Jl_so3(w) for the same w with ONE sin / cos evaluation (the transcendental calls are the long poles of the
// per-observation chain); every other operation is the one of the two functions above, so the results are bit-identical.
__device__ void exp_and_Jl(const V3 &w, M3 &Rexp, M3 &J) {
  const M3 wx = skew3(w);
  const double theta = vnorm(w);
  const double st = sin(theta), ct = cos(theta);
  double A, B;
  if (theta < 1e-7) {
    A = 1;
    B = 0.5;
  } else {
    A = st / theta;
    B = (1 - ct) / (theta * theta);
  }
  Rexp = theta == 0 ? eye3() : ma(ma(eye3(), ms(wx, A)), ms(mm(wx, wx), B));
  if (theta < 1e-6) {
    J = eye3();
    return;
  }
  const V3 a = vsc(w, 1.0 / theta);
  M3 aat;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) aat(i, j) = a[i] * a[j];
  J = ma(ma(ms(eye3(), st / theta), ms(aat, 1 - st / theta)), ms(skew3(a), (1 - ct) / theta));
}
__device__ __forceinline__ M3 ldM(const double *p) {
  M3 m;
#pragma unroll
  for (int i = 0; i < 9; ++i) m.m[i] = p[i];
  return m;
}
__device__ __forceinline__ V3 ldV(const double *p) { return V3{{p[0], p[1], p[2]}}; }

// State::bounding_times + bounding_poses_n (order 3)   REF: State.cpp:1023-1136
__device__ int bounding_start(const JacParams &P, double t) {
  const int N = P.n_clones;
  if (N < 4) return -1;
  const double *ct = P.clone_time;
  if (t < ct[0] - P.dt_exp || t > ct[N - 1] + P.dt_exp) return -1;
  if (t > ct[N - 1]) return -1;  // State.cpp:852-855
  int n_b = -1;
  for (int i = 0; i < N - 1; ++i)
    if (ct[i] - P.dt_exp <= t && t <= ct[i + 1] + P.dt_exp) {
      n_b = i;
      break;
    }
  if (n_b < 0) return -1;
  int start = n_b - 1;
  if (n_b - 1 < 0)
    start = 0;
  else if (n_b + 2 >= N)
    start = N - 4;
  if (start < 0 || start + 4 > N) return -1;
  return start;
}

struct Interp {
  M3 R;
  V3 p;
  M3 Ho[4];       // orientation blocks of dT/dx for the 4 poses
  double lam[4];  // position blocks are lam I  (lam[0] = 1 - sum)
  double dtj[6];
};

// polynomial through clones s0..s0+3 at time t.  REF: State.cpp:631-723, 881-958
__device__ void interpolate(const JacParams &P, int s0, double t, bool fej, bool want_jac, Interp &o) {
  const double *Rs = fej ? P.clone_R_fej : P.clone_R, *ps = fej ? P.clone_p_fej : P.clone_p;
  const M3 R0 = ldM(Rs + 9 * s0);
  const V3 p0 = ldV(ps + 3 * s0);
  V3 th[3], dp[3];
  M3 Rw[3], V;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    Rw[w] = mm(ldM(Rs + 9 * (s0 + 1 + w)), tp(R0));
    th[w] = log_so3(Rw[w]);
    dp[w] = vsub(ldV(ps + 3 * (s0 + 1 + w)), p0);
    const double d = P.clone_time[s0 + 1 + w] - P.clone_time[s0];
    V(w, 0) = d;  // REF uses std::pow(d, i); products differ from it by at most one rounding
    V(w, 1) = d * d;
    V(w, 2) = d * d * d;
  }
  const M3 Vi = inv3(V);
  const double dtm = t - P.clone_time[s0];
  const double pw[4] = {1.0, dtm, dtm * dtm, dtm * dtm * dtm};
  double lam[3], lamd[3];
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    lam[w] = lamd[w] = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      lam[w] += pw[i + 1] * Vi(i, w);
      lamd[w] += (double)(i + 1) * pw[i] * Vi(i, w);
    }
  }
  V3 A_ori{{0, 0, 0}}, A_pos{{0, 0, 0}};
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    A_ori = vadd(A_ori, vsc(th[w], lam[w]));
    A_pos = vadd(A_pos, vsc(dp[w], lam[w]));
  }
  const M3 Rio = exp_so3(A_ori);
  o.R = mm(Rio, R0);
  o.p = vadd(p0, A_pos);
  if (!want_jac) return;
  const M3 Jl = Jl_so3(A_ori);
  M3 H0o = Rio;
  double lsum = 0;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const M3 JinvW = inv3(Jl_so3(th[w]));
    H0o = ma(H0o, ms(mm(Jl, mm(JinvW, Rw[w])), -lam[w]));
    o.Ho[w + 1] = ms(mm(Jl, JinvW), lam[w]);
    o.lam[w + 1] = lam[w];
    lsum += lam[w];
  }
  o.Ho[0] = H0o;
  o.lam[0] = 1.0 - lsum;
  V3 dori{{0, 0, 0}}, dpos{{0, 0, 0}};
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    dori = vadd(dori, vsc(th[w], lamd[w]));
    dpos = vadd(dpos, vsc(dp[w], lamd[w]));
  }
  const V3 top = vsc(mv(Jl, dori), -1.0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.dtj[i] = top[i];
    o.dtj[3 + i] = dpos[i];
  }
}


// ---- window tables: everything in `interpolate` that depends only on the interpolation window (the four
// clones s0..s0+3) and not on the observation time: R0, p0, the three relative rotations with their logs and
// inverse left Jacobians, the position differences, the inverse Vandermonde matrix.  A feature's 15
// observations share at most n_clones - 3 windows, and the transcendental-heavy part (3 log_so3, 3 Jl_so3, 4
// 3x3 inverses per call) is two thirds of the per-observation work: the table is built once per workgroup by
// otherwise idle lanes (three lanes per window and variant), the observation lanes then only evaluate the
// polynomial.  Same expressions in the same order as interpolate(): results are bit-identical.
struct WinTab {
  M3 R0, Rw[3], JinvW[3], Vi;
  V3 p0, th[3], dp[3];
  double vrow[3][3];  // rows of the Vandermonde matrix (scratch until Vi is formed)
};
#define JAC_MAX_WIN 40  // (n_clones - 3) * 2 variants must fit

// Rt != null: the estimate-variant tables of a second state of the same window (clone rotations Rt, positions pt; clone times
// shared) follow at tab[2 * nwin + s0] (the line update triangulates on the state before the point correction and linearises on
// the corrected one).  The second state comes in as two pointer VALUES: a pointer to its JacParams keeps that whole struct — and
// the caller's — in scratch memory (a select between members of two structs becomes a select of their addresses: 48 bytes of
// scratch stores per lane at the top of every workgroup, 1.1 MB per line launch in round 5's WRITE_SIZE).
__device__ void build_window_tables(const JacParams &P, WinTab *tab, const double *Rt = nullptr, const double *pt = nullptr) {
  const int nwin = max(P.n_clones - 3, 0);
  const int ntask = nwin * (Rt ? 3 : 2);
  const double *const R_est = P.clone_R, *const R_fej = P.clone_R_fej, *const p_est = P.clone_p, *const p_fej = P.clone_p_fej;
  for (int idx = threadIdx.x; idx < ntask * 3; idx += blockDim.x) {
    const int w = idx % 3, e = idx / 3;
    const bool second = e >= 2 * nwin;
    const int s0 = second ? e - 2 * nwin : e >> 1, fej = second ? 0 : e & 1;
    const double *Rs = second ? Rt : (fej ? R_fej : R_est), *ps = second ? pt : (fej ? p_fej : p_est);
    WinTab &T = tab[e];
    const M3 R0 = ldM(Rs + 9 * s0);
    const V3 p0 = ldV(ps + 3 * s0);
    const M3 Rw = mm(ldM(Rs + 9 * (s0 + 1 + w)), tp(R0));
    const V3 th = log_so3(Rw);
    T.Rw[w] = Rw;
    T.th[w] = th;
    T.dp[w] = vsub(ldV(ps + 3 * (s0 + 1 + w)), p0);
    const double d = P.clone_time[s0 + 1 + w] - P.clone_time[s0];
    T.vrow[w][0] = d;
    T.vrow[w][1] = d * d;
    T.vrow[w][2] = d * d * d;
    if (fej) T.JinvW[w] = inv3(Jl_so3(th));
    if (w == 0) {
      T.R0 = R0;
      T.p0 = p0;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ntask; e += blockDim.x) {
    M3 V;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int c = 0; c < 3; ++c) V(w, c) = tab[e].vrow[w][c];
    tab[e].Vi = inv3(V);
  }
  __syncthreads();
}

__device__ void interpolate_tab(const JacParams &P, const WinTab &T, int s0, double t, bool want_jac, Interp &o) {
  const double dtm = t - P.clone_time[s0];
  const double pw[4] = {1.0, dtm, dtm * dtm, dtm * dtm * dtm};
  double lam[3], lamd[3];
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    lam[w] = lamd[w] = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      lam[w] += pw[i + 1] * T.Vi(i, w);
      lamd[w] += (double)(i + 1) * pw[i] * T.Vi(i, w);
    }
  }
  V3 A_ori{{0, 0, 0}}, A_pos{{0, 0, 0}};
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    A_ori = vadd(A_ori, vsc(T.th[w], lam[w]));
    A_pos = vadd(A_pos, vsc(T.dp[w], lam[w]));
  }
  M3 Rio, Jl;
  if (want_jac)
    exp_and_Jl(A_ori, Rio, Jl);
  else
    Rio = exp_so3(A_ori);
  o.R = mm(Rio, T.R0);
  o.p = vadd(T.p0, A_pos);
  if (!want_jac) return;
  M3 H0o = Rio;
  double lsum = 0;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const M3 JinvW = T.JinvW[w];
    H0o = ma(H0o, ms(mm(Jl, mm(JinvW, T.Rw[w])), -lam[w]));
    o.Ho[w + 1] = ms(mm(Jl, JinvW), lam[w]);
    o.lam[w + 1] = lam[w];
    lsum += lam[w];
  }
  o.Ho[0] = H0o;
  o.lam[0] = 1.0 - lsum;
  V3 dori{{0, 0, 0}}, dpos{{0, 0, 0}};
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    dori = vadd(dori, vsc(T.th[w], lamd[w]));
    dpos = vadd(dpos, vsc(T.dp[w], lamd[w]));
  }
  const V3 top = vsc(mv(Jl, dori), -1.0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.dtj[i] = top[i];
    o.dtj[3 + i] = dpos[i];
  }
}

__device__ void jacobian_rows(const JacParams &P, const WinTab *tab, int f, int o, int s0, double tm, int c, double *hf, double *hx,
                              double *rs, int cstr, int rstr);
__device__ void jacobian_rows_core(const JacParams &P, int f, int o, int s0, double tm, int c, const Interp &jac, M3 R_GtoI, V3 p_IinG, double *hf,
                                   double *hx, double *rs, int cstr, int rstr);

// One workgroup (one wave) per feature, one lane per observation.  The feature's slice of the
// batch [Hf | Hx | res] is zero-filled here (no separate memset of the 1.8 MB batch), the row slot
// of an observation is the number of valid observations in front of it (ballot prefix).
__global__ void __launch_bounds__(128) jacobian_kernel(JacParams P) {
  __shared__ WinTab tab[JAC_MAX_WIN];
  const int f = blockIdx.x;
  const int ld = P.ld, k = P.k;
  double *hf = P.Hf + (size_t)f * 3 * ld, *hx = P.Hx + (size_t)f * k * ld, *rs = P.res + (size_t)f * ld;
  for (int i = threadIdx.x; i < 3 * ld; i += blockDim.x) hf[i] = 0.0;
  for (int i = threadIdx.x; i < k * ld; i += blockDim.x) hx[i] = 0.0;
  for (int i = threadIdx.x; i < ld; i += blockDim.x) rs[i] = 0.0;
  if (f == 0 && P.cols_out)
    for (int i = threadIdx.x; i < k; i += blockDim.x) P.cols_out[i] = P.cols_in[i];
  build_window_tables(P, tab);  // (ends with a barrier: also orders the zero fill before the row writes)
  if (threadIdx.x >= 64) return;  // observation lanes = wave 0
  const int o0 = P.obs_ptr[f], o1 = P.obs_ptr[f + 1];
  int base = 0;
  for (int ob = o0; ob < o1; ob += 64) {  // (more than 64 observations of one feature: next chunk)
    const int o = ob + threadIdx.x;
    const bool have = o < o1;
    const double tm = (have ? P.obs_time[o] : 0.0) + P.cam_dt;
    const int s0 = have ? bounding_start(P, tm) : -1;
    const unsigned long long vmask = __ballot(s0 >= 0);
    const int c = base + __popcll(vmask & ((1ull << threadIdx.x) - 1ull));
    base += __popcll(vmask);
    if (s0 >= 0 && 2 * c + 2 <= ld) jacobian_rows(P, tab, f, o, s0, tm, c, hf, hx, rs, ld, 1);
  }
  if (threadIdx.x == 0) P.rows[f] = 2 * base;
}

// The reference's selection loop on the device: candidate f is taken when it passes its own tests and fewer than max_sel
// candidates before it did.  Called by every thread of the workgroup.
__device__ bool candidate_selected(const JacParams &P, int f) {
  auto base = [&](int g) { return P.sel_flags[g] && P.tri_ok[g] && (!P.tri_err || P.tri_err[g] < 3.0); };
  int before = 0;
  for (int g0 = 0; g0 < f; g0 += blockDim.x) {
    const int g = g0 + threadIdx.x;
    before += __syncthreads_count(g < f && base(g));
  }
  return base(f) && before < P.max_sel;
}

// jacobian_kernel + nullspace_kernel in one launch for the resident update path: the feature's [Hf | Hx | res] block is built
// row-major in LDS, projected there (nullspace_core.hpp) and only the projected block goes to global memory — one launch, one
// 1.7 MB write and one 1.7 MB read less on the update chain.  The covariance gathers of the update ride on it as extra workgroups
// (they read the column map from the packed input block: the resident copy is being written by workgroup 0).
__device__ void triangulate_feature(const JacParams &P, int f, int o0, int o1, double *poses, unsigned char *valid, const float *uvn, const float *uv,
                                    const plv_tri_options &opt, double *__restrict__ p_out, unsigned char *__restrict__ ok_out,
                                    double *__restrict__ err_out, int max_obs, double *tri_smem, double *tot, bool poses_ready = false,
                                    double *res_l = nullptr);
// tri.on: the workgroup's first wave triangulates the feature before the Jacobians are built (what triangulate_kernel did in a launch
// of its own) — while the selection loop has no cap to enforce (n_feat <= max_sel) a candidate is taken on its own verdict.
#define TRI_TERMS 10
__host__ __device__ inline int tri_smem_doubles(int max_obs) {  // TriObs [max_obs] | terms [max_obs][TRI_TERMS] | list [max_obs] (ints)
  return max_obs * (15 + TRI_TERMS) + (max_obs + 1) / 2 + 2;
}
struct PointTriStage {
  int on, max_obs;
  double *poses;          // [n_obs][12] scratch
  unsigned char *valid;   // [n_obs]
  const float *uvn;       // [n_obs][2]
  plv_tri_options opt;
  double *p_out;          // [F][3] == P.p_FinG of the Jacobian stage
  unsigned char *ok_out;  // [F]
  double *err_out;        // [F]
};
// REF: CamHelper.cpp:217-224 (and LineHelper's twin): R += H_ Q H_^T * mlt with H_ = HI * blockdiag(I, R_clone_fej^T), HI the 2 x 6
// Jacobian of the measurement in the interpolated pose, Q the CPI covariance of that pose.
__device__ __forceinline__ void add_imu_cov(const JacParams &P, int o, const double *HI, double *Rn) {
  const double *Q = P.res_Q + 36 * (size_t)o;
  const double *Rc = P.clone_R_fej + 9 * (size_t)P.res_clone[o];  // Rot_fej of the clone, row-major; H_cpi(3:6,3:6) = its transpose
  double Hc[12];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Hc[6 * i + j] = HI[6 * i + j];
      Hc[6 * i + 3 + j] = HI[6 * i + 3] * Rc[3 * j] + HI[6 * i + 4] * Rc[3 * j + 1] + HI[6 * i + 5] * Rc[3 * j + 2];
    }
  double HQ[12];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) s += Hc[6 * i + q] * Q[6 * q + j];
      HQ[6 * i + j] = s;
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) s += HQ[6 * i + q] * Hc[6 * j + q];
      Rn[2 * i + j] += s * P.intr_err_mlt;
    }
}

// Element (row, col) of a block goes to base[col * cstr + row * rstr]: (ld, 1) for the batch in global memory (column-major per
// feature), (1, ncol) for the row-major LDS image the fused kernel projects in place.
__device__ void jacobian_rows(const JacParams &P, const WinTab *tab, int f, int o, int s0, double tm, int c, double *hf, double *hx,
                              double *rs, int cstr, int rstr) {
  Interp jac;
  interpolate_tab(P, tab[2 * s0 + 1], s0, tm, true, jac);
  M3 R_GtoI;
  V3 p_IinG;
  if (P.res_R) {
    R_GtoI = ldM(P.res_R + 9 * o);
    p_IinG = ldV(P.res_p + 3 * o);
  } else {
    Interp est;
    interpolate_tab(P, tab[2 * s0], s0, tm, false, est);
    R_GtoI = est.R;
    p_IinG = est.p;
  }
  jacobian_rows_core(P, f, o, s0, tm, c, jac, R_GtoI, p_IinG, hf, hx, rs, cstr, rstr);
}
// ---- the rows of one observation behind the two interpolations, in pieces (jacobian_rows_core runs them one after the other on one
// lane; the fused launch spreads them over the four waves of the workgroup, jacobian_rows_split).  Every value is formed by the same
// expression wherever its piece runs.
// (1) residual at the estimate pose (CamBase::distort_d rounds through float both ways) + distortion Jacobians at the estimate's
//     normalised coordinates
__device__ __forceinline__ void rows_est_part(const JacParams &P, const V3 &pf, const float *uv /* the observation's image point */, const M3 &R_GtoI,
                                              const V3 &p_IinG, double *r2, double *dzn, double *dzeta) {
  const M3 R_ItoC = ldM(P.R_ItoC);
  const V3 p_IinC = ldV(P.p_IinC);
  const double *K = P.K;
  const V3 p_FinI = mv(R_GtoI, vsub(pf, p_IinG));
  const V3 p_FinC = vadd(mv(R_ItoC, p_FinI), p_IinC);
  const double un = p_FinC[0] / p_FinC[2], vn = p_FinC[1] / p_FinC[2];
  {
    const double x = (double)(float)un, y = (double)(float)vn;
    const double r = sqrt(x * x + y * y), r_2 = r * r, r_4 = r_2 * r_2;
    const double x1 = x * (1 + K[4] * r_2 + K[5] * r_4) + 2 * K[6] * x * y + K[7] * (r_2 + 2 * x * x);
    const double y1 = y * (1 + K[4] * r_2 + K[5] * r_4) + K[6] * (r_2 + 2 * y * y) + 2 * K[7] * x * y;
    r2[0] = (double)uv[0] - (double)(float)(K[0] * x1 + K[2]);
    r2[1] = (double)uv[1] - (double)(float)(K[1] * y1 + K[3]);
  }
  {
    const double x = un, y = vn;
    const double r = sqrt(x * x + y * y), r_2 = r * r, r_4 = r_2 * r_2;
    const double x_2 = x * x, y_2 = y * y, x_y = x * y;
    dzn[0] = K[0] * ((1 + K[4] * r_2 + K[5] * r_4) + (2 * K[4] * x_2 + 4 * K[5] * x_2 * r_2) + 2 * K[6] * y + (2 * K[7] * x + 4 * K[7] * x));
    dzn[1] = K[0] * (2 * K[4] * x_y + 4 * K[5] * x_y * r_2 + 2 * K[6] * x + 2 * K[7] * y);
    dzn[2] = K[1] * (2 * K[4] * x_y + 4 * K[5] * x_y * r_2 + 2 * K[6] * x + 2 * K[7] * y);
    dzn[3] = K[1] * ((1 + K[4] * r_2 + K[5] * r_4) + (2 * K[4] * y_2 + 4 * K[5] * y_2 * r_2) + 2 * K[7] * x + (2 * K[6] * y + 4 * K[6] * y));
    const double x1 = x * (1 + K[4] * r_2 + K[5] * r_4) + 2 * K[6] * x * y + K[7] * (r_2 + 2 * x * x);
    const double y1 = y * (1 + K[4] * r_2 + K[5] * r_4) + K[6] * (r_2 + 2 * y * y) + 2 * K[7] * x * y;
#pragma unroll
    for (int i = 0; i < 16; ++i) dzeta[i] = 0;
    dzeta[0] = x1;
    dzeta[2] = 1;
    dzeta[4] = K[0] * x * r_2;
    dzeta[5] = K[0] * x * r_4;
    dzeta[6] = 2 * K[0] * x * y;
    dzeta[7] = K[0] * (r_2 + 2 * x * x);
    dzeta[9] = y1;
    dzeta[11] = 1;
    dzeta[12] = K[1] * y * r_2;
    dzeta[13] = K[1] * y * r_4;
    dzeta[14] = K[1] * (r_2 + 2 * y * y);
    dzeta[15] = 2 * K[1] * x * y;
  }
}
// (2) projection chain at the first estimates: dznp = d(normalised)/d(p_FinC), dpC_dpG, dpC_dI = [R_ItoC skew(p_FinI) | -dpC_dpG],
//     lever = p_FinC - p_IinC (extrinsic block)
__device__ __forceinline__ void rows_fej_part(const JacParams &P, const V3 &pf_fej, const M3 &R_GtoI, const V3 &p_IinG, double *dznp, M3 &dpC_dpG,
                                              double *dpC_dI, V3 &lever) {
  const M3 R_ItoC = ldM(P.R_ItoC);
  const V3 p_IinC = ldV(P.p_IinC);
  const V3 p_FinI = mv(R_GtoI, vsub(pf_fej, p_IinG));
  const V3 p_FinC = vadd(mv(R_ItoC, p_FinI), p_IinC);
  const double iz = 1 / p_FinC[2];
  dznp[0] = iz, dznp[1] = 0, dznp[2] = -p_FinC[0] / (p_FinC[2] * p_FinC[2]);
  dznp[3] = 0, dznp[4] = iz, dznp[5] = -p_FinC[1] / (p_FinC[2] * p_FinC[2]);
  dpC_dpG = mm(R_ItoC, R_GtoI);
  const M3 left = mm(R_ItoC, skew3(p_FinI));
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dpC_dI[6 * i + j] = left(i, j);
      dpC_dI[6 * i + 3 + j] = -dpC_dpG(i, j);
    }
  lever = vsub(p_FinC, p_IinC);
}
__device__ __forceinline__ bool rows_at_clone(const JacParams &P, double tm) {
  bool at_clone = false;
  for (int i = 0; i < P.n_clones; ++i) at_clone = at_clone || P.clone_time[i] == tm;
  return at_clone;
}
// (3) measurement Jacobian in the interpolated pose, noise (REF :207-239 incl. the `R_llt.llt().solve(I)` form), whitening:
//     Wm (2 x 2), wz = Wm dz_dpC (2 x 3), WI = wz dpC_dI (2 x 6)
__device__ __forceinline__ void rows_whiten_part(const JacParams &P, int o, bool at_clone, const double *dzn, const double *dznp, const double *dpC_dI,
                                                 double *Wm, double *wz, double *WI) {
  double dz_dpC[6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) dz_dpC[3 * i + j] = dzn[2 * i] * dznp[j] + dzn[2 * i + 1] * dznp[3 + j];
  double HI[12];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) HI[6 * i + j] = dz_dpC[3 * i] * dpC_dI[j] + dz_dpC[3 * i + 1] * dpC_dI[6 + j] + dz_dpC[3 * i + 2] * dpC_dI[12 + j];
  double Rn[4] = {P.sigma_pix * P.sigma_pix, 0, 0, P.sigma_pix * P.sigma_pix};
  if (!at_clone && P.use_pol_cov) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        double s = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) s += HI[6 * i + q] * (q < 3 ? P.intr_ori_cov : P.intr_pos_cov) * HI[6 * j + q];
        Rn[2 * i + j] += s;
      }
  } else if (!at_clone && P.use_imu_cov) {
    add_imu_cov(P, o, HI, Rn);
  }
  const double l00 = sqrt(Rn[0]), l10 = Rn[2] / l00, l11 = sqrt(Rn[3] - l10 * l10);
  const double m00 = sqrt(l00), m10 = l10 / m00, m11 = sqrt(l11 - m10 * m10);
#pragma unroll
  for (int col = 0; col < 2; ++col) {
    const double b0 = col == 0 ? 1.0 : 0.0, b1 = col == 1 ? 1.0 : 0.0;
    const double y0 = b0 / m00, y1 = (b1 - m10 * y0) / m11;
    const double x1 = y1 / m11, x0 = (y0 - m10 * x1) / m00;
    Wm[col] = x0;
    Wm[2 + col] = x1;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    wz[j] = Wm[0] * dz_dpC[j] + Wm[1] * dz_dpC[3 + j];
    wz[3 + j] = Wm[2] * dz_dpC[j] + Wm[3] * dz_dpC[3 + j];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) WI[6 * i + j] = wz[3 * i] * dpC_dI[j] + wz[3 * i + 1] * dpC_dI[6 + j] + wz[3 * i + 2] * dpC_dI[12 + j];
}
// (4) the blocks.  Element (row, col) goes to base[col * cstr + row * rstr].  The slice was zero-filled by this workgroup and every
// (row, column) is written once (the clones of a window, the time offset, the extrinsics and the intrinsics are distinct state
// blocks), so these are plain stores: an accumulate would put a load in front of every one of them on the observation's chain.
__device__ __forceinline__ void rows_write_res(int c, const double *Wm, const double *r2, double *rs, int rstr) {
  rs[(2 * c) * rstr] = Wm[0] * r2[0] + Wm[1] * r2[1];
  rs[(2 * c + 1) * rstr] = Wm[2] * r2[0] + Wm[3] * r2[1];
}
__device__ __forceinline__ void rows_write_hf(const JacParams &P, const V3 &pf_fej, int c, const double *wz, const M3 &dpC_dpG, double *hf, int cstr,
                                              int rstr) {
  M3 G = dpC_dpG;
  if (P.feat_rep == PLV_FEAT_GLOBAL_FULL_INVERSE_DEPTH) {  // REF: CamHelper.cpp:29-51
    const double g_rho = 1 / vnorm(pf_fej);
    const double g_phi = acos(g_rho * pf_fej[2]);
    const double g_theta = atan2(pf_fej[1], pf_fej[0]);
    const double sin_th = sin(g_theta), cos_th = cos(g_theta), sin_phi = sin(g_phi), cos_phi = cos(g_phi), rho = g_rho;
    M3 H;
    H(0, 0) = -(1.0 / rho) * sin_th * sin_phi;
    H(0, 1) = (1.0 / rho) * cos_th * cos_phi;
    H(0, 2) = -(1.0 / (rho * rho)) * cos_th * sin_phi;
    H(1, 0) = (1.0 / rho) * cos_th * sin_phi;
    H(1, 1) = (1.0 / rho) * sin_th * cos_phi;
    H(1, 2) = -(1.0 / (rho * rho)) * sin_th * sin_phi;
    H(2, 0) = 0.0;
    H(2, 1) = -(1.0 / rho) * sin_phi;
    H(2, 2) = -(1.0 / (rho * rho)) * cos_phi;
    G = mm(dpC_dpG, H);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) hf[(size_t)j * cstr + (2 * c + i) * rstr] = wz[3 * i] * G(0, j) + wz[3 * i + 1] * G(1, j) + wz[3 * i + 2] * G(2, j);
}
__device__ __forceinline__ void rows_write_pose(int col, int c, const double *WI, const M3 &Ho, double lam, double *hx, int cstr, int rstr) {
  if (col < 0) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double so = WI[6 * i] * Ho(0, j) + WI[6 * i + 1] * Ho(1, j) + WI[6 * i + 2] * Ho(2, j);
      hx[(size_t)(col + j) * cstr + (2 * c + i) * rstr] = so;
      hx[(size_t)(col + 3 + j) * cstr + (2 * c + i) * rstr] = WI[6 * i + 3 + j] * lam;
    }
}
__device__ __forceinline__ void rows_write_dt(const JacParams &P, int c, const double *WI, const double *dtj, double *hx, int cstr, int rstr) {
  if (P.col_dt < 0) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    double s = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) s += WI[6 * i + q] * dtj[q];
    hx[(size_t)P.col_dt * cstr + (2 * c + i) * rstr] = s;
  }
}
__device__ __forceinline__ void rows_write_ext(const JacParams &P, int c, const double *wz, const V3 &lever, double *hx, int cstr, int rstr) {
  if (P.col_ext < 0) return;
  const M3 sk = skew3(lever);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      hx[(size_t)(P.col_ext + j) * cstr + (2 * c + i) * rstr] = wz[3 * i] * sk(0, j) + wz[3 * i + 1] * sk(1, j) + wz[3 * i + 2] * sk(2, j);
      hx[(size_t)(P.col_ext + 3 + j) * cstr + (2 * c + i) * rstr] = wz[3 * i + j];
    }
}
__device__ __forceinline__ void rows_write_int(const JacParams &P, int c, const double *Wm, const double *dzeta, double *hx, int cstr, int rstr) {
  if (P.col_int < 0) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hx[(size_t)(P.col_int + j) * cstr + (2 * c) * rstr] = Wm[0] * dzeta[j] + Wm[1] * dzeta[8 + j];
    hx[(size_t)(P.col_int + j) * cstr + (2 * c + 1) * rstr] = Wm[2] * dzeta[j] + Wm[3] * dzeta[8 + j];
  }
}

// Everything of an observation's rows behind the two interpolations: jac = the first-estimate polynomial with its Jacobians, (R_GtoI,
// p_IinG) = the estimate pose the residual is taken at.
__device__ void jacobian_rows_core(const JacParams &P, int f, int o, int s0, double tm, int c, const Interp &jac, M3 R_GtoI, V3 p_IinG, double *hf,
                                   double *hx, double *rs, int cstr, int rstr) {
  double r2[2], dzn[4], dzeta[16], dznp[6], dpC_dI[18], Wm[4], wz[6], WI[12];
  M3 dpC_dpG;
  V3 lever;
  const V3 pf = ldV(P.p_FinG + 3 * f), pf_fej = ldV(P.p_FinG_fej + 3 * f);
  rows_est_part(P, pf, P.obs_uv + 2 * o, R_GtoI, p_IinG, r2, dzn, dzeta);
  rows_fej_part(P, pf_fej, jac.R, jac.p, dznp, dpC_dpG, dpC_dI, lever);
  rows_whiten_part(P, o, rows_at_clone(P, tm), dzn, dznp, dpC_dI, Wm, wz, WI);
  rows_write_res(c, Wm, r2, rs, rstr);
  rows_write_hf(P, pf_fej, c, wz, dpC_dpG, hf, cstr, rstr);
#pragma unroll
  for (int w = 0; w < 4; ++w) rows_write_pose(P.clone_col[s0 + w], c, WI, jac.Ho[w], jac.lam[w], hx, cstr, rstr);
  rows_write_dt(P, c, WI, jac.dtj, hx, cstr, rstr);
  rows_write_ext(P, c, wz, lever, hx, cstr, rstr);
  rows_write_int(P, c, Wm, dzeta, hx, cstr, rstr);
}

// Per-observation scratch of the fused launches in LDS, one slot per row pair (stride PRE_STRIDE doubles: one lane per slot reads and
// writes without bank conflicts): the first-estimate interpolation with its Jacobians (what jacobian_rows_core takes as `jac`) and
// the estimate pose.  Neither depends on the feature's position, so three waves fill the slots while the first wave triangulates.
__device__ __forceinline__ void tri_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#define PRE_STRIDE 127
#define LPRE_STRIDE 113  // lines: the same slots up to PRE_PE, then the exchange of line_rows_split
enum { LPRE_R2 = 70, LPRE_DZLI = 72, LPRE_ATC = 84, LPRE_WM = 85, LPRE_WLI = 89, LPRE_WI = 101 };
enum { PRE_R = 0, PRE_P = 9, PRE_HO = 12, PRE_LAM = 48, PRE_DTJ = 52, PRE_RE = 58, PRE_PE = 67,
       // exchange between the waves of jacobian_rows_split
       PRE_R2 = 70, PRE_DZN = 72, PRE_DZETA = 76, PRE_ATC = 92, PRE_WM = 93, PRE_WZ = 97, PRE_WI = 103, PRE_G = 115, PRE_LEVER = 124 };
__device__ __forceinline__ void pre_store_jac(double *pre, const Interp &j) {
#pragma unroll
  for (int i = 0; i < 9; ++i) pre[PRE_R + i] = j.R.m[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) pre[PRE_P + i] = j.p[i];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
#pragma unroll
    for (int i = 0; i < 9; ++i) pre[PRE_HO + 9 * w + i] = j.Ho[w].m[i];
    pre[PRE_LAM + w] = j.lam[w];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) pre[PRE_DTJ + i] = j.dtj[i];
}
__device__ __forceinline__ void pre_load_jac(const double *pre, Interp &j) {
#pragma unroll
  for (int i = 0; i < 9; ++i) j.R.m[i] = pre[PRE_R + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) j.p[i] = pre[PRE_P + i];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
#pragma unroll
    for (int i = 0; i < 9; ++i) j.Ho[w].m[i] = pre[PRE_HO + 9 * w + i];
    j.lam[w] = pre[PRE_LAM + w];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) j.dtj[i] = pre[PRE_DTJ + i];
}
// Row slots of a feature's observations (wave 0 of the workgroup): slot = number of observations with bounding clones in front of it
// (ballot prefix), -1 for an observation without a row pair.  s0 = first clone of its interpolation window or -1.  Returns the
// number of observations with bounding clones.
__device__ __forceinline__ int assign_row_slots(const JacParams &P, int o0, int o1, int ld, const double *tm_l /* times of o0 .. */, int *s0_l, int *slot_l) {
  const int lane = threadIdx.x & 63;
  int base = 0;
  for (int ob = o0; ob < o1; ob += 64) {
    const int o = ob + lane;
    const bool have = o < o1;
    const double tm = (have ? tm_l[o - o0] : 0.0) + P.cam_dt;
    const int s0 = have ? bounding_start(P, tm) : -1;
    const unsigned long long vmask = __ballot(s0 >= 0);
    const int c = base + __popcll(vmask & ((1ull << lane) - 1ull));
    base += __popcll(vmask);
    if (have) {
      s0_l[o - o0] = s0;
      slot_l[o - o0] = (s0 >= 0 && 2 * c + 2 <= ld) ? c : -1;
    }
  }
  return base;
}
// estimate pose of observation o (CamHelper::get_imu_poses) from the window tables; the same values as campose_one's
__device__ __forceinline__ void est_pose_tab(const JacParams &P, const WinTab &T, int o, int s0, double tm, M3 &R_GtoI, V3 &p_IinG) {
  if (P.res_R) {
    R_GtoI = ldM(P.res_R + 9 * o);
    p_IinG = ldV(P.res_p + 3 * o);
  } else {
    Interp est;
    interpolate_tab(P, T, s0, tm, false, est);
    R_GtoI = est.R;
    p_IinG = est.p;
  }
}

// The rows of every observation of the workgroup's feature from the LDS slots, the pieces of jacobian_rows_core spread over the four
// waves (lane = observation in all of them; two barriers):
//   A  wave 0: projection chain at the first estimates | wave 1: residual + distortion Jacobians | wave 2: "observed at a clone time"
//   B  wave 0: Jacobian in the interpolated pose, noise, whitening (the long pole: four square roots and six divisions in a row)
//   C  wave w: the block of interpolation pose w; + wave 0: residual rows, time offset | 1: extrinsics | 2: intrinsics | 3: Hf
// One lane doing all of it in sequence took 17 k cycles (a wave retires a dependent fp64 operation every ~8 cycles whatever its lanes
// do); this form ~7 k.  Called by all 256 threads.
__device__ __forceinline__ void jacobian_rows_split(const JacParams &P, const V3 &pf, const V3 &pf_fej, int o0, int n_o, const int *s0_l,
                                                    const int *slot_l, const double *tm_l, const float *uv_l, double *pre, double *X, int ncol, int k) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *hf = X, *hx = X + 3, *rs = X + 3 + k;
  for (int ib = 0; ib < n_o; ib += 64) {
    const int i = ib + lane;
    const int c = i < n_o ? slot_l[i] : -1;
    const int o = o0 + i;
    double *pr = pre + (size_t)max(c, 0) * PRE_STRIDE;
    double dznp[6], dpC_dI[18];  // (wave 0, stage A -> B)
    if (c >= 0) {
      if (wave == 0) {
        M3 dpC_dpG;
        V3 lever;
        rows_fej_part(P, pf_fej, ldM(pr + PRE_R), ldV(pr + PRE_P), dznp, dpC_dpG, dpC_dI, lever);
#pragma unroll
        for (int q = 0; q < 9; ++q) pr[PRE_G + q] = dpC_dpG.m[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) pr[PRE_LEVER + q] = lever[q];
      } else if (wave == 1) {
        double r2[2], dzn[4], dzeta[16];
        rows_est_part(P, pf, uv_l + 2 * i, ldM(pr + PRE_RE), ldV(pr + PRE_PE), r2, dzn, dzeta);
        pr[PRE_R2] = r2[0], pr[PRE_R2 + 1] = r2[1];
#pragma unroll
        for (int q = 0; q < 4; ++q) pr[PRE_DZN + q] = dzn[q];
#pragma unroll
        for (int q = 0; q < 16; ++q) pr[PRE_DZETA + q] = dzeta[q];
      } else if (wave == 2) {
        pr[PRE_ATC] = rows_at_clone(P, tm_l[i] + P.cam_dt) ? 1.0 : 0.0;
      }
    }
    __syncthreads();
    if (c >= 0 && wave == 0) {
      double dzn[4], Wm[4], wz[6], WI[12];
#pragma unroll
      for (int q = 0; q < 4; ++q) dzn[q] = pr[PRE_DZN + q];
      rows_whiten_part(P, o, pr[PRE_ATC] != 0.0, dzn, dznp, dpC_dI, Wm, wz, WI);
#pragma unroll
      for (int q = 0; q < 4; ++q) pr[PRE_WM + q] = Wm[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) pr[PRE_WZ + q] = wz[q];
#pragma unroll
      for (int q = 0; q < 12; ++q) pr[PRE_WI + q] = WI[q];
    }
    __syncthreads();
    if (c >= 0) {
      double WI[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) WI[q] = pr[PRE_WI + q];
      rows_write_pose(P.clone_col[s0_l[i] + wave], c, WI, ldM(pr + PRE_HO + 9 * wave), pr[PRE_LAM + wave], hx, 1, ncol);
      if (wave == 0) {
        double Wm[4], r2[2] = {pr[PRE_R2], pr[PRE_R2 + 1]}, dtj[6];
#pragma unroll
        for (int q = 0; q < 4; ++q) Wm[q] = pr[PRE_WM + q];
#pragma unroll
        for (int q = 0; q < 6; ++q) dtj[q] = pr[PRE_DTJ + q];
        rows_write_res(c, Wm, r2, rs, ncol);
        rows_write_dt(P, c, WI, dtj, hx, 1, ncol);
      } else if (wave == 1) {
        double wz[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) wz[q] = pr[PRE_WZ + q];
        rows_write_ext(P, c, wz, ldV(pr + PRE_LEVER), hx, 1, ncol);
      } else if (wave == 2) {
        double Wm[4], dzeta[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) Wm[q] = pr[PRE_WM + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) dzeta[q] = pr[PRE_DZETA + q];
        rows_write_int(P, c, Wm, dzeta, hx, 1, ncol);
      } else {
        double wz[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) wz[q] = pr[PRE_WZ + q];
        rows_write_hf(P, pf_fej, c, wz, ldM(pr + PRE_G), hf, 1, ncol);
      }
    }
    __syncthreads();  // (a second pass reuses nothing of the slots, but its stage A must not overtake this pass's readers of X rows: distinct rows — kept for the slots)
  }
}

// jacobian_kernel + nullspace_kernel (+ triangulation in front, + the gate behind) in one launch, one workgroup of four waves per
// batch entry.  Round 4 order of work (stamps: PLV_KNOB_KERNEL_STAMPS):
//   1. window tables (all threads), row slots (wave 0)
//   2. wave 0: estimate poses of the observations from the tables, then the triangulation (serial in the observations: LM);
//      waves 1-3 meanwhile: the first-estimate interpolation + its Jacobians of every observation (position independent) into LDS,
//      and the rows of the covariance the gate will read are pulled into this XCD's L2
//   3. one lane per observation finishes its two rows from the LDS slots (residual, projection chain, whitening, blocks)
//   4. null space (compact WY: the three reflectors from the Hf panel by one wave, one pass over the other columns), 5. gate.
// Round 3 ran 2 -> 1 -> 3 with both interpolations inside step 3 and the poses of step 2 from the untabulated polynomial:
// 69 us mean per workgroup, of it poses 10, rows 13, null space 8 (profiles/r04).
// Dynamic LDS of the fused launches (doubles): X [ld][ncol] | null-space scratch | { window tables | triangulation scratch | slots
// [min(ld / 2, max_obs)][PRE_STRIDE] | camera poses [max_obs][12] | s0, slot (ints), valid (bytes) [max_obs] }; the gate's block
// (GateLds) overlays everything behind the null-space scratch: tables, scratch and slots are dead when the gate starts.
struct FusedLds {
  int ns, tab, tri, pre, cam, idx, end;  // offsets in doubles
};
// fdim 3 (points: one pose array, per observation time + uv + uvn = 3 doubles, 2 ints + 1 byte) or 6 (lines: camera and IMU poses, time
// + two segments = 5 doubles, 3 ints + 1 byte)
__host__ __device__ inline FusedLds fused_lds_layout(int ld, int ncol, int fdim, int n_tab, int max_obs, bool with_tri) {
  const bool lines = fdim == 6;
  FusedLds L;
  L.ns = ld * ncol;
  L.tab = (L.ns + nullspace_wy_scratch(ld, fdim) + 7) & ~7;
  L.tri = L.tab + n_tab * (int)(sizeof(WinTab) / 8);
  L.pre = L.tri + (with_tri && !lines ? tri_smem_doubles(max_obs) : 0);
  L.cam = L.pre + (ld / 2 < max_obs ? ld / 2 : max_obs) * (lines ? LPRE_STRIDE : PRE_STRIDE);
  L.idx = L.cam + max_obs * 12 * (lines ? 2 : 1);
  L.end = L.idx + (lines ? 5 : 3) * max_obs + (max_obs * (lines ? 13 : 9) + 7) / 8 + 1;
  return L;
}
static_assert(sizeof(WinTab) % 8 == 0, "fused_lds_layout counts WinTab in doubles");

__global__ void __launch_bounds__(256) jacobian_nullspace_kernel(JacParams P, int F, GatherArgs g, PointTriStage tri, GateStage gate) {
  extern __shared__ double jsm[];
  __shared__ int s_rows, s_base;
  __shared__ double tri_tot[32], s_tri[5];
  __shared__ double s_ct[JAC_MAX_WIN / 2 + 3];
  __shared__ int s_ccol[JAC_MAX_WIN / 2 + 3];
  if ((int)blockIdx.x >= F) {
    gather_cov_block(g, blockIdx.x - F);
    return;
  }
  int f = blockIdx.x;
  if (P.spec_order) {  // speculative submission: one workgroup per pool entry (spec_select_kernel's list; that kernel also does what
                       // workgroup 0 does for a launch — the column map's copy, the other gate counter zeroed: done here, behind a
                       // test of blockIdx.x, the same lines put the whole parameter block into scratch memory, 700 bytes per lane)
    if ((int)blockIdx.x >= *P.spec_count) return;
    f = P.spec_order[blockIdx.x];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double touch = 0.0;  // (one load per 128-byte line of the input block, see JacParams::in_base; never stored)
  for (int off = threadIdx.x * 128; off < P.in_bytes; off += 256 * 128) touch += *(const volatile double *)(P.in_base + off);
  const int ld = P.ld, k = P.k, ncol = 3 + k + 1, max_obs = tri.max_obs;
  const FusedLds lay = fused_lds_layout(ld, ncol, 3, 2 * max(P.n_clones - 3, 0), max_obs, tri.on != 0);
  double *X = jsm, *piv = jsm + lay.ns, *tri_smem = jsm + lay.tri, *pre = jsm + lay.pre, *cam = jsm + lay.cam;
  WinTab *tab = reinterpret_cast<WinTab *>(jsm + lay.tab);
  double *tm_l = jsm + lay.idx;
  float *uv_l = reinterpret_cast<float *>(tm_l + max_obs), *uvn_l = uv_l + 2 * max_obs;
  int *s0_l = reinterpret_cast<int *>(uvn_l + 2 * max_obs), *slot_l = s0_l + max_obs;
  unsigned char *valid_l = reinterpret_cast<unsigned char *>(slot_l + max_obs);
  jac_stamp(0);
  // What the workgroup reads more than once goes to LDS in one round of loads — clone times and columns, its observations' times and
  // image points: a dependent load from memory is 500+ cycles even when it hits, and the bounding-clone search, the row pieces and
  // the triangulation's passes are chains of them.  The pointers are then redirected (same indices as before).
  const int o0 = P.obs_ptr[f], o1 = P.obs_end ? P.obs_end[f] : P.obs_ptr[f + 1];
  if ((int)threadIdx.x < P.n_clones) s_ct[threadIdx.x] = P.clone_time[threadIdx.x], s_ccol[threadIdx.x] = P.clone_col[threadIdx.x];
  for (int i = threadIdx.x; i < o1 - o0; i += blockDim.x) {
    tm_l[i] = P.obs_time[o0 + i];
    uv_l[2 * i] = P.obs_uv[2 * (o0 + i)], uv_l[2 * i + 1] = P.obs_uv[2 * (o0 + i) + 1];
    if (tri.on) uvn_l[2 * i] = tri.uvn[2 * (o0 + i)], uvn_l[2 * i + 1] = tri.uvn[2 * (o0 + i) + 1];
  }
  for (int i = threadIdx.x; i < ld * ncol; i += blockDim.x) X[i] = 0.0;
  if (f == 0 && P.cols_out)
    for (int i = threadIdx.x; i < k; i += blockDim.x) P.cols_out[i] = P.cols_in[i];
  __syncthreads();
  P.clone_time = s_ct, P.clone_col = s_ccol;  // (same indices; what is indexed by observation is read from the LDS copies by local index)
  if (wave == 0) {
    const int base = assign_row_slots(P, o0, o1, ld, tm_l, s0_l, slot_l);
    if (lane == 0) s_base = base;
    jac_stamp(15);
  }
  build_window_tables(P, tab);  // (ends with a barrier: also orders the zero fill and the slots before what follows)
  jac_stamp(2);
  if (wave == 0) {
    // estimate poses: IMU pose into the observation's slot (rows), camera pose for the triangulation
    for (int i = lane; i < o1 - o0; i += 64) {
      const int s0 = s0_l[i], c = slot_l[i];
      valid_l[i] = s0 >= 0;
      if (s0 < 0) continue;
      M3 R_GtoI;
      V3 p_IinG;
      est_pose_tab(P, tab[2 * s0], o0 + i, s0, tm_l[i] + P.cam_dt, R_GtoI, p_IinG);
      if (c >= 0) {
        double *pr = pre + (size_t)c * PRE_STRIDE;
#pragma unroll
        for (int q = 0; q < 9; ++q) pr[PRE_RE + q] = R_GtoI.m[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) pr[PRE_PE + q] = p_IinG[q];
      }
      if (tri.on) {  // REF: CamHelper::get_cam_poses (as campose_one)
        const M3 R_GtoC = mm(ldM(P.R_ItoC), R_GtoI);
        const V3 p_CinG = vsub(p_IinG, mv(tp(R_GtoC), ldV(P.p_IinC)));
#pragma unroll
        for (int q = 0; q < 9; ++q) cam[12 * i + q] = R_GtoC.m[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) cam[12 * i + 9 + q] = p_CinG[q];
      }
    }
    if (tri.on) {
      tri_wave_sync();
      triangulate_feature(P, f, 0, o1 - o0, cam, valid_l, uvn_l, uv_l, tri.opt, tri.p_out, tri.ok_out, tri.err_out, max_obs, tri_smem, tri_tot, true, s_tri);
    }
  } else {
    for (int i = threadIdx.x - 64; i < o1 - o0; i += 192) {
      const int c = slot_l[i];
      if (c < 0) continue;
      Interp jac;
      const int s0 = s0_l[i];
      interpolate_tab(P, tab[2 * s0 + 1], s0, tm_l[i] + P.cam_dt, true, jac);
      pre_store_jac(pre + (size_t)c * PRE_STRIDE, jac);
    }
    if (gate.on) gate_stage_prior(gate, reinterpret_cast<double *>(reinterpret_cast<char *>(jsm) + gate.ps_off), P.cols_in, k, threadIdx.x - 64, 192);
  }
  __threadfence_block();
  __syncthreads();
  jac_stamp(1);
  bool selected;
  V3 pf, pf_fej;
  if (tri.on) {  // (the workgroup's own result from LDS: no trip through memory)
    selected = P.sel_flags[f] && s_tri[3] != 0.0 && s_tri[4] < 3.0;
    pf = pf_fej = ldV(s_tri);  // MSCKF features: FEJ value = estimate (REF CamHelper.cpp:556-557)
  } else {
    selected = !P.tri_ok || candidate_selected(P, f);
    pf = ldV(P.p_FinG + 3 * f), pf_fej = ldV(P.p_FinG_fej + 3 * f);
  }
  if (selected) {  // (block-uniform)
    jacobian_rows_split(P, pf, pf_fej, o0, o1 - o0, s0_l, slot_l, tm_l, uv_l, pre, X, ncol, k);
    if (threadIdx.x == 0) {
      s_rows = min(2 * s_base, ld & ~1);
      P.rows[f] = 2 * s_base;
      if (P.spec_pass) atomicAdd(P.spec_pass, 1);  // (see JacParams::spec_pass)
    }
  } else if (threadIdx.x == 0) {
    s_rows = 0;
    P.rows[f] = 0;
  }
  __syncthreads();
  jac_stamp(3);
  const int rows = s_rows;
  GateLds &gl = *reinterpret_cast<GateLds *>(reinterpret_cast<char *>(jsm) + gate.lds_off);
  const double *gPs = reinterpret_cast<const double *>(reinterpret_cast<char *>(jsm) + gate.ps_off);
  double *gT = reinterpret_cast<double *>(reinterpret_cast<char *>(jsm) + gate.t_off);
  if ((tri.on || P.tri_ok) && rows == 0) {  // one-submission update: a candidate the selection did not take is an empty system (rows[f] = 0) that
                                            // nothing reads — its padded block is not written (rocprofv3, round 2: 1.6 MB per launch, mostly these)
    if (gate.on) gate_tail(gate, gl, gPs, gT, f, X, ncol, 3, 0, 0, k, P.cols_in);  // (verdict "not accepted" + its share of the probe block)
    return;
  }
  const int shift = rows > 3 ? 3 : 0;  // (a block with no more rows than Hf has columns is left as it is, as nullspace_kernel does)
  if (shift) nullspace_householder_wy<3>(X, piv, rows, ncol);
  jac_stamp(4);
  double *hf = P.Hf + (size_t)f * 3 * ld, *hx = P.Hx + (size_t)f * k * ld, *rs = P.res + (size_t)f * ld;
  // (with the gate as this launch's tail nothing reads the projected block from memory any more: the gate takes it from LDS and
  //  leaves the accepted rows in the stack — 0.7 MB of writes per launch less, rocprofv3 WRITE_SIZE)
  for (int j = gate.on ? ncol : (int)threadIdx.x; j < ncol; j += blockDim.x) {
    double *dst = j < 3 ? hf + j * ld : (j < 3 + k ? hx + (size_t)(j - 3) * ld : rs);
    const int off = j < 3 ? 0 : shift;
    for (int i0 = 0; i0 < ld; i0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = i0 + u + off;
        v[u] = r < ld ? X[r * ncol + j] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < ld) dst[i0 + u] = v[u];
    }
  }
  jac_stamp(5);
  if (gate.on) gate_tail(gate, gl, gPs, gT, f, X, ncol, 3, shift, min(rows, ld), k, P.cols_in);  // (X is only read from here on)
  if (touch == 1.2345678e300) P.rows[f] = -1;  // (never: keeps the touch loads)
  jac_stamp(9);
}

// ------------------------------------------------------------------------------------------
// a18/a19: camera pose of every observation (thread per observation), then linear triangulation +
// Levenberg-Marquardt refinement + reprojection error (thread per feature).
//   CamHelper::get_imu_poses / get_cam_poses          REF: PL/update/cam/CamHelper.cpp:327-395
//   FeatureInitializer::single_triangulation          REF: OV/feat/FeatureInitializer.cpp:30-112
//   FeatureInitializer::single_gaussnewton            REF: OV/feat/FeatureInitializer.cpp:197-375
//   CamHelper::moving_consistency (mean reprojection) REF: PL/update/cam/CamHelper.cpp:426-483
__device__ void campose_one(const JacParams &P, int o, double *__restrict__ poses /*[n_obs][12]*/, unsigned char *__restrict__ valid,
                            double *__restrict__ imu) {
  const double tm = P.obs_time[o] + P.cam_dt;
  const int s0 = bounding_start(P, tm);
  valid[o] = s0 >= 0;
  if (s0 < 0) return;
  M3 R_GtoI;
  V3 p_IinG;
  if (P.res_R) {
    R_GtoI = ldM(P.res_R + 9 * o);
    p_IinG = ldV(P.res_p + 3 * o);
  } else {
    Interp est;
    interpolate(P, s0, tm, false, false, est);
    R_GtoI = est.R;
    p_IinG = est.p;
  }
  if (imu) {
#pragma unroll
    for (int i = 0; i < 9; ++i) imu[12 * o + i] = R_GtoI.m[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) imu[12 * o + 9 + i] = p_IinG[i];
  }
  const M3 R_GtoC = mm(ldM(P.R_ItoC), R_GtoI);
  const V3 p_CinG = vsub(p_IinG, mv(tp(R_GtoC), ldV(P.p_IinC)));
#pragma unroll
  for (int i = 0; i < 9; ++i) poses[12 * o + i] = R_GtoC.m[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) poses[12 * o + 9 + i] = p_CinG[i];
}

// a19, CPI branch: State::get_interpolated_pose_imu (REF: State.cpp:1138-1155) over have_cpi's first two stages
// (:273-355): the record stored at exactly t, else create_new_cpi_linear between the neighbouring records of the
// same clone.  Thread per query; the table is sorted by time (State::cpis is a std::map).  The reference inserts
// every interpolated record back into the map; a later query between two interpolated records then interpolates
// along the same geodesic / the same line, so answering every query from the original table gives the same pose
// up to rounding.  ok = 0 where the reference would fall through to create_new_cpi_integrate (needs the IMU
// buffer: SURVEY 8(f) rank 2).
__device__ int cpi_find(const double *t, int n, double x) {  // index of the record stored at exactly x, or -1
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (t[mid] < x) lo = mid + 1; else hi = mid;
  }
  return (lo < n && t[lo] == x) ? lo : -1;
}
__device__ int clone_find(const CpiParams &C, double x) {
  for (int i = 0; i < C.n_clones; ++i)
    if (C.clone_time[i] == x) return i;
  return -1;
}
__global__ void __launch_bounds__(64) cpi_pose_kernel(CpiParams C, const double *__restrict__ tq, double *__restrict__ Rout,
                                                      double *__restrict__ pout, unsigned char *__restrict__ ok) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= C.n_q) return;
  ok[q] = 0;
  const double t = tq[q];
  const int n = C.n;
  M3 R_I0toIk;
  V3 alpha;
  double clone_t, dt;
  const int e = cpi_find(C.t, n, t);
  if (e >= 0 && clone_find(C, C.clone_t[e]) >= 0) {  // :275-277
    R_I0toIk = ldM(C.R + 9 * e);
    alpha = ldV(C.alpha + 3 * e);
    clone_t = C.clone_t[e];
    dt = C.dt[e];
  } else {  // create_new_cpi_linear :286-355
    if (n == 0 || t < C.t[0] || t > C.t[n - 1]) return;
    int lo = 0, hi = n;  // lower_bound(t)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (C.t[mid] < t) lo = mid + 1; else hi = mid;
    }
    const int i0 = (t == C.t[0]) ? 0 : lo - 1;  // :312-318 equal-or-lower ... strictly lower unless t is the first key
    int up = lo;                                 // upper_bound(t)
    while (up < n && C.t[up] <= t) ++up;
    const int i1 = (t == C.t[n - 1]) ? n - 1 : up;  // :321-328
    if (C.clone_t[i0] != C.clone_t[i1]) return;     // :334-338
    if (C.clone_t[i0] < C.clone_time[0]) return;    // :340-344
    const double lambda = (t - C.t[i0]) / (C.t[i1] - C.t[i0]);
    const M3 R0 = ldM(C.R + 9 * i0), R1 = ldM(C.R + 9 * i1);
    R_I0toIk = mm(exp_so3(vsc(log_so3(mm(R1, tp(R0))), lambda)), R0);
    const V3 a0 = ldV(C.alpha + 3 * i0), a1 = ldV(C.alpha + 3 * i1);
    alpha = vadd(vsc(a0, 1 - lambda), vsc(a1, lambda));
    clone_t = C.clone_t[i0];
    dt = t - clone_t;
  }
  const int ci = clone_find(C, clone_t), vi = cpi_find(C.t, n, clone_t);
  if (ci < 0 || vi < 0) return;  // clones.at / cpis.at would throw
  const M3 RGtoI0 = ldM(C.clone_R + 9 * ci);
  const V3 p0 = ldV(C.clone_p + 3 * ci), v0 = ldV(C.v + 3 * vi);
  const V3 g{{C.gravity[0], C.gravity[1], C.gravity[2]}};
  const M3 RGtoI = mm(R_I0toIk, RGtoI0);
  V3 p = vadd(p0, vsc(v0, dt));                   // :1153
  p = vsub(p, vsc(vsc(vsc(g, 0.5), dt), dt));
  p = vadd(p, mv(tp(RGtoI0), alpha));
#pragma unroll
  for (int i = 0; i < 9; ++i) Rout[9 * q + i] = RGtoI.m[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) pout[3 * q + i] = p[i];
  ok[q] = 1;
}

int launch_cpi_poses(plv_ctx *ctx, const CpiParams &C, const double *d_tq, double *d_R, double *d_p, unsigned char *d_ok) {
  ProfScope ps(ctx->prof, "cpi_pose_kernel", ctx->stream);
  hipLaunchKernelGGL(cpi_pose_kernel, dim3((C.n_q + 63) / 64), dim3(64), 0, ctx->stream, C, d_tq, d_R, d_p, d_ok);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

__device__ bool solve3(const M3 &A, const V3 &b, V3 &x) {
  double a[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) a[i][j] = A(i, j);
    a[i][3] = b[i];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int piv = c;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (r > c && fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    double pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pv[j] = piv == 0 ? a[0][j] : (piv == 1 ? a[1][j] : a[2][j]);
    if (pv[c] == 0) return false;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (r == piv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[r][j] = a[c][j];
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) a[c][j] = pv[j];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (r > c) {
        const double f = a[r][c] / a[c][c];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j >= c) a[r][j] -= f * a[c][j];
      }
  }
  x[2] = a[2][3] / a[2][2];
  x[1] = (a[1][3] - a[1][2] * x[2]) / a[1][1];
  x[0] = (a[0][3] - a[0][1] * x[1] - a[0][2] * x[2]) / a[0][0];
  return true;
}
__device__ void sym_eig3(const M3 &A, double ev[3]) {
  const double PI = 3.14159265358979323846;
  const double p1 = A(0, 1) * A(0, 1) + A(0, 2) * A(0, 2) + A(1, 2) * A(1, 2);
  const double q = (A(0, 0) + A(1, 1) + A(2, 2)) / 3;
  const double p2 = (A(0, 0) - q) * (A(0, 0) - q) + (A(1, 1) - q) * (A(1, 1) - q) + (A(2, 2) - q) * (A(2, 2) - q) + 2 * p1;
  const double p = sqrt(p2 / 6);
  if (p == 0) {
    ev[0] = ev[1] = ev[2] = q;
    return;
  }
  const M3 B = ms(ma(A, ms(eye3(), -q)), 1 / p);
  const double detB = B(0, 0) * (B(1, 1) * B(2, 2) - B(1, 2) * B(2, 1)) - B(0, 1) * (B(1, 0) * B(2, 2) - B(1, 2) * B(2, 0)) +
                      B(0, 2) * (B(1, 0) * B(2, 1) - B(1, 1) * B(2, 0));
  const double r = detB / 2;
  const double phi = r <= -1 ? PI / 3 : (r >= 1 ? 0 : acos(r) / 3);
  ev[0] = q + 2 * p * cos(phi);
  ev[2] = q + 2 * p * cos(phi + (2 * PI / 3));
  ev[1] = 3 * q - ev[0] - ev[2];
}

// a18: FeatureInitializer::single_triangulation + single_gaussnewton and the reprojection check of CamHelper::feature_triangulation
// (REF: open_vins/ov_core/src/feat/FeatureInitializer.cpp:29-309, PL-VIWO/src/update/cam/CamHelper.cpp:397-483).
// One wave per feature.  Every pass of the algorithm (linear system, cost of a trial point, Hessian + gradient, baseline, mean
// reprojection error) is "a term per observation, summed": the lanes compute the terms of their observations into LDS, then lane c
// adds up component c over the observations IN OBSERVATION ORDER, so the sums are the ones of the serial loop, bit for bit
// (both sides are built -ffp-contract=off), and the Levenberg-Marquardt control flow, which every lane executes on the same totals,
// takes the same branches as a thread-per-feature version (160 us for a pool of ~20 features; this form: a few us).
struct TriObs {     // per valid observation, relative to the anchor pose (the newest observation)
  double R[9];      // R_AtoCi
  double pa[3];     // p_AinCi
  double pc[3];     // p_CiinA
};
static_assert(sizeof(TriObs) == 15 * 8, "tri_smem_doubles counts 15 doubles per TriObs");
// One feature, ONE wave (the 64 lanes that call it; other waves of the workgroup must not): poses of its observations, linear
// triangulation, Levenberg-Marquardt refinement, reprojection error.  tri_smem: max_obs * (sizeof(TriObs) + TRI_TERMS * 8 + 4) + 16 bytes
// of LDS, tot: 32 doubles of LDS.  Wave-level synchronisation only (the lanes run in lockstep; the fences order the LDS traffic).
// Observations o0 .. o1 - 1 index poses / valid / uvn / uv: the global arrays of triangulate_kernel (o0 = obs_ptr[f]) or a workgroup's
// LDS copies of its own feature's observations (o0 = 0, poses_ready: the caller has filled poses and valid).
__device__ void triangulate_feature(const JacParams &P, int f, int o0, int o1, double *poses, unsigned char *valid, const float *uvn, const float *uv,
                                    const plv_tri_options &opt, double *__restrict__ p_out, unsigned char *__restrict__ ok_out,
                                    double *__restrict__ err_out, int max_obs, double *tri_smem, double *tot, bool poses_ready,
                                    double *res_l /* LDS copy of the result for the caller's workgroup: p [3], ok, err */) {
  const int lane = threadIdx.x & 63;
  if (!poses_ready) {  // camera poses of this feature's observations (CamHelper::get_imu_poses / get_cam_poses), one lane each: no separate launch
    for (int o = o0 + lane; o < o1; o += 64) campose_one(P, o, poses, valid, nullptr);
    __threadfence_block();
    tri_wave_sync();
  }
  jac_stamp(10);
  TriObs *ob = reinterpret_cast<TriObs *>(tri_smem);                       // [max_obs]
  double *term = tri_smem + (size_t)max_obs * (sizeof(TriObs) / 8);        // [max_obs][TRI_TERMS]
  int *list = reinterpret_cast<int *>(term + (size_t)max_obs * TRI_TERMS);  // [max_obs] indices of the valid observations
  // ordered compaction of the valid observations
  int M = 0;
  for (int base = o0; base < o1; base += 64) {
    const int o = base + lane;
    const bool v = o < o1 && valid[o];
    const unsigned long long m = __ballot(v);
    if (v) list[M + __popcll(m & ((1ull << lane) - 1ull))] = o;
    M += __popcll(m);
  }
  if (lane == 0) {
    p_out[3 * f] = p_out[3 * f + 1] = p_out[3 * f + 2] = 0;
    ok_out[f] = 0;
    if (err_out) err_out[f] = 0;
    if (res_l) res_l[0] = res_l[1] = res_l[2] = res_l[3] = res_l[4] = 0.0;
    if (P.tri_dbg) P.tri_dbg[4 * f] = P.tri_dbg[4 * f + 1] = P.tri_dbg[4 * f + 2] = P.tri_dbg[4 * f + 3] = NAN;
  }
  if (M < 2) return;
  tri_wave_sync();
  const int last = list[M - 1];
  const M3 R_GtoA = ldM(poses + 12 * last);  // anchor = newest observation (FeatureInitializer.cpp:44-45)
  const V3 p_AinG = ldV(poses + 12 * last + 9);
  // sums `n` terms per observation in observation order; afterwards tot[0..n) holds the totals for every lane
  auto reduce = [&](int n) {
    tri_wave_sync();
    if (lane < n) {  // (the terms of eight observations are requested together, then added in observation order)
      double s = 0;
      int q = 0;
      for (; q + 8 <= M; q += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = term[(q + u) * TRI_TERMS + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
      }
      for (; q < M; ++q) s += term[q * TRI_TERMS + lane];
      tot[lane] = s;
    }
    tri_wave_sync();
  };
  // ---- linear triangulation: A = sum Bp^T Bp, b = sum Ai p_CiinA
  for (int q = lane; q < M; q += 64) {
    const int o = list[q];
    const M3 R_AtoCi = mm(ldM(poses + 12 * o), tp(R_GtoA));
    const V3 p_CiinA = mv(R_GtoA, vsub(ldV(poses + 12 * o + 9), p_AinG));
    const V3 p_AinCi = vsc(mv(R_AtoCi, p_CiinA), -1.0);
#pragma unroll
    for (int i = 0; i < 9; ++i) ob[q].R[i] = R_AtoCi.m[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ob[q].pa[i] = p_AinCi[i], ob[q].pc[i] = p_CiinA[i];
    V3 bi = mv(tp(R_AtoCi), V3{{(double)uvn[2 * o], (double)uvn[2 * o + 1], 1.0}});
    bi = vsc(bi, 1.0 / vnorm(bi));
    const M3 Bp = skew3(bi);
    const M3 Ai = mm(tp(Bp), Bp);
    const V3 bq = mv(Ai, p_CiinA);
    double *tq = term + q * TRI_TERMS;
    tq[0] = Ai(0, 0), tq[1] = Ai(0, 1), tq[2] = Ai(0, 2), tq[3] = Ai(1, 1), tq[4] = Ai(1, 2), tq[5] = Ai(2, 2);
    tq[6] = bq[0], tq[7] = bq[1], tq[8] = bq[2];
  }
  reduce(9);
  M3 A;
  A(0, 0) = tot[0], A(0, 1) = A(1, 0) = tot[1], A(0, 2) = A(2, 0) = tot[2], A(1, 1) = tot[3], A(1, 2) = A(2, 1) = tot[4], A(2, 2) = tot[5];
  const V3 b{{tot[6], tot[7], tot[8]}};
  V3 pf;
  if (!solve3(A, b, pf)) return;
  double ev[3];
  sym_eig3(A, ev);
  const double condA = ev[0] / ev[2];
  if (P.tri_dbg && lane == 0) P.tri_dbg[4 * f] = fabs(condA), P.tri_dbg[4 * f + 1] = pf[2];
  if (fabs(condA) > opt.max_cond_number || pf[2] < opt.min_dist || pf[2] > opt.max_dist || isnan(vnorm(pf))) return;
  jac_stamp(11);
  auto tri_error = [&](double alpha, double beta, double rho) {
    for (int q = lane; q < M; q += 64) {
      const TriObs &c = ob[q];
      const int o = list[q];
      const double hi1 = c.R[0] * alpha + c.R[1] * beta + c.R[2] + rho * c.pa[0];
      const double hi2 = c.R[3] * alpha + c.R[4] * beta + c.R[5] + rho * c.pa[1];
      const double hi3 = c.R[6] * alpha + c.R[7] * beta + c.R[8] + rho * c.pa[2];
      const float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
      const float r0 = uvn[2 * o] - z0, r1 = uvn[2 * o + 1] - z1;
      const float nrm = sqrtf(r0 * r0 + r1 * r1);
      term[q * TRI_TERMS] = (double)nrm * (double)nrm;
    }
    reduce(1);
    return tot[0];
  };
  if (opt.refine_features) {
    double rho = 1 / pf[2], alpha = pf[0] / pf[2], beta = pf[1] / pf[2];
    double lam = 1e-3, eps = 10000;
    int runs = 0;
    bool recompute = true;
    M3 Hess{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
    V3 grad{{0, 0, 0}};
    double cost_old = tri_error(alpha, beta, rho);
    int lm_pass = 0;
    const bool spec = M <= 16;  // (uniform) room for four damping factors side by side: 16 lanes each
    const int grp = lane >> 4, ql = lane & 15;
    bool stop = false;
    while (runs < 5 && lam < 1e10 && eps > 1e-6) {
      jac_stamp(16 + min(lm_pass++, 11));
      if (recompute) {
        for (int q = lane; q < M; q += 64) {
          const TriObs &c = ob[q];
          const int o = list[q];
          const double hi1 = c.R[0] * alpha + c.R[1] * beta + c.R[2] + rho * c.pa[0];
          const double hi2 = c.R[3] * alpha + c.R[4] * beta + c.R[5] + rho * c.pa[1];
          const double hi3 = c.R[6] * alpha + c.R[7] * beta + c.R[8] + rho * c.pa[2];
          const double h3s = pow(hi3, 2.0);
          const double Hj[6] = {(c.R[0] * hi3 - hi1 * c.R[6]) / h3s, (c.R[1] * hi3 - hi1 * c.R[7]) / h3s, (c.pa[0] * hi3 - hi1 * c.pa[2]) / h3s,
                                (c.R[3] * hi3 - hi2 * c.R[6]) / h3s, (c.R[4] * hi3 - hi2 * c.R[7]) / h3s, (c.pa[1] * hi3 - hi2 * c.pa[2]) / h3s};
          const float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
          const double r0 = (double)(uvn[2 * o] - z0), r1 = (double)(uvn[2 * o + 1] - z1);
          double *tq = term + q * TRI_TERMS;
          tq[0] = Hj[0] * r0 + Hj[3] * r1, tq[1] = Hj[1] * r0 + Hj[4] * r1, tq[2] = Hj[2] * r0 + Hj[5] * r1;
          tq[3] = Hj[0] * Hj[0] + Hj[3] * Hj[3], tq[4] = Hj[0] * Hj[1] + Hj[3] * Hj[4], tq[5] = Hj[0] * Hj[2] + Hj[3] * Hj[5];
          tq[6] = Hj[1] * Hj[1] + Hj[4] * Hj[4], tq[7] = Hj[1] * Hj[2] + Hj[4] * Hj[5], tq[8] = Hj[2] * Hj[2] + Hj[5] * Hj[5];
        }
        reduce(9);
        grad = V3{{tot[0], tot[1], tot[2]}};
        Hess(0, 0) = tot[3], Hess(0, 1) = Hess(1, 0) = tot[4], Hess(0, 2) = Hess(2, 0) = tot[5];
        Hess(1, 1) = tot[6], Hess(1, 2) = Hess(2, 1) = tot[7], Hess(2, 2) = tot[8];
      }
      if (spec) {
        // Four damping factors at once (round 4): a failed step changes nothing but lam (x 10), so the steps the serial loop would
        // try next — lam, 10 lam, 100 lam, 1000 lam on the same Hessian — are independent.  Lane group g (16 lanes, one per
        // observation) solves for 10^g lam and forms its cost terms; the four costs are summed in observation order as before
        // (component g of reduce); then the serial loop's decisions are replayed on them: the first group whose step the loop
        // would have reached AND taken decides, exactly as if the groups before it had been tried one by one.  A streak of failed
        // steps (the tail of most refinements: ~1.5 passes of 4.7 on average, 8 of 12 for the slowest) costs one pass per four.
        double lamg = lam;
        for (int i = 0; i < grp; ++i) lamg = lamg * 10;
        M3 Hl = Hess;
#pragma unroll
        for (int r = 0; r < 3; ++r) Hl(r, r) *= (1.0 + lamg);
        V3 dxg{{0, 0, 0}};
        const bool okg = solve3(Hl, grad, dxg);
        if (ql < M) {
          double tv = 0.0;
          if (okg) {
            const TriObs &c = ob[ql];
            const int o = list[ql];
            const double a2 = alpha + dxg[0], b2 = beta + dxg[1], r2 = rho + dxg[2];
            const double hi1 = c.R[0] * a2 + c.R[1] * b2 + c.R[2] + r2 * c.pa[0];
            const double hi2 = c.R[3] * a2 + c.R[4] * b2 + c.R[5] + r2 * c.pa[1];
            const double hi3 = c.R[6] * a2 + c.R[7] * b2 + c.R[8] + r2 * c.pa[2];
            const float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
            const float r0 = uvn[2 * o] - z0, r1 = uvn[2 * o + 1] - z1;
            const float nrm = sqrtf(r0 * r0 + r1 * r1);
            tv = (double)nrm * (double)nrm;
          }
          term[ql * TRI_TERMS + grp] = tv;
        }
        if (ql == 0) {
          tot[16 + 4 * grp] = dxg[0], tot[17 + 4 * grp] = dxg[1], tot[18 + 4 * grp] = dxg[2];
          tot[19 + 4 * grp] = okg ? 1.0 : 0.0;
        }
        reduce(4);
        bool leave = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (leave || !(lam < 1e10)) {  // (the while condition in front of the pass the serial loop would run now; runs and eps have not moved)
            leave = true;
            continue;
          }
          if (tot[19 + 4 * j] == 0.0) {  // solve3 failed: break
            stop = leave = true;
            continue;
          }
          const double cost = tot[j];
          const V3 dx{{tot[16 + 4 * j], tot[17 + 4 * j], tot[18 + 4 * j]}};
          if (cost <= cost_old && (cost_old - cost) / cost_old < 1e-6) {
            alpha += dx[0];
            beta += dx[1];
            rho += dx[2];
            eps = 0;
            stop = leave = true;
          } else if (cost <= cost_old) {
            recompute = true;
            cost_old = cost;
            alpha += dx[0];
            beta += dx[1];
            rho += dx[2];
            runs++;
            lam = lam / 10;
            eps = vnorm(dx);
            leave = true;  // (a new linearisation: the other groups' steps belong to the old one)
          } else {
            recompute = false;
            lam = lam * 10;
          }
        }
        tri_wave_sync();  // (tot is rewritten by the next pass)
        if (stop) break;
        continue;
      }
      M3 Hl = Hess;
#pragma unroll
      for (int r = 0; r < 3; ++r) Hl(r, r) *= (1.0 + lam);
      V3 dx;
      if (!solve3(Hl, grad, dx)) break;
      const double cost = tri_error(alpha + dx[0], beta + dx[1], rho + dx[2]);
      if (cost <= cost_old && (cost_old - cost) / cost_old < 1e-6) {
        alpha += dx[0];
        beta += dx[1];
        rho += dx[2];
        eps = 0;
        break;
      }
      if (cost <= cost_old) {
        recompute = true;
        cost_old = cost;
        alpha += dx[0];
        beta += dx[1];
        rho += dx[2];
        runs++;
        lam = lam / 10;
        eps = vnorm(dx);
      } else {
        recompute = false;
        lam = lam * 10;
      }
    }
    jac_stamp(12);
    pf = V3{{alpha / rho, beta / rho, 1 / rho}};
    const V3 dir = vsc(pf, 1.0 / vnorm(pf));
    for (int q = lane; q < M; q += 64) {
      const V3 p_CiinA{{ob[q].pc[0], ob[q].pc[1], ob[q].pc[2]}};
      const double along = p_CiinA[0] * dir[0] + p_CiinA[1] * dir[1] + p_CiinA[2] * dir[2];
      term[q * TRI_TERMS] = vnorm(vsub(p_CiinA, vsc(dir, along)));
    }
    tri_wave_sync();
    double base_max = 0;
    for (int q = 0; q < M; ++q) base_max = fmax(base_max, term[q * TRI_TERMS]);
    tri_wave_sync();
    if (P.tri_dbg && lane == 0) P.tri_dbg[4 * f + 2] = pf[2], P.tri_dbg[4 * f + 3] = vnorm(pf) / base_max;
    if (pf[2] < opt.min_dist || pf[2] > opt.max_dist || (vnorm(pf) / base_max) > opt.max_baseline || isnan(vnorm(pf))) return;
  }
  const V3 pg = vadd(mv(tp(R_GtoA), pf), p_AinG);
  double e = 0;
  if (err_out) {  // mean pixel reprojection error (CamHelper.cpp:441-470), float round trip of distort_d included
    const double *K = P.K;
    for (int q = lane; q < M; q += 64) {
      const int o = list[q];
      const V3 pC = mv(ldM(poses + 12 * o), vsub(pg, ldV(poses + 12 * o + 9)));
      const double x = (double)(float)(pC[0] / pC[2]), y = (double)(float)(pC[1] / pC[2]);
      const double r = sqrt(x * x + y * y), r_2 = r * r, r_4 = r_2 * r_2;
      const double x1 = x * (1 + K[4] * r_2 + K[5] * r_4) + 2 * K[6] * x * y + K[7] * (r_2 + 2 * x * x);
      const double y1 = y * (1 + K[4] * r_2 + K[5] * r_4) + K[6] * (r_2 + 2 * y * y) + 2 * K[7] * x * y;
      const double r0 = (double)uv[2 * o] - (double)(float)(K[0] * x1 + K[2]);
      const double r1 = (double)uv[2 * o + 1] - (double)(float)(K[1] * y1 + K[3]);
      term[q * TRI_TERMS] = sqrt(r0 * r0 + r1 * r1);
    }
    reduce(1);
    e = tot[0];
  }
  if (lane == 0) {
    p_out[3 * f] = pg[0];
    p_out[3 * f + 1] = pg[1];
    p_out[3 * f + 2] = pg[2];
    ok_out[f] = 1;
    if (err_out) err_out[f] = e / M;
    if (res_l) res_l[0] = pg[0], res_l[1] = pg[1], res_l[2] = pg[2], res_l[3] = 1.0, res_l[4] = e / M;
  }
}

__global__ void __launch_bounds__(64) triangulate_kernel(JacParams P, double *poses, unsigned char *valid, const float *__restrict__ uvn,
                                                         plv_tri_options opt, double *__restrict__ p_out,
                                                         unsigned char *__restrict__ ok_out, double *__restrict__ err_out, int max_obs) {
  extern __shared__ double tri_smem[];
  __shared__ double tot[32];  // (totals of a pass + the four candidate steps of a speculative one)
  triangulate_feature(P, blockIdx.x, P.obs_ptr[blockIdx.x], P.obs_ptr[blockIdx.x + 1], poses, valid, uvn, P.obs_uv, opt, p_out, ok_out, err_out, max_obs, tri_smem,
                      tot);
}


// ------------------------------------------------------------------------------------------ lines
// a28: LineHelper::get_line_feature_jacobian_full   REF: PL-VIWO/src/update/cam/linefeat/LineHelper.cpp:733-1024
// (point-line coupling off, UpdaterCamera.cpp:373).  One workgroup per line, one lane per
// observation, same slot / zero-fill scheme as jacobian_kernel.  The reference's arithmetic is
// kept: dz/dl starts from Identity(2,3) (third column stays zero) and ln_2 = l0^2 + l1 + l1 (:921-928);
// the pose written back by get_interpolated_jacobian (first estimates) feeds dli_dI (:898,940-945)
// while G_to_I keeps the estimate pose (:846-850).
__device__ __forceinline__ V3 cross3(const V3 &a, const V3 &b) {
  return V3{{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}
__device__ __forceinline__ double dot3(const V3 &a, const V3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ---- the rows of one line observation in pieces (line_rows runs them one after the other on one lane; the fused launch spreads
// them over the four waves, line_rows_split).  Every value is formed by the same expression wherever its piece runs.
// (1) residual (signed distances of the segment's end points to the projected line) and dzli = d(residual)/d(line in the IMU frame,
//     [n; v]) at the estimate pose; Rsk = -Re skew(pe) (Hf needs it with Re)
__device__ __forceinline__ void line_est_part(const JacParams &P, const V3 &nG, const V3 &vG, const float *seg /* the observed segment: x1 y1 x2 y2 */, const M3 &Re,
                                              const V3 &pe, double *r2, double *dzli, M3 &Rsk) {
  const M3 R_ItoC = ldM(P.R_ItoC);
  const V3 p_IinC = ldV(P.p_IinC);
  const double *Kc = P.K;
  const double Kl[9] = {Kc[1], 0, 0, 0, Kc[0], 0, -Kc[1] * Kc[2], -Kc[0] * Kc[3], Kc[0] * Kc[1]};
  Rsk = ms(mm(Re, skew3(pe)), -1.0);
  const V3 nI = vadd(mv(Re, nG), mv(Rsk, vG)), vI = mv(Re, vG);
  const M3 SR = mm(skew3(p_IinC), R_ItoC);
  const V3 nC = vadd(mv(R_ItoC, nI), mv(SR, vI));
  const double l3[3] = {Kl[0] * nC[0] + Kl[1] * nC[1] + Kl[2] * nC[2], Kl[3] * nC[0] + Kl[4] * nC[1] + Kl[5] * nC[2],
                        Kl[6] * nC[0] + Kl[7] * nC[1] + Kl[8] * nC[2]};
  const double us[3] = {(double)seg[0], (double)seg[1], 1.0};
  const double ue[3] = {(double)seg[2], (double)seg[3], 1.0};
  const double lnorm = sqrt(l3[0] * l3[0] + l3[1] * l3[1]);
  const double ds = us[0] * l3[0] + us[1] * l3[1] + us[2] * l3[2], de = ue[0] * l3[0] + ue[1] * l3[1] + ue[2] * l3[2];
  r2[0] = ds / lnorm, r2[1] = de / lnorm;
  const double ln_2 = l3[0] * l3[0] + l3[1] + l3[1];
  double dzl[6] = {1, 0, 0, 0, 1, 0};
  dzl[0] = us[0] - (l3[0] * ds) / ln_2;
  dzl[1] = us[1] - (l3[1] * ds) / ln_2;
  dzl[3] = ue[0] - (l3[0] * de) / ln_2;
  dzl[4] = ue[1] - (l3[1] * de) / ln_2;
  const double isq = 1 / sqrt(ln_2);
#pragma unroll
  for (int i = 0; i < 6; ++i) dzl[i] *= isq;
  double dzK[6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) dzK[3 * i + j] = dzl[3 * i] * Kl[j] + dzl[3 * i + 1] * Kl[3 + j] + dzl[3 * i + 2] * Kl[6 + j];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dzli[6 * i + j] = dzK[3 * i] * R_ItoC(0, j) + dzK[3 * i + 1] * R_ItoC(1, j) + dzK[3 * i + 2] * R_ItoC(2, j);
      dzli[6 * i + 3 + j] = dzK[3 * i] * SR(0, j) + dzK[3 * i + 1] * SR(1, j) + dzK[3 * i + 2] * SR(2, j);
    }
}
// (2) d(line in the IMU frame)/d(IMU pose) at the first estimates: dli_dI = [A00 A03; A30 0]
__device__ __forceinline__ void line_fej_part(const V3 &nG, const V3 &vG, const M3 &Rf, const V3 &pf, M3 &A00, M3 &A30, M3 &A03) {
  A00 = skew3(mv(Rf, vsub(nG, mv(skew3(pf), vG))));
  A30 = skew3(mv(Rf, vG));
  A03 = mm(Rf, skew3(vG));
}
// x = a * dli_dI for a 2 x 6 matrix a
__device__ __forceinline__ void line_times_dli(const double *a, const M3 &A00, const M3 &A30, const M3 &A03, double *x) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double so = 0, sp = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        so += a[6 * i + q] * A00(q, j);
        sp += a[6 * i + q] * A03(q, j);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        so += a[6 * i + 3 + q] * A30(q, j);
        sp += a[6 * i + 3 + q] * 0.0;
      }
      x[6 * i + j] = so;
      x[6 * i + 3 + j] = sp;
    }
}
// (3) Jacobian in the interpolated pose, noise, whitening: Wm (2 x 2), wli = Wm dzli (2 x 6), WI = wli dli_dI (2 x 6)
__device__ __forceinline__ void line_whiten_part(const JacParams &P, int o, bool at_clone, const double *dzli, const M3 &A00, const M3 &A30, const M3 &A03,
                                                 double *Wm, double *wli, double *WI) {
  double HI[12];
  line_times_dli(dzli, A00, A30, A03, HI);
  double Rn[4] = {P.sigma_pix * P.sigma_pix, 0, 0, P.sigma_pix * P.sigma_pix};
  if (!at_clone && P.use_pol_cov) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        double s = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) s += HI[6 * i + q] * (q < 3 ? P.intr_ori_cov : P.intr_pos_cov) * HI[6 * j + q];
        Rn[2 * i + j] += s;
      }
  } else if (!at_clone && P.use_imu_cov) {
    add_imu_cov(P, o, HI, Rn);
  }
  const double l00 = sqrt(Rn[0]), l10 = Rn[2] / l00, l11 = sqrt(Rn[3] - l10 * l10);
  const double m00 = sqrt(l00), m10 = l10 / m00, m11 = sqrt(l11 - m10 * m10);
#pragma unroll
  for (int col = 0; col < 2; ++col) {
    const double b0 = col == 0 ? 1.0 : 0.0, b1 = col == 1 ? 1.0 : 0.0;
    const double y0 = b0 / m00, y1 = (b1 - m10 * y0) / m11;
    const double x1 = y1 / m11, x0 = (y0 - m10 * x1) / m00;
    Wm[col] = x0;
    Wm[2 + col] = x1;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    wli[j] = Wm[0] * dzli[j] + Wm[1] * dzli[6 + j];
    wli[6 + j] = Wm[2] * dzli[j] + Wm[3] * dzli[6 + j];
  }
  line_times_dli(wli, A00, A30, A03, WI);
}
// (4) Hf = wli * G_to_I, G_to_I = [Re Rsk; 0 Re]
__device__ __forceinline__ void line_write_hf(int c, const double *wli, const M3 &Re, const M3 &Rsk, double *hf, int cstr, int rstr) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double a = 0, b = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a += wli[6 * i + q] * Re(q, j);
        b += wli[6 * i + q] * Rsk(q, j);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a += wli[6 * i + 3 + q] * 0.0;
        b += wli[6 * i + 3 + q] * Re(q, j);
      }
      hf[(size_t)j * cstr + (2 * c + i) * rstr] = a;
      hf[(size_t)(3 + j) * cstr + (2 * c + i) * rstr] = b;
    }
}

// Element (row, col) of a block goes to base[col * cstr + row * rstr], as in jacobian_rows.
__device__ void line_rows(const JacParams &P, int l, int o, int s0, double tm, int c, double *hf, double *hx, double *rs, int cstr,
                          int rstr, const WinTab *tab = nullptr) {
  const V3 nG = ldV(P.line_FinG + 6 * l), vG = ldV(P.line_FinG + 6 * l + 3);
  Interp jac;
  if (tab)  // (window tables: the same values, see build_window_tables)
    interpolate_tab(P, tab[2 * s0 + 1], s0, tm, true, jac);
  else
    interpolate(P, s0, tm, true, true, jac);
  M3 Re;
  V3 pe;
  if (P.res_R) {
    Re = ldM(P.res_R + 9 * o);
    pe = ldV(P.res_p + 3 * o);
  } else {
    Interp est;
    if (tab)
      interpolate_tab(P, tab[2 * s0], s0, tm, false, est);
    else
      interpolate(P, s0, tm, false, false, est);
    Re = est.R;
    pe = est.p;
  }
  double r2[2], dzli[12], Wm[4], wli[12], WI[12];
  M3 Rsk, A00, A30, A03;
  line_est_part(P, nG, vG, P.seg_uv + 4 * o, Re, pe, r2, dzli, Rsk);
  line_fej_part(nG, vG, jac.R, jac.p, A00, A30, A03);
  line_whiten_part(P, o, rows_at_clone(P, tm), dzli, A00, A30, A03, Wm, wli, WI);
  rows_write_res(c, Wm, r2, rs, rstr);
  line_write_hf(c, wli, Re, Rsk, hf, cstr, rstr);
#pragma unroll
  for (int w = 0; w < 4; ++w) rows_write_pose(P.clone_col[s0 + w], c, WI, jac.Ho[w], jac.lam[w], hx, cstr, rstr);  // (written once: plain stores)
  rows_write_dt(P, c, WI, jac.dtj, hx, cstr, rstr);
}

__global__ void __launch_bounds__(64) line_jacobian_kernel(JacParams P) {
  const int l = blockIdx.x;
  const int ld = P.ld, k = P.k;
  double *hf = P.Hf + (size_t)l * 6 * ld, *hx = P.Hx + (size_t)l * k * ld, *rs = P.res + (size_t)l * ld;
  for (int i = threadIdx.x; i < 6 * ld; i += 64) hf[i] = 0.0;
  for (int i = threadIdx.x; i < k * ld; i += 64) hx[i] = 0.0;
  for (int i = threadIdx.x; i < ld; i += 64) rs[i] = 0.0;
  __syncthreads();
  if (P.tri_ok && !candidate_selected(P, l)) {
    if (threadIdx.x == 0) P.rows[l] = 0;
    return;
  }
  const int o0 = P.obs_ptr[l], o1 = P.obs_ptr[l + 1];
  int base = 0;
  for (int ob = o0; ob < o1; ob += 64) {
    const int o = ob + threadIdx.x;
    const bool have = o < o1;
    const double tm = (have ? P.obs_time[o] : 0.0) + P.cam_dt;
    const int s0 = have ? bounding_start(P, tm) : -1;
    const unsigned long long vmask = __ballot(s0 >= 0);
    const int c = base + __popcll(vmask & ((1ull << threadIdx.x) - 1ull));
    base += __popcll(vmask);
    if (s0 >= 0 && 2 * c + 2 <= ld) line_rows(P, l, o, s0, tm, c, hf, hx, rs, ld, 1);
  }
  if (threadIdx.x == 0) P.rows[l] = 2 * base;
}

// line_jacobian_kernel + the null-space projection in one launch for the resident update path (the line twin of
// jacobian_nullspace_kernel): the line's [Hf (6) | Hx | res] block is built row-major in LDS, projected there by six Householder
// reflections and only the projected block goes to global memory.  The covariance gathers of the update ride on it as extra
// workgroups, and workgroup 0 publishes the column map.
__device__ void line_triangulate_one(const JacParams &P, int l, int o0, int o1, const double *cam, const double *imu, const unsigned char *valid,
                                     const float *seg_uvn, const double *anchor, bool has_anchor, double *out, unsigned char &ok);
// tri.on: the line is triangulated first, by this very workgroup, on the state Pt (LineHelper::get_line_features runs on the state
// before the point update, lines_update linearises on the updated one: two views of the same window) — one launch for what were
// line_triangulate_kernel + this one.  Only while the selection loop has no cap to enforce (n_feat <= max_sel): a line is then
// taken on its own merits and needs no count over the lines before it.
struct LineTriStage {
  int on, max_obs;
  double *cam, *imu;      // [n_obs][12] camera / IMU poses of the observations (scratch)
  unsigned char *valid;   // [n_obs]
  double *out_g;          // [L][6]  == P.line_FinG of the Jacobian stage
  unsigned char *ok_g;    // [L]
};
// The rows of every observation of the workgroup's line from the LDS slots, the pieces of line_rows spread over the four waves
// (the line twin of jacobian_rows_split; lane = observation, two barriers):
//   A  wave 0: d(line)/d(pose) at the first estimates | wave 1: residual + its Jacobian in the line at the estimate pose | wave 2: "at a clone"
//   B  wave 0: Jacobian in the interpolated pose, noise, whitening
//   C  wave w: the block of interpolation pose w; + wave 0: residual rows, time offset | wave 1: Hf
__device__ __forceinline__ void line_rows_split(const JacParams &P, const V3 &nG, const V3 &vG, int o0, int n_o, const int *s0_l, const int *slot_l,
                                                const double *tm_l, const float *seg_l, double *pre, double *X, int ncol, int k) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *hf = X, *hx = X + 6, *rs = X + 6 + k;
  for (int ib = 0; ib < n_o; ib += 64) {
    const int i = ib + lane;
    const int c = i < n_o ? slot_l[i] : -1;
    const int o = o0 + i;
    double *pr = pre + (size_t)max(c, 0) * LPRE_STRIDE;
    M3 A00, A30, A03;  // (wave 0, stage A -> B)
    M3 Re, Rsk;        // (wave 1, stage A -> C)
    if (c >= 0) {
      if (wave == 0) {
        line_fej_part(nG, vG, ldM(pr + PRE_R), ldV(pr + PRE_P), A00, A30, A03);
      } else if (wave == 1) {
        double r2[2], dzli[12];
        Re = ldM(pr + PRE_RE);
        line_est_part(P, nG, vG, seg_l + 4 * i, Re, ldV(pr + PRE_PE), r2, dzli, Rsk);
        pr[LPRE_R2] = r2[0], pr[LPRE_R2 + 1] = r2[1];
#pragma unroll
        for (int q = 0; q < 12; ++q) pr[LPRE_DZLI + q] = dzli[q];
      } else if (wave == 2) {
        pr[LPRE_ATC] = rows_at_clone(P, tm_l[i] + P.cam_dt) ? 1.0 : 0.0;
      }
    }
    __syncthreads();
    if (c >= 0 && wave == 0) {
      double dzli[12], Wm[4], wli[12], WI[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) dzli[q] = pr[LPRE_DZLI + q];
      line_whiten_part(P, o, pr[LPRE_ATC] != 0.0, dzli, A00, A30, A03, Wm, wli, WI);
#pragma unroll
      for (int q = 0; q < 4; ++q) pr[LPRE_WM + q] = Wm[q];
#pragma unroll
      for (int q = 0; q < 12; ++q) pr[LPRE_WLI + q] = wli[q], pr[LPRE_WI + q] = WI[q];
    }
    __syncthreads();
    if (c >= 0) {
      double WI[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) WI[q] = pr[LPRE_WI + q];
      rows_write_pose(P.clone_col[s0_l[i] + wave], c, WI, ldM(pr + PRE_HO + 9 * wave), pr[PRE_LAM + wave], hx, 1, ncol);
      if (wave == 0) {
        double Wm[4], r2[2] = {pr[LPRE_R2], pr[LPRE_R2 + 1]}, dtj[6];
#pragma unroll
        for (int q = 0; q < 4; ++q) Wm[q] = pr[LPRE_WM + q];
#pragma unroll
        for (int q = 0; q < 6; ++q) dtj[q] = pr[PRE_DTJ + q];
        rows_write_res(c, Wm, r2, rs, ncol);
        rows_write_dt(P, c, WI, dtj, hx, 1, ncol);
      } else if (wave == 1) {
        double wli[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) wli[q] = pr[LPRE_WLI + q];
        line_write_hf(c, wli, Re, Rsk, hf, 1, ncol);
      }
    }
    __syncthreads();
  }
}

// ov_type::JPLQuat::update on the device, operation for operation what plv_jpl_left_update (init_api.cpp) does on the host
// (REF: open_vins/ov_core/src/types/JPLQuat.h:62-73, utils/quat_ops.h:152-157, 232-252): both sides are built -ffp-contract=off and
// use + - * / sqrt only, so the state a chained launch forms is bit for bit the one the host forms when it applies the same dx.
__device__ __forceinline__ void jpl_left_update_dev(const double *Qin, const double *d, double *M) {
  double a[3] = {0.5 * d[0], 0.5 * d[1], 0.5 * d[2]}, b = 1.0;
  const double nd = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + 1.0);
  a[0] /= nd, a[1] /= nd, a[2] /= nd, b /= nd;
  const double v[3] = {Qin[0], Qin[1], Qin[2]}, w0 = Qin[3];
  double r[4];
  r[0] = b * v[0] - (a[1] * v[2] - a[2] * v[1]) + a[0] * w0;
  r[1] = b * v[1] - (a[2] * v[0] - a[0] * v[2]) + a[1] * w0;
  r[2] = b * v[2] - (a[0] * v[1] - a[1] * v[0]) + a[2] * w0;
  r[3] = -(a[0] * v[0] + a[1] * v[1] + a[2] * v[2]) + b * w0;
  if (r[3] < 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = -r[i];
  }
  const double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  double Q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) Q[i] = r[i] / nr;
  const double x = Q[0], y = Q[1], z = Q[2], w = Q[3], c = 2 * w * w - 1;
  M[0] = c + 2 * x * x, M[1] = 2 * w * z + 2 * x * y, M[2] = -2 * w * y + 2 * x * z;
  M[3] = -2 * w * z + 2 * y * x, M[4] = c + 2 * y * y, M[5] = 2 * w * x + 2 * y * z;
  M[6] = 2 * w * y + 2 * z * x, M[7] = -2 * w * x + 2 * z * y, M[8] = c + 2 * z * z;
}

// line_jacobian_kernel + the null-space projection (+ triangulation in front, + the gate behind) in one launch: the line twin of
// jacobian_nullspace_kernel, same order of work (inputs to LDS, window tables, [wave 0: poses on the state Pt + plane intersection |
// waves 1-3: both interpolations on the state P into the LDS slots], rows over four waves, compact-WY null space with six
// reflectors, gate).  tri.on: the line is triangulated on the state Pt (LineHelper::get_line_features runs on the state before the
// point update, lines_update linearises on the updated one: two views of the same window).  Only while the selection loop has no
// cap to enforce (n_feat <= max_sel): a line is then taken on its own merits and needs no count over the lines before it.
__global__ void __launch_bounds__(256) line_jacobian_nullspace_kernel(JacParams P, int L, GatherArgs g, JacParams Pt, LineTriStage tri, GateStage gate) {
  extern __shared__ double jsm[];
  __shared__ int s_rows, s_base, s_ok;
  __shared__ double s_line[6];
  __shared__ double s_ct[JAC_MAX_WIN / 2 + 3];
  __shared__ int s_ccol[JAC_MAX_WIN / 2 + 3];
  __shared__ double s_cR[(JAC_MAX_WIN / 2 + 3) * 9], s_cp[(JAC_MAX_WIN / 2 + 3) * 3], s_cal[9 + 3 + 8 + 1], s_anchor[3];
  __shared__ unsigned char s_has;
  if (P.chain_dx && *P.chain_status != 0) return;  // (every workgroup, the gathers included: see JacParams::chain_status)
  if ((int)blockIdx.x >= L) {
    gather_cov_block(g, blockIdx.x - L);
    return;
  }
  const int l = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double touch = 0.0;  // (one load per 128-byte line of the input block, see JacParams::in_base; never stored)
  for (int off = threadIdx.x * 128; off < P.in_bytes; off += 256 * 128) touch += *(const volatile double *)(P.in_base + off);
  const int ld = P.ld, k = P.k, ncol = 6 + k + 1, max_obs = tri.max_obs, nwin = max(P.n_clones - 3, 0);
  const FusedLds lay = fused_lds_layout(ld, ncol, 6, (tri.on ? 3 : 2) * nwin, max_obs, tri.on != 0);
  double *X = jsm, *piv = jsm + lay.ns, *pre = jsm + lay.pre, *cam = jsm + lay.cam, *imu = cam + (size_t)max_obs * 12;
  WinTab *tab = reinterpret_cast<WinTab *>(jsm + lay.tab);
  double *tm_l = jsm + lay.idx;
  float *uv_l = reinterpret_cast<float *>(tm_l + max_obs), *uvn_l = uv_l + 4 * max_obs;
  int *s0_l = reinterpret_cast<int *>(uvn_l + 4 * max_obs), *slot_l = s0_l + max_obs, *s0t_l = slot_l + max_obs;
  unsigned char *valid_l = reinterpret_cast<unsigned char *>(s0t_l + max_obs);
  jac_stamp(0);
  const int o0 = P.obs_ptr[l], o1 = P.obs_ptr[l + 1];
  if ((int)threadIdx.x < P.n_clones) s_ct[threadIdx.x] = P.clone_time[threadIdx.x], s_ccol[threadIdx.x] = P.clone_col[threadIdx.x];
  for (int i = threadIdx.x; i < o1 - o0; i += blockDim.x) {
    tm_l[i] = P.obs_time[o0 + i];
#pragma unroll
    for (int q = 0; q < 4; ++q) uv_l[4 * i + q] = P.seg_uv[4 * (o0 + i) + q];
    if (tri.on)
#pragma unroll
      for (int q = 0; q < 4; ++q) uvn_l[4 * i + q] = Pt.seg_uvn[4 * (o0 + i) + q];
  }
  for (int i = threadIdx.x; i < ld * ncol; i += blockDim.x) X[i] = 0.0;
  if (l == 0 && P.cols_out)
    for (int i = threadIdx.x; i < k; i += blockDim.x) P.cols_out[i] = P.cols_in[i];
  __syncthreads();
  P.clone_time = s_ct, P.clone_col = s_ccol, Pt.clone_time = s_ct;  // (the two states share the window: stage_line_inputs)
  if (P.chain_dx && *P.chain_applied != 0) {  // (block-uniform) the state of the linearisation = the staged state (+) dx, see JacParams::chain_dx
    const double *dx = P.chain_dx;
    const int N = P.n_clones, t = threadIdx.x;
    if (t < N) {
      const int id = P.chain_id[t];
#pragma unroll
      for (int q = 0; q < 3; ++q) s_cp[3 * t + q] = id >= 0 ? P.clone_p[3 * t + q] + dx[id + 3 + q] : P.clone_p[3 * t + q];
      if (id >= 0) {
        jpl_left_update_dev(P.chain_q + 4 * t, dx + id, s_cR + 9 * t);
      } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) s_cR[9 * t + q] = P.clone_R[9 * t + q];
      }
    } else if (t == 64) {
      const int id = P.chain_id[N];
      if (id >= 0) {
        jpl_left_update_dev(P.chain_qe, dx + id, s_cal);
#pragma unroll
        for (int q = 0; q < 3; ++q) s_cal[9 + q] = P.p_IinC[q] + dx[id + 3 + q];
      } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) s_cal[q] = P.R_ItoC[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) s_cal[9 + q] = P.p_IinC[q];
      }
    } else if (t == 65) {
      const int id = P.chain_id[N + 1];
#pragma unroll
      for (int q = 0; q < 8; ++q) s_cal[12 + q] = id >= 0 ? P.K[q] + dx[id + q] : P.K[q];
    } else if (t == 66) {
      const int id = P.chain_id[N + 2];
      s_cal[20] = id >= 0 ? P.cam_dt + dx[id] : P.cam_dt;
    }
    __syncthreads();
    P.clone_R = s_cR, P.clone_p = s_cp;
#pragma unroll
    for (int q = 0; q < 9; ++q) P.R_ItoC[q] = s_cal[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) P.p_IinC[q] = s_cal[9 + q];
#pragma unroll
    for (int q = 0; q < 8; ++q) P.K[q] = s_cal[12 + q];
    P.cam_dt = s_cal[20];
  }
  if (wave == 0) {
    const int base = assign_row_slots(P, o0, o1, ld, tm_l, s0_l, slot_l);
    if (lane == 0) s_base = base;
    jac_stamp(15);
  } else if (wave == 1 && tri.on) {
    for (int i = lane; i < o1 - o0; i += 64) s0t_l[i] = bounding_start(Pt, tm_l[i] + Pt.cam_dt);
  }
  build_window_tables(P, tab, tri.on ? Pt.clone_R : nullptr, tri.on ? Pt.clone_p : nullptr);  // (ends with a barrier: also orders the zero fill and the slots before what follows)
  jac_stamp(2);
  if (wave == 0) {
    if (tri.on) {
      // poses of the observations on the state of the triangulation (CamHelper::get_imu_poses / get_cam_poses, as campose_one)
      for (int i = lane; i < o1 - o0; i += 64) {
        const int s0 = s0t_l[i];
        valid_l[i] = s0 >= 0;
        if (s0 < 0) continue;
        M3 R_GtoI;
        V3 p_IinG;
        est_pose_tab(Pt, tab[2 * nwin + s0], o0 + i, s0, tm_l[i] + Pt.cam_dt, R_GtoI, p_IinG);
#pragma unroll
        for (int q = 0; q < 9; ++q) imu[12 * i + q] = R_GtoI.m[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) imu[12 * i + 9 + q] = p_IinG[q];
        const M3 R_GtoC = mm(ldM(Pt.R_ItoC), R_GtoI);
        const V3 p_CinG = vsub(p_IinG, mv(tp(R_GtoC), ldV(Pt.p_IinC)));
#pragma unroll
        for (int q = 0; q < 9; ++q) cam[12 * i + q] = R_GtoC.m[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) cam[12 * i + 9 + q] = p_CinG[q];
      }
      if (Pt.anc_ptr && lane == 0) {  // the line's anchor: its first point that is triangulated (JacParams::anc_ptr)
        unsigned char has = 0;
        for (int c = Pt.anc_ptr[l]; c < Pt.anc_ptr[l + 1] && !has; ++c) {
          const int pf = Pt.anc_f[c];
          if (pf >= 0 && Pt.anc_tri_ok[pf]) {
            s_anchor[0] = Pt.anc_tri_p[3 * pf], s_anchor[1] = Pt.anc_tri_p[3 * pf + 1], s_anchor[2] = Pt.anc_tri_p[3 * pf + 2];
            has = 1;
          } else if (Pt.anc_has_old[c]) {
            s_anchor[0] = Pt.anc_old[3 * c], s_anchor[1] = Pt.anc_old[3 * c + 1], s_anchor[2] = Pt.anc_old[3 * c + 2];
            has = 1;
          }
        }
        s_has = has;
      }
      tri_wave_sync();
      jac_stamp(10);
      double out_l[6] = {0, 0, 0, 0, 0, 0};
      unsigned char ok_l = 0;
      const bool has_anchor = Pt.anc_ptr ? s_has != 0 : (Pt.has_pt && Pt.has_pt[l]);
      line_triangulate_one(Pt, l, 0, o1 - o0, cam, imu, valid_l, uvn_l, Pt.anc_ptr ? s_anchor : Pt.anchor_pt + 3 * l, has_anchor, out_l, ok_l);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) tri.out_g[6 * l + i] = out_l[i], s_line[i] = out_l[i];
        tri.ok_g[l] = ok_l;
        s_ok = ok_l;
      }
    }
  } else {
    for (int i = threadIdx.x - 64; i < o1 - o0; i += 192) {
      const int c = slot_l[i];
      if (c < 0) continue;
      const int s0 = s0_l[i];
      double *pr = pre + (size_t)c * LPRE_STRIDE;
      Interp jac;
      interpolate_tab(P, tab[2 * s0 + 1], s0, tm_l[i] + P.cam_dt, true, jac);
      pre_store_jac(pr, jac);
      M3 Re;
      V3 pe;
      est_pose_tab(P, tab[2 * s0], o0 + i, s0, tm_l[i] + P.cam_dt, Re, pe);
#pragma unroll
      for (int q = 0; q < 9; ++q) pr[PRE_RE + q] = Re.m[q];
#pragma unroll
      for (int q = 0; q < 3; ++q) pr[PRE_PE + q] = pe[q];
    }
    if (gate.on) gate_stage_prior(gate, reinterpret_cast<double *>(reinterpret_cast<char *>(jsm) + gate.ps_off), P.cols_in, k, threadIdx.x - 64, 192);
  }
  __threadfence_block();
  __syncthreads();
  jac_stamp(1);
  bool selected;
  V3 nG, vG;
  if (tri.on) {  // (the workgroup's own result from LDS)
    selected = P.sel_flags[l] && s_ok;
    nG = ldV(s_line), vG = ldV(s_line + 3);
  } else {
    selected = !P.tri_ok || candidate_selected(P, l);
    nG = ldV(P.line_FinG + 6 * l), vG = ldV(P.line_FinG + 6 * l + 3);
  }
  if (selected) {  // (block-uniform)
    line_rows_split(P, nG, vG, o0, o1 - o0, s0_l, slot_l, tm_l, uv_l, pre, X, ncol, k);
    if (threadIdx.x == 0) {
      s_rows = min(2 * s_base, ld & ~1);
      P.rows[l] = 2 * s_base;
    }
  } else if (threadIdx.x == 0) {
    s_rows = 0;
    P.rows[l] = 0;
  }
  __syncthreads();
  jac_stamp(3);
  const int rows = s_rows;
  GateLds &gl = *reinterpret_cast<GateLds *>(reinterpret_cast<char *>(jsm) + gate.lds_off);
  const double *gPs = reinterpret_cast<const double *>(reinterpret_cast<char *>(jsm) + gate.ps_off);
  double *gT = reinterpret_cast<double *>(reinterpret_cast<char *>(jsm) + gate.t_off);
  if ((tri.on || P.tri_ok) && rows == 0) {  // (an unselected pool line: empty system, nothing reads its block)
    if (gate.on) gate_tail(gate, gl, gPs, gT, l, X, ncol, 6, 0, 0, k, P.cols_in);
    return;
  }
  const int shift = rows > 6 ? 6 : 0;  // (a block with no more rows than Hf has columns is left as it is, as nullspace_kernel does)
  if (shift) nullspace_householder_wy<6>(X, piv, rows, ncol);
  jac_stamp(4);
  double *hf = P.Hf + (size_t)l * 6 * ld, *hx = P.Hx + (size_t)l * k * ld, *rs = P.res + (size_t)l * ld;
  for (int j = gate.on ? ncol : (int)threadIdx.x; j < ncol; j += blockDim.x) {  // (not written when the gate follows in this launch)
    double *dst = j < 6 ? hf + j * ld : (j < 6 + k ? hx + (size_t)(j - 6) * ld : rs);
    const int off = j < 6 ? 0 : shift;
    for (int i0 = 0; i0 < ld; i0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = i0 + u + off;
        v[u] = r < ld ? X[r * ncol + j] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < ld) dst[i0 + u] = v[u];
    }
  }
  jac_stamp(5);
  if (gate.on) gate_tail(gate, gl, gPs, gT, l, X, ncol, 6, shift, min(rows, ld), k, P.cols_in);
  if (touch == 1.2345678e300) P.rows[l] = -1;  // (never: keeps the touch loads)
  jac_stamp(9);
}

__device__ void line_triangulate_one(const JacParams &P, int l, int o0, int o1, const double *cam, const double *imu, const unsigned char *valid,
                                     const float *seg_uvn, const double *anchor, bool has_anchor, double *out, unsigned char &ok);
// a27: LineHelper::line_triangulation   REF: LineHelper.cpp:202-293, 372-495, 615-650.  One wave per line: the lanes first
// compute the camera / IMU poses of the line's observations (one each), then every lane runs the (short, serial) plane
// intersection on them and lane 0 stores the result.
__global__ void __launch_bounds__(64) line_triangulate_kernel(JacParams P, double *cam, double *imu, unsigned char *valid, double *out_g,
                                                              unsigned char *ok_g) {
  const int l = blockIdx.x;
  const int o0 = P.obs_ptr[l], o1 = P.obs_ptr[l + 1];
  for (int o = o0 + (int)threadIdx.x; o < o1; o += 64) campose_one(P, o, cam, valid, imu);
  __threadfence_block();
  __syncthreads();
  double out_l[6] = {0, 0, 0, 0, 0, 0};
  unsigned char ok_l = 0;
  line_triangulate_one(P, l, o0, o1, cam, imu, valid, P.seg_uvn, P.anchor_pt + 3 * l, P.has_pt && P.has_pt[l], out_l, ok_l);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) out_g[6 * l + i] = out_l[i];
    ok_g[l] = ok_l;
  }
}
// Observations o0 .. o1 - 1 index cam / imu / valid / seg_uvn (global arrays, or the workgroup's LDS copies with o0 = 0).
__device__ void line_triangulate_one(const JacParams &P, int l, int o0, int o1, const double *cam, const double *imu, const unsigned char *valid,
                                     const float *seg_uvn, const double *anchor, bool has_anchor, double *out /*[6], zero on entry*/, unsigned char &ok) {
  int first = -1, nvalid = 0;
  for (int o = o0; o < o1; ++o)
    if (valid[o]) {
      if (first < 0) first = o;
      ++nvalid;
    }
  if (nvalid < 2) return;
  const int D = P.lineD ? P.lineD[l] : 0;
  if (D > 0 && has_anchor) {
    const M3 R = ldM(imu + 12 * first);
    const V3 e{{D == 1 ? 1.0 : 0.0, D == 2 ? 1.0 : 0.0, D == 3 ? 1.0 : 0.0}};
    const V3 dir = mv(tp(R), e);
    const V3 mom = cross3(ldV(anchor), dir);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      out[i] = mom[i];
      out[3 + i] = dir[i];
    }
    ok = 1;
    return;
  }
  const M3 R0 = ldM(cam + 12 * first);
  const V3 p0 = ldV(cam + 12 * first + 9);
  const float *u0 = seg_uvn + 4 * first;
  const V3 p11{{(double)u0[0], (double)u0[1], 1.0}}, p12{{(double)u0[2], (double)u0[3], 1.0}};
  // plane through (a, b, c3): [ (a-c3) x (b-c3), -c3 . (a x b) ]
  const V3 n0 = cross3(p11, p12);
  const double pl0[4] = {n0[0], n0[1], n0[2], 0.0};  // the first camera centre is the origin of its own frame
  V3 dsum{{0, 0, 0}}, nsum{{0, 0, 0}};
  double dnorm = 0;
  int cnt = 0;
  for (int o = first + 1; o < o1; ++o) {
    if (!valid[o]) continue;
    const M3 Ri = ldM(cam + 12 * o);
    const V3 pi = ldV(cam + 12 * o + 9);
    const M3 R0i = mm(Ri, tp(R0));
    const V3 pi0 = mv(R0, vsub(pi, p0));
    const float *um = seg_uvn + 4 * o;
    V3 p31{{(double)um[0], (double)um[1], 1.0}}, p32{{(double)um[2], (double)um[3], 1.0}};
    p31 = vadd(mv(tp(R0i), p31), pi0);
    p32 = vadd(mv(tp(R0i), p32), pi0);
    const V3 nn = cross3(vsub(p31, pi0), vsub(p32, pi0));
    const double pl1[4] = {nn[0], nn[1], nn[2], -dot3(pi0, cross3(p31, p32))};
    V3 n1{{pl0[0], pl0[1], pl0[2]}}, n2{{pl1[0], pl1[1], pl1[2]}};
    n1 = vsc(n1, 1 / vnorm(n1));
    n2 = vsc(n2, 1 / vnorm(n2));
    const double cth = dot3(n1, n2) / (vnorm(n1) * vnorm(n2));
    if (fabs(cth) >= 0.99) continue;
#define PLV_DP(i, j) (pl0[i] * pl1[j] - pl1[i] * pl0[j])
    const V3 head{{PLV_DP(0, 3), PLV_DP(1, 3), PLV_DP(2, 3)}}, tail{{-PLV_DP(1, 2), PLV_DP(0, 2), -PLV_DP(0, 1)}};
#undef PLV_DP
    dsum = vadd(dsum, tail);
    nsum = vadd(nsum, head);
    dnorm += vnorm(tail);
    ++cnt;
  }
  if (cnt == 0) return;
  const V3 rhead = vsc(dsum, 1 / dnorm), rtail = vsc(nsum, 1.0 / cnt);
  const M3 R0t = tp(R0);
  const V3 vW = mv(R0t, rhead);
  const V3 nW = vadd(mv(R0t, rtail), mv(skew3(p0), mv(R0t, rhead)));
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    out[i] = nW[i];
    out[3 + i] = vW[i];
  }
  ok = 1;
}

// PLV_KNOB_KERNEL_STAMPS: per-phase cycle stamps of the projected Jacobian launches (see jac_stamp): offsets from the workgroup's start,
// mean and max over the workgroups that went all the way (entries the selection took), printed when the library unloads.
#define TRY_STAMP(x)          \
  do {                        \
    const int _rc = (x);      \
    if (_rc != PLV_OK) return _rc; \
  } while (0)
struct JacStampHost {
  bool on = plv::knob(plv::PLV_KNOB_KERNEL_STAMPS);
  long long *d = nullptr;
  int cap = 0;
  struct Acc {
    const char *name;
    double sum[JAC_NSTAMP] = {}, mx[JAC_NSTAMP] = {}, slowest = 0;
    long n[JAC_NSTAMP] = {}, launches = 0;
  } acc[2] = {{"tri_jacobian_nullspace_kernel"}, {"line_tri_jacobian_nullspace_kernel"}};
  int arm(hipStream_t s, int blocks) {
    if (blocks > cap) {
      if (d) (void)hipFree(d);
      cap = blocks + 64;
      PLV_HIP_CHECK(hipMalloc(&d, (size_t)cap * JAC_NSTAMP * 8));
      PLV_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_jac_stamps), &d, sizeof(d)));
    }
    PLV_HIP_CHECK(hipMemsetAsync(d, 0, (size_t)blocks * JAC_NSTAMP * 8, s));
    return PLV_OK;
  }
  int collect(hipStream_t s, int blocks, int which) {
    PLV_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<long long> h((size_t)blocks * JAC_NSTAMP);
    PLV_HIP_CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    Acc &a = acc[which];
    double slow = 0;
    for (int b = 0; b < blocks; ++b) {
      const long long *t = &h[(size_t)b * JAC_NSTAMP];
      if (!t[0] || !t[9]) continue;
      slow = std::max(slow, (double)(t[9] - t[0]));
      if (!t[4]) continue;  // (not taken: no rows)
      for (int i = 1; i < JAC_NSTAMP; ++i)
        if (t[i]) a.sum[i] += (double)(t[i] - t[0]), a.mx[i] = std::max(a.mx[i], (double)(t[i] - t[0])), ++a.n[i];
    }
    a.slowest += slow;
    ++a.launches;
    return PLV_OK;
  }
  ~JacStampHost() {
    if (!on) return;
    static const char *label[JAC_NSTAMP] = {"start", "triangulated", "window tables", "rows built", "null space", "block written", "gate: T", "gate: S",
                                            "gate: factor", "end", "tri: camera poses", "tri: linear solve", "tri: refined", "gate: map staged", "gate: T first column", "row slots",
                                            "LM pass 1", "LM pass 2", "LM pass 3", "LM pass 4", "LM pass 5", "LM pass 6", "LM pass 7", "LM pass 8", "LM pass 9", "LM pass 10", "LM pass 11", "LM pass 12+",
                                            "", "", "", ""};
    static const int order[] = {15, 2, 10, 11, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 12, 1, 3, 4, 5, 13, 14, 6, 7, 8, 9};
    for (const Acc &a : acc) {
      if (!a.launches) continue;
      fprintf(stderr, "[plv stamps] %s: %ld launches, slowest workgroup %.0f cycles on average\n", a.name, a.launches, a.slowest / (double)a.launches);
      for (int i : order)
        if (a.n[i]) fprintf(stderr, "[plv stamps]   %-20s @ %8.0f mean  %8.0f max  (%ld workgroups)\n", label[i], a.sum[i] / (double)a.n[i], a.mx[i], a.n[i]);
    }
  }
};
static JacStampHost &jac_stamps() {
  static JacStampHost h;
  return h;
}

int launch_line_jacobians(plv_ctx *ctx, const JacParams &P) {
  ProfScope ps(ctx->prof, "line_jacobian_kernel", ctx->stream);
  hipLaunchKernelGGL(line_jacobian_kernel, dim3(P.n_feat), dim3(64), 0, ctx->stream, P);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_line_jacobians_projected(plv_ctx *ctx, const JacParams &P, const GatherArgs *g, int gather_blocks, const JacParams *Pt, double *d_cam,
                                    double *d_imu, unsigned char *d_valid, double *d_lines, unsigned char *d_ok, int max_obs) {
  ProfScope ps(ctx->prof, Pt ? "line_tri_jacobian_nullspace_kernel" : "line_jacobian_nullspace_kernel", ctx->stream);
  if (2 * (P.n_clones - 3) > JAC_MAX_WIN) {
    set_last_error("line jacobians: %d clones exceed the window table (%d)", P.n_clones, JAC_MAX_WIN / 2 + 3);
    return PLV_E_CAPACITY;
  }
  max_obs = std::max(max_obs, 1);
  const FusedLds lay = fused_lds_layout(P.ld, 6 + P.k + 1, 6, (Pt ? 3 : 2) * std::max(P.n_clones - 3, 0), max_obs, Pt != nullptr);
  size_t shm = (size_t)lay.end * sizeof(double);
  const size_t lds_cap = 160 * 1024 - static_smem_bytes((const void *)line_jacobian_nullspace_kernel);
  if (shm > lds_cap) {
    set_last_error("line jacobians: block of %zu bytes exceeds LDS", shm);
    return PLV_E_CAPACITY;
  }
  GatherArgs none{};
  LineTriStage tri{Pt ? 1 : 0, max_obs, d_cam, d_imu, d_valid, d_lines, d_ok};
  GateStage gate = ctx->gate_stage;
  ctx->gate_stage.on = 0;  // (one launch takes it)
  const size_t gate_off = ((size_t)lay.tab * sizeof(double) + 63) & ~(size_t)63;  // (overlays tables and slots: fused_lds_layout)
  const size_t t_off = (gate_off + sizeof(GateLds) + 63) & ~(size_t)63;
  const size_t ps_off = (std::max(shm, t_off + (size_t)GATE_MMAX * GATE_TLD * sizeof(double)) + 63) & ~(size_t)63;
  const size_t gate_end = ps_off + (size_t)gate_ps_doubles(P.k) * sizeof(double);
  if (P.k > GATE_KMAX || gate_end > lds_cap) gate.on = 0;
  if (gate.on) {
    gate.lds_off = (int)gate_off, gate.ps_off = (int)ps_off, gate.t_off = (int)t_off;
    shm = gate_end;
  }
  ctx->gate_stage_taken = gate.on != 0;
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)line_jacobian_nullspace_kernel, (int)shm));
  if (jac_stamps().on) TRY_STAMP(jac_stamps().arm(ctx->stream, P.n_feat));
  hipLaunchKernelGGL(line_jacobian_nullspace_kernel, dim3(P.n_feat + (g ? gather_blocks : 0)), dim3(256), shm, ctx->stream, P, P.n_feat,
                     g ? *g : none, Pt ? *Pt : P, tri, gate);
  PLV_HIP_CHECK(hipGetLastError());
  if (jac_stamps().on) TRY_STAMP(jac_stamps().collect(ctx->stream, P.n_feat, 1));
  return PLV_OK;
}

int launch_triangulate_lines(plv_ctx *ctx, const JacParams &P, double *d_poses, double *d_imu, unsigned char *d_valid,
                             double *d_lines, unsigned char *d_ok) {
  {
    ProfScope ps(ctx->prof, "line_triangulate_kernel", ctx->stream);
    hipLaunchKernelGGL(line_triangulate_kernel, dim3(P.n_feat), dim3(64), 0, ctx->stream, P, d_poses, d_imu, d_valid, d_lines, d_ok);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_triangulate(plv_ctx *ctx, const JacParams &P, double *d_poses, unsigned char *d_valid, const float *d_uvn,
                       const plv_tri_options &opt, double *d_p, unsigned char *d_ok, double *d_err, int max_obs) {
  {
    ProfScope ps(ctx->prof, "triangulate_kernel", ctx->stream);
    const size_t shm = (size_t)std::max(max_obs, 1) * (sizeof(TriObs) + TRI_TERMS * 8 + 4) + 16;
    if (shm > 150 * 1024) {
      set_last_error("triangulation: a track of %d observations exceeds LDS", max_obs);
      return PLV_E_CAPACITY;
    }
    PLV_HIP_CHECK(ensure_dyn_smem((const void *)triangulate_kernel, (int)shm));
    hipLaunchKernelGGL(triangulate_kernel, dim3(P.n_feat), dim3(64), shm, ctx->stream, P, d_poses, d_valid, d_uvn, opt, d_p, d_ok, d_err,
                       std::max(max_obs, 1));
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_jacobians_projected(plv_ctx *ctx, const JacParams &P, const GatherArgs *g, int gather_blocks, const plv_tri_options *tri_opt,
                               double *d_poses, unsigned char *d_valid, const float *d_uvn, double *d_p, unsigned char *d_ok, double *d_err, int max_obs) {
  ProfScope ps(ctx->prof, tri_opt ? "tri_jacobian_nullspace_kernel" : "jacobian_nullspace_kernel", ctx->stream);
  if (2 * (P.n_clones - 3) > JAC_MAX_WIN) {
    set_last_error("jacobians: %d clones exceed the window table (%d)", P.n_clones, JAC_MAX_WIN / 2 + 3);
    return PLV_E_CAPACITY;
  }
  max_obs = std::max(max_obs, 1);
  const FusedLds lay = fused_lds_layout(P.ld, 3 + P.k + 1, 3, 2 * std::max(P.n_clones - 3, 0), max_obs, tri_opt != nullptr);
  size_t shm = (size_t)lay.end * sizeof(double);
  const size_t lds_cap = 160 * 1024 - static_smem_bytes((const void *)jacobian_nullspace_kernel);
  if (shm > lds_cap) {
    set_last_error("jacobians: feature block of %zu bytes exceeds LDS", shm);
    return PLV_E_CAPACITY;
  }
  GatherArgs none{};
  PointTriStage tri{};
  tri.max_obs = max_obs;
  if (tri_opt) tri = PointTriStage{1, max_obs, d_poses, d_valid, d_uvn, *tri_opt, d_p, d_ok, d_err};
  GateStage gate = ctx->gate_stage;
  ctx->gate_stage.on = 0;  // (one launch takes it)
  // the gate's LDS block overlays tables, scratch and slots (fused_lds_layout): taken where it fits behind the entry's block
  const size_t gate_off = ((size_t)lay.tab * sizeof(double) + 63) & ~(size_t)63;
  // (GateLds and T overlay tables, scratch and slots: dead when the gate starts; the prior block, staged early, sits behind them)
  const size_t t_off = (gate_off + sizeof(GateLds) + 63) & ~(size_t)63;  // (T too: it is written when the scratch is dead)
  const size_t ps_off = (std::max(shm, t_off + (size_t)GATE_MMAX * GATE_TLD * sizeof(double)) + 63) & ~(size_t)63;
  const size_t gate_end = ps_off + (size_t)gate_ps_doubles(P.k) * sizeof(double);
  if (P.k > GATE_KMAX || gate_end > lds_cap) gate.on = 0;  // (rows: plv_update_gate_prepare)
  if (gate.on) {
    gate.lds_off = (int)gate_off, gate.ps_off = (int)ps_off, gate.t_off = (int)t_off;
    shm = gate_end;
  }
  ctx->gate_stage_taken = gate.on != 0;
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)jacobian_nullspace_kernel, (int)shm));
  if (jac_stamps().on) TRY_STAMP(jac_stamps().arm(ctx->stream, P.n_feat));
  // (a speculative batch: one workgroup per pool entry, at most spec_grid(max_sel) of them — spec_select_kernel's list — not one per candidate)
  const int wg_feat = P.spec_order ? std::max(1, std::min(P.n_feat, spec_grid(P.max_sel))) : P.n_feat;
  hipLaunchKernelGGL(jacobian_nullspace_kernel, dim3(wg_feat + (g ? gather_blocks : 0)), dim3(256), shm, ctx->stream, P, wg_feat,
                     g ? *g : none, tri, gate);
  PLV_HIP_CHECK(hipGetLastError());
  if (jac_stamps().on) TRY_STAMP(jac_stamps().collect(ctx->stream, P.n_feat, 0));
  return PLV_OK;
}

// (see SpecSelectArgs, jacobian_kernels.hpp) one workgroup: a pass over the candidates for membership and the pool's size, a second
// one that writes the ranges, the flags and the survivors' new observation
__global__ void __launch_bounds__(1024) spec_select_kernel(SpecSelectArgs A) {
  __shared__ int s_wave[16], s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  // pass 1: membership, and the pool's candidates listed in batch order (a prefix count over the candidates, 1024 at a time)
  for (int f0 = 0; f0 < A.F; f0 += blockDim.x) {
    const int f = f0 + threadIdx.x;
    bool in_pool = false;
    if (f < A.F) {
      const int li = A.li[f], m = A.meta[f];
      bool survived = false;
      if (li >= 0 && li < A.n_flow && A.flow_mask[li]) {
        const float x = A.flow_p1[2 * li], y = A.flow_p1[2 * li + 1];
        survived = !(x < 0 || y < 0 || (int)x >= A.W || (int)y >= A.H);  // REF TrackKLT.cpp:161-163
      }
      const int n_old = A.obs_ptr[f + 1] - A.obs_ptr[f] - (li >= 0 ? 1 : 0);  // (a tracked point's range ends with the slot of this frame)
      const int n_use = n_old + ((survived && (m & 2)) ? 1 : 0);
      in_pool = ((m & 1) || !((m & 8) || survived)) && n_use >= 2;
    }
    const unsigned long long b = __ballot(in_pool);
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    const int rank = before + __popcll(b & ((1ull << lane) - 1ull));
    if (in_pool && rank < A.grid) A.order[rank] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  const int count = s_base;
  const bool over = count > A.grid;
  if (threadIdx.x == 0) A.words[0] = count, A.words[1] = over ? 1 : 0, A.words[2] = over ? 0 : count, A.words[3] = (!over && count > A.max_sel) ? 1 : 0, A.words[4] = 0;
  // what workgroup 0 of the Jacobian launch does for the launch as a whole (that launch may not hold candidate 0 at all)
  if (threadIdx.x == 0 && A.zero_word) *A.zero_word = 0;
  if (A.cols_out)
    for (int i = threadIdx.x; i < A.k; i += blockDim.x) A.cols_out[i] = A.cols_in[i];
  // pass 2: the ranges, the flags, the survivors' new observation; empty outputs for what the Jacobian launch will not work on
  for (int f = threadIdx.x; f < A.F; f += blockDim.x) {
    const int li = A.li[f], m = A.meta[f], o0 = A.obs_ptr[f], o1 = A.obs_ptr[f + 1];
    bool survived = false;
    float x = 0.f, y = 0.f;
    if (li >= 0 && li < A.n_flow && A.flow_mask[li]) {
      x = A.flow_p1[2 * li], y = A.flow_p1[2 * li + 1];
      survived = !(x < 0 || y < 0 || (int)x >= A.W || (int)y >= A.H);
    }
    const int n_old = o1 - o0 - (li >= 0 ? 1 : 0);
    const bool with_new = survived && (m & 2);
    const int n_use = n_old + (with_new ? 1 : 0);
    const bool in_pool = ((m & 1) || !((m & 8) || survived)) && n_use >= 2;
    const int n_valid = (int)A.prevalid[f] + ((with_new && (m & 4)) ? 1 : 0);
    A.member[f] = in_pool ? 1 : 0;
    A.sel_flags[f] = (in_pool && !over && n_valid >= 2) ? 1 : 0;
    A.obs_end[f] = (in_pool && !over) ? o0 + n_use : o0;
    if (in_pool && with_new) {
      const int o = o0 + n_old;
      A.obs_uv[2 * o] = x, A.obs_uv[2 * o + 1] = y;
      A.obs_uvn[2 * o] = A.flow_n1[2 * li], A.obs_uvn[2 * o + 1] = A.flow_n1[2 * li + 1];
    }
    if (!in_pool || over) {
      A.rows_out[f] = 0;
      A.tri_p[3 * f] = A.tri_p[3 * f + 1] = A.tri_p[3 * f + 2] = 0.0;
      A.tri_err[f] = 0.0;
      A.tri_ok[f] = 0;
      if (A.chi2) A.chi2[f] = NAN;
      if (A.accepted) A.accepted[f] = 0;
      if (A.acc_rows) A.acc_rows[f] = 0;
    }
  }
}
int launch_spec_select(plv_ctx *ctx, const SpecSelectArgs &A) {
  ProfScope ps(ctx->prof, "spec_select_kernel", ctx->stream);
  hipLaunchKernelGGL(spec_select_kernel, dim3(1), dim3(1024), 0, ctx->stream, A);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_jacobians(plv_ctx *ctx, const JacParams &P) {
  ProfScope ps(ctx->prof, "jacobian_kernel", ctx->stream);
  if (2 * (P.n_clones - 3) > JAC_MAX_WIN) {
    set_last_error("jacobians: %d clones exceed the window table (%d)", P.n_clones, JAC_MAX_WIN / 2 + 3);
    return PLV_E_CAPACITY;
  }
  hipLaunchKernelGGL(jacobian_kernel, dim3(P.n_feat), dim3(128), 0, ctx->stream, P);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

}  // namespace plv
