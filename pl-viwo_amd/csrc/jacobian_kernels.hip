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

__device__ void build_window_tables(const JacParams &P, WinTab *tab) {
  const int nwin = max(P.n_clones - 3, 0);
  for (int idx = threadIdx.x; idx < nwin * 2 * 3; idx += blockDim.x) {
    const int w = idx % 3, e = idx / 3, s0 = e >> 1, fej = e & 1;
    const double *Rs = fej ? P.clone_R_fej : P.clone_R, *ps = fej ? P.clone_p_fej : P.clone_p;
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
  for (int e = threadIdx.x; e < nwin * 2; e += blockDim.x) {
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
__device__ void triangulate_feature(const JacParams &P, int f, double *poses, unsigned char *valid, const float *__restrict__ uvn,
                                    const plv_tri_options &opt, double *__restrict__ p_out, unsigned char *__restrict__ ok_out,
                                    double *__restrict__ err_out, int max_obs, double *tri_smem, double *tot);
// tri.on: the workgroup's first wave triangulates the feature before the Jacobians are built (what triangulate_kernel did in a launch
// of its own) — while the selection loop has no cap to enforce (n_feat <= max_sel) a candidate is taken on its own verdict.
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
__global__ void __launch_bounds__(256) jacobian_nullspace_kernel(JacParams P, int F, GatherArgs g, PointTriStage tri, GateStage gate) {
  extern __shared__ double jsm[];  // X [ld][ncol] | piv [ld] | triangulation scratch
  __shared__ WinTab tab[JAC_MAX_WIN];
  __shared__ int s_rows;
  __shared__ double tri_tot[10];
  if ((int)blockIdx.x >= F) {
    gather_cov_block(g, blockIdx.x - F);
    return;
  }
  const int f = blockIdx.x;
  const int ld = P.ld, k = P.k, ncol = 3 + k + 1;
  double *X = jsm, *piv = jsm + ld * ncol;
  for (int i = threadIdx.x; i < ld * ncol; i += blockDim.x) X[i] = 0.0;
  if (f == 0 && P.cols_out)
    for (int i = threadIdx.x; i < k; i += blockDim.x) P.cols_out[i] = P.cols_in[i];
  bool selected;
  if (tri.on) {
    if (threadIdx.x < 64) triangulate_feature(P, f, tri.poses, tri.valid, tri.uvn, tri.opt, tri.p_out, tri.ok_out, tri.err_out, tri.max_obs, piv + ld, tri_tot);
    __threadfence_block();
    __syncthreads();
    selected = P.sel_flags[f] && tri.ok_out[f] && tri.err_out[f] < 3.0;
  } else {
    selected = !P.tri_ok || candidate_selected(P, f);
  }
  build_window_tables(P, tab);  // (ends with a barrier: also orders the zero fill before the row writes)
  if (!selected) {
    if (threadIdx.x == 0) {
      s_rows = 0;
      P.rows[f] = 0;
    }
  } else if (threadIdx.x < 64) {
    const int o0 = P.obs_ptr[f], o1 = P.obs_ptr[f + 1];
    int base = 0;
    for (int ob = o0; ob < o1; ob += 64) {
      const int o = ob + threadIdx.x;
      const bool have = o < o1;
      const double tm = (have ? P.obs_time[o] : 0.0) + P.cam_dt;
      const int s0 = have ? bounding_start(P, tm) : -1;
      const unsigned long long vmask = __ballot(s0 >= 0);
      const int c = base + __popcll(vmask & ((1ull << threadIdx.x) - 1ull));
      base += __popcll(vmask);
      if (s0 >= 0 && 2 * c + 2 <= ld) jacobian_rows(P, tab, f, o, s0, tm, c, X, X + 3, X + 3 + k, 1, ncol);
    }
    if (threadIdx.x == 0) {
      s_rows = min(2 * base, ld & ~1);
      P.rows[f] = 2 * base;
    }
  }
  __syncthreads();
  const int rows = s_rows;
  if ((tri.on || P.tri_ok) && rows == 0) {  // one-submission update: a candidate the selection did not take is an empty system (rows[f] = 0) that
                                            // nothing reads — its padded block is not written (rocprofv3, round 2: 1.6 MB per launch, mostly these)
    if (gate.on) gate_tail(gate, *reinterpret_cast<GateLds *>(reinterpret_cast<char *>(jsm) + gate.lds_off), f, X, ncol, 3, 0, 0, k, P.cols_in);  // (verdict "not accepted" + its share of the probe block)
    return;
  }
  const int shift = rows > 3 ? 3 : 0;  // (a block with no more rows than Hf has columns is left as it is, as nullspace_kernel does)
  if (shift) nullspace_householder(X, piv, rows, ncol, 3);
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
  if (gate.on) {  // (X is only read from here on: no barrier needed between the write-out and the gate)
    gate_tail(gate, *reinterpret_cast<GateLds *>(reinterpret_cast<char *>(jsm) + gate.lds_off), f, X, ncol, 3, shift, min(rows, ld), k, P.cols_in);
  }
}

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
  const M3 R_ItoC = ldM(P.R_ItoC);
  const V3 p_IinC = ldV(P.p_IinC);
  const double *K = P.K;
  const V3 pf = ldV(P.p_FinG + 3 * f), pf_fej = ldV(P.p_FinG_fej + 3 * f);

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
  // ---- residual (estimate pose); CamBase::distort_d rounds through float both ways
  V3 p_FinI = mv(R_GtoI, vsub(pf, p_IinG));
  V3 p_FinC = vadd(mv(R_ItoC, p_FinI), p_IinC);
  const double un = p_FinC[0] / p_FinC[2], vn = p_FinC[1] / p_FinC[2];
  double r2[2];
  {
    const double x = (double)(float)un, y = (double)(float)vn;
    const double r = sqrt(x * x + y * y), r_2 = r * r, r_4 = r_2 * r_2;
    const double x1 = x * (1 + K[4] * r_2 + K[5] * r_4) + 2 * K[6] * x * y + K[7] * (r_2 + 2 * x * x);
    const double y1 = y * (1 + K[4] * r_2 + K[5] * r_4) + K[6] * (r_2 + 2 * y * y) + 2 * K[7] * x * y;
    r2[0] = (double)P.obs_uv[2 * o] - (double)(float)(K[0] * x1 + K[2]);
    r2[1] = (double)P.obs_uv[2 * o + 1] - (double)(float)(K[1] * y1 + K[3]);
  }
  // ---- distortion Jacobians at the estimate's normalised coordinates
  double dzn[4], dzeta[16];
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
  // ---- chain at the first estimates
  R_GtoI = jac.R;
  p_IinG = jac.p;
  p_FinI = mv(R_GtoI, vsub(pf_fej, p_IinG));
  p_FinC = vadd(mv(R_ItoC, p_FinI), p_IinC);
  const double iz = 1 / p_FinC[2];
  const double dznp[6] = {iz, 0, -p_FinC[0] / (p_FinC[2] * p_FinC[2]), 0, iz, -p_FinC[1] / (p_FinC[2] * p_FinC[2])};
  const M3 dpC_dpG = mm(R_ItoC, R_GtoI);
  const M3 left = mm(R_ItoC, skew3(p_FinI));
  double dpC_dI[18];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dpC_dI[6 * i + j] = left(i, j);
      dpC_dI[6 * i + 3 + j] = -dpC_dpG(i, j);
    }
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
  // ---- noise + whitening (REF :207-239 incl. the `R_llt.llt().solve(I)` form)
  double Rn[4] = {P.sigma_pix * P.sigma_pix, 0, 0, P.sigma_pix * P.sigma_pix};
  bool at_clone = false;
  for (int i = 0; i < P.n_clones; ++i) at_clone = at_clone || P.clone_time[i] == tm;
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
  double Wm[4];
#pragma unroll
  for (int col = 0; col < 2; ++col) {
    const double b0 = col == 0 ? 1.0 : 0.0, b1 = col == 1 ? 1.0 : 0.0;
    const double y0 = b0 / m00, y1 = (b1 - m10 * y0) / m11;
    const double x1 = y1 / m11, x0 = (y0 - m10 * x1) / m00;
    Wm[col] = x0;
    Wm[2 + col] = x1;
  }
  double wz[6], wzeta[16];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    wz[j] = Wm[0] * dz_dpC[j] + Wm[1] * dz_dpC[3 + j];
    wz[3 + j] = Wm[2] * dz_dpC[j] + Wm[3] * dz_dpC[3 + j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    wzeta[j] = Wm[0] * dzeta[j] + Wm[1] * dzeta[8 + j];
    wzeta[8 + j] = Wm[2] * dzeta[j] + Wm[3] * dzeta[8 + j];
  }
  rs[(2 * c) * rstr] = Wm[0] * r2[0] + Wm[1] * r2[1];
  rs[(2 * c + 1) * rstr] = Wm[2] * r2[0] + Wm[3] * r2[1];
  // ---- Hf
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
  // ---- Hx: four interpolation poses.  The slice was zero-filled by this workgroup and every (row, column) below is written
  // once (the clones of a window, the time offset, the extrinsics and the intrinsics are distinct state blocks), so these are
  // plain stores: an accumulate would put a global load in front of every one of them on the observation's chain.
  double WI[12];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) WI[6 * i + j] = wz[3 * i] * dpC_dI[j] + wz[3 * i + 1] * dpC_dI[6 + j] + wz[3 * i + 2] * dpC_dI[12 + j];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int col = P.clone_col[s0 + w];
    if (col < 0) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double so = WI[6 * i] * jac.Ho[w](0, j) + WI[6 * i + 1] * jac.Ho[w](1, j) + WI[6 * i + 2] * jac.Ho[w](2, j);
        hx[(size_t)(col + j) * cstr + (2 * c + i) * rstr] = so;
        hx[(size_t)(col + 3 + j) * cstr + (2 * c + i) * rstr] = WI[6 * i + 3 + j] * jac.lam[w];
      }
  }
  if (P.col_dt >= 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) s += WI[6 * i + q] * jac.dtj[q];
      hx[(size_t)P.col_dt * cstr + (2 * c + i) * rstr] = s;
    }
  if (P.col_ext >= 0) {
    const M3 sk = skew3(vsub(p_FinC, p_IinC));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        hx[(size_t)(P.col_ext + j) * cstr + (2 * c + i) * rstr] = wz[3 * i] * sk(0, j) + wz[3 * i + 1] * sk(1, j) + wz[3 * i + 2] * sk(2, j);
        hx[(size_t)(P.col_ext + 3 + j) * cstr + (2 * c + i) * rstr] = wz[3 * i + j];
      }
  }
  if (P.col_int >= 0)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) hx[(size_t)(P.col_int + j) * cstr + (2 * c + i) * rstr] = wzeta[8 * i + j];
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
#define TRI_TERMS 10
// One feature, ONE wave (the 64 lanes that call it; other waves of the workgroup must not): poses of its observations, linear
// triangulation, Levenberg-Marquardt refinement, reprojection error.  tri_smem: max_obs * (sizeof(TriObs) + TRI_TERMS * 8 + 4) + 16 bytes
// of LDS, tot: TRI_TERMS doubles of LDS.  Wave-level synchronisation only (the lanes run in lockstep; the fences order the LDS traffic).
__device__ __forceinline__ void tri_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ void triangulate_feature(const JacParams &P, int f, double *poses, unsigned char *valid, const float *__restrict__ uvn,
                                    const plv_tri_options &opt, double *__restrict__ p_out, unsigned char *__restrict__ ok_out,
                                    double *__restrict__ err_out, int max_obs, double *tri_smem, double *tot) {
  const int lane = threadIdx.x & 63;
  {  // camera poses of this feature's observations (CamHelper::get_imu_poses / get_cam_poses), one lane each: no separate launch
    const int o0 = P.obs_ptr[f], o1 = P.obs_ptr[f + 1];
    for (int o = o0 + lane; o < o1; o += 64) campose_one(P, o, poses, valid, nullptr);
    __threadfence_block();
    tri_wave_sync();
  }
  TriObs *ob = reinterpret_cast<TriObs *>(tri_smem);                       // [max_obs]
  double *term = tri_smem + (size_t)max_obs * (sizeof(TriObs) / 8);        // [max_obs][TRI_TERMS]
  int *list = reinterpret_cast<int *>(term + (size_t)max_obs * TRI_TERMS);  // [max_obs] indices of the valid observations
  const int o0 = P.obs_ptr[f], o1 = P.obs_ptr[f + 1];
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
  }
  if (M < 2) return;
  tri_wave_sync();
  const int last = list[M - 1];
  const M3 R_GtoA = ldM(poses + 12 * last);  // anchor = newest observation (FeatureInitializer.cpp:44-45)
  const V3 p_AinG = ldV(poses + 12 * last + 9);
  // sums `n` terms per observation in observation order; afterwards tot[0..n) holds the totals for every lane
  auto reduce = [&](int n) {
    tri_wave_sync();
    if (lane < n) {
      double s = 0;
      for (int q = 0; q < M; ++q) s += term[q * TRI_TERMS + lane];
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
  if (fabs(condA) > opt.max_cond_number || pf[2] < opt.min_dist || pf[2] > opt.max_dist || isnan(vnorm(pf))) return;
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
    while (runs < 5 && lam < 1e10 && eps > 1e-6) {
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
      const double r0 = (double)P.obs_uv[2 * o] - (double)(float)(K[0] * x1 + K[2]);
      const double r1 = (double)P.obs_uv[2 * o + 1] - (double)(float)(K[1] * y1 + K[3]);
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
  }
}

__global__ void __launch_bounds__(64) triangulate_kernel(JacParams P, double *poses, unsigned char *valid, const float *__restrict__ uvn,
                                                         plv_tri_options opt, double *__restrict__ p_out,
                                                         unsigned char *__restrict__ ok_out, double *__restrict__ err_out, int max_obs) {
  extern __shared__ double tri_smem[];
  __shared__ double tot[TRI_TERMS];
  triangulate_feature(P, blockIdx.x, poses, valid, uvn, opt, p_out, ok_out, err_out, max_obs, tri_smem, tot);
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

// Element (row, col) of a block goes to base[col * cstr + row * rstr], as in jacobian_rows.
__device__ void line_rows(const JacParams &P, int l, int o, int s0, double tm, int c, double *hf, double *hx, double *rs, int cstr,
                          int rstr, const WinTab *tab = nullptr) {
  const M3 R_ItoC = ldM(P.R_ItoC);
  const V3 p_IinC = ldV(P.p_IinC);
  const double *Kc = P.K;
  const double Kl[9] = {Kc[1], 0, 0, 0, Kc[0], 0, -Kc[1] * Kc[2], -Kc[0] * Kc[3], Kc[0] * Kc[1]};
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
  const M3 Rsk = ms(mm(Re, skew3(pe)), -1.0);
  const V3 nI = vadd(mv(Re, nG), mv(Rsk, vG)), vI = mv(Re, vG);
  const M3 SR = mm(skew3(p_IinC), R_ItoC);
  const V3 nC = vadd(mv(R_ItoC, nI), mv(SR, vI));
  const double l3[3] = {Kl[0] * nC[0] + Kl[1] * nC[1] + Kl[2] * nC[2], Kl[3] * nC[0] + Kl[4] * nC[1] + Kl[5] * nC[2],
                        Kl[6] * nC[0] + Kl[7] * nC[1] + Kl[8] * nC[2]};
  const double us[3] = {(double)P.seg_uv[4 * o], (double)P.seg_uv[4 * o + 1], 1.0};
  const double ue[3] = {(double)P.seg_uv[4 * o + 2], (double)P.seg_uv[4 * o + 3], 1.0};
  const double lnorm = sqrt(l3[0] * l3[0] + l3[1] * l3[1]);
  const double ds = us[0] * l3[0] + us[1] * l3[1] + us[2] * l3[2], de = ue[0] * l3[0] + ue[1] * l3[1] + ue[2] * l3[2];
  const double r2[2] = {ds / lnorm, de / lnorm};
  const double ln_2 = l3[0] * l3[0] + l3[1] + l3[1];
  double dzl[6] = {1, 0, 0, 0, 1, 0};
  dzl[0] = us[0] - (l3[0] * ds) / ln_2;
  dzl[1] = us[1] - (l3[1] * ds) / ln_2;
  dzl[3] = ue[0] - (l3[0] * de) / ln_2;
  dzl[4] = ue[1] - (l3[1] * de) / ln_2;
  const double isq = 1 / sqrt(ln_2);
#pragma unroll
  for (int i = 0; i < 6; ++i) dzl[i] *= isq;
  double dzK[6], dzli[12];
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
  const M3 Rf = jac.R;
  const V3 pf = jac.p;
  const M3 A00 = skew3(mv(Rf, vsub(nG, mv(skew3(pf), vG)))), A30 = skew3(mv(Rf, vG)), A03 = mm(Rf, skew3(vG));
  // HI = dzli * dli_dI, dli_dI = [A00 A03; A30 0]
  double HI[12];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double so = 0, sp = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        so += dzli[6 * i + q] * A00(q, j);
        sp += dzli[6 * i + q] * A03(q, j);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        so += dzli[6 * i + 3 + q] * A30(q, j);
        sp += dzli[6 * i + 3 + q] * 0.0;
      }
      HI[6 * i + j] = so;
      HI[6 * i + 3 + j] = sp;
    }
  double Rn[4] = {P.sigma_pix * P.sigma_pix, 0, 0, P.sigma_pix * P.sigma_pix};
  bool at_clone = false;
  for (int i = 0; i < P.n_clones; ++i) at_clone = at_clone || P.clone_time[i] == tm;
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
  double Wm[4];
#pragma unroll
  for (int col = 0; col < 2; ++col) {
    const double b0 = col == 0 ? 1.0 : 0.0, b1 = col == 1 ? 1.0 : 0.0;
    const double y0 = b0 / m00, y1 = (b1 - m10 * y0) / m11;
    const double x1 = y1 / m11, x0 = (y0 - m10 * x1) / m00;
    Wm[col] = x0;
    Wm[2 + col] = x1;
  }
  rs[(2 * c) * rstr] = Wm[0] * r2[0] + Wm[1] * r2[1];
  rs[(2 * c + 1) * rstr] = Wm[2] * r2[0] + Wm[3] * r2[1];
  double wli[12];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    wli[j] = Wm[0] * dzli[j] + Wm[1] * dzli[6 + j];
    wli[6 + j] = Wm[2] * dzli[j] + Wm[3] * dzli[6 + j];
  }
  // Hf = wli * G_to_I, G_to_I = [Re Rsk; 0 Re]
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
  double WI[12];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double so = 0, sp = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        so += wli[6 * i + q] * A00(q, j);
        sp += wli[6 * i + q] * A03(q, j);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        so += wli[6 * i + 3 + q] * A30(q, j);
        sp += wli[6 * i + 3 + q] * 0.0;
      }
      WI[6 * i + j] = so;
      WI[6 * i + 3 + j] = sp;
    }
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int col = P.clone_col[s0 + w];
    if (col < 0) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double so = WI[6 * i] * jac.Ho[w](0, j) + WI[6 * i + 1] * jac.Ho[w](1, j) + WI[6 * i + 2] * jac.Ho[w](2, j);
        hx[(size_t)(col + j) * cstr + (2 * c + i) * rstr] = so;  // (written once: plain stores, as in jacobian_rows)
        hx[(size_t)(col + 3 + j) * cstr + (2 * c + i) * rstr] = WI[6 * i + 3 + j] * jac.lam[w];
      }
  }
  if (P.col_dt >= 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) s += WI[6 * i + q] * jac.dtj[q];
      hx[(size_t)P.col_dt * cstr + (2 * c + i) * rstr] = s;
    }
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
                                     double *out, unsigned char &ok);
// tri.on: the line is triangulated first, by this very workgroup, on the state Pt (LineHelper::get_line_features runs on the state
// before the point update, lines_update linearises on the updated one: two views of the same window) — one launch for what were
// line_triangulate_kernel + this one.  Only while the selection loop has no cap to enforce (n_feat <= max_sel): a line is then
// taken on its own merits and needs no count over the lines before it.
struct LineTriStage {
  int on;
  double *cam, *imu;      // [n_obs][12] camera / IMU poses of the observations (scratch)
  unsigned char *valid;   // [n_obs]
  double *out_g;          // [L][6]  == P.line_FinG of the Jacobian stage
  unsigned char *ok_g;    // [L]
};
__global__ void __launch_bounds__(256) line_jacobian_nullspace_kernel(JacParams P, int L, GatherArgs g, JacParams Pt, LineTriStage tri, GateStage gate) {
  extern __shared__ double jsm[];  // X [ld][ncol] | piv [ld]
  __shared__ WinTab tab[JAC_MAX_WIN];
  __shared__ int s_rows, s_ok;
  if ((int)blockIdx.x >= L) {
    gather_cov_block(g, blockIdx.x - L);
    return;
  }
  const int l = blockIdx.x;
  const int ld = P.ld, k = P.k, ncol = 6 + k + 1;
  double *X = jsm, *piv = jsm + ld * ncol;
  for (int i = threadIdx.x; i < ld * ncol; i += blockDim.x) X[i] = 0.0;
  if (l == 0 && P.cols_out)
    for (int i = threadIdx.x; i < k; i += blockDim.x) P.cols_out[i] = P.cols_in[i];
  bool selected;
  if (tri.on) {
    const int o0 = Pt.obs_ptr[l], o1 = Pt.obs_ptr[l + 1];
    for (int o = o0 + (int)threadIdx.x; o < o1; o += blockDim.x) campose_one(Pt, o, tri.cam, tri.valid, tri.imu);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 64) {
      double out_l[6] = {0, 0, 0, 0, 0, 0};
      unsigned char ok_l = 0;
      line_triangulate_one(Pt, l, o0, o1, tri.cam, tri.imu, tri.valid, out_l, ok_l);
      if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) tri.out_g[6 * l + i] = out_l[i];
        tri.ok_g[l] = ok_l;
        s_ok = ok_l;
      }
    }
    __threadfence_block();
    __syncthreads();
    selected = P.sel_flags[l] && s_ok;
  } else {
    selected = !P.tri_ok || candidate_selected(P, l);
  }
  if (selected)
    build_window_tables(P, tab);  // (block-uniform; ends with a barrier: also orders the zero fill before the row writes)
  else
    __syncthreads();
  if (!selected) {
    if (threadIdx.x == 0) {
      s_rows = 0;
      P.rows[l] = 0;
    }
  } else if (threadIdx.x < 64) {
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
      if (s0 >= 0 && 2 * c + 2 <= ld) line_rows(P, l, o, s0, tm, c, X, X + 6, X + 6 + k, 1, ncol, tab);
    }
    if (threadIdx.x == 0) {
      s_rows = min(2 * base, ld & ~1);
      P.rows[l] = 2 * base;
    }
  }
  __syncthreads();
  const int rows = s_rows;
  if ((tri.on || P.tri_ok) && rows == 0) {  // (an unselected pool line: empty system, nothing reads its block)
    if (gate.on) gate_tail(gate, *reinterpret_cast<GateLds *>(reinterpret_cast<char *>(jsm) + gate.lds_off), l, X, ncol, 6, 0, 0, k, P.cols_in);
    return;
  }
  const int shift = rows > 6 ? 6 : 0;  // (a block with no more rows than Hf has columns is left as it is, as nullspace_kernel does)
  if (shift) nullspace_householder(X, piv, rows, ncol, 6);
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
  if (gate.on) gate_tail(gate, *reinterpret_cast<GateLds *>(reinterpret_cast<char *>(jsm) + gate.lds_off), l, X, ncol, 6, shift, min(rows, ld), k, P.cols_in);
}

__device__ void line_triangulate_one(const JacParams &P, int l, int o0, int o1, const double *cam, const double *imu, const unsigned char *valid,
                                     double *out, unsigned char &ok);
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
  line_triangulate_one(P, l, o0, o1, cam, imu, valid, out_l, ok_l);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) out_g[6 * l + i] = out_l[i];
    ok_g[l] = ok_l;
  }
}
__device__ void line_triangulate_one(const JacParams &P, int l, int o0, int o1, const double *cam, const double *imu, const unsigned char *valid,
                                     double *out /*[6], zero on entry*/, unsigned char &ok) {
  int first = -1, nvalid = 0;
  for (int o = o0; o < o1; ++o)
    if (valid[o]) {
      if (first < 0) first = o;
      ++nvalid;
    }
  if (nvalid < 2) return;
  const int D = P.lineD ? P.lineD[l] : 0;
  if (D > 0 && P.has_pt && P.has_pt[l]) {
    const M3 R = ldM(imu + 12 * first);
    const V3 e{{D == 1 ? 1.0 : 0.0, D == 2 ? 1.0 : 0.0, D == 3 ? 1.0 : 0.0}};
    const V3 dir = mv(tp(R), e);
    const V3 mom = cross3(ldV(P.anchor_pt + 3 * l), dir);
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
  const float *u0 = P.seg_uvn + 4 * first;
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
    const float *um = P.seg_uvn + 4 * o;
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

int launch_line_jacobians(plv_ctx *ctx, const JacParams &P) {
  ProfScope ps(ctx->prof, "line_jacobian_kernel", ctx->stream);
  hipLaunchKernelGGL(line_jacobian_kernel, dim3(P.n_feat), dim3(64), 0, ctx->stream, P);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_line_jacobians_projected(plv_ctx *ctx, const JacParams &P, const GatherArgs *g, int gather_blocks, const JacParams *Pt, double *d_cam,
                                    double *d_imu, unsigned char *d_valid, double *d_lines, unsigned char *d_ok) {
  ProfScope ps(ctx->prof, Pt ? "line_tri_jacobian_nullspace_kernel" : "line_jacobian_nullspace_kernel", ctx->stream);
  if (2 * (P.n_clones - 3) > JAC_MAX_WIN) {
    set_last_error("line jacobians: %d clones exceed the window table (%d)", P.n_clones, JAC_MAX_WIN / 2 + 3);
    return PLV_E_CAPACITY;
  }
  size_t shm = (size_t)(P.ld * (6 + P.k + 1) + P.ld) * sizeof(double);
  if (shm + sizeof(WinTab) * JAC_MAX_WIN + 64 > 160 * 1024) {
    set_last_error("line jacobians: block of %zu bytes exceeds LDS", shm);
    return PLV_E_CAPACITY;
  }
  GatherArgs none{};
  LineTriStage tri{Pt ? 1 : 0, d_cam, d_imu, d_valid, d_lines, d_ok};
  GateStage gate = ctx->gate_stage;
  ctx->gate_stage.on = 0;  // (one launch takes it)
  const size_t gate_off = (shm + 63) & ~(size_t)63;
  if (P.k > GATE_KMAX || gate_off + sizeof(GateLds) + sizeof(WinTab) * JAC_MAX_WIN + 512 > 160 * 1024) gate.on = 0;
  if (gate.on) {
    gate.lds_off = (int)gate_off;
    shm = gate_off + sizeof(GateLds);
  }
  ctx->gate_stage_taken = gate.on != 0;
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)line_jacobian_nullspace_kernel, (int)shm));
  hipLaunchKernelGGL(line_jacobian_nullspace_kernel, dim3(P.n_feat + (g ? gather_blocks : 0)), dim3(256), shm, ctx->stream, P, P.n_feat,
                     g ? *g : none, Pt ? *Pt : P, tri, gate);
  PLV_HIP_CHECK(hipGetLastError());
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
  const size_t tri_shm = tri_opt ? (size_t)std::max(max_obs, 1) * (sizeof(TriObs) + TRI_TERMS * 8 + 4) + 16 : 0;
  size_t shm = (size_t)(P.ld * (3 + P.k + 1) + P.ld) * sizeof(double) + tri_shm;
  if (shm + sizeof(WinTab) * JAC_MAX_WIN + 256 > 160 * 1024) {
    set_last_error("jacobians: feature block of %zu bytes exceeds LDS", shm);
    return PLV_E_CAPACITY;
  }
  GatherArgs none{};
  PointTriStage tri{};
  if (tri_opt) tri = PointTriStage{1, std::max(max_obs, 1), d_poses, d_valid, d_uvn, *tri_opt, d_p, d_ok, d_err};
  GateStage gate = ctx->gate_stage;
  ctx->gate_stage.on = 0;  // (one launch takes it)
  // the gate's LDS block sits behind the entry's block in the launch's dynamic shared memory: taken only where both fit
  const size_t gate_off = (shm + 63) & ~(size_t)63;
  if (P.k > GATE_KMAX || gate_off + sizeof(GateLds) + sizeof(WinTab) * JAC_MAX_WIN + 512 > 160 * 1024) gate.on = 0;  // (rows: plv_update_gate_prepare)
  if (gate.on) {
    gate.lds_off = (int)gate_off;
    shm = gate_off + sizeof(GateLds);
  }
  ctx->gate_stage_taken = gate.on != 0;
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)jacobian_nullspace_kernel, (int)shm));
  hipLaunchKernelGGL(jacobian_nullspace_kernel, dim3(P.n_feat + (g ? gather_blocks : 0)), dim3(256), shm, ctx->stream, P, P.n_feat,
                     g ? *g : none, tri, gate);
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
