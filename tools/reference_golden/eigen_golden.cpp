// Pins for the update half of the oracle from REAL Eigen (absent from this image; see make_reference_golden.py): the three steps of
// the MSCKF update as the reference performs them with Eigen — a Givens elimination of the feature Jacobian applied to [Hx | res]
// (StateHelper::nullspace_project_inplace, REF: PL-VIWO/src/state/StateHelper.cpp:616-651), the Givens triangularisation of the stacked
// system (measurement_compress_inplace, :602-614, :653-672) and the EKF step with Eigen's LLT (EKFUpdate, :94-173) — on seeded inputs
// this file generates itself with a 64-bit LCG (so that the consumer, tests/test_reference_golden.py, regenerates them bit for bit
// without Eigen).  Written against Eigen's public API (JacobiRotation::makeGivens / applyOnTheLeft, LLT); no reference source is copied.
//   g++ -O2 -I/usr/include/eigen3 eigen_golden.cpp -o eigen_golden && ./eigen_golden tests/golden
#include <Eigen/Dense>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

using Eigen::MatrixXd;
using Eigen::VectorXd;

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static double lcg() {  // uniform in (-1, 1); the consumer mirrors these two lines
  g_s = g_s * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(int64_t)(g_s >> 11) / (double)(1ll << 52) - 1.0;
}
static MatrixXd rnd(int r, int c) {
  MatrixXd M(r, c);
  for (int j = 0; j < c; ++j)
    for (int i = 0; i < r; ++i) M(i, j) = lcg();  // column-major fill
  return M;
}
static void dump(const std::string &dir, const char *name, const MatrixXd &M) {
  FILE *f = fopen((dir + "/ref_update_" + name + ".bin").c_str(), "wb");
  const int64_t hdr[2] = {M.rows(), M.cols()};
  fwrite(hdr, 8, 2, f);
  fwrite(M.data(), 8, (size_t)M.size(), f);  // column-major
  fclose(f);
}
// eliminate column n of A below the diagonal from the bottom up with Givens rotations, carrying B (and c) along
static void givens_columns(MatrixXd &A, MatrixXd &B, VectorXd *c) {
  Eigen::JacobiRotation<double> G;
  for (int n = 0; n < A.cols(); ++n)
    for (int m = (int)A.rows() - 1; m > n; --m) {
      G.makeGivens(A(m - 1, n), A(m, n));
      A.block(m - 1, n, 2, A.cols() - n).applyOnTheLeft(0, 1, G.adjoint());
      B.block(m - 1, 0, 2, B.cols()).applyOnTheLeft(0, 1, G.adjoint());
      if (c) c->segment(m - 1, 2).applyOnTheLeft(0, 1, G.adjoint());
    }
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  // (1) null-space projection of one feature: Hf 30 x 3, Hx 30 x 40, res 30
  MatrixXd Hf = rnd(30, 3), Hx = rnd(30, 40);
  VectorXd res = rnd(30, 1);
  dump(dir, "ns_Hf", Hf), dump(dir, "ns_Hx", Hx), dump(dir, "ns_res", res);
  givens_columns(Hf, Hx, &res);
  dump(dir, "ns_Hx_out", Hx.bottomRows(27)), dump(dir, "ns_res_out", res.tail(27));
  // (2) compression of a 400 x 40 stacked system
  MatrixXd H = rnd(400, 40), dummy(400, 0);
  VectorXd r = rnd(400, 1);
  dump(dir, "cp_H", H), dump(dir, "cp_res", r);
  {
    MatrixXd R1 = r;
    givens_columns(H, R1, nullptr);
    r = R1.col(0);
  }
  dump(dir, "cp_H_out", H.topRows(40)), dump(dir, "cp_res_out", r.head(40));
  // (3) EKF step: P = A A^T + 1e-3 I (n = 60), H 40 x 40 on columns 15..54, R = I
  const int n = 60, k = 40;
  MatrixXd A = rnd(n, n) * 0.1, P = A * A.transpose() + 1e-3 * MatrixXd::Identity(n, n);
  MatrixXd Hk = H.topRows(k);
  VectorXd rk = r.head(k);
  MatrixXd M = P.middleCols(15, k) * Hk.transpose();
  MatrixXd S = Hk * P.block(15, 15, k, k) * Hk.transpose() + MatrixXd::Identity(k, k);
  MatrixXd Sinv = S.selfadjointView<Eigen::Upper>().llt().solve(MatrixXd::Identity(k, k));
  MatrixXd K = M * Sinv;
  VectorXd dx = K * rk;
  MatrixXd Pn = P - K * M.transpose();
  Pn = Pn.selfadjointView<Eigen::Upper>();
  dump(dir, "ekf_P", P), dump(dir, "ekf_P_out", Pn), dump(dir, "ekf_dx", dx);
  FILE *f = fopen((dir + "/ref_update.json").c_str(), "w");
  fprintf(f, "{\"eigen_version\": \"%d.%d.%d\", \"ekf_cols_first\": 15, \"ekf_k\": %d}\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION, k);
  fclose(f);
  return 0;
}
