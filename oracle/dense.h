// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may link or call anything under oracle/.
//
// dense.h — a minimal column-major fp64 matrix (the restatement's stand-in for Eigen::MatrixXd,
// which is absent from this image) plus the few factorizations the reference calls on Eigen.
// PARITY UNPINNED: the reference holds no golden vectors for this path; this
// restatement is cross-checked against numpy/scipy in tests/ instead.
#pragma once
#include <cassert>
#include <cmath>
#include <cstring>
#include <vector>

namespace orc {

struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double &operator()(int i, int j) { return a[(size_t)j * r + i]; }
  double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
  static Mat from(const double *p, int rows, int cols, int ld) {
    Mat m(rows, cols);
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i < rows; ++i) m(i, j) = p[(size_t)j * ld + i];
    return m;
  }
  void to(double *p, int ld) const {
    for (int j = 0; j < c; ++j)
      for (int i = 0; i < r; ++i) p[(size_t)j * ld + i] = (*this)(i, j);
  }
  static Mat identity(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; ++i) m(i, i) = 1.0;
    return m;
  }
};

inline Mat matmul(const Mat &A, const Mat &B) {
  assert(A.c == B.r);
  Mat C(A.r, B.c);
  for (int j = 0; j < B.c; ++j)
    for (int k = 0; k < A.c; ++k) {
      double b = B(k, j);
      if (b == 0.0) continue;
      for (int i = 0; i < A.r; ++i) C(i, j) += A(i, k) * b;
    }
  return C;
}
inline Mat transpose(const Mat &A) {
  Mat T(A.c, A.r);
  for (int j = 0; j < A.c; ++j)
    for (int i = 0; i < A.r; ++i) T(j, i) = A(i, j);
  return T;
}

// In-place lower Cholesky of the symmetric matrix whose UPPER triangle is valid (what
// `S.selfadjointView<Upper>().llt()` consumes).  Returns false when a pivot is <= 0 or NaN.
inline bool cholesky_lower_from_upper(const Mat &S, Mat &L) {
  int n = S.r;
  L = Mat(n, n);
  for (int j = 0; j < n; ++j) {
    double d = S(j, j);
    for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
    if (!(d > 0.0)) return false;
    double ljj = std::sqrt(d);
    L(j, j) = ljj;
    for (int i = j + 1; i < n; ++i) {
      double s = S(j, i);  // upper element (j,i) == symmetric (i,j)
      for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
      L(i, j) = s / ljj;
    }
  }
  return true;
}

// Solve (L L^T) X = B in place.
inline void cholesky_solve(const Mat &L, Mat &B) {
  int n = L.r;
  for (int c = 0; c < B.c; ++c) {
    for (int i = 0; i < n; ++i) {
      double s = B(i, c);
      for (int k = 0; k < i; ++k) s -= L(i, k) * B(k, c);
      B(i, c) = s / L(i, i);
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = B(i, c);
      for (int k = i + 1; k < n; ++k) s -= L(k, i) * B(k, c);
      B(i, c) = s / L(i, i);
    }
  }
}

// General inverse by partial-pivot Gauss-Jordan (the role Eigen's `.inverse()` plays at
// UpdaterStatistics.cpp:111 and State.cpp:687).  Returns false if singular.
inline bool inverse(const Mat &A, Mat &Ainv) {
  int n = A.r;
  Mat M = A;
  Ainv = Mat::identity(n);
  for (int col = 0; col < n; ++col) {
    int piv = col;
    double best = std::fabs(M(col, col));
    for (int i = col + 1; i < n; ++i)
      if (std::fabs(M(i, col)) > best) best = std::fabs(M(i, col)), piv = i;
    if (!(best > 0.0)) return false;
    if (piv != col)
      for (int j = 0; j < n; ++j) {
        std::swap(M(piv, j), M(col, j));
        std::swap(Ainv(piv, j), Ainv(col, j));
      }
    double d = 1.0 / M(col, col);
    for (int j = 0; j < n; ++j) M(col, j) *= d, Ainv(col, j) *= d;
    for (int i = 0; i < n; ++i) {
      if (i == col) continue;
      double f = M(i, col);
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) M(i, j) -= f * M(col, j), Ainv(i, j) -= f * Ainv(col, j);
    }
  }
  return true;
}

}  // namespace orc
