// Relative error of v_rcp_f64 / v_rsq_f64 raw and after one / two Newton steps (against the IEEE result), over 1e6 inputs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double e = fma(-v, r0, 1.0);
  double r1 = fma(r0, e, r0);
  e = fma(-v, r1, 1.0);
  double r2 = fma(r1, e, r1);
  double q0 = __builtin_amdgcn_rsq(v);
  double f = fma(-v * q0, q0, 1.0);
  double q1 = fma(0.5 * q0, f, q0);
  out[5 * i] = r0, out[5 * i + 1] = r1, out[5 * i + 2] = r2, out[5 * i + 3] = q0, out[5 * i + 4] = q1;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(5 * n);
  for (int i = 0; i < n; ++i) x[i] = std::exp(-20.0 + 40.0 * (i + 0.5) / n) * (1.0 + 1e-3 * std::sin(i));
  double *dx, *dout;
  hipMalloc(&dx, n * 8);
  hipMalloc(&dout, 5 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
  double m[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double rc = 1.0 / x[i], rq = 1.0 / std::sqrt(x[i]);
    for (int j = 0; j < 3; ++j) m[j] = std::fmax(m[j], std::fabs(o[5 * i + j] - rc) / rc);
    for (int j = 3; j < 5; ++j) m[j] = std::fmax(m[j], std::fabs(o[5 * i + j] - rq) / rq);
  }
  printf("max rel err: rcp raw %.3g, +1 Newton %.3g, +2 Newton %.3g; rsq raw %.3g, +1 Newton %.3g  (eps = 2.2e-16)\n", m[0], m[1], m[2], m[3], m[4]);
  return 0;
}
