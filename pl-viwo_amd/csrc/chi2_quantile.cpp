// chi2_quantile.cpp — 95 % chi-square quantile, replacing
// boost::math::quantile(chi_squared(k), 0.95)   REF: PL/update/UpdaterStatistics.cpp:31-37,113-116
// Host-only numerics: regularised lower incomplete gamma P(a,x) (series / Lentz continued
// fraction) inverted by safeguarded Newton from a Wilson–Hilferty start.
#include <cmath>

namespace plv {

static double gamma_p(double a, double x) {
  if (x <= 0.0) return 0.0;
  const double lg = std::lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 2000; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
    }
    return sum * std::exp(-x + a * std::log(x) - lg);
  }
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 2000; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (std::fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < 1e-17) break;
  }
  return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}

double chi2_quantile(int dof, double p) {
  if (dof < 1) return 0.0;
  const double a = 0.5 * dof;
  // Wilson–Hilferty start (z for p = 0.95; a generic rational approximation is not needed here)
  const double z = 1.6448536269514722;
  double t = 2.0 / (9.0 * dof);
  double x = dof * std::pow(1.0 - t + z * std::sqrt(t), 3.0);
  if (x <= 0.0) x = 0.5 * dof;
  double lo = 0.0, hi = 1e300;
  for (int it = 0; it < 100; ++it) {
    double f = gamma_p(a, 0.5 * x) - p;
    if (f > 0.0) hi = x; else lo = x;
    // pdf of chi2
    double logpdf = (a - 1.0) * std::log(0.5 * x) - 0.5 * x - std::lgamma(a) - std::log(2.0);
    double step = f / std::exp(logpdf);
    double xn = x - step;
    if (!(xn > lo && xn < hi)) xn = (hi < 1e299) ? 0.5 * (lo + hi) : 2.0 * x;
    if (std::fabs(xn - x) <= 1e-15 * std::fabs(x)) {
      x = xn;
      break;
    }
    x = xn;
  }
  return x;
}

}  // namespace plv
