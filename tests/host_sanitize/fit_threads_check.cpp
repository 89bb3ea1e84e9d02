// default_fit_threads() of pl-viwo_amd/csrc/line_host.hpp under affinities of 1 .. N CPUs: prints "<cpus> <helpers>"
// per line.  Built and run by tests/test_host_sanitize.py::test_helper_threads_are_clamped_to_the_allowed_cpus (no device code).
#include <cstdio>

#include "../../pl-viwo_amd/csrc/line_host.hpp"

int main() {
  cpu_set_t all;
  CPU_ZERO(&all);
  if (sched_getaffinity(0, sizeof all, &all) != 0) return 2;
  int cpus[CPU_SETSIZE], n = 0;
  for (int c = 0; c < CPU_SETSIZE; ++c)
    if (CPU_ISSET(c, &all)) cpus[n++] = c;
  for (int k = 1; k <= n; ++k) {
    cpu_set_t s;
    CPU_ZERO(&s);
    for (int i = 0; i < k; ++i) CPU_SET(cpus[i], &s);
    if (sched_setaffinity(0, sizeof s, &s) != 0) return 3;
    printf("%d %d\n", k, plv::linehost::default_fit_threads());
  }
  return 0;
}
