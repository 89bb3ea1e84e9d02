// Stand-in for ov_core's print macros (REF: open_vins/ov_core/src/utils/print.h:54,96-100; PL-VIWO/src/utils/Print_Logger.h:52-56)
#pragma once
#include <cstdio>
#define RED "\033[0;31m"
#define RESET "\033[0m"
#define PRINT_ERROR(...) std::printf(__VA_ARGS__)
#define PRINT4(...) std::printf(__VA_ARGS__)
