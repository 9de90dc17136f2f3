// Host build of sage_amd/csrc/crlog.h for tests/test_crlog.py (the same header the kernels compile).
#include <math.h>
#include <stdint.h>

#include "../../sage_amd/csrc/crlog.h"

extern "C" {
// mode 0: production (two phases), 1: accurate phase only, 2: fast phase only (NaN where it cannot decide)
void emu_cr_log(int mode, const double* x, uint64_t n, double* out) {
    for (uint64_t i = 0; i < n; i++)
        out[i] = mode == 1 ? sagecore::cr_log<1>(x[i]) : mode == 2 ? sagecore::cr_log<2>(x[i]) : sagecore::cr_log<0>(x[i]);
}
void emu_libm_log(const double* x, uint64_t n, double* out) {
    for (uint64_t i = 0; i < n; i++) out[i] = log(x[i]);
}
}
