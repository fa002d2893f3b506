/* mathwrap.c -- test access to include/svo_math.h as the ORACLE build compiles it (gcc, -ffp-contract=off):
 * tests/test_svo_math.py measures its accuracy against numpy's long double and tests/test_gpu_math.py checks that
 * the HIP library evaluates the same functions to the same bits on the device.  Test infrastructure. */
#include "svo_oracle.h"

void orc_math_eval(int fn, const double *x, int n, double *y)
{
    for (int i = 0; i < n; i++) {
        const double v = x[i];
        y[i] = fn == 0 ? svo_sin(v) : fn == 1 ? svo_cos(v) : fn == 2 ? svo_acos(v) : fn == 3 ? svo_cbrt(v) : svo_log(v);
    }
}
