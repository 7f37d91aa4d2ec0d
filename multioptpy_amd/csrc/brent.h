// Brent's method with SciPy's exact update order (scipy/optimize/Zeros/brentq.c), as called by
// the reference at multioptpy/Optimizer/rsirfo.py:1042.  Returns 0 and the root, -1 when
// f(a), f(b) do not bracket (SciPy raises ValueError), -2 when maxiter is exhausted (RuntimeError).
#pragma once
#include <math.h>

#ifndef MOPT_HD
#if defined(__HIPCC__)
#define MOPT_HD __host__ __device__
#else
#define MOPT_HD
#endif
#endif

namespace mopt {

template <class F>
MOPT_HD int brentq(F &&f, double xa, double xb, double xtol, double rtol, int maxiter, double *root) {
    double xpre = xa, xcur = xb, xblk = 0.0, fblk = 0.0, spre = 0.0, scur = 0.0;
    double fpre = f(xpre);
    double fcur = f(xcur);
    if (fpre == 0.0) { *root = xpre; return 0; }
    if (fcur == 0.0) { *root = xcur; return 0; }
    if (signbit(fpre) == signbit(fcur)) return -1;
    for (int i = 0; i < maxiter; ++i) {
        if (fpre != 0.0 && fcur != 0.0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre; fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        double delta = (xtol + rtol * fabs(xcur)) / 2.0;
        double sbis = (xblk - xcur) / 2.0;
        if (fcur == 0.0 || fabs(sbis) < delta) { *root = xcur; return 0; }
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                double dpre = (fpre - fcur) / (xpre - xcur);
                double dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            double lim1 = fabs(spre), lim2 = 3.0 * fabs(sbis) - delta;
            if (2.0 * fabs(stry) < (lim1 < lim2 ? lim1 : lim2)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0.0 ? delta : -delta);
        fcur = f(xcur);
    }
    return -2;
}

}  // namespace mopt
