// Ayala-Schlegel curvature along the band (calculate_gamma, multioptpy/MEP/pathopt_bneb_force.py:161-222): quintic fit through the energies and the
// gradient projections of an image and its two neighbours; gamma = second derivative at the image.  Shared by k_gamma (band.hip) and k_bneb_force.
#pragma once
#include "common.hpp"

namespace mopt {

// 6x6 solve by LU with partial pivoting (numpy.linalg.solve = dgesv); returns false on an exactly zero pivot.
__device__ inline bool solve6(double A[6][6], double b[6]) {
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double mx = fabs(A[k][k]);
        for (int r = k + 1; r < 6; ++r) if (fabs(A[r][k]) > mx) { mx = fabs(A[r][k]); p = r; }
        if (A[p][k] == 0.0) return false;
        if (p != k) {
            for (int c = 0; c < 6; ++c) { double t = A[k][c]; A[k][c] = A[p][c]; A[p][c] = t; }
            double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        for (int r = k + 1; r < 6; ++r) {
            const double f = A[r][k] / A[k][k];
            for (int c = k; c < 6; ++c) A[r][c] -= f * A[k][c];
            b[r] -= f * b[k];
        }
    }
    for (int k = 5; k >= 0; --k) {
        double s = b[k];
        for (int c = k + 1; c < 6; ++c) s -= A[k][c] * b[c];
        b[k] = s / A[k][k];
    }
    return true;
}

// The five partial sums of a thread (stride = workgroup size).  NOT inlined on purpose: k_gamma and k_bneb_force must run the same
// instruction sequence (with -ffp-contract=fast the compiler is free to fuse a product into the following sum in one place and not in another),
// or an image-sharded band and a whole band would differ in the last bit of gamma.
__device__ __noinline__ void ayala_partial_sums(int n3, const double *xm, const double *x0, const double *xp, const double *gm, const double *g0,
                                                const double *gp, const double *t, double *v) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    for (int k = threadIdx.x; k < n3; k += blockDim.x) {
        const double dm = x0[k] - xm[k], dp = xp[k] - x0[k];
        a0 += dm * dm; a1 += dp * dp; a2 += gm[k] * dm; a3 += g0[k] * t[k]; a4 += gp[k] * dp;
    }
    v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3; v[4] = a4;
}

// d2prev = |x_i - x_{i-1}|^2, d2next = |x_{i+1} - x_i|^2, gm_dm = g_{i-1} . (x_i - x_{i-1}), g0_t = g_i . tau_i, gp_dp = g_{i+1} . (x_{i+1} - x_i)
__device__ inline double ayala_gamma(double d2prev, double d2next, double gm_dm, double g0_t, double gp_dp, double Em, double E0, double Ep) {
    const double dprev = sqrt(d2prev), dnext = sqrt(d2next);
    double gam = 0.0;
    if (!(dprev < 1e-6 || dnext < 1e-6)) {
        const double s[3] = {-dprev, 0.0, dnext};
        double A[6][6], b[6];
        for (int r = 0; r < 3; ++r) {
            double pw = 1.0;
            for (int c = 0; c < 6; ++c) { A[r][c] = pw; pw *= s[r]; }
            A[3 + r][0] = 0.0;
            pw = 1.0;
            for (int c = 1; c < 6; ++c) { A[3 + r][c] = c * pw; pw *= s[r]; }
        }
        b[0] = Em; b[1] = E0; b[2] = Ep;
        b[3] = gm_dm / dprev; b[4] = g0_t; b[5] = gp_dp / dnext;
        if (solve6(A, b)) gam = 2.0 * b[2];
    }
    return gam;
}

}  // namespace mopt
