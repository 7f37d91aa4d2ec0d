// Divide-and-conquer eigen-solver of a symmetric tridiagonal matrix (Cuppen; Gu & Eisenstat's stable eigenvector formula):
// the scalar logic shared by the HIP kernels of k_dc.hip (one wave per secular root, reductions through a wave reducer) and
// by the host-side logic tests (serial reducer, tests/hostsim).  It replaces bisection + inverse iteration for matrices
// whose tridiagonal form does not split into small blocks -- dense molecular Hessians (the reference enters the RFO-NEB path
// with them: neb.py:794-804), where inverse iteration walks eigenvalue clusters in sequence and the vectors need
// re-orthogonalisation; here orthogonality follows from the formulas, clusters deflate.
// The eigen-decomposition itself is numpy.linalg.eigh in the reference (rsirfo.py:606): any orthonormal eigenbasis is a valid
// result, the method is free.  Algorithmic layout follows LAPACK's published dstedc family (dlaed1/2/3/4), restated.
//
//   T = [T1 0; 0 T2] + |rho| u u^T,  u = e_cut-1 + sign(rho) e_cut,  T1, T2 with their corner entries reduced by |rho|
//   T1 = Q1 D1 Q1^T, T2 = Q2 D2 Q2^T  ->  T = Q (D + rho' z z^T) Q^T,  z = (last row of Q1, sign * first row of Q2)/sqrt 2, rho' = 2|rho|
//   deflation (tiny z_i; nearly equal d_i, d_j by a Givens rotation)  ->  k x k secular problem
//   roots of 1/rho' + sum z_i^2 / (d_i - lambda) = 0 with the differences d_i - lambda_j kept to full relative accuracy
//   zhat_i^2 = prod_j (lambda_j - d_i) / prod_{j != i} (d_j - d_i)  (Loewner)  ->  u_j = (zhat_i / (d_i - lambda_j))_i / norm
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MOPT_DC_HD __host__ __device__
#else
#define MOPT_DC_HD
#endif

namespace mopt {

// quotient inside the secular sums: on the device a hardware reciprocal estimate with two Newton steps (|delta| is a normal number;
// ~1 ulp, far inside the 8 eps margin of the stopping criterion), IEEE division on the host
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double dc_quot(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return a * r;
}
// square root of the discriminants of the rational steps (they only steer the iteration: the converged delta does not depend on them)
__device__ __forceinline__ double dc_sqrt_fast(double x) {
    if (!(x > 1e-300)) return sqrt(x);
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return x * y;
}
#else
inline double dc_quot(double a, double b) { return a / b; }
inline double dc_sqrt_fast(double x) { return sqrt(x); }
#endif

constexpr double DC_EPS = 1.1102230246251565e-16;      // unit round-off (LAPACK dlamch('E'))
constexpr int DC_LEAF = 32;                             // largest leaf (solved whole by the wave-level Jacobi)
constexpr int DC_MAXIT = 80;

// number of halving levels so that every leaf has at most DC_LEAF rows; leaf boundaries lb(i) = (i n) >> L
MOPT_DC_HD inline int dc_levels(int n) { int L = 0; while (((n + (1 << L) - 1) >> L) > DC_LEAF) ++L; return L; }
// the same with leaves of at most DC_LEAF_WG rows, each solved by one workgroup with the in-LDS divide and conquer of dc_block_body.hpp
// (round 4: one level of the launch-per-stage chain less -- n = 300: 8 leaves of 37 / 38 rows instead of 16 of 18 / 19)
constexpr int DC_LEAF_WG = 40;
MOPT_DC_HD inline int dc_levels_wg(int n) { int L = 0; while (((n + (1 << L) - 1) >> L) > DC_LEAF_WG) ++L; return L; }
MOPT_DC_HD inline int dc_bound(int n, int L, int i) { return (int)(((long long)i * n) >> L); }

struct DcRot { int a, b; double c, s; };               // rows a, b of the eigenvector table: x_a' = c x_a + s x_b, x_b' = c x_b - s x_a

// Deflation scan of one merge (LAPACK dlaed2's loop, restated).  Sequential, one thread.
//   m        size of the merged problem, m1 = size of the upper part (types: index < m1 upper, else lower)
//   d[m]     eigenvalues of the two parts (modified in place by the rotations)
//   z[m]     z vector, already scaled by 1/sqrt 2 (modified in place)
//   rho      2 |rho| > 0 (rho == 0: everything deflates)
//   order[m] indices of d in ascending order
// Out: k non-deflated; nd[0..k) their indices in ascending d (dl, wv the matching poles and weights); typ[0..k) 1 upper only,
// 2 mixed, 3 lower only; defl[0..m-k) the deflated indices; rot[0..nrot) the rotations in application order.
MOPT_DC_HD inline void dc_deflate(int m, int m1, double *d, double *z, double rho, const int *order, int *k_out, int *nd, double *dl,
                                  double *wv, int *typ, int *defl, DcRot *rot, int *nrot_out, int *coltyp /* m scratch */) {
    int k = 0, ndefl = 0, nrot = 0;
    double zmax = 0.0, dmax = 0.0;
    for (int i = 0; i < m; ++i) { zmax = fmax(zmax, fabs(z[i])); dmax = fmax(dmax, fabs(d[i])); coltyp[i] = (i < m1) ? 1 : 3; }
    const double tol = 8.0 * DC_EPS * fmax(dmax, zmax);
    if (rho * zmax <= tol) {
        for (int j = 0; j < m; ++j) defl[ndefl++] = order[j];
        *k_out = 0; *nrot_out = 0;
        return;
    }
    int pj = -1;
    for (int j = 0; j < m; ++j) {
        const int nj = order[j];
        if (rho * fabs(z[nj]) <= tol) { defl[ndefl++] = nj; continue; }
        if (pj < 0) { pj = nj; continue; }
        double s = z[pj], c = z[nj];
        const double tau = hypot(c, s);
        const double t = d[nj] - d[pj];
        c /= tau; s = -s / tau;
        if (fabs(t * c * s) <= tol) {
            // the two poles are numerically equal: rotate the weight of pj into nj and deflate pj
            z[nj] = tau; z[pj] = 0.0;
            if (coltyp[nj] != coltyp[pj]) coltyp[nj] = 2;
            coltyp[pj] = 4;
            rot[nrot].a = pj; rot[nrot].b = nj; rot[nrot].c = c; rot[nrot].s = s; ++nrot;
            const double tt = d[pj] * c * c + d[nj] * s * s;
            d[nj] = d[pj] * s * s + d[nj] * c * c;
            d[pj] = tt;
            defl[ndefl++] = pj;
            pj = nj;
        } else {
            nd[k] = pj; dl[k] = d[pj]; wv[k] = z[pj]; typ[k] = coltyp[pj]; ++k;
            pj = nj;
        }
    }
    if (pj >= 0) { nd[k] = pj; dl[k] = d[pj]; wv[k] = z[pj]; typ[k] = coltyp[pj]; ++k; }
    *k_out = k; *nrot_out = nrot;
}

// The same scan restricted to its events, for the kernels' parallel layout: the survivors surv[0 .. nk) (ascending poles, tiny
// weights already removed) are given, cand has bit i set when the pair (surv[i-1], surv[i]) passed the cheap closeness test
//   |t z_a z_b| <= 2 tol (z_a^2 + z_b^2)      (t = d_b - d_a; twice the scan's tolerance, so an unset bit proves "no rotation")
// with the ORIGINAL d and z.  In the scan the partner of surv[i] is always surv[i-1] (untouched, or just rotated), so only
// flagged pairs and the pairs right behind a rotation need the scan's full test.  Marks removed[i] = 1 for survivors that a
// rotation deflates, applies the rotations to d, z, coltyp and lists them.  One thread.
MOPT_DC_HD inline void dc_walk_candidates(int nk, const int *surv, const unsigned long long *cand, double *d, double *z, double tol,
                                          int *coltyp, int *removed, DcRot *rot, int *nrot_out) {
    int nrot = 0;
    int i = 1;
    while (i < nk) {
        // next flagged pair at or behind i
        int w = i >> 6;
        unsigned long long bits = cand[w] & ~((1ull << (i & 63)) - 1ull);
        const int nwords = (nk + 63) >> 6;
        while (bits == 0ull && ++w < nwords) bits = cand[w];
        if (w >= nwords) break;
        int lowbit = 0;
        while (!((bits >> lowbit) & 1ull)) ++lowbit;
        i = (w << 6) + lowbit;
        if (i >= nk) break;
        // chain: test pair i; a rotation dirties surv[i], so the next pair is tested too
        for (;;) {
            const int pj = surv[i - 1], nj = surv[i];
            double s = z[pj], c = z[nj];
            const double tau = hypot(c, s);
            const double t = d[nj] - d[pj];
            c /= tau; s = -s / tau;
            if (!(fabs(t * c * s) <= tol)) { ++i; break; }
            z[nj] = tau; z[pj] = 0.0;
            if (coltyp[nj] != coltyp[pj]) coltyp[nj] = 2;
            coltyp[pj] = 4;
            rot[nrot].a = pj; rot[nrot].b = nj; rot[nrot].c = c; rot[nrot].s = s; ++nrot;
            const double tt = d[pj] * c * c + d[nj] * s * s;
            d[nj] = d[pj] * s * s + d[nj] * c * c;
            d[pj] = tt;
            removed[i - 1] = 1;
            ++i;
            if (i >= nk) break;
        }
    }
    *nrot_out = nrot;
}

// One root of the secular equation  1/rho + sum_i w_i^2 / (dl_i - lambda) = 0,  dl ascending, all w_i != 0, rho > 0:
// root j lies in (dl_j, dl_j+1) (j < k-1) or in (dl_k-1, dl_k-1 + rho |w|^2).  The iteration works on tau = lambda - origin with
// the origin at the nearer pole, so delta_i = dl_i - lambda is obtained as (dl_i - origin) - tau without cancellation
// (LAPACK dlaed4's scheme: two-pole rational interpolation, Newton fall-back, bisection safeguard).
//   R: reducer with  template<int K, class F> void sumk(int m, F f, double (&out)[K])  -- f(i, acc) adds the terms of index i.
//   delta[k]: out, dl_i - lambda_j.  Returns the iteration count, negative when the stopping criterion was not met.
template <class R>
MOPT_DC_HD inline int dc_secular(const R &red, int k, int j, const double *dl, const double *w, double rho, double *delta, double *lam) {
    const double rhoinv = 1.0 / rho;
    if (k == 1) {
        const double t = rho * w[0] * w[0];
        red.for_each(1, [&](int) { delta[0] = -t; });
        *lam = dl[0] + t;
        return 0;
    }
    int ii;                      // origin pole
    double tau, lb, ub;
    const bool last = (j == k - 1);
    const int pa = last ? k - 2 : j, pb = last ? k - 1 : j + 1;            // the two poles of the rational model
    const double wa2 = w[pa] * w[pa], wb2 = w[pb] * w[pb];
    const double del = dl[pb] - dl[pa];
    bool orgati = true;
    if (!last) {
        const double mid = 0.5 * del;
        const double dj = dl[j];
        double s1[1];
        red.template sumk<1>(k, [&](int i, double *acc) { if (i != pa && i != pb) acc[0] += w[i] * w[i] / ((dl[i] - dj) - mid); }, s1);
        const double c = rhoinv + s1[0];
        const double wmid = c - wa2 / mid + wb2 / mid;
        if (wmid > 0.0) {
            orgati = true; ii = j;
            const double a = c * del + wa2 + wb2, b = wa2 * del;
            const double disc = sqrt(fabs(a * a - 4.0 * b * c));
            tau = (a > 0.0) ? 2.0 * b / (a + disc) : (a - disc) / (2.0 * c);
            lb = 0.0; ub = mid;
        } else {
            orgati = false; ii = j + 1;
            const double a = c * del - wa2 - wb2, b = wb2 * del;
            const double disc = sqrt(fabs(a * a + 4.0 * b * c));
            tau = (a < 0.0) ? 2.0 * b / (a - disc) : -(a + disc) / (2.0 * c);
            lb = -mid; ub = 0.0;
        }
        if (!(tau >= lb && tau <= ub)) tau = 0.5 * (lb + ub);          // (also catches NaN)
    } else {
        ii = k - 1;
        const double midpt = 0.5 * rho;
        const double dk = dl[k - 1];
        double s1[1];
        red.template sumk<1>(k, [&](int i, double *acc) { if (i < k - 2) acc[0] += w[i] * w[i] / ((dl[i] - dk) - midpt); }, s1);
        const double c = rhoinv + s1[0];
        const double wl = c + wa2 / ((dl[k - 2] - dk) - midpt) - wb2 / midpt;
        const double a = -c * del + wa2 + wb2, b = wb2 * del;
        const double disc = sqrt(fabs(a * a + 4.0 * b * c));
        const double t2 = (a < 0.0) ? 2.0 * b / (disc - a) : (a + disc) / (2.0 * c);
        if (wl <= 0.0) {
            const double temp = wa2 / (del + rho) + wb2 / rho;
            tau = (c <= temp) ? rho : t2;
            lb = midpt; ub = rho;
        } else {
            tau = t2;
            lb = 0.0; ub = midpt;
        }
        if (!(tau >= lb && tau <= ub)) tau = 0.5 * (lb + ub);
    }
    const double dorg = dl[ii];
    red.for_each(k, [&](int i) { delta[i] = (dl[i] - dorg) - tau; });
    red.sync();
    int it = 0;
    bool ok = false;
    for (; it < DC_MAXIT; ++it) {
        // psi: poles below the origin side, phi: above; the origin pole is kept apart (interior roots)
        double sm[5];      // psi, dpsi, phi, dphi, sum |terms|
        red.template sumk<5>(k, [&](int i, double *acc) {
            const double t = dc_quot(w[i], delta[i]), tt = w[i] * t;
            if (last) {
                if (i < k - 1) { acc[0] += tt; acc[1] += t * t; } else { acc[2] += tt; acc[3] += t * t; }
                acc[4] += fabs(tt);
            } else if (i != ii) {
                if (i < ii) { acc[0] += tt; acc[1] += t * t; } else { acc[2] += tt; acc[3] += t * t; }
                acc[4] += fabs(tt);
            }
        }, sm);
        double wv, dw, erretm;
        const double da = delta[pa], db = delta[pb];
        if (last) {
            wv = rhoinv + sm[2] + sm[0];
            dw = sm[1] + sm[3];
            erretm = 8.0 * sm[4] + 2.0 * rhoinv + fabs(tau) * dw;
        } else {
            const double t = dc_quot(w[ii], delta[ii]), tt = w[ii] * t;
            wv = rhoinv + sm[2] + sm[0] + tt;
            dw = sm[1] + sm[3] + t * t;
            erretm = 8.0 * sm[4] + 2.0 * rhoinv + 3.0 * fabs(tt) + fabs(tau) * dw;
        }
        if (fabs(wv) <= DC_EPS * erretm) { ok = true; break; }
        if (wv <= 0.0) lb = fmax(lb, tau); else ub = fmin(ub, tau);
        double eta;
        if (last) {
            double c = wv - da * sm[1] - db * sm[3];
            const double a = (da + db) * wv - da * db * (sm[1] + sm[3]);
            const double b = da * db * wv;
            if (c < 0.0) c = fabs(c);
            if (c == 0.0) eta = ub - tau;
            else {
                const double disc = dc_sqrt_fast(fabs(a * a - 4.0 * b * c));
                eta = (a >= 0.0) ? dc_quot(a + disc, 2.0 * c) : dc_quot(2.0 * b, a - disc);
            }
            if (wv * eta > 0.0) eta = -dc_quot(wv, dw);
        } else {
            double c;
            if (orgati) { const double t = dc_quot(w[pa], da); c = wv - db * dw - (dl[pa] - dl[pb]) * t * t; }
            else { const double t = dc_quot(w[pb], db); c = wv - da * dw - (dl[pb] - dl[pa]) * t * t; }
            double a = (da + db) * wv - da * db * dw;
            const double b = da * db * wv;
            if (c == 0.0) {
                if (a == 0.0) a = orgati ? w[pa] * w[pa] + db * db * (sm[1] + sm[3]) : w[pb] * w[pb] + da * da * (sm[1] + sm[3]);
                eta = dc_quot(b, a);
            } else {
                const double disc = dc_sqrt_fast(fabs(a * a - 4.0 * b * c));
                eta = (a <= 0.0) ? dc_quot(a - disc, 2.0 * c) : dc_quot(2.0 * b, a + disc);
            }
            if (wv * eta >= 0.0) eta = -dc_quot(wv, dw);
        }
        double tn = tau + eta;
        if (!(tn <= ub && tn >= lb)) {                 // outside the bracket (or NaN): bisect it
            eta = (wv < 0.0) ? 0.5 * (ub - tau) : 0.5 * (lb - tau);
            tn = tau + eta;
        }
        if (tn == tau) { ok = (ub - lb) <= 4.0 * DC_EPS * fmax(fabs(lb), fabs(ub)); if (!ok) ok = fabs(wv) <= 64.0 * DC_EPS * erretm; break; }
        red.sync();
        red.for_each(k, [&](int i) { delta[i] -= eta; });
        red.sync();
        tau = tn;
    }
    *lam = dorg + tau;
    return ok ? it : -1 - it;
}

}  // namespace mopt
