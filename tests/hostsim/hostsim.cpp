// Host build of the scalar control logic in multioptpy_amd/csrc/logic.h with a serial reducer.
// TEST-ONLY: lets the CPU test-suite check the exact code the kernels run (update-rule
// coefficients, secular solver, alpha search, Brent iterates, trust radius) against the oracle
// without a GPU.  Never loaded by the product package.
#include <vector>
#include <cstring>
#include "../../multioptpy_amd/csrc/logic.h"
#include "../../multioptpy_amd/csrc/dc_logic.h"
#include <algorithm>
#include <numeric>

namespace {
struct SerialRed {
    template <class F> double sum(int m, F f) const { double a = 0.0; for (int i = 0; i < m; ++i) a += f(i); return a; }
    template <class F> void sum2(int m, F f, double *s0, double *s1) const { double a0 = 0.0, a1 = 0.0; for (int i = 0; i < m; ++i) f(i, a0, a1); *s0 = a0; *s1 = a1; }
    template <class P> int min_index(int m, P p) const { for (int i = 0; i < m; ++i) if (p(i)) return i; return m; }
    template <class F> void for_each(int m, F f) const { for (int i = 0; i < m; ++i) f(i); }
    template <int K, class F> void sumk(int m, F f, double (&out)[K]) const {
        for (int q = 0; q < K; ++q) out[q] = 0.0;
        for (int i = 0; i < m; ++i) f(i, out);
    }
    void sync() const {}
};

// small dense symmetric eigen-solver for the leaves of the host simulation (cyclic Jacobi; test-only)
void leaf_jacobi(int m, std::vector<double> &A, std::vector<double> &V, std::vector<double> &w) {
    V.assign((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) V[(size_t)i * m + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < m; ++p) for (int q = p + 1; q < m; ++q) off += A[(size_t)p * m + q] * A[(size_t)p * m + q];
        if (off < 1e-300) break;
        for (int p = 0; p < m; ++p)
            for (int q = p + 1; q < m; ++q) {
                const double apq = A[(size_t)p * m + q];
                if (apq == 0.0) continue;
                const double th = (A[(size_t)q * m + q] - A[(size_t)p * m + p]) / (2.0 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int r = 0; r < m; ++r) {
                    const double arp = A[(size_t)r * m + p], arq = A[(size_t)r * m + q];
                    A[(size_t)r * m + p] = c * arp - sn * arq; A[(size_t)r * m + q] = sn * arp + c * arq;
                }
                for (int r = 0; r < m; ++r) {
                    const double apr = A[(size_t)p * m + r], aqr = A[(size_t)q * m + r];
                    A[(size_t)p * m + r] = c * apr - sn * aqr; A[(size_t)q * m + r] = sn * apr + c * aqr;
                }
                for (int r = 0; r < m; ++r) {
                    const double vrp = V[(size_t)r * m + p], vrq = V[(size_t)r * m + q];
                    V[(size_t)r * m + p] = c * vrp - sn * vrq; V[(size_t)r * m + q] = sn * vrp + c * vrq;
                }
            }
    }
    w.resize(m);
    for (int i = 0; i < m; ++i) w[i] = A[(size_t)i * m + i];
}
}

extern "C" {

int sim_update_coefficients(int rule, const double *dots9, double *out9) {
    mopt::UpdateDots d;
    d.ss = dots9[0]; d.sy = dots9[1]; d.yy = dots9[2]; d.syd = dots9[3]; d.su = dots9[4];
    d.AA = dots9[5]; d.As = dots9[6]; d.zz = dots9[7]; d.zs = dots9[8];
    mopt::UpdateCoef c = mopt::update_coefficients(rule, d);
    out9[0] = c.c_yy; out9[1] = c.c_uu; out9[2] = c.c_AA; out9[3] = c.c_As; out9[4] = c.c_ss;
    out9[5] = c.applied; out9[6] = c.block_sym; out9[7] = c.mix; out9[8] = c.branch;
    return 0;
}

double sim_dd_theta(double ss, double sy, double thr) { return mopt::dd_theta(ss, sy, thr); }
int sim_rule_is_dd(int rule) { return mopt::rule_is_dd(rule); }
int sim_rule_is_block(int rule) { return mopt::rule_is_block(rule); }

int sim_rs_step(const mopt_rsirfo_params *p, double trust_radius, const double *w, const double *gt, int m,
                double *step, double *res5) {
    std::vector<double> best(m > 0 ? m : 1), cg2(m > 0 ? m : 1), cw(m > 0 ? m : 1);      // scratch as in k_rfo_scalar (cached alpha quotients)
    SerialRed red;
    mopt::RsResult r = mopt::rs_step(red, *p, trust_radius, w, gt, m, step, best.data(), cg2.data(), cw.data());
    res5[0] = r.branch; res5[1] = r.alpha; res5[2] = r.lambda_aug0; res5[3] = r.initial_norm; res5[4] = r.secular_iters0;
    return 0;
}

double sim_solve_rfo(const double *w, const double *gt, int m, double alpha, double *step, int *iters) {
    SerialRed red; double n2;
    return mopt::solve_rfo(red, w, gt, m, alpha, step, &n2, iters);
}

int sim_adjust_trust_radius(const mopt_rsirfo_params *p, mopt_rsirfo_state *st, double actual, double predicted,
                            double min_eig, double grad_norm) {
    mopt::adjust_trust_radius(*p, *st, actual, predicted, min_eig, grad_norm);
    return 0;
}

// Brent on a few fixed test functions (ids match tests/golden/make_golden.py BRENT_FUNCS order)
int sim_brentq(int func, double a, double b, double xtol, double rtol, int maxiter, double *root, double *xs, int *nx) {
    int count = 0;
    auto f = [&](double t) {
        double v;
        switch (func) {
        case 0: v = t * t * t - 2 * t - 5; break;
        case 1: v = cos(t) - t; break;
        case 2: v = 1.0 / t - 5.0; break;
        case 3: v = tanh(40 * (t - 0.3)) + 0.2; break;
        default: v = 0.81 / ((1 + t) * (1 + t)) - 0.25; break;
        }
        xs[count++] = t;
        return v;
    };
    int rc = mopt::brentq(f, a, b, xtol, rtol, maxiter, root);
    *nx = count;
    return rc;
}


// Divide and conquer on a symmetric tridiagonal (D, E): the orchestration the kernels of k_dc.hip perform, serially, on top
// of the shared scalar logic of dc_logic.h.  W: eigenvalues (unsorted), Zt: eigenvectors, vector-major (row j belongs to W[j]).
// info[5] = merges in which the staged deflation of the kernels disagreed with the scan (must be 0)
// info[0] = secular solves that missed the stopping criterion, info[1] = largest iteration count, info[2] = rotations, info[3] = deflated (top level)
int sim_dc_eigh(int n, const double *D, const double *E, double *W, double *Zt, int *info) {
    using namespace mopt;
    SerialRed red;
    const int L = dc_levels(n);
    // scaled to unit max-norm (the deflation tolerance mixes |d| and |z| <= 1; dstedc does the same); eigenvalues scaled back at the end
    double orgnrm = 0.0;
    for (int i = 0; i < n; ++i) { orgnrm = fmax(orgnrm, fabs(D[i])); if (i < n - 1) orgnrm = fmax(orgnrm, fabs(E[i])); }
    if (orgnrm == 0.0) { for (int i = 0; i < n; ++i) { W[i] = 0.0; for (int j = 0; j < n; ++j) Zt[(size_t)i * n + j] = (i == j); } info[0] = info[1] = info[2] = info[3] = info[4] = info[5] = 0; return 0; }
    std::vector<double> Ds(n), Es(n, 0.0);
    for (int i = 0; i < n; ++i) { Ds[i] = D[i] / orgnrm; if (i < n - 1) Es[i] = E[i] / orgnrm; }
    D = Ds.data(); E = Es.data();
    std::vector<double> D2(D, D + n), dcur(n), dnew(n), Q((size_t)n * n, 0.0), Qn((size_t)n * n, 0.0);
    for (int i = 1; i < (1 << L); ++i) { const int c = dc_bound(n, L, i); const double r = fabs(E[c - 1]); D2[c - 1] -= r; D2[c] -= r; }
    for (int i = 0; i < (1 << L); ++i) {
        const int s = dc_bound(n, L, i), t = dc_bound(n, L, i + 1), m = t - s;
        std::vector<double> A((size_t)m * m, 0.0), V, w;
        for (int r = 0; r < m; ++r) { A[(size_t)r * m + r] = D2[s + r]; if (r + 1 < m) { A[(size_t)r * m + r + 1] = E[s + r]; A[(size_t)(r + 1) * m + r] = E[s + r]; } }
        leaf_jacobi(m, A, V, w);
        for (int c = 0; c < m; ++c) { dcur[s + c] = w[c]; for (int r = 0; r < m; ++r) Q[(size_t)(s + c) * n + s + r] = V[(size_t)r * m + c]; }
    }
    info[0] = info[1] = info[2] = info[3] = info[4] = info[5] = 0;
    for (int lev = 1; lev <= L; ++lev) {
        const int nm = 1 << (L - lev);
        for (int q = 0; q < nm; ++q) {
            const int s = dc_bound(n, L, q << lev), c = dc_bound(n, L, (q << lev) + (1 << (lev - 1))), t = dc_bound(n, L, (q + 1) << lev);
            const int m = t - s, m1 = c - s;
            const double rho0 = E[c - 1];
            std::vector<double> d(m), z(m), dl(m), wv(m), what(m), S((size_t)m * m), U((size_t)m * m), lam(m);
            std::vector<int> order(m), nd(m), typ(m), defl(m), coltyp(m);
            std::vector<DcRot> rot(m);
            for (int i = 0; i < m; ++i) {
                d[i] = dcur[s + i];
                z[i] = (i < m1 ? Q[(size_t)(s + i) * n + (c - 1)] : (rho0 < 0 ? -1.0 : 1.0) * Q[(size_t)(s + i) * n + c]) * 0.70710678118654752440;
            }
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
            int k = 0, nrot = 0;
            {
                // the kernel's staged form (parallel phases emulated serially + dc_walk_candidates) must agree with the scan
                std::vector<double> d2(d), z2(z);
                const double rho2 = 2.0 * fabs(rho0);
                double zmx = 0.0, dmx = 0.0;
                for (int i = 0; i < m; ++i) { zmx = fmax(zmx, fabs(z2[i])); dmx = fmax(dmx, fabs(d2[i])); }
                const double tol = 8.0 * DC_EPS * fmax(dmx, zmx);
                std::vector<int> surv, defl2, ct(m), rem(m, 0), nd2;
                std::vector<DcRot> rot2(m);
                int nrot2 = 0;
                if (rho2 * zmx <= tol) { for (int j = 0; j < m; ++j) defl2.push_back(order[j]); }
                else {
                    for (int j = 0; j < m; ++j) { const int nj = order[j]; if (rho2 * fabs(z2[nj]) <= tol) defl2.push_back(nj); else surv.push_back(nj); }
                    const int nk = (int)surv.size();
                    std::vector<unsigned long long> cand((nk + 63) / 64 + 1, 0ull);
                    for (int i = 1; i < nk; ++i) {
                        const double za = z2[surv[i - 1]], zb = z2[surv[i]], tt = d2[surv[i]] - d2[surv[i - 1]];
                        if (fabs(tt * za * zb) <= 2.0 * tol * (za * za + zb * zb)) cand[i >> 6] |= 1ull << (i & 63);
                    }
                    for (int i = 0; i < m; ++i) ct[i] = (i < m1) ? 1 : 3;
                    dc_walk_candidates(nk, surv.data(), cand.data(), d2.data(), z2.data(), tol, ct.data(), rem.data(), rot2.data(), &nrot2);
                    for (int i = 0; i < nk; ++i) { if (rem[i]) defl2.push_back(surv[i]); else nd2.push_back(surv[i]); }
                }
                info[4] = 0;
                dc_deflate(m, m1, d.data(), z.data(), 2.0 * fabs(rho0), order.data(), &k, nd.data(), dl.data(), wv.data(), typ.data(), defl.data(), rot.data(), &nrot, coltyp.data());
                bool same = ((int)nd2.size() == k) && nrot2 == nrot;
                for (int i = 0; same && i < k; ++i) same = (nd2[i] == nd[i]) && (d2[nd[i]] == dl[i]) && (z2[nd[i]] == wv[i]) && (ct[nd[i]] == typ[i]);
                for (int r = 0; same && r < nrot; ++r) same = rot2[r].a == rot[r].a && rot2[r].b == rot[r].b && rot2[r].c == rot[r].c && rot2[r].s == rot[r].s;
                std::vector<int> da(defl.begin(), defl.begin() + (m - k)), db(defl2);
                std::sort(da.begin(), da.end()); std::sort(db.begin(), db.end());
                same = same && da == db;
                if (!same) info[5]++;
            }
            info[2] += nrot;
            if (lev == L) info[3] = m - k;
            for (int r = 0; r < nrot; ++r) {
                double *xa = &Q[(size_t)(s + rot[r].a) * n], *xb = &Q[(size_t)(s + rot[r].b) * n];
                for (int x = s; x < t; ++x) { const double va = xa[x], vb = xb[x]; xa[x] = rot[r].c * va + rot[r].s * vb; xb[x] = rot[r].c * vb - rot[r].s * va; }
            }
            for (int j = 0; j < k; ++j) {
                const int itc = dc_secular(red, k, j, dl.data(), wv.data(), 2.0 * fabs(rho0), &S[(size_t)j * m], &lam[j]);
                if (itc < 0) info[0]++;
                info[1] = std::max(info[1], itc < 0 ? -1 - itc : itc);
            }
            for (int i = 0; i < k; ++i) {
                double p = -S[(size_t)i * m + i];
                for (int j = 0; j < k; ++j) if (j != i) p *= S[(size_t)j * m + i] / (dl[i] - dl[j]);
                what[i] = copysign(sqrt(fabs(p)), wv[i]);
            }
            for (int j = 0; j < k; ++j) {
                double nr = 0.0;
                for (int i = 0; i < k; ++i) { U[(size_t)j * m + i] = what[i] / S[(size_t)j * m + i]; nr += U[(size_t)j * m + i] * U[(size_t)j * m + i]; }
                nr = 1.0 / sqrt(nr);
                for (int i = 0; i < k; ++i) U[(size_t)j * m + i] *= nr;
            }
            for (int j = 0; j < k; ++j) {
                double *out = &Qn[(size_t)(s + j) * n];
                for (int x = s; x < t; ++x) out[x] = 0.0;
                for (int i = 0; i < k; ++i) { const double u = U[(size_t)j * m + i]; const double *in = &Q[(size_t)(s + nd[i]) * n]; for (int x = s; x < t; ++x) out[x] += u * in[x]; }
                dnew[s + j] = lam[j];
            }
            for (int qd = 0; qd < m - k; ++qd) {
                const double *in = &Q[(size_t)(s + defl[qd]) * n];
                double *out = &Qn[(size_t)(s + k + qd) * n];
                for (int x = s; x < t; ++x) out[x] = in[x];
                dnew[s + k + qd] = d[defl[qd]];
            }
        }
        std::swap(Q, Qn); std::swap(dcur, dnew);
    }
    for (int i = 0; i < n; ++i) W[i] = dcur[i] * orgnrm;
    std::copy(Q.begin(), Q.end(), Zt);
    return 0;
}

}
