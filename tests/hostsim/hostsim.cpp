// Host build of the scalar control logic in multioptpy_amd/csrc/logic.h with a serial reducer.
// TEST-ONLY: lets the CPU test-suite check the exact code the kernels run (update-rule
// coefficients, secular solver, alpha search, Brent iterates, trust radius) against the oracle
// without a GPU.  Never loaded by the product package.
#include <vector>
#include <cstring>
#include "../../multioptpy_amd/csrc/logic.h"

namespace {
struct SerialRed {
    template <class F> double sum(int m, F f) const { double a = 0.0; for (int i = 0; i < m; ++i) a += f(i); return a; }
    template <class F> void sum2(int m, F f, double *s0, double *s1) const { double a0 = 0.0, a1 = 0.0; for (int i = 0; i < m; ++i) f(i, a0, a1); *s0 = a0; *s1 = a1; }
    template <class P> int min_index(int m, P p) const { for (int i = 0; i < m; ++i) if (p(i)) return i; return m; }
    template <class F> void for_each(int m, F f) const { for (int i = 0; i < m; ++i) f(i); }
    void sync() const {}
};
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

}
