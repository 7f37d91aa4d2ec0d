// Scalar control logic of the optimizer step, shared by the HIP kernels (one wave per image,
// reductions through a wave reducer) and by the host-side logic tests (serial reducer).
// Everything here follows multioptpy/Optimizer/rsirfo.py and hessian_update.py /
// block_hessian_update.py of the reference; line numbers are quoted per function.
#pragma once
#include <math.h>
#include <stdint.h>
#include "../../include/mopt_hip.h"

#if defined(__HIPCC__)
#define MOPT_HD __host__ __device__
#else
#define MOPT_HD
#endif

namespace mopt {

// ---------------------------------------------------------------------------------------
// Hessian update: every rule is  dH = c_yy y'y'^T + c_uu uu^T + c_AA A A^T + c_As (A s^T + s A^T) + c_ss ss^T
// with u = H s, y' = (possibly Powell-damped) y, A = y' - u  (unscaled; CFD's factor 2 is folded in).
// ---------------------------------------------------------------------------------------
struct UpdateDots {
    double ss, sy, yy;      // s.s, s.y, y.y (original y)
    double syd;             // s.y'  (damped y, == sy when no damping)
    double su;              // s.H s
    double AA, As;          // A.A, A.s   with A = y' - H s
    double zz, zs;          // flowchart only: z = y - H y
};

struct UpdateCoef {
    double c_yy, c_uu, c_AA, c_As, c_ss;
    int applied;            // 0: guards skipped the update (RSIRFO.update_hessian, rsirfo.py:1326-1335)
    int block_sym;          // block rules hand back symm(B + dH) - B (block_hessian_update.py:118,255)
    double mix;             // phi / phi^2 / weight used (diagnostic)
    int branch;             // flowchart: 1 SR1, 2 BFGS, 3 FSB
};

MOPT_HD inline bool rule_needs_Hy(int rule) { return rule == MOPT_RULE_FLOWCHART; }

MOPT_HD inline bool rule_is_dd(int rule) {
    return rule == MOPT_RULE_BLOCK_CFD_FSB_DD || rule == MOPT_RULE_BLOCK_BFGS_DD || rule == MOPT_RULE_BLOCK_FSB_DD ||
           rule == MOPT_RULE_BFGS_DD || rule == MOPT_RULE_CFD_FSB_DD || rule == MOPT_RULE_FSB_DD;
}

MOPT_HD inline bool rule_is_block(int rule) {
    return rule >= MOPT_RULE_BLOCK_CFD_FSB_DD && rule <= MOPT_RULE_BLOCK_BOFILL;
}

// Powell damping with B = I (hessian_update.py:200-242, block_hessian_update.py:565-595).
// Returns theta; y' = theta*y + (1-theta)*s.  theta = 1 means "not active".
MOPT_HD inline double dd_theta(double ss, double sy, double thr) {
    const double mu2 = 0.2;
    if (sy < mu2 * ss) {
        double den = ss - sy;
        double th = (fabs(den) < thr) ? 0.1 : (1.0 - mu2) * ss / den;
        if (th < 0.0) th = 0.0;
        if (th > 1.0) th = 1.0;
        return th;
    }
    return 1.0;
}

MOPT_HD inline double safe_inv1(double m) {   // block_hessian_update.py:12-21 on a 1x1 matrix
    return (m == 0.0) ? 1.0 / (m + 1e-10) : 1.0 / m;
}

struct Acc5 {
    double yy, uu, AA, As, ss;
    MOPT_HD void add(double w, const Acc5 &o) { yy += w * o.yy; uu += w * o.uu; AA += w * o.AA; As += w * o.As; ss += w * o.ss; }
};

MOPT_HD inline Acc5 zero5() { Acc5 a; a.yy = a.uu = a.AA = a.As = a.ss = 0.0; return a; }

// hessian_update.py:35-65
MOPT_HD inline Acc5 std_bfgs(const UpdateDots &d) {
    Acc5 a = zero5();
    if (fabs(d.syd) >= 1e-10) a.yy = 1.0 / d.syd;
    if (fabs(d.su) >= 1e-10) a.uu = -1.0 / d.su;
    return a;
}
// hessian_update.py:67-85 with A_ref = k * A
MOPT_HD inline Acc5 std_sr1(const UpdateDots &d, double k) {
    Acc5 a = zero5();
    double den = k * d.As;
    if (fabs(den) >= 1e-10) a.AA = k * k / den;
    return a;
}
// hessian_update.py:87-104
MOPT_HD inline Acc5 std_psb(const UpdateDots &d) {
    Acc5 a = zero5();
    if (fabs(d.ss) >= 1e-10) {
        a.ss = -d.As / (d.ss * d.ss);
        a.As = 1.0 / d.ss;
    }
    return a;
}
// hessian_update.py:106-130
MOPT_HD inline double std_bofill_const(const UpdateDots &d, double k) {
    double num = (k * d.As) * (k * d.As);
    double den = (k * k * d.AA) * d.ss;
    return (fabs(den) >= 1e-10) ? num / den : 0.0;
}

// block_hessian_update.py:75-118 / :619-641 on (sc*s, sc*y')
MOPT_HD inline Acc5 blk_bfgs(const UpdateDots &d, double sc, bool damped_variant) {
    Acc5 a = zero5();
    if (!(fabs(sc) * sqrt(d.ss) > 1e-8)) return a;
    if (!damped_variant && (sc * sc * d.syd) <= 1e-12) return a;
    a.uu = -(sc * sc) * safe_inv1(sc * sc * d.su);
    a.yy = (sc * sc) * safe_inv1(sc * sc * d.syd);
    return a;
}
// block_hessian_update.py:159-184
MOPT_HD inline Acc5 blk_sr1(const UpdateDots &d, double sc, double k) {
    Acc5 a = zero5();
    a.AA = (k * sc) * (k * sc) * safe_inv1(k * sc * sc * d.As);
    return a;
}
// block_hessian_update.py:120-157
MOPT_HD inline Acc5 blk_psb(const UpdateDots &d, double sc) {
    Acc5 a = zero5();
    if (!(fabs(sc) * sqrt(d.ss) > 1e-8)) return a;
    double ssp = sc * sc * d.ss;
    if (fabs(ssp) >= 1e-8) {
        a.ss = -(sc * sc * d.As) * (sc * sc) / (ssp * ssp);
        a.As = (sc * sc) / ssp;
    }
    return a;
}
// block_hessian_update.py:190-231
MOPT_HD inline double blk_weight(const UpdateDots &d, double k, bool bofill_logic) {
    double num = (k * d.As) * (k * d.As);
    double den = (k * k * d.AA) * d.ss;
    double c = (fabs(den) > 1e-12) ? num / den : 0.0;
    if (c != c) c = 0.0;
    if (c < 0.0) c = 0.0;
    if (c > 1.0) c = 1.0;
    return bofill_logic ? c : sqrt(c);
}

// Coefficients for one (s, y) pair.  `d` must hold the dots computed with the (possibly damped) y'.
MOPT_HD inline UpdateCoef update_coefficients(int rule, const UpdateDots &d, bool require_positive_sy = true) {
    UpdateCoef r;
    r.c_yy = r.c_uu = r.c_AA = r.c_As = r.c_ss = 0.0;
    r.applied = 0; r.block_sym = rule_is_block(rule) ? 1 : 0; r.mix = 0.0; r.branch = 0;
    // RSIRFO.update_hessian guards (rsirfo.py:1322-1335), on the ORIGINAL y
    if (sqrt(d.ss) < 1e-10 || sqrt(d.yy) < 1e-10) return r;
    if (require_positive_sy && d.sy <= 0.0) return r;      // EnhancedRSPRFO.update_hessian has no such guard (rsprfo.py:1205-1209)
    r.applied = 1;
    Acc5 a = zero5();
    switch (rule) {
    case MOPT_RULE_BFGS: case MOPT_RULE_BFGS_DD:
        a = std_bfgs(d); break;
    case MOPT_RULE_SR1:
        a = std_sr1(d, 1.0); break;
    case MOPT_RULE_PSB:
        a = std_psb(d); break;
    case MOPT_RULE_FSB: case MOPT_RULE_FSB_DD: case MOPT_RULE_CFD_FSB: case MOPT_RULE_CFD_FSB_DD: {
        double k = (rule == MOPT_RULE_CFD_FSB || rule == MOPT_RULE_CFD_FSB_DD) ? 2.0 : 1.0;
        double phi = sqrt(std_bofill_const(d, k));
        a.add(1.0 - phi, std_bfgs(d));
        a.add(phi, std_sr1(d, k));
        r.mix = phi;
    } break;
    case MOPT_RULE_BOFILL: case MOPT_RULE_CFD_BOFILL: case MOPT_RULE_PCFD_BOFILL: {
        double k = (rule == MOPT_RULE_BOFILL) ? 1.0 : 2.0;
        double c = std_bofill_const(d, k);
        a.add(1.0 - c, std_psb(d));
        a.add(c, std_sr1(d, k));
        r.mix = c;
        if (rule == MOPT_RULE_PCFD_BOFILL) {
            // hessian_update.py:321-343: + sum_ij (o_j^T D o_i)(o_i o_j^T + o_j o_i^T) = 2 P D P, P = I - ss^T/(s.s);
            // P s = 0, P A = A - (A.s/s.s) s  ->  only the AA part of D survives.
            double t = d.As / d.ss;
            double aa = a.AA;
            a.AA += 2.0 * aa;
            a.As += -2.0 * aa * t;
            a.ss += 2.0 * aa * t * t;
        }
    } break;
    case MOPT_RULE_MSP: {
        double den = sqrt(d.AA) * sqrt(d.ss);
        double c = 0.0;
        if (den >= 1e-10) {
            c = d.As / den;
            if (c < -1.0) c = -1.0;
            if (c > 1.0) c = 1.0;
        }
        double phi = 1.0 - c * c;
        a.add(phi, std_psb(d));
        a.add(1.0 - phi, std_sr1(d, 1.0));
        r.mix = phi;
    } break;
    case MOPT_RULE_FLOWCHART: {
        // hessian_update.py:163-194 (z = y - H y as written there)
        double zden = sqrt(d.ss) * sqrt(d.zz);
        if (fabs(zden) < 1e-10) zden += 1e-10;
        double zs = d.zs / zden;
        double yden = sqrt(d.ss) * sqrt(d.yy);
        if (fabs(yden) < 1e-10) yden += 1e-10;
        double ys = d.sy / yden;
        if (zs < -0.1) { a = std_sr1(d, 1.0); r.branch = 1; }
        else if (ys > 0.1) { a = std_bfgs(d); r.branch = 2; }
        else {
            double phi = sqrt(std_bofill_const(d, 1.0));
            a.add(1.0 - phi, std_bfgs(d));
            a.add(phi, std_sr1(d, 1.0));
            r.mix = phi; r.branch = 3;
        }
    } break;
    case MOPT_RULE_BLOCK_BFGS:
        a = blk_bfgs(d, 1.0, false); break;
    case MOPT_RULE_BLOCK_BFGS_DD:
        a = blk_bfgs(d, 1.0, true); break;
    case MOPT_RULE_BLOCK_FSB: case MOPT_RULE_BLOCK_FSB_DD: case MOPT_RULE_BLOCK_CFD_FSB: case MOPT_RULE_BLOCK_CFD_FSB_DD:
    case MOPT_RULE_BLOCK_BOFILL: case MOPT_RULE_BLOCK_CFD_BOFILL: {
        bool cfd = (rule == MOPT_RULE_BLOCK_CFD_FSB || rule == MOPT_RULE_BLOCK_CFD_FSB_DD || rule == MOPT_RULE_BLOCK_CFD_BOFILL);
        bool bof = (rule == MOPT_RULE_BLOCK_BOFILL || rule == MOPT_RULE_BLOCK_CFD_BOFILL);
        double k = cfd ? 2.0 : 1.0;
        double w = blk_weight(d, k, bof || cfd);
        a.add(w, blk_sr1(d, 1.0, k));
        a.add(1.0 - w, bof ? blk_psb(d, 1.0) : blk_bfgs(d, 1.0, false));
        r.mix = w;
    } break;
    case MOPT_RULE_BLOCK_FSB_WEIGHTED: case MOPT_RULE_BLOCK_CFD_FSB_WEIGHTED:
    case MOPT_RULE_BLOCK_BOFILL_WEIGHTED: case MOPT_RULE_BLOCK_CFD_BOFILL_WEIGHTED: {
        bool cfd = (rule == MOPT_RULE_BLOCK_CFD_FSB_WEIGHTED || rule == MOPT_RULE_BLOCK_CFD_BOFILL_WEIGHTED);
        bool bof = (rule == MOPT_RULE_BLOCK_BOFILL_WEIGHTED || rule == MOPT_RULE_BLOCK_CFD_BOFILL_WEIGHTED);
        double k = cfd ? 2.0 : 1.0;
        double w = blk_weight(d, k, bof || cfd);
        a.add(1.0, blk_sr1(d, w, k));
        a.add(1.0, bof ? blk_psb(d, 1.0 - w) : blk_bfgs(d, 1.0 - w, false));
        r.mix = w;
    } break;
    default:
        r.applied = 0;
        break;
    }
    r.c_yy = a.yy; r.c_uu = a.uu; r.c_AA = a.AA; r.c_As = a.As; r.c_ss = a.ss;
    return r;
}

// ---------------------------------------------------------------------------------------
// Trust radius (rsirfo.py:660-887)
// ---------------------------------------------------------------------------------------
MOPT_HD inline double dmin(double a, double b) { return a < b ? a : b; }
MOPT_HD inline double dmax(double a, double b) { return a > b ? a : b; }

MOPT_HD inline void adjust_trust_radius_adaptive(const mopt_rsirfo_params &p, mopt_rsirfo_state &st,
                                                 double actual, double predicted, double min_eig) {
    if (fabs(predicted) < 1e-10) return;
    double ratio = actual / predicted;
    double ae = fabs(min_eig);
    double cf = (ae > 1e-6) ? dmin(p.max_curvature_factor, 1.0 / dmax(ae, 0.1)) : 1.5;
    if (p.saddle_order > 0 && min_eig < -1e-6) cf *= p.negative_curvature_safety;
    double tr = st.trust_radius;
    if (ratio > 0.75) {
        double inc = dmin(1.5 * cf, p.max_curvature_factor);
        tr = dmin(tr * inc, p.trust_radius_max);
    } else if (ratio > 0.5) {
        double inc = dmin(1.1 * cf, 1.5);
        tr = dmin(tr * inc, p.trust_radius_max);
    } else if (ratio > 0.25) {
        if (cf > 1.2) tr = dmin(tr * 1.05, p.trust_radius_max);
    } else if (ratio > 0.1) {
        tr = dmax(tr * 0.5, p.trust_radius_min);
    } else {
        tr = dmax(tr * 0.25, p.trust_radius_min);
    }
    // np.clip(x, lo, hi) = minimum(maximum(x, lo), hi)
    tr = dmin(dmax(tr, p.trust_radius_min), p.trust_radius_max);
    st.trust_radius = tr;
}

MOPT_HD inline void adjust_trust_radius(const mopt_rsirfo_params &p, mopt_rsirfo_state &st, double actual,
                                        double predicted, double min_eig, double grad_norm) {
    if (p.use_adaptive_trust_radius) {
        if (grad_norm < p.adaptive_trust_gradient_norm_threshold) {
            adjust_trust_radius_adaptive(p, st, actual, predicted, min_eig);
            return;
        }
    }
    if (fabs(predicted) < 1e-10) return;
    double ratio = actual / predicted;
    if (ratio > p.good_step_threshold)
        st.trust_radius = dmin(st.trust_radius * p.trust_radius_increase_factor, p.trust_radius_max);
    else if (ratio < p.poor_step_threshold)
        st.trust_radius = dmax(st.trust_radius * p.trust_radius_decrease_factor, p.trust_radius_min);
}

MOPT_HD inline void push3(double *arr, int32_t &n, double v) {   // "if len >= 3: pop(0); append"
    if (n >= 3) { arr[0] = arr[1]; arr[1] = arr[2]; n = 2; }
    arr[n++] = v;
}

// ---------------------------------------------------------------------------------------
// Secular equation and alpha search (rsirfo.py:924-1313, 1374-1575, 1688-1715).
// R is a reducer: R.sum(m, f) = sum_i f(i); R.min_index(m, pred) = first i with pred(i) or m;
// R.for_each(m, f) runs f(i) for every i (lane-strided on the device).
// ---------------------------------------------------------------------------------------
// Quotient inside the secular sums.  On the device: hardware reciprocal estimate + two Newton steps (|den| >= 1e-30 is a normal number;
// ~1 ulp).  The sums only steer Newton / bisection iterates whose stopping tests are 1e-10-relative, and the step itself is formed
// with IEEE division (solve_rfo); on the host (logic tests): IEEE division.  One f64 division is ~30 dependent instructions on gfx950,
// and k_rfo_scalar spent 57 of its 72 us in ~40 secular evaluations of 10 divisions per lane.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double sec_quot(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return a * r;
}
#else
inline double sec_quot(double a, double b) { return a / b; }
#endif

MOPT_HD inline double safe_den(double d, double floor_) {
    if (fabs(d) < floor_) {
        double sg = (d > 0.0) ? 1.0 : ((d < 0.0) ? -1.0 : 0.0);
        d = sg * floor_;
    }
    if (d == 0.0) d = floor_;
    return d;
}

// The secular function divides every term three times (g/alpha, w/alpha, g'^2/(w' - lambda)); alpha is fixed inside one
// solve, so with scratch (cg2, cw: m doubles each, or null) the two alpha quotients are formed once per solve and every
// evaluation is one division per term -- the same IEEE operations on the same operands, bitwise the same sums.
template <class R>
MOPT_HD double secular_f(const R &red, double lmd, const double *w, const double *gt, int m, double alpha, const double *cg2 = nullptr,
                         const double *cw = nullptr) {
    if (cg2) return lmd + red.sum(m, [=](int i) { return sec_quot(cg2[i], safe_den(cw[i] - lmd, 1e-30)); });
    return lmd + red.sum(m, [=](int i) {
        double gp = gt[i] / alpha;
        return (gp * gp) / safe_den(w[i] / alpha - lmd, 1e-30);
    });
}

template <class R>
MOPT_HD double secular_fp(const R &red, double lmd, const double *w, const double *gt, int m, double alpha, const double *cg2 = nullptr,
                          const double *cw = nullptr) {
    if (cg2) return 1.0 + red.sum(m, [=](int i) { double dd = safe_den(cw[i] - lmd, 1e-30); return sec_quot(cg2[i], dd * dd); });
    return 1.0 + red.sum(m, [=](int i) {
        double gp = gt[i] / alpha;
        double dd = safe_den(w[i] / alpha - lmd, 1e-30);
        return (gp * gp) / (dd * dd);
    });
}

// _solve_secular_more_sorensen + _solve_secular_safeguarded (rsirfo.py:1505-1575, 1374-1503)
template <class R>
MOPT_HD double solve_secular(const R &red, const double *w, const double *gt, int m, double alpha, int *iters, double *cg2 = nullptr,
                             double *cw = nullptr) {
    if (iters) *iters = 0;
    if (cg2) {
        red.sync();                                             // (readers of the previous solve are done)
        red.for_each(m, [=](int i) { double gp = gt[i] / alpha; cg2[i] = gp * gp; cw[i] = w[i] / alpha; });
        red.sync();
    }
    double gnorm_sq = cg2 ? red.sum(m, [=](int i) { return cg2[i]; }) : red.sum(m, [=](int i) { double gp = gt[i] / alpha; return gp * gp; });
    int first = cg2 ? red.min_index(m, [=](int i) { return cg2[i] > 1e-20; })
                    : red.min_index(m, [=](int i) { double gp = gt[i] / alpha; return gp * gp > 1e-20; });
    if (first >= m) return w[0] / alpha;                       // hard case
    double pole = w[first] / alpha;
    double guess = 0.5 * (pole - sqrt(dmax(0.0, pole * pole + 4.0 * gnorm_sq)));
    double b = pole, a = guess;
    double fa = secular_f(red, a, w, gt, m, alpha, cg2, cw);
    double gnorm = sqrt(gnorm_sq);
    int limit = 10;
    while (fa > 0.0 && limit > 0) {
        a = a - dmax(dmax(gnorm, fabs(a) * 0.1), 1e-8);
        fa = secular_f(red, a, w, gt, m, alpha, cg2, cw);
        --limit;
    }
    if (fa > 0.0) return guess;
    double lk = guess;
    if (lk <= a || lk >= b) lk = (a + b) / 2.0;
    const double tol = 1e-10 * fabs(pole) + 1e-12;
    for (int it = 0; it < 250; ++it) {
        double fl, fpl = 0.0;
        if (cg2) {                                             // f and f' in one pass over the terms (same operations, two sums)
            double s0, s1;
            red.sum2(m, [=](int i, double &a0, double &a1) { double dd = safe_den(cw[i] - lk, 1e-30); const double q = sec_quot(cg2[i], dd); a0 += q; a1 += sec_quot(q, dd); }, &s0, &s1);
            fl = lk + s0; fpl = 1.0 + s1;
        } else fl = secular_f(red, lk, w, gt, m, alpha, cg2, cw);
        if (fabs(fl) < tol) { if (iters) *iters = it + 1; return lk; }
        if (!cg2) fpl = secular_fp(red, lk, w, gt, m, alpha, cg2, cw);
        double dn = (fabs(fpl) > 1e-20) ? -fl / fpl : 0.0;
        double ln = lk + dn;
        double lb = (a + b) / 2.0;
        double nxt = (dn != 0.0 && ln > a && ln < b) ? ln : lb;
        if (fl > 0.0) b = lk; else a = lk;
        lk = nxt;
        if (fabs(b - a) < tol) { if (iters) *iters = it + 1; return (a + b) / 2.0; }
    }
    if (iters) *iters = 250;
    return (a + b) / 2.0;
}

// step of solve_rfo for a given lambda_aug and its squared norm in one pass (every index is visited by exactly one lane, which squares
// what it just wrote)
template <class R>
MOPT_HD double form_rfo_step(const R &red, const double *w, const double *gt, int m, double alpha, double lam, double *step) {
    const double n2 = red.sum(m, [=](int i) {
        double den = safe_den(w[i] / alpha - lam, 1e-20);
        double v = -(gt[i] / alpha) / den;
        step[i] = v;
        return v * v;
    });
    red.sync();
    return n2;
}

// solve_rfo (rsirfo.py:1688-1715): writes step[0..m), returns ||step||^2 via *norm2; returns lambda_aug
template <class R>
MOPT_HD double solve_rfo(const R &red, const double *w, const double *gt, int m, double alpha, double *step,
                         double *norm2, int *iters, double *cg2 = nullptr, double *cw = nullptr) {
    double lam = solve_secular(red, w, gt, m, alpha, iters, cg2, cw);
    *norm2 = form_rfo_step(red, w, gt, m, alpha, lam, step);
    return lam;
}

// get_step_derivative (rsirfo.py:1250-1313)
template <class R>
MOPT_HD double step_derivative(const R &red, double alpha, const double *w, const double *gt, int m, double lam) {
    int nvalid = (int)red.sum(m, [=](int i) {
        double d = w[i] - lam * alpha;
        if (fabs(d) < 1e-8) { double sg = (d > 0.0) ? 1.0 : ((d < 0.0) ? -1.0 : 0.0); d = sg * 1e-8; }
        return (fabs(d * d * d) > 1e-10) ? 1.0 : 0.0;
    });
    if (nvalid == 0) return 1e-8;
    double sum = red.sum(m, [=](int i) {
        double d = w[i] - lam * alpha;
        if (fabs(d) < 1e-8) { double sg = (d > 0.0) ? 1.0 : ((d < 0.0) ? -1.0 : 0.0); d = sg * 1e-8; }
        double d3 = d * d * d;
        if (!(fabs(d3) > 1e-10)) return 0.0;
        double t = (gt[i] * gt[i]) / d3;
        if (fabs(t) > 1e20) t = (t > 0.0 ? 1e20 : -1e20);
        return t;
    });
    double res = 2.0 * lam * sum;
    if (!isfinite(res) || fabs(res) > 1e20) {
        if (res != 0.0) res = (res > 0.0 ? 1.0 : (res < 0.0 ? -1.0 : NAN)) * 1e20; else res = 1e-8;
    }
    return res;
}

}  // namespace mopt

// scipy.optimize.brentq state machine is in brent.h (needs the objective as a callable)
#include "brent.h"

namespace mopt {

struct RsResult {
    int branch;          // 0 within, 1 brent, 2 newton, 3 steepest-descent fallback inside the loop
    double alpha;
    double lambda_aug0;
    double initial_norm;
    int secular_iters0;
};

template <class R>
MOPT_HD RsResult rs_step_rest(const R &red, const mopt_rsirfo_params &p, double trust_radius, const double *w, const double *gt, int m, double *step,
                              double *best, double *cg2, double *cw, double lam0, double n2_0, int its0, const double *pre_lo, const double *pre_hi);

// get_rs_step + compute_rsprfo_step (rsirfo.py:924-1248) in the eigenbasis.
// w, gt: the m kept eigenvalues / transformed gradient; step: output (m); best: scratch (m).
template <class R>
MOPT_HD RsResult rs_step(const R &red, const mopt_rsirfo_params &p, double trust_radius, const double *w,
                         const double *gt, int m, double *step, double *best, double *cg2 = nullptr, double *cw = nullptr) {
    double n2;
    int its = 0;
    const double lam0 = solve_rfo(red, w, gt, m, p.alpha0, step, &n2, &its, cg2, cw);
    return rs_step_rest(red, p, trust_radius, w, gt, m, step, best, cg2, cw, lam0, n2, its, nullptr, nullptr);
}

// Everything behind the first solve (lambda_aug, ||step||^2 and iteration count of solve_rfo(alpha0) are handed in; `step` holds that
// step).  pre_lo / pre_hi: ||step||^2 of the two bracket solves (alpha = 1e-6 and alpha_max) when the caller has formed them already --
// k_rfo_scalar runs them on two further waves next to the first solve; they are independent of it, and the step they would leave in
// `step` is formed again by the first micro cycle.
template <class R>
MOPT_HD RsResult rs_step_rest(const R &red, const mopt_rsirfo_params &p, double trust_radius, const double *w, const double *gt, int m, double *step,
                              double *best, double *cg2, double *cw, double lam0, double n2_0, int its0, const double *pre_lo, const double *pre_hi) {
    RsResult rr;
    rr.branch = 0; rr.alpha = p.alpha0;
    rr.lambda_aug0 = lam0;
    rr.secular_iters0 = its0;
    rr.initial_norm = sqrt(n2_0);
    const double R_ = trust_radius;
    if (rr.initial_norm <= R_) return rr;

    // --- compute_rsprfo_step ---
    double alpha = p.alpha0;
    {
        double n2lo, n2hi;
        if (pre_lo) n2lo = *pre_lo; else solve_rfo(red, w, gt, m, 1e-6, step, &n2lo, nullptr, cg2, cw);
        double nlo = sqrt(n2lo);
        double obj_lo = nlo * nlo - R_ * R_;
        if (pre_hi) n2hi = *pre_hi; else solve_rfo(red, w, gt, m, p.alpha_max, step, &n2hi, nullptr, cg2, cw);
        double nhi = sqrt(n2hi);
        double obj_hi = nhi * nhi - R_ * R_;
        if (obj_lo * obj_hi < 0.0) {
            double root = 0.0;
            auto obj = [&](double a) {
                double q;
                solve_rfo(red, w, gt, m, a, step, &q, nullptr, cg2, cw);
                return q - R_ * R_;
            };
            int ok = brentq(obj, 1e-6, p.alpha_max, 1e-6, 1e-6, 50, &root);
            if (ok == 0) {
                double q;
                solve_rfo(red, w, gt, m, root, step, &q, nullptr, cg2, cw);
                double nrm = sqrt(q);
                if (fabs(nrm - R_) < p.step_norm_tolerance) { rr.branch = 1; rr.alpha = root; return rr; }
                alpha = root;
            } else {
                alpha = p.alpha0;
            }
        }
    }
    rr.branch = 2;
    // Newton refinement (rsirfo.py:1080-1246)
    const int MAXH = 64;
    double hist[MAXH];
    int hc = 0;
    bool have_best = false;
    double best_diff = INFINITY;
    bool hasL = false, hasR = false;
    double aL = 0.0, aR = 0.0;
    int max_mc = p.max_micro_cycles < MAXH ? p.max_micro_cycles : MAXH;
    bool broke = false;
    for (int mu = 0; mu < max_mc; ++mu) {
        double q;
        double lam;
        if (mu == 0 && alpha == p.alpha0) {
            // the first micro cycle repeats the very first solve (same alpha, same inputs: the reference computes the same numbers
            // twice, rsirfo.py:945 and 1085): its lambda is kept, only the step -- overwritten by the two bracket solves -- is formed again
            lam = rr.lambda_aug0;
            q = form_rfo_step(red, w, gt, m, alpha, lam, step);
        } else {
            lam = solve_rfo(red, w, gt, m, alpha, step, &q, nullptr, cg2, cw);
        }
        double nrm = sqrt(q);
        double diff = fabs(nrm - R_);
        if (diff < best_diff) {
            red.for_each(m, [=](int i) { best[i] = step[i]; });
            red.sync();
            have_best = true;
            best_diff = diff;
        }
        double objv = nrm * nrm - R_ * R_;
        if (objv < 0.0 && (!hasL || alpha > aL)) { aL = alpha; hasL = true; }
        else if (objv > 0.0 && (!hasR || alpha < aR)) { aR = alpha; hasR = true; }
        if (fabs(objv) < 1e-8 || diff < p.step_norm_tolerance) { broke = true; break; }
        if (hc < max_mc) hist[hc++] = nrm;
        double dd = step_derivative(red, alpha, w, gt, m, lam);
        double anew;
        if (fabs(dd) < 1e-10) {
            if (hasL && hasR) anew = (aL + aR) / 2.0;
            else if (objv > 0.0) anew = dmax(alpha / 2.0, 1e-6);
            else anew = dmin(alpha * 2.0, p.alpha_max);
        } else {
            double raw = -objv / dd;
            double astep = dmin(dmax(raw, -p.alpha_step_max), p.alpha_step_max);
            anew = alpha + astep;
            if (hasL && hasR) anew = dmax(dmin(anew, aR * 0.99), aL * 1.01);
        }
        alpha = dmin(dmax(anew, 1e-6), p.alpha_max);
        if (alpha == p.alpha_max || alpha == 1e-6) { broke = true; break; }
        if (hc >= 3) {
            int i = hc - 1;
            if (fabs(hist[i] - hist[i - 1]) < 1e-6 && fabs(hist[i - 1] - hist[i - 2]) < 1e-6) { broke = true; break; }
        }
    }
    if (!broke) {
        // for-else: micro cycles exhausted (rsirfo.py:1213-1246)
        double bn = have_best ? sqrt(red.sum(m, [=](int i) { return best[i] * best[i]; })) : 0.0;
        if (have_best && fabs(bn - R_) < p.step_norm_tolerance * 1.1) {
            red.for_each(m, [=](int i) { step[i] = best[i]; });
        } else {
            double gn = sqrt(red.sum(m, [=](int i) { return gt[i] * gt[i]; }));
            if (gn > 1e-10) red.for_each(m, [=](int i) { step[i] = -gt[i] / gn * R_; });
            else red.for_each(m, [=](int i) { step[i] = 0.0; });
            rr.branch = 3;
        }
        red.sync();
    }
    rr.alpha = alpha;
    return rr;
}

}  // namespace mopt
