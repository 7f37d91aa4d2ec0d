// EnhancedRSPRFO's scalar stages as device functions (k_prfo.hip describes them), shared with k_small.hip.
// Reference: multioptpy/Optimizer/rsprfo.py:287-355, 357-432, 514-711, 908-1071, 1097-1188.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "logic.h"
#include "rfo_scalar_body.hpp"      // WaveRedLocal

namespace mopt {

// _process_previous_step (rsprfo.py:908-962) behind the three dot products g_prev . s_prev, s_prev . Hp s_prev, s_prev . s_prev: one thread
__device__ inline void prfo_prev_logic(const mopt_rsprfo_params &p, mopt_rsprfo_state &st, double gs, double sHs, double ss, double B_e, double pre_move_norm) {
    const double actual = B_e - st.prev_energy;
    st.last_actual = actual;
    const double psn = (pre_move_norm >= 0.0) ? pre_move_norm : sqrt(ss);
    const double pred_red = -(gs + 0.5 * sHs);
    double ratio;
    if (fabs(pred_red) < 1e-14) ratio = (fabs(actual) < 1e-14) ? 1.0 : 0.0;
    else { ratio = actual / pred_red; if (!isfinite(ratio)) ratio = 0.0; }
    st.last_ratio = ratio;
    const bool at_boundary = psn >= st.trust_radius * 0.95;
    if (p.adapt_trust_radius) {
        if (ratio < p.eta_2) st.trust_radius = dmax(p.gamma_1 * psn, p.trust_radius_min);
        else if (ratio > p.eta_3 && at_boundary) st.trust_radius = dmin(p.gamma_2 * st.trust_radius, p.trust_radius_max);
    }
    st.step_accepted = 1;
    if (p.step_rejection_enabled && ratio < p.rejection_threshold) {
        st.consecutive_rejections += 1;
        if (st.consecutive_rejections >= p.max_consecutive_rejections) st.consecutive_rejections = 0;
        else st.step_accepted = 0;
    } else {
        st.consecutive_rejections = 0;
    }
}

// Smallest eigenvalue of the arrowhead matrix [[diag(d), b], [b^T, 0]] (mode > 0: largest), d_i = lam_i/alpha, b_i = g_i/alpha,
// restricted to the index list idx[0..m).  Root of  mu + sum b_i^2/(d_i - mu) = 0  left of every pole (right, mirrored).
// Safeguarded Newton to full double precision.  step_i = b_i/(d_i - mu) (= -v_i/nu of rsprfo.py:1160-1166).
// d_k = sg lam_k / alpha and b_k^2 = (g_k / alpha)^2 are formed once per call into the scratch arrays sd, sb2 (m entries each): an evaluation
// of f and f' then costs one division per entry and ONE pass with two sums (it was six divisions and two passes).
template <class R>
__device__ inline double arrow_extreme(const R &red, const double *lam, const double *g, const int *idx, int m, double alpha,
                                       bool maximise, double *sd, double *sb2) {
    const double sg = maximise ? -1.0 : 1.0;      // largest root of (D, b) = - smallest root of (-D, b)
    if (m == 1) {
        // one pole (the usual maximisation subspace of a first-order saddle search): mu^2 - D mu - b^2 = 0 in closed form, written so that
        // no cancellation occurs for either sign of D
        const double D = sg * lam[idx[0]] / alpha, b = g[idx[0]] / alpha, rt = sqrt(D * D + 4.0 * b * b);
        const double mu = (D > 0.0) ? -2.0 * b * b / (D + rt) : 0.5 * (D - rt);
        return sg * mu;
    }
    double dmin_ = INFINITY, bbl = 0.0;
    for (int k = threadIdx.x & 63; k < m; k += WAVE) {
        const double d = sg * lam[idx[k]] / alpha, b = g[idx[k]] / alpha;
        sd[k] = d; sb2[k] = b * b;
        dmin_ = fmin(dmin_, d); bbl += b * b;
    }
    red.sync();
    dmin_ = -wave_max(-dmin_);
    const double bb = wave_sum(bbl);
    if (bb == 0.0) return sg * fmin(dmin_, 0.0);       // decoupled: eigenvalues are d_i and 0
    auto ffp = [&](double mu, double *fv, double *fpv) {
        double s0, s1;
        red.sum2(m, [=](int k, double &a0, double &a1) { const double r = 1.0 / (sd[k] - mu), t = sb2[k] * r; a0 += t; a1 += t * r; }, &s0, &s1);
        *fv = mu + s0; *fpv = 1.0 + s1;
    };
    auto f = [&](double mu) { double a, b; ffp(mu, &a, &b); return a; };
    double t0 = 0.5 * (fabs(dmin_) + sqrt(dmin_ * dmin_ + 4.0 * bb));
    double lo = dmin_ - 1.01 * t0 - 1e-300, hi = dmin_;
    for (int it = 0; it < 60 && !(f(lo) < 0.0); ++it) lo = dmin_ - 2.0 * (dmin_ - lo);
    double mu = 0.5 * (lo + hi);
    if (mu >= hi) mu = lo;
    for (int it = 0; it < 200; ++it) {
        double fv, fpv;
        ffp(mu, &fv, &fpv);
        if (fv == 0.0) break;
        if (fv > 0.0) hi = mu; else lo = mu;
        double nx = mu - fv / fpv;
        if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
        if (nx == mu || !(hi - lo > 4.0 * 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi)))) { mu = nx; break; }
        mu = nx;
    }
    red.sync();                                               // (the scratch arrays are free again)
    return sg * mu;
}

// FUSED = false: the single wave of k_prfo_scalar.  FUSED = true: called by ONE wave of a larger workgroup (k_small.hip): barriers are wave-level.
// sm: 7 * n doubles + 4 * n ints of LDS
template <bool FUSED>
__device__ __forceinline__ void prfo_scalar_body(int n, const mopt_rsprfo_params *params, mopt_rsprfo_state *state, const double *W, const int *perm, const int *einfo,
                                                 const double *gt_all, const double *ov_all, int have_ts, double *coef_out, int *ts_row_out, double *sm) {
    auto SYNC = [] { if constexpr (FUSED) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); };
    double *w = sm;              // [n] eigenvalues after shifting, ascending
    double *gt = w + n;          // [n] gradient in that basis
    double *ov = gt + n;         // [n] |overlap| with the tracked TS vector
    double *stp = ov + n;        // [n] step in the eigenbasis
    double *best = stp + n;      // [n]
    double *key = best + n;      // [n] scratch (unsorted shifted values; later d_k of arrow_extreme)
    double *sb2 = key + n;       // [n] scratch (b_k^2 of arrow_extreme)
    int *vidx = (int *)(sb2 + n);        // [n] Vt row of sorted position k
    int *ismax = vidx + n;               // [n]
    int *mxi = ismax + n;                // [n] max index list (first nmax) ...
    int *mni = mxi + n;                  // [n] min index list
    const int lane = threadIdx.x & 63;
    using RedT = typename std::conditional<FUSED, WaveRedLocal, WaveRed>::type;
    const RedT red;
    const mopt_rsprfo_params p = params[0];
    mopt_rsprfo_state st = state[0];
    const bool bad = (einfo[1] & 1) != 0;
    // ---- first eigh (ascending) + shifting (rsprfo.py:287-355); "eigh again" of V diag V^T = same vectors, re-sorted values
    for (int k = lane; k < n; k += WAVE) { key[k] = bad ? 1.0 : W[k]; }
    SYNC();
    bool shifted = false;
    if (p.hessian_shift_enabled) {
        if (p.saddle_order == 0) {
            const double mn = key[0];
            if (mn < p.min_positive_eigval) {
                const double sh = p.min_positive_eigval - mn;
                for (int k = lane; k < n; k += WAVE) key[k] = key[k] + sh;
                shifted = true;
            }
        } else {
            int any = 0;
            for (int k = lane; k < n; k += WAVE) {
                if (k < p.saddle_order) { if (key[k] > p.min_negative_eigval) { key[k] = p.min_negative_eigval; any = 1; } }
                else if (key[k] < 1e-6) { key[k] = p.min_positive_eigval; any = 1; }
            }
            if constexpr (FUSED) shifted = __ballot(any != 0) != 0ull; else shifted = __syncthreads_or(any) != 0;
        }
    }
    SYNC();
    // stable rank sort of the shifted values
    for (int k = lane; k < n; k += WAVE) {
        const double kv = key[k];
        int r = 0;
        for (int j = 0; j < n; ++j) { const double kj = key[j]; r += (kj < kv || (kj == kv && j < k)) ? 1 : 0; }
        w[r] = kv;
        const int row = bad ? k : perm[k];
        vidx[r] = row;
        gt[r] = gt_all[row];
        ov[r] = have_ts ? fabs(ov_all[row]) : 0.0;
    }
    SYNC();
    // ---- mode selection (rsprfo.py:964-1071); sorted_indices == identity here
    int nmax = 0;
    if (lane == 0) {
        for (int k = 0; k < n; ++k) ismax[k] = 0;
        const int so = p.saddle_order < n ? p.saddle_order : n;
        if (so > 0) {
            int first = -1;          // the tracked mode, listed first in max_indices
            if (!p.mode_following_enabled) {
                first = -2;
            } else if (st.ts_mode_idx < 0 || !have_ts) {
                st.ts_mode_idx = 0;
                first = -2;
            } else {
                int bi = 0; double bo = ov[0];
                for (int k = 1; k < n; ++k) if (ov[k] > bo) { bo = ov[k]; bi = k; }
                if (bo > p.overlap_threshold) { st.ts_mode_idx = bi; first = bi; }
                else if (p.eigvec_following) {
                    int bsel = -1; double bw_ = -1.0;
                    for (int k = 0; k < n; ++k) if (ov[k] > p.mixing_threshold) {
                        const double wt = ov[k] * ov[k] * (w[k] < 0.0 ? 1.0 : 0.1);
                        if (wt > bw_) { bw_ = wt; bsel = k; }
                    }
                    if (bsel < 0) { st.ts_mode_idx = 0; first = -2; }
                    else { st.ts_mode_idx = bsel; first = bsel; }
                } else { st.ts_mode_idx = 0; first = -2; }
            }
            if (first == -2) { for (int k = 0; k < so; ++k) { ismax[k] = 1; mxi[nmax++] = k; } }
            else {
                ismax[first] = 1; mxi[nmax++] = first;
                for (int k = 0; k < n && nmax < so; ++k) if (k != first) { ismax[k] = 1; mxi[nmax++] = k; }
            }
        }
        int c = 0;
        for (int k = 0; k < n; ++k) if (!ismax[k]) mni[c++] = k;
        best[0] = (double)nmax;
        best[1] = (double)st.ts_mode_idx;
    }
    SYNC();
    nmax = (int)best[0];
    st.ts_mode_idx = (int)best[1];
    const int nmin = n - nmax;
    SYNC();
    // ---- micro cycles (rsprfo.py:514-662)
    const double gnorm = st.gradient_norm;       // |projected gradient| written by the host wrapper
    double R = st.trust_radius;
    if (p.adaptive_trust_enabled && gnorm < p.gradient_scaling_threshold) {
        double r = p.gradient_trust_coupling * gnorm / p.gradient_scaling_threshold * p.trust_radius_max;
        r = dmax(r, p.trust_radius_min);
        R = dmin(r, st.trust_radius);
    }
    const double Reff = R;
    double alpha = p.alpha0;
    bool have_best = false, converged = false, done = false;
    double best_diff = INFINITY;
    double hist[64];
    int hc = 0;
    const int maxmc = p.max_micro_cycles < 64 ? p.max_micro_cycles : 64;
    auto use_best = [&]() {
        const double bn = sqrt(red.sum(n, [=](int k) { return best[k] * best[k]; }));
        const double sc = (bn > Reff) ? Reff / bn : 1.0;
        for (int k = lane; k < n; k += WAVE) stp[k] = (bn > Reff) ? best[k] * sc : best[k];
        SYNC();
    };
    for (int mu = 0; mu < maxmc && !done; ++mu) {
        double emax = 0.0, emin = 0.0;
        for (int k = lane; k < n; k += WAVE) stp[k] = 0.0;
        SYNC();
        if (nmax > 0) {
            emax = arrow_extreme(red, w, gt, mxi, nmax, alpha, true, key, sb2);
            for (int k = lane; k < nmax; k += WAVE) { const int i = mxi[k]; stp[i] = (gt[i] / alpha) / (w[i] / alpha - emax); }
        }
        if (nmin > 0) {
            emin = arrow_extreme(red, w, gt, mni, nmin, alpha, false, key, sb2);
            for (int k = lane; k < nmin; k += WAVE) { const int i = mni[k]; stp[i] = (gt[i] / alpha) / (w[i] / alpha - emin); }
        }
        SYNC();
        const double nmx2 = red.sum(nmax, [=](int k) { double s = stp[mxi[k]]; return s * s; });
        const double nmn2 = red.sum(nmin, [=](int k) { double s = stp[mni[k]]; return s * s; });
        const double nrm = sqrt(red.sum(n, [=](int k) { return stp[k] * stp[k]; }));
        const double smaxn = sqrt(nmx2), sminn = sqrt(nmn2);
        if (hc < 64) hist[hc++] = nrm;
        const double diff = fabs(nrm - Reff);
        if (diff < best_diff) {
            for (int k = lane; k < n; k += WAVE) best[k] = stp[k];
            SYNC();
            have_best = true; best_diff = diff;
        }
        if (nrm <= Reff) { use_best(); converged = false; done = true; break; }            // break -> "did not converge" exit (:594-596, 652-660)
        if (nrm > 0.0 && diff / nrm < p.micro_cycle_rtol) {
            for (int k = lane; k < n; k += WAVE) stp[k] = stp[k] * (Reff / nrm);
            SYNC();
            converged = true; done = true; break;
        }
        if (hc >= 3 && fabs(hist[hc - 1] - hist[hc - 2]) < p.micro_cycle_atol && fabs(hist[hc - 2] - hist[hc - 3]) < p.micro_cycle_atol) {
            use_best(); converged = true; done = true; break;
        }
        // _compute_alpha_step (:664-711)
        const double eps = 1e-12;
        double dd = 0.0;
        if (nmax > 0 && smaxn > eps) {
            const double sf = 1.0 + smaxn * smaxn * alpha;
            if (fabs(sf) > eps) {
                const double q = red.sum(nmax, [=](int k) { const int i = mxi[k]; double den = w[i] - emax * alpha;
                    if (fabs(den) < eps) den = (den > 0.0 ? 1.0 : (den < 0.0 ? -1.0 : 0.0)) * eps; return gt[i] * gt[i] / (den * den * den); });
                dd += 2.0 * emax / sf * q;
            }
        }
        if (nmin > 0 && sminn > eps) {
            const double sf = 1.0 + sminn * sminn * alpha;
            if (fabs(sf) > eps) {
                const double q = red.sum(nmin, [=](int k) { const int i = mni[k]; double den = w[i] - emin * alpha;
                    if (fabs(den) < eps) den = (den > 0.0 ? 1.0 : (den < 0.0 ? -1.0 : 0.0)) * eps; return gt[i] * gt[i] / (den * den * den); });
                dd += 2.0 * emin / sf * q;
            }
        }
        double astep;
        if (fabs(dd) < eps) astep = (nrm > Reff) ? alpha * 0.5 : 0.0;
        else astep = (Reff * Reff - nrm * nrm) / dd;
        if (fabs(astep) > p.alpha_step_max) astep *= p.alpha_step_max / fabs(astep);
        alpha = dmin(dmax(alpha + astep, p.alpha_min), p.alpha_max);
        if (alpha == p.alpha_max || alpha == p.alpha_min) { use_best(); converged = true; done = true; break; }
    }
    if (!done) { if (have_best) use_best(); else { for (int k = lane; k < n; k += WAVE) stp[k] = 0.0; SYNC(); } converged = false; }
    // ---- NaN fallback (:820-840), scaling (:846-861), prediction (:868-870)
    const double bad_cnt = red.sum(n, [=](int k) { return isfinite(stp[k]) ? 0.0 : 1.0; });
    if (bad_cnt > 0.0) {
        const double sdn = sqrt(red.sum(n, [=](int k) { return gt[k] * gt[k]; }));
        const double tgt = dmin(sdn, st.trust_radius);
        for (int k = lane; k < n; k += WAVE) stp[k] = (sdn > 1e-12) ? -gt[k] * (tgt / sdn) : 0.0;
        SYNC();
        converged = false;
    }
    double sn = sqrt(red.sum(n, [=](int k) { return stp[k] * stp[k]; }));
    double scale = 1.0;
    if (p.gradient_scaling_enabled && !(gnorm < 1e-10 || sn < 1e-10)) {
        const double ratio = sn / gnorm;
        if (ratio > 50.0) scale = dmax(50.0 / ratio, p.min_step_scale);
    }
    double tot = 1.0;
    if (scale < 1.0) { tot *= scale; sn = sn * scale; }
    if (sn > Reff * 1.01) { tot *= (Reff / sn); sn = Reff; }
    const double lin = red.sum(n, [=](int k) { return gt[k] * stp[k]; });
    const double quad = red.sum(n, [=](int k) { return w[k] * stp[k] * stp[k]; });
    const double pred = tot * lin + 0.5 * tot * tot * quad;
    for (int k = lane; k < n; k += WAVE) coef_out[k] = 0.0;
    SYNC();
    // k_backtransform writes -(sum coef_i v_i): store -step so that the move vector comes out with its own sign
    for (int k = lane; k < n; k += WAVE) coef_out[vidx[k]] = -(stp[k] * tot);
    if (lane == 0) {
        st.new_predicted = pred;
        st.shifted = shifted ? 1 : 0;
        st.micro_converged = converged ? 1 : 0;
        st.n_max = nmax;
        st.step_norm = sn;
        st.jacobi_sweeps = (einfo[1] & 4) ? -1 : einfo[0];
        st.iter += 1;
        state[0] = st;
        ts_row_out[0] = (p.saddle_order > 0 && p.mode_following_enabled && st.ts_mode_idx >= 0) ? vidx[st.ts_mode_idx] : -1;
    }
}


}  // namespace mopt
