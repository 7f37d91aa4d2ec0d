// EnhancedRSPRFO.run (multioptpy/Optimizer/rsprfo.py:713-886) on the device, single geometry.
//   k_prfo_prev    _process_previous_step (:908-962): reduction ratio (:434-472), trust radius (:474-512), rejection counter
//   k_prfo_scalar  one wave: eigenvalue shifting (:287-355), mode following (:964-1071), micro cycles with the
//                  augmented-Hessian solves replaced by the secular equation of the arrowhead matrix (:514-711, 1097-1168),
//                  gradient-based scaling (:357-399), adaptive trust radius (:401-432), predicted energy change (:1170-1188)
// The eigen-solve, gradient projection, Hessian update and back-transformation reuse the RSIRFO kernels.
#include <cstring>
#include <vector>

#include "pipeline.h"

namespace mopt {

// grid 1, block 256: dots for the reduction ratio.  u = Hp * s_prev was computed by k_matvec.
__global__ void __launch_bounds__(256) k_prfo_prev(int n, const mopt_rsprfo_params *__restrict__ params, mopt_rsprfo_state *__restrict__ state,
                                                   const double *__restrict__ gprev, const double *__restrict__ sprev,
                                                   const double *__restrict__ u, double B_e, double pre_move_norm) {
    __shared__ double red[3 * 16];
    double v[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) { v[0] += gprev[i] * sprev[i]; v[1] += sprev[i] * u[i]; v[2] += sprev[i] * sprev[i]; }
    block_sum<3>(v, red);
    if (threadIdx.x != 0) return;
    const mopt_rsprfo_params p = params[0];
    mopt_rsprfo_state st = state[0];
    const double actual = B_e - st.prev_energy;
    st.last_actual = actual;
    const double psn = (pre_move_norm >= 0.0) ? pre_move_norm : sqrt(v[2]);
    const double pred_red = -(v[0] + 0.5 * v[1]);
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
    state[0] = st;
}

// Smallest eigenvalue of the arrowhead matrix [[diag(d), b], [b^T, 0]] (mode > 0: largest), d_i = lam_i/alpha, b_i = g_i/alpha,
// restricted to the index list idx[0..m).  Root of  mu + sum b_i^2/(d_i - mu) = 0  left of every pole (right, mirrored).
// Safeguarded Newton to full double precision.  step_i = b_i/(d_i - mu) (= -v_i/nu of rsprfo.py:1160-1166).
__device__ inline double arrow_extreme(const WaveRed &red, const double *lam, const double *g, const int *idx, int m, double alpha,
                                       bool maximise) {
    const double sg = maximise ? -1.0 : 1.0;      // largest root of (D, b) = - smallest root of (-D, b)
    double dmin_ = INFINITY;
    for (int k = threadIdx.x & 63; k < m; k += WAVE) dmin_ = fmin(dmin_, sg * lam[idx[k]] / alpha);
    dmin_ = -wave_max(-dmin_);
    const double bb = red.sum(m, [=](int k) { double b = g[idx[k]] / alpha; return b * b; });
    if (bb == 0.0) return sg * fmin(dmin_, 0.0);       // decoupled: eigenvalues are d_i and 0
    auto f = [&](double mu) { return mu + red.sum(m, [=](int k) { double b = g[idx[k]] / alpha; return b * b / (sg * lam[idx[k]] / alpha - mu); }); };
    auto fp = [&](double mu) { return 1.0 + red.sum(m, [=](int k) { double b = g[idx[k]] / alpha; double d = sg * lam[idx[k]] / alpha - mu; return b * b / (d * d); }); };
    double t0 = 0.5 * (fabs(dmin_) + sqrt(dmin_ * dmin_ + 4.0 * bb));
    double lo = dmin_ - 1.01 * t0 - 1e-300, hi = dmin_;
    for (int it = 0; it < 60 && !(f(lo) < 0.0); ++it) lo = dmin_ - 2.0 * (dmin_ - lo);
    double mu = 0.5 * (lo + hi);
    if (mu >= hi) mu = lo;
    for (int it = 0; it < 200; ++it) {
        const double fv = f(mu);
        if (fv == 0.0) break;
        if (fv > 0.0) hi = mu; else lo = mu;
        double nx = mu - fv / fp(mu);
        if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
        if (nx == mu || !(hi - lo > 4.0 * 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi)))) { mu = nx; break; }
        mu = nx;
    }
    return sg * mu;
}

// dynamic LDS: 6 * n doubles + 4 * n ints
__global__ void __launch_bounds__(64) k_prfo_scalar(int n, const mopt_rsprfo_params *__restrict__ params, mopt_rsprfo_state *__restrict__ state,
                                                    const double *__restrict__ W, const int *__restrict__ perm, const int *__restrict__ einfo,
                                                    const double *__restrict__ gt_all, const double *__restrict__ ov_all, int have_ts,
                                                    double *__restrict__ coef_out, int *__restrict__ ts_row_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *w = sm;              // [n] eigenvalues after shifting, ascending
    double *gt = w + n;          // [n] gradient in that basis
    double *ov = gt + n;         // [n] |overlap| with the tracked TS vector
    double *stp = ov + n;        // [n] step in the eigenbasis
    double *best = stp + n;      // [n]
    double *key = best + n;      // [n] scratch (unsorted shifted values)
    int *vidx = (int *)(key + n);        // [n] Vt row of sorted position k
    int *ismax = vidx + n;               // [n]
    int *mxi = ismax + n;                // [n] max index list (first nmax) ...
    int *mni = mxi + n;                  // [n] min index list
    const int lane = threadIdx.x;
    const WaveRed red;
    const mopt_rsprfo_params p = params[0];
    mopt_rsprfo_state st = state[0];
    const bool bad = (einfo[1] & 1) != 0;
    // ---- first eigh (ascending) + shifting (rsprfo.py:287-355); "eigh again" of V diag V^T = same vectors, re-sorted values
    for (int k = lane; k < n; k += WAVE) { key[k] = bad ? 1.0 : W[k]; }
    __syncthreads();
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
            shifted = __syncthreads_or(any) != 0;
        }
    }
    __syncthreads();
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
    __syncthreads();
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
    __syncthreads();
    nmax = (int)best[0];
    st.ts_mode_idx = (int)best[1];
    const int nmin = n - nmax;
    __syncthreads();
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
        __syncthreads();
    };
    for (int mu = 0; mu < maxmc && !done; ++mu) {
        double emax = 0.0, emin = 0.0;
        for (int k = lane; k < n; k += WAVE) stp[k] = 0.0;
        __syncthreads();
        if (nmax > 0) {
            emax = arrow_extreme(red, w, gt, mxi, nmax, alpha, true);
            for (int k = lane; k < nmax; k += WAVE) { const int i = mxi[k]; stp[i] = (gt[i] / alpha) / (w[i] / alpha - emax); }
        }
        if (nmin > 0) {
            emin = arrow_extreme(red, w, gt, mni, nmin, alpha, false);
            for (int k = lane; k < nmin; k += WAVE) { const int i = mni[k]; stp[i] = (gt[i] / alpha) / (w[i] / alpha - emin); }
        }
        __syncthreads();
        const double nmx2 = red.sum(nmax, [=](int k) { double s = stp[mxi[k]]; return s * s; });
        const double nmn2 = red.sum(nmin, [=](int k) { double s = stp[mni[k]]; return s * s; });
        const double nrm = sqrt(red.sum(n, [=](int k) { return stp[k] * stp[k]; }));
        const double smaxn = sqrt(nmx2), sminn = sqrt(nmn2);
        if (hc < 64) hist[hc++] = nrm;
        const double diff = fabs(nrm - Reff);
        if (diff < best_diff) {
            for (int k = lane; k < n; k += WAVE) best[k] = stp[k];
            __syncthreads();
            have_best = true; best_diff = diff;
        }
        if (nrm <= Reff) { use_best(); converged = false; done = true; break; }            // break -> "did not converge" exit (:594-596, 652-660)
        if (nrm > 0.0 && diff / nrm < p.micro_cycle_rtol) {
            for (int k = lane; k < n; k += WAVE) stp[k] = stp[k] * (Reff / nrm);
            __syncthreads();
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
    if (!done) { if (have_best) use_best(); else { for (int k = lane; k < n; k += WAVE) stp[k] = 0.0; __syncthreads(); } converged = false; }
    // ---- NaN fallback (:820-840), scaling (:846-861), prediction (:868-870)
    const double bad_cnt = red.sum(n, [=](int k) { return isfinite(stp[k]) ? 0.0 : 1.0; });
    if (bad_cnt > 0.0) {
        const double sdn = sqrt(red.sum(n, [=](int k) { return gt[k] * gt[k]; }));
        const double tgt = dmin(sdn, st.trust_radius);
        for (int k = lane; k < n; k += WAVE) stp[k] = (sdn > 1e-12) ? -gt[k] * (tgt / sdn) : 0.0;
        __syncthreads();
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
    __syncthreads();
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

__global__ void k_norm1(int n, const double *__restrict__ v, double *__restrict__ out) {
    __shared__ double red[16];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += v[i] * v[i];
    a = block_sum1(a, red);
    if (threadIdx.x == 0) out[0] = sqrt(a);
}

__global__ void k_set_gnorm(mopt_rsprfo_state *st, const double *gn) { st[0].gradient_norm = gn[0]; }
// the tracked TS eigenvector (row *row of Vt; the unit vector when the solver fell back to the identity eigen-system) into a fixed buffer,
// so that it travels with the other results instead of behind a second synchronisation
__global__ void __launch_bounds__(256) k_prfo_tsrow(int n, const double *__restrict__ Vt, const int *__restrict__ row, const int *__restrict__ einfo, double *__restrict__ out) {
    const int r = row[0];
    if (r < 0) return;
    const bool ident = (einfo[1] & 1) != 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = ident ? (i == r ? 1.0 : 0.0) : Vt[(size_t)r * n + i];
}

}  // namespace mopt

using namespace mopt;

namespace {
struct DB {
    void *p = nullptr;
    ~DB() { if (p) hipFree(p); }
    int alloc(mopt_ctx *ctx, size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { ctx->err = "hipMalloc failed"; p = nullptr; return MOPT_E_NOMEM; }
        return MOPT_OK;
    }
    template <class T> T *as() { return (T *)p; }
};
}  // namespace

#define PTRY(x) do { int rc__ = (x); if (rc__ != MOPT_OK) return rc__; } while (0)
#define PH2D(dst, src, bytes) MOPT_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream))
#define PD2H(dst, src, bytes) MOPT_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream))

static int g_prfo_attr = 0;
void *mopt_pin_get(mopt_ctx *ctx, size_t bytes);      // pinned staging of the context (mopt_abi.hip)

static int rsprfo_run_once(mopt_ctx *ctx, int n, const mopt_rsprfo_params *params, mopt_rsprfo_state *state,
                           const double *hessian, const double *bias_hessian, const double *geom, const double *B_g,
                           const double *pre_geom, const double *pre_B_g, double B_e, const double *prev_gradient,
                           const double *prev_move_vector, double pre_move_norm, const double *ts_vec, double *step,
                           double *H_out, double *ts_vec_out, bool *timed_out);

extern "C" int mopt_rsprfo_run(mopt_ctx *ctx, int n, const mopt_rsprfo_params *params, mopt_rsprfo_state *state,
                               const double *hessian, const double *bias_hessian, const double *geom, const double *B_g,
                               const double *pre_geom, const double *pre_B_g, double B_e, const double *prev_gradient,
                               const double *prev_move_vector, double pre_move_norm, const double *ts_vec, double *step,
                               double *H_out, double *ts_vec_out) {
    if (!ctx || !state) return MOPT_E_ARG;
    // A hand-off time-out of the persistent tridiagonalisation (GPU shared with a long-running kernel) is not a numerical failure: the
    // call is repeated once from the caller's inputs with one launch per Householder step (H_out may alias `hessian`: keep a copy).
    const mopt_rsprfo_state st0 = *state;
    std::vector<double> hkeep;
    if (H_out == hessian && hessian && n > 0) hkeep.assign(hessian, hessian + (size_t)n * n);
    bool timed_out = false;
    int rc = rsprfo_run_once(ctx, n, params, state, hessian, bias_hessian, geom, B_g, pre_geom, pre_B_g, B_e, prev_gradient, prev_move_vector,
                             pre_move_norm, ts_vec, step, H_out, ts_vec_out, &timed_out);
    if (rc == MOPT_OK && timed_out) {
        *state = st0;
        const bool saved = ctx->tri_no_persist;
        ctx->tri_no_persist = true;
        rc = rsprfo_run_once(ctx, n, params, state, hkeep.empty() ? hessian : hkeep.data(), bias_hessian, geom, B_g, pre_geom, pre_B_g, B_e, prev_gradient,
                             prev_move_vector, pre_move_norm, ts_vec, step, H_out, ts_vec_out, &timed_out);
        ctx->tri_no_persist = saved;
    }
    return rc;
}

static int rsprfo_run_once(mopt_ctx *ctx, int n, const mopt_rsprfo_params *params, mopt_rsprfo_state *state,
                           const double *hessian, const double *bias_hessian, const double *geom, const double *B_g,
                           const double *pre_geom, const double *pre_B_g, double B_e, const double *prev_gradient,
                           const double *prev_move_vector, double pre_move_norm, const double *ts_vec, double *step,
                           double *H_out, double *ts_vec_out, bool *timed_out) {
    *timed_out = false;
    if (!ctx) return MOPT_E_ARG;
    if (n < 3 || n % 3 != 0 || n > MOPT_MAX_N) { ctx->err = "mopt_rsprfo_run: bad n"; return MOPT_E_ARG; }
    if (!params || !state || !geom || !B_g || !step || !H_out) { ctx->err = "mopt_rsprfo_run: null argument"; return MOPT_E_ARG; }
    if (!hessian) { ctx->err = "Hessian matrix must be set before running optimization"; return MOPT_E_STATE; }
    if (params->rule < 0 || params->rule >= MOPT_RULE_COUNT) { ctx->err = "unknown update rule"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    RsirfoWork wk;
    PTRY(wk.alloc(ctx, n, 1, true, bias_hessian != nullptr, false));
    const size_t nn = (size_t)n * n * 8, nv = (size_t)n * 8;
    DB dpar, dst, dprevg, dprevs, dts, dov, dtsrow, dgn;
    PTRY(dpar.alloc(ctx, sizeof(mopt_rsprfo_params))); PTRY(dst.alloc(ctx, sizeof(mopt_rsprfo_state)));
    PTRY(dprevg.alloc(ctx, nv)); PTRY(dprevs.alloc(ctx, nv)); PTRY(dts.alloc(ctx, nv)); PTRY(dov.alloc(ctx, nv));
    PTRY(dtsrow.alloc(ctx, 16)); PTRY(dgn.alloc(ctx, 16));
    mopt_rsprfo_state st = *state;
    st.last_ratio = NAN; st.last_actual = NAN; st.step_accepted = 1; st.hess_updated = 0;
    PH2D(wk.H, hessian, nn);
    if (bias_hessian) PH2D(wk.Bias, bias_hessian, nn);
    PH2D(wk.x, geom, nv); PH2D(wk.bg, B_g, nv);
    PH2D(dpar.p, params, sizeof(mopt_rsprfo_params)); PH2D(dst.p, &st, sizeof(st));
    // ---- 1. previous step (rsprfo.py:753-760)
    if (st.has_prev && prev_gradient && prev_move_vector) {
        PH2D(dprevg.p, prev_gradient, nv); PH2D(dprevs.p, prev_move_vector, nv);
        PTRY(mopt_launch_project_hess(ctx, n, 1, wk.H, wk.Bias, wk.x, wk.Q, wk.W6, wk.A));
        PTRY(mopt_launch_matvec(ctx, n, 1, wk.A, dprevs.as<double>(), wk.u));
        hipLaunchKernelGGL(k_prfo_prev, dim3(1), dim3(256), 0, ctx->stream, n, dpar.as<mopt_rsprfo_params>(), dst.as<mopt_rsprfo_state>(),
                           dprevg.as<double>(), dprevs.as<double>(), wk.u, B_e, pre_move_norm);
    }
    // ---- 2. Hessian update with (B_g - pre_B_g) (rsprfo.py:765-767, 1190-1260)
    bool updated = false;
    if (pre_geom && pre_B_g) {
        PH2D(wk.px, pre_geom, nv); PH2D(wk.pg, pre_B_g, nv);
        int rule = params->rule, one = 1;
        PH2D(wk.rules, &rule, sizeof(int)); PH2D(wk.do_update, &one, sizeof(int));      // (the locals outlive the copies: this block synchronises below)
        // candidate = sym(H + dH) into wk.A; accepted unless max |eig| > 1e6 (rsprfo.py:1236-1250)
        PTRY(mopt_launch_hess_update(ctx, n, 1, wk.rules, wk.do_update, rule == MOPT_RULE_FLOWCHART, wk.x, wk.px, wk.bg, wk.pg, wk.s, wk.y,
                                     wk.u, wk.Hy, wk.yd, wk.Av, wk.coef, wk.info, wk.H, nullptr, wk.A, false, /*guard_sy=*/false));
        // Gershgorin: max |eig| <= ||H||_inf, so the spectrum itself is only needed when that bound exceeds 1e6
        PTRY(mopt_launch_norm_inf(ctx, n, 1, wk.A, wk.sigma));
        UpdateCoef hc;
        double sig = 0.0;
        PD2H(&hc, wk.coef, sizeof(UpdateCoef));
        PD2H(&sig, wk.sigma, sizeof(double));
        MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (hc.applied) {
            bool accept = true;
            if (!(sig / 1.0625 <= 1e6)) {
                PTRY(mopt_eigh_auto(ctx, wk, n, 1, false));
                std::vector<double> ev(n);
                PD2H(ev.data(), wk.eigw, nv);
                MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                double mx = 0.0;
                for (int i = 0; i < n; ++i) mx = mx > fabs(ev[i]) ? mx : fabs(ev[i]);
                accept = !(mx > 1e6);
            }
            if (accept) {
                MOPT_HIP_TRY(ctx, hipMemcpyAsync(wk.H, wk.A, nn, hipMemcpyDeviceToDevice, ctx->stream));
                updated = true;
            }
        }
    }
    // ---- 3. projected gradient and its norm
    PTRY(mopt_launch_project_grad(ctx, n, 1, wk.bg, wk.x, wk.gp, /*variant=*/1));
    hipLaunchKernelGGL(k_norm1, dim3(1), dim3(256), 0, ctx->stream, n, wk.gp, dgn.as<double>());
    hipLaunchKernelGGL(k_set_gnorm, dim3(1), dim3(1), 0, ctx->stream, dst.as<mopt_rsprfo_state>(), dgn.as<double>());
    // ---- 4. eigh(H + bias), not TR/rot projected (rsprfo.py:780-783)
    PTRY(mopt_launch_sym_add(ctx, n, 1, wk.H, wk.Bias, wk.A));
    PTRY(mopt_eigh_auto(ctx, wk, n, 1, false));
    // ---- 5. g~ = V^T g, overlaps with the tracked TS vector, scalar logic, back-transformation
    const int have_ts = (ts_vec != nullptr && st.ts_mode_idx >= 0) ? 1 : 0;
    if (have_ts) PH2D(dts.p, ts_vec, nv);
    PTRY(mopt_launch_vtg(ctx, n, 1, wk.Vt, wk.gp, wk.einfo, wk.gt_all));
    if (have_ts) PTRY(mopt_launch_vtg(ctx, n, 1, wk.Vt, dts.as<double>(), wk.einfo, dov.as<double>()));
    const size_t lds = (size_t)6 * n * 8 + (size_t)4 * n * 4 + 16;
    if (!g_prfo_attr) {
        MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_prfo_scalar, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
        g_prfo_attr = 1;
    }
    hipLaunchKernelGGL(k_prfo_scalar, dim3(1), dim3(64), lds, ctx->stream, n, dpar.as<mopt_rsprfo_params>(), dst.as<mopt_rsprfo_state>(), wk.eigw,
                       wk.perm, wk.einfo, wk.gt_all, dov.as<double>(), have_ts, wk.coefout, dtsrow.as<int>());
    PTRY(mopt_launch_backtransform(ctx, n, 1, wk.Vt, wk.coefout, wk.einfo, wk.step));
    MOPT_HIP_TRY(ctx, hipGetLastError());
    int ts_row = -1, bad = 0;
    std::vector<double> ts_stage;
    {
        // the results come back through the context's pinned staging buffer: none of the copies blocks on pageable memory, one
        // synchronisation (four pageable copies were 70 us of a 0.77 ms call at n = 21); Hessians above 1 MB go straight to the caller
        const bool stage_H = nn <= ((size_t)1 << 20);
        const size_t o_st = (nv + 255) & ~(size_t)255, o_row = o_st + 256 + ((sizeof(st) + 255) & ~(size_t)255), o_bad = o_row + 256, o_ts = o_bad + 256,
                     o_H = o_ts + ((nv + 255) & ~(size_t)255);
        char *pin = (char *)mopt_pin_get(ctx, o_H + (stage_H ? nn : 0));
        if (!pin) { ctx->err = "hipHostMalloc failed (staging buffer)"; return MOPT_E_NOMEM; }
        PD2H(pin, wk.step, nv);
        PD2H(pin + o_st, dst.p, sizeof(st));
        PD2H(pin + o_row, dtsrow.p, sizeof(int));
        PD2H(pin + o_bad, wk.einfo + 1, sizeof(int));
        if (ts_vec_out) {
            hipLaunchKernelGGL(k_prfo_tsrow, dim3(1), dim3(256), 0, ctx->stream, n, wk.Vt, dtsrow.as<int>(), wk.einfo, dov.as<double>());      // (dov is free behind k_prfo_scalar)
            PD2H(pin + o_ts, dov.p, nv);
        }
        if (updated) PD2H(stage_H ? (void *)(pin + o_H) : (void *)H_out, wk.H, nn);
        MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(step, pin, nv);
        std::memcpy((void *)&st, pin + o_st, sizeof(st));
        std::memcpy(&ts_row, pin + o_row, sizeof(int));
        std::memcpy(&bad, pin + o_bad, sizeof(int));
        if (ts_vec_out) { ts_stage.resize(n); std::memcpy(ts_stage.data(), pin + o_ts, nv); }
        if (stage_H && updated) std::memcpy(H_out, pin + o_H, nn);
        if (!updated && H_out != hessian) std::memcpy(H_out, hessian, nn);      // (no update: the device copy is the caller's matrix bit for bit)
    }
    if (ts_vec_out && ts_row >= 0) {
        std::memcpy(ts_vec_out, ts_stage.data(), nv);
    }
    st.hess_updated = updated ? 1 : 0;
    *state = st;
    *timed_out = (bad & 8) != 0;
    if (mopt_debug_poison_on()) PTRY(wk.check_canaries());
    return MOPT_OK;
}
