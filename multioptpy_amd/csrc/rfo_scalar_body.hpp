// Scalar stage of the RS-I-RFO step (one wave per geometry): the body of k_rfo_scalar (k_rfo.hip), shared with k_small.hip.
// Reference: multioptpy/Optimizer/rsirfo.py:285-490, 492-657, 924-1313.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "logic.h"

namespace mopt {

// WaveRed with a wave-level sync(): for the section of k_rfo_scalar in which the three waves of the workgroup run different solves
// (a workgroup barrier there would have to be reached equally often by all three)
struct WaveRedLocal : WaveRed {
    __device__ void sync() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
};

// rsirfo.py:492-551 on an ascending array (wave-parallel)
__device__ inline bool ill_conditioned(const WaveRed &red, const double *w, int n, double thresh) {
    if (n < 2) return false;
    int cnt = (int)red.sum(n, [=](int i) { return fabs(w[i]) > 1e-10 ? 1.0 : 0.0; });
    if (cnt < 2) return true;
    double mx = 0.0, mn = INFINITY;
    for (int i = threadIdx.x & 63; i < n; i += WAVE) {
        double a = fabs(w[i]);
        if (a > 1e-10) { mx = fmax(mx, a); mn = fmin(mn, a); }
    }
    mx = wave_max(mx);
    mn = -wave_max(-mn);
    if (mn < 1e-15) return true;
    return (mx / mn) > thresh;
}

// Three waves per image: all of them run the (cheap, identical) set-up; then wave 0 solves the RFO equations for alpha0 while waves 1 and 2
// solve them for the two bracket values of alpha (1e-6, alpha_max) -- three of the five secular solves of a step side by side --
// and wave 0 finishes alone (logic.h rs_step_rest).  dynamic LDS: 15 * n doubles + 3 * n ints
// MODE 0: the three waves of k_rfo_scalar's workgroup (workgroup barriers).  MODE 1: called by ONE wave of a larger workgroup; MODE 2: called by waves
// 0 .. 2 of a larger workgroup (k_small.hip), wave index in `wv`, each wave with its OWN scratch `sm` -- they run the same set-up side by side on the
// shared read-only inputs, waves 1 and 2 solve the two bracket problems and raise done[0] / done[1] (LDS ints the caller has zeroed), wave 0 waits
// for both behind its own first solve.  In modes 1 and 2 every
// barrier of the body is then a wave-level one and the wave runs the two bracket solves itself.  sm: 15 * n doubles + 3 * n ints of LDS.
template <int MODE>
__device__ __forceinline__ void rfo_scalar_body(int n, int b, const mopt_rsirfo_params *params, mopt_rsirfo_state *states, const double *W, const int *perm,
                                                const int *einfo, const double *gt_all_, const double *Bg, const double *Be, const UpdateCoef *coef,
                                                double *coef_out_, int dbg, double *sm, const int *mf = nullptr, int nmf = 0, int wv = 0, int *done = nullptr) {
    constexpr bool FUSED = MODE != 0;
    // mf != null: MF_RSIRFO (mf_rsirfo.py:128-265) -- the modes to reflect are the nmf ascending positions in mf (chosen by the mode follower on the
    // host) instead of the first saddle_order non-zero ones, and the trust-radius update takes the curvature of the first of them
    auto SYNC = [] { if constexpr (FUSED) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); };
    // one latency-bound wave per image, usually running next to the back-transformation (pipeline.hip): highest issue priority on its SIMD
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wid = (MODE == 0) ? (int)(threadIdx.x >> 6) : (MODE == 2 ? wv : 0);
    __shared__ double s_n2[2];
    double *w0 = sm;             // [n] eigenvalues of the projected Hessian (ascending)
    double *g0 = w0 + n;         // [n] v_i . g in that order
    double *ws = g0 + n;         // [n] image-function eigenvalues, ascending, filtered (first m entries)
    double *gs = ws + n;         // [n] matching gradient components
    double *stp = gs + n;        // [n] step in the filtered eigenbasis
    double *best = stp + n;      // [n] scratch
    double *fac = best + n;      // [n] sign factor of the image function per sorted position (1, -1 or 0)
    double *cg2 = fac + n;       // [n] (g'/alpha)^2 of the current secular solve
    double *cw = cg2 + n;        // [n] lambda'/alpha of the current secular solve
    double *ex = cw + n;         // [2][3][n] cg2, cw, step of the bracket solves (waves 1, 2)
    int *vidx = (int *)(ex + 6 * n);     // [n] eigenvector (Vt row) of sorted position k
    int *kidx = vidx + n;                // [n] sorted position k of kept entry
    using RedT = typename std::conditional<FUSED, WaveRedLocal, WaveRed>::type;
    const RedT red;
    const long long tp0 = dbg ? wall_clock64() : 0;
    long long tp1 = 0, tp2 = 0, tp3 = 0;
    const mopt_rsirfo_params p = params[b];
    mopt_rsirfo_state st = states[b];
    const bool bad_eig = (einfo[2 * b + 1] & 1) != 0;
    const double *gt_all = gt_all_ + (size_t)b * n;
    double *coef_out = coef_out_ + (size_t)b * n;
    if (einfo[2 * b + 1] & 8) {
        // hand-off time-out of the persistent tridiagonalisation (GPU shared with a long-running kernel): not a numerical event.  The
        // optimizer state of this image stays as it was, so the host can solve the image again (mopt_rsirfo_retry_timeouts) and run this
        // stage for it a second time; the step is zero until then.  jacobi_sweeps = -2 tells the two kinds of failure apart.
        if (wid == 0) {
            for (int k = lane; k < n; k += WAVE) coef_out[k] = 0.0;
            if (lane == 0) states[b].jacobi_sweeps = -2;
        }
        return;
    }

    for (int k = lane; k < n; k += WAVE) {
        if (bad_eig) { w0[k] = 1.0; vidx[k] = k; }
        else { w0[k] = W[(size_t)b * n + k]; vidx[k] = perm[(size_t)b * n + k]; }
        g0[k] = gt_all[vidx[k]];
        fac[k] = 1.0;
    }
    SYNC();
    // --- level shift bookkeeping (rsirfo.py:553-657).  The reference re-runs eigh on H + shift*I and subtracts the
    // shift again; a Jacobi solve is invariant under a uniform diagonal shift (the rotation angles depend on
    // a_pp - a_qq and a_pq only), so only the decision flag is reproduced here.
    bool shifted = false;
    if (p.use_level_shift) shifted = true;
    else if (p.auto_level_shift) shifted = ill_conditioned(red, w0, n, p.condition_number_threshold);
    // --- gradient norm of the UNPROJECTED biased gradient (rsirfo.py:312)
    const double gnorm = sqrt(red.sum(n, [=](int i) { double v = Bg[(size_t)b * n + i]; return v * v; }));
    if (gnorm < p.gradient_norm_threshold) st.converged = 1;
    if (st.n_actual > 0 && fabs(st.actual[st.n_actual - 1]) < p.energy_change_threshold) st.converged = 1;
    // --- trust radius (rsirfo.py:380-398)
    const double energy = Be[b];
    if (st.has_prev) {
        const double actual = energy - st.prev_energy;
        push3(st.actual, st.n_actual, actual);
        if (st.n_predicted > 0) adjust_trust_radius(p, st, actual, st.predicted[st.n_predicted - 1], (mf && nmf > 0 && mf[0] >= 0 && mf[0] < n) ? w0[mf[0]] : w0[0], gnorm);
    }
    // --- image function (rsirfo.py:404-427): reflect (or, in NEB mode, remove) the first saddle_order modes with |lambda| > 1e-10.
    // P = I - c v v^T commutes with H on its eigenvectors: H* has the same eigenvectors, eigenvalue (1-c) lambda, g* = (1-c) g~.
    {
        int found = 0;
        const double f = p.neb_mode ? 0.0 : -1.0;
        if (mf) {
            for (int t = 0; t < nmf; ++t) { const int i = mf[t]; if (i >= 0 && i < n && fabs(w0[i]) > 1e-10 && lane == 0) fac[i] = f; }      // mf_rsirfo.py:207-214
        } else {
            for (int i = 0; i < n && found < p.saddle_order; ++i) {
                if (fabs(w0[i]) > 1e-10) { if (lane == 0) fac[i] = f; ++found; }
            }
        }
        SYNC();
    }
    // second decomposition: the level-shift decision is taken again on the image Hessian's spectrum
    bool shifted2 = false;
    if (p.saddle_order > 0) {
        for (int k = lane; k < n; k += WAVE) stp[k] = fac[k] * w0[k];
        SYNC();
        if (p.use_level_shift) shifted2 = true;
        else if (p.auto_level_shift) shifted2 = ill_conditioned(red, stp, n, p.condition_number_threshold);
    } else {
        shifted2 = shifted;
    }
    st.level_shift_applied = shifted2 ? 1 : 0;
    if (dbg) tp1 = wall_clock64();
    // --- ascending order of the image eigenvalues + 1e-6 filter (rsirfo.py:265-283,440)
    // rank of entry k among the starred values (stable); saddle_order is tiny so only those entries move.
    // The reference sorts the image eigenvalues ascending (stable) and then drops |lambda*| < 1e-6.  The starred values
    // differ from the ascending w0 only at the <= saddle_order reflected entries ("moved"), so the sorted position of every
    // entry follows from prefix counts: computed by the whole wave with ballots (a serial walk by one lane cost 80 us here).
    //   unmoved u at position c among the unmoved:  rank = c + #{moved t : val_t < val_u}
    //   moved t (t-th in index order):               rank = #{unmoved u : val_u <= val_t} + #{t' < t : val_t' <= val_t} + #{t' > t : val_t' < val_t}
    int m = 0;
    int *srt = kidx + n;                 // [n] index at sorted position
    // (MODE 2: three waves run this set-up side by side, each on its own scratch -- the list of moved entries lives there too, in the `ex`
    // area that only MODE 0 uses, so that no wave reads what another one is writing)
    __shared__ int s_moved_wg[8];
    __shared__ int s_nmoved_wg;
    int *const s_moved = (MODE == 2) ? (int *)ex : s_moved_wg;
    int &s_nmoved = (MODE == 2) ? ((int *)ex)[8] : s_nmoved_wg;
    {
        // moved entries, in index order (at most 8 are honoured, as before)
        int off = 0;
        for (int base = 0; base < n; base += WAVE) {
            const int k = base + lane;
            const bool f = (k < n) && (fac[k] != 1.0);
            const unsigned long long mk = __ballot(f);
            const int pos = off + __popcll(mk & ((1ull << lane) - 1ull));
            if (f && pos < 8) s_moved[pos] = k;
            off += __popcll(mk);
        }
        if (lane == 0) s_nmoved = off < 8 ? off : 8;
        SYNC();
        const int nmoved = s_nmoved;
        // unmoved entries
        int offu = 0;
        for (int base = 0; base < n; base += WAVE) {
            const int k = base + lane;
            const bool f = (k < n) && (fac[k] == 1.0);
            const unsigned long long mk = __ballot(f);
            if (f) {
                int r = offu + __popcll(mk & ((1ull << lane) - 1ull));
                const double val = w0[k];
                for (int t = 0; t < nmoved; ++t) { const int i = s_moved[t]; if (fac[i] * w0[i] < val) ++r; }
                srt[r] = k;
            }
            offu += __popcll(mk);
        }
        // moved entries (entries beyond the eighth moved one keep fac != 1 but are not placed, as in the serial form)
        for (int t = 0; t < nmoved; ++t) {
            const int i = s_moved[t];
            const double val = fac[i] * w0[i];
            int cnt = 0;
            for (int k = lane; k < n; k += WAVE) cnt += (fac[k] == 1.0 && w0[k] <= val) ? 1 : 0;
            cnt = (int)(wave_sum((double)cnt) + 0.5);
            for (int t2 = 0; t2 < nmoved; ++t2) {
                if (t2 == t) continue;
                const int i2 = s_moved[t2];
                const double v2 = fac[i2] * w0[i2];
                if (t2 < t ? (v2 <= val) : (v2 < val)) ++cnt;
            }
            if (lane == 0) srt[cnt] = i;
        }
        SYNC();
        const int ntot = offu + nmoved;
        // filter, keeping the order
        int offk = 0;
        for (int base = 0; base < ntot; base += WAVE) {
            const int c = base + lane;
            bool f = false;
            int i = 0;
            if (c < ntot) { i = srt[c]; f = !(fabs(fac[i] * w0[i]) < p.small_eigval_thresh); }
            const unsigned long long mk = __ballot(f);
            if (f) kidx[offk + __popcll(mk & ((1ull << lane) - 1ull))] = i;
            offk += __popcll(mk);
        }
        m = offk;
        SYNC();
    }
    for (int k = lane; k < m; k += WAVE) {
        const int i = kidx[k];
        ws[k] = fac[i] * w0[i];
        gs[k] = fac[i] * g0[i];
    }
    SYNC();
    st.prev_eigvec_size = m;
    st.n_kept = m;
    if (dbg) tp2 = wall_clock64();
    // --- RS step (rsirfo.py:924-984)
    RsResult rr;
    rr.branch = 0; rr.alpha = p.alpha0; rr.lambda_aug0 = 0.0; rr.initial_norm = 0.0; rr.secular_iters0 = 0;
    {
        // (up to here all three waves have computed the same things; m, ws, gs are complete behind the barrier above)
        const WaveRedLocal redl;
        double lam0 = 0.0, n2_0 = 0.0;
        int its0 = 0;
        if constexpr (MODE == 1) {
            if (m > 0) rr = rs_step(redl, p, st.trust_radius, ws, gs, m, stp, best, cg2, cw);
        } else if constexpr (MODE == 2) {
            if (wid != 0) {
                double q = 0.0;
                if (m > 0) solve_rfo(redl, ws, gs, m, wid == 1 ? 1e-6 : p.alpha_max, stp, &q, nullptr, cg2, cw);      // (this wave's own scratch)
                // (raised in every case: wave 0 does not write states[b] before both of the others have read it and are done)
                if (lane == 0) { s_n2[wid - 1] = q; __threadfence_block(); ((volatile int *)done)[wid - 1] = 1; }
                return;
            }
            if (m > 0) lam0 = solve_rfo(redl, ws, gs, m, p.alpha0, stp, &n2_0, &its0, cg2, cw);
            while (((volatile int *)done)[0] == 0 || ((volatile int *)done)[1] == 0) __builtin_amdgcn_s_sleep(1);
            __threadfence_block();
            if (m > 0) rr = rs_step_rest(redl, p, st.trust_radius, ws, gs, m, stp, best, cg2, cw, lam0, n2_0, its0, &s_n2[0], &s_n2[1]);
        } else {
        if (m > 0) {
            if (wid == 0) lam0 = solve_rfo(redl, ws, gs, m, p.alpha0, stp, &n2_0, &its0, cg2, cw);
            else {
                double *c2 = ex + (size_t)(wid - 1) * 3 * n, *cwx = c2 + n, *stx = cwx + n, q;
                solve_rfo(redl, ws, gs, m, wid == 1 ? 1e-6 : p.alpha_max, stx, &q, nullptr, c2, cwx);
                if (lane == 0) s_n2[wid - 1] = q;
            }
        }
        SYNC();
        if (wid != 0) return;                                // (wave 0 alone from here on: the barriers below are its own)
        if (m > 0) rr = rs_step_rest(red, p, st.trust_radius, ws, gs, m, stp, best, cg2, cw, lam0, n2_0, its0, &s_n2[0], &s_n2[1]);
        }
    }
    SYNC();
    if (dbg) tp3 = wall_clock64();
    // --- finite check (rsirfo.py:455-463) and steepest-descent fallback in the full eigenbasis
    double nonfinite = (m > 0) ? red.sum(m, [=](int k) { return isfinite(stp[k]) ? 0.0 : 1.0; }) : 0.0;
    for (int k = lane; k < n; k += WAVE) coef_out[k] = 0.0;
    SYNC();
    double pred = 0.0;
    if (nonfinite > 0.0 || m == 0) {
        // move = -gradient (projected), clipped to the trust radius; in the eigenbasis: coef_i = -g0_i * scale
        const double gn = sqrt(red.sum(n, [=](int k) { return g0[k] * g0[k]; }));
        double scale = 1.0;
        if (nonfinite > 0.0 && gn > st.trust_radius) scale = st.trust_radius / gn;
        if (m == 0 && nonfinite == 0.0) scale = 0.0;       // no usable mode: V_kept is empty, the step is the zero vector
        for (int k = lane; k < n; k += WAVE) coef_out[vidx[k]] = -g0[k] * scale;
        pred = red.sum(n, [=](int k) { double c = -g0[k] * scale; return g0[k] * c + 0.5 * w0[k] * c * c; });
        if (nonfinite > 0.0) rr.branch = 3;
    } else {
        for (int k = lane; k < m; k += WAVE) coef_out[vidx[kidx[k]]] = stp[k];
        // rfo_model (rsirfo.py:1717-1720) with the un-reflected gradient and Hessian: g.s + 0.5 s^T H s in the eigenbasis
        pred = red.sum(m, [=](int k) { const int i = kidx[k]; const double c = stp[k]; return g0[i] * c + 0.5 * w0[i] * c * c; });
    }
    push3(st.predicted, st.n_predicted, pred);
    st.prev_energy = energy;
    st.has_prev = 1;
    st.iteration += 1;
    st.branch = rr.branch;
    st.alpha = rr.alpha;
    st.lambda_aug = rr.lambda_aug0;
    st.initial_step_norm = rr.initial_norm;
    st.secular_iters = rr.secular_iters0;
    st.gradient_norm = gnorm;
    st.jacobi_sweeps = (einfo[2 * b + 1] & 4) ? -1 : einfo[2 * b];       // -1: solver failure, surfaced as MOPT_E_NUMERIC
    st.hess_updated = coef ? coef[b].applied : 0;
    if (lane == 0) states[b] = st;
    if (dbg && lane == 0 && (b == 0 || b == 15))
        printf("[rfo_prof] image %d: setup %.1f us, order+filter %.1f us, rs_step %.1f us (branch %d, secular iters %d), rest %.1f us\n", b,
               (tp1 - tp0) * 0.01, (tp2 - tp1) * 0.01, (tp3 - tp2) * 0.01, rr.branch, rr.secular_iters0, (wall_clock64() - tp3) * 0.01);
}

}  // namespace mopt
