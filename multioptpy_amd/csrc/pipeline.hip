// Batched RSIRFO.run pipeline on device buffers (multioptpy/Optimizer/rsirfo.py:285-490).
#include <cstdlib>
#include "pipeline.h"

RsirfoWork::~RsirfoWork() {
    // (the sentinel bookkeeping of k_tri_early is per address: memory that goes back to the allocator is not "known clean" any more)
    if (ctx && block && ctx->early_clean == (const void *)tri.D) ctx->early_clean = nullptr;
    if (block) hipFree(block);
}

int RsirfoWork::alloc(mopt_ctx *c, int n_, int batch_, bool own_hessian, bool with_bias, bool with_inplace) {
    ctx = c; n = n_; batch = batch_; owns_H = own_hessian;
    const size_t nn = (size_t)batch * n * n, nv = (size_t)batch * n;
    auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    size_t tot = 0;
    const int nmat = 3 + (own_hessian ? 1 : 0) + (with_bias ? 1 : 0) + (with_inplace ? 1 : 0);
    tot += nmat * al(nn * 8);
    tot += 17 * al(nv * 8);            // vectors
    tot += 2 * al(nv * 6 * 8);         // Q, W6
    tot += al(nv * 4);                 // perm
    tot += 3 * al((size_t)batch * 8) + al((size_t)batch * 8 * 8);
    tot += al((size_t)batch * 2 * 4) + al((size_t)batch * (mopt::JAC_MAX_SWEEPS + 1) * 4) + 2 * al((size_t)batch * 4);
    tot += 5 * al(nv * 8) + 2 * al(nv * 4) + al(2 * nv * 4) + al((size_t)2 * batch * 4) + al((size_t)batch * 4);      // reduction-route eigen-solver scratch
    tot += al((size_t)batch * ((size_t)n * n + MOPT_TRI_T_DOUBLES(n)) * 8);                                              // ... its second eigenvector buffer and the T factors
    tot += al(batch * sizeof(mopt::UpdateCoef)) + al(batch * sizeof(mopt_rsirfo_params)) + al(batch * sizeof(mopt_rsirfo_state));
    if (hipMalloc(&block, tot) != hipSuccess) { c->err = "hipMalloc failed (RSIRFO workspace)"; block = nullptr; return MOPT_E_NOMEM; }
    char *p = (char *)block;
    auto take = [&](size_t bytes) { void *r = p; p += al(bytes); return r; };
    if (own_hessian) H = (double *)take(nn * 8);
    if (with_bias) Bias = (double *)take(nn * 8);
    if (with_inplace) Hinpl = (double *)take(nn * 8);
    A = (double *)take(nn * 8); Gt = (double *)take(nn * 8); Vt = (double *)take(nn * 8);
    double **vecs[17] = {&x, &bg, &g, &px, &pg, &s, &y, &u, &Hy, &yd, &Av, &gp, &eigw, &gt_all, &coefout, &step, nullptr};
    for (int i = 0; i < 16; ++i) *vecs[i] = (double *)take(nv * 8);
    take(nv * 8);
    Q = (double *)take(nv * 6 * 8); W6 = (double *)take(nv * 6 * 8);
    perm = (int *)take(nv * 4);
    be = (double *)take((size_t)batch * 8); sigma = (double *)take((size_t)batch * 8); take((size_t)batch * 8);
    info = (double *)take((size_t)batch * 8 * 8);
    einfo = (int *)take((size_t)batch * 2 * 4);
    rotcount = (int *)take((size_t)batch * (mopt::JAC_MAX_SWEEPS + 1) * 4);
    rules = (int *)take((size_t)batch * 4); do_update = (int *)take((size_t)batch * 4);
    coef = (mopt::UpdateCoef *)take(batch * sizeof(mopt::UpdateCoef));
    params = (mopt_rsirfo_params *)take(batch * sizeof(mopt_rsirfo_params));
    states = (mopt_rsirfo_state *)take(batch * sizeof(mopt_rsirfo_state));
    tri.D = (double *)take(nv * 8); tri.E = (double *)take(nv * 8); tri.TAU = (double *)take(nv * 8); tri.WT = (double *)take(nv * 8);
    tri.onenrm = (double *)take(nv * 8);
    tri.bstart = (int *)take(nv * 4); tri.bend = (int *)take(nv * 4); tri.cstart = (int *)take(2 * nv * 4);     // tight + loose cluster tables
    tri.ncl = (int *)take((size_t)2 * batch * 4); tri.fail = (int *)take((size_t)batch * 4);
    tri.Z2 = (double *)take((size_t)batch * ((size_t)n * n + MOPT_TRI_T_DOUBLES(n)) * 8);     // Loewdin scratch + T factors
    return MOPT_OK;
}

int mopt_rsirfo_pipeline(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, bool any_update, bool any_flowchart,
                         double *d_Hinpl, double *d_Hsym, bool warm_eigh, const mopt::FusedBand *fb) {
    int rc;
    const mopt::UpdateCoef *coef = nullptr;
    if (!fb) {                                               // (fused band path: already forked by the band, ahead of its force kernel)
        rc = mopt_launch_trrot_basis_side(ctx, n, batch, wk.x, wk.Q);                                // joined in front of k_proj_w
        if (rc != MOPT_OK) return rc;
    }
    rc = mopt_launch_project_grad(ctx, n, batch, wk.bg, wk.x, wk.gp, 0, /*side_stream=*/true);      // joined in front of the RFO stage
    if (rc != MOPT_OK) return rc;
    const bool clear_for_eigh = (fb != nullptr) && mopt_eigh_algo(n) == 1 && n * n >= 8 * n + 64;
    if (fb) {
        mopt::FusedBand fbl = *fb;
        fbl.clear_for_eigh = clear_for_eigh;
        rc = mopt_launch_fused_update_project(ctx, wk, n, batch, fbl);
        if (rc != MOPT_OK) return rc;
        coef = wk.coef;
    } else if (any_update) {
        rc = mopt_launch_hess_update(ctx, n, batch, wk.rules, wk.do_update, any_flowchart, wk.x, wk.px, wk.g, wk.pg, wk.s, wk.y,
                                     wk.u, wk.Hy, wk.yd, wk.Av, wk.coef, wk.info, wk.H, d_Hinpl, d_Hsym, false);
        if (rc != MOPT_OK) return rc;
        coef = wk.coef;
    }
    if (!fb) {
        rc = mopt_launch_project_hess(ctx, n, batch, d_Hsym, wk.Bias, wk.x, wk.Q, wk.W6, wk.A, /*basis_ready=*/true);
        if (rc != MOPT_OK) return rc;
    }
    // The scalar stage of the step (k_rfo_scalar: one wave per image) runs next to the back-transformation when the solver takes the
    // hook: g~ = V^T g = Z^T (Q^T g) needs the tridiagonal form's eigenvectors only (k_applyq.hip, EighRfoHook in pipeline.h).
    struct HookUser { RsirfoWork *wk; int n, batch; const mopt::UpdateCoef *coef; } hu = {&wk, n, batch, coef};
    EighRfoHook hook;
    hook.gp = wk.gp; hook.gq = wk.Hy; hook.ev_gp = ctx->ev_join2; hook.user = &hu;
    hook.vout = wk.Gt;                                       // (free on the reduction route: the eigenvectors V land there, Z stays in wk.Vt)
    hook.pre_cleared = clear_for_eigh;                       // k_fused_final has zeroed the failure flags and the exchange area
    hook.front = [](mopt_ctx *c, void *u, hipStream_t st, const double *d_Zt, const double *d_gq) -> int {
        HookUser *h = (HookUser *)u;
        mopt_scope sc(c, "rfo_scalar", 1, st);
        return mopt_launch_vtg(c, h->n, h->batch, d_Zt, d_gq, h->wk->einfo, h->wk->gt_all, h->wk->gp, st);
    };
    hook.scalar = [](mopt_ctx *c, void *u, hipStream_t st) -> int {
        HookUser *h = (HookUser *)u;
        RsirfoWork &w = *h->wk;
        mopt_scope sc(c, "rfo_scalar", 1, st);               // (helper stream: overlaps eigh_backtrans)
        return mopt_launch_rfo_scalar(c, st, h->n, h->batch, w.params, w.states, w.eigw, w.perm, w.einfo, w.gt_all, w.bg, w.be, h->coef, w.coefout);
    };
    rc = mopt_eigh_auto(ctx, wk, n, batch, warm_eigh, /*preserve_A=*/false, &hook);
    if (rc != MOPT_OK) return rc;
    if (hook.used) {
        mopt_scope sc(ctx, "rfo", 1);
        return mopt_launch_backtransform(ctx, n, batch, hook.vout, wk.coefout, wk.einfo, wk.step);
    }
    MOPT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join2, 0));       // the projected gradient (side stream)
    rc = mopt_launch_rfo(ctx, n, batch, wk.params, wk.states, wk.eigw, wk.perm, wk.einfo, wk.Vt, wk.gp, wk.bg, wk.be, coef,
                         wk.gt_all, wk.coefout, wk.step);
    return rc;
}

// eigh(wk.A) -> wk.eigw (ascending through wk.perm), wk.Vt (vector-major), wk.einfo; wk.A is preserved unless the caller
// says it is not needed any more (the RS-I-RFO pipeline: the reflectors then overwrite it in place, one n^2 copy less)
int mopt_eigh_auto(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, bool warm, bool preserve_A, EighRfoHook *hook) {
    if (hook) hook->used = false;
    if (mopt_eigh_algo(n) == 1) {
        // reduction route: the reflectors overwrite the input, eigenvectors land in Vt
        double *work = wk.A;
        if (preserve_A) {
            MOPT_HIP_TRY(ctx, hipMemcpyAsync(wk.Gt, wk.A, (size_t)batch * n * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            work = wk.Gt;
        }
        return mopt_launch_eigh_tridiag(ctx, n, batch, work, wk.Vt, wk.tri.D, wk.tri.E, wk.tri.TAU, wk.tri.WT, wk.tri.onenrm, wk.tri.bstart,
                                        wk.tri.bend, wk.tri.cstart, wk.tri.ncl, wk.tri.fail, wk.eigw, wk.perm, wk.einfo, wk.tri.Z2, hook);
    }
    return mopt_launch_eigh(ctx, n, batch, wk.A, wk.sigma, wk.Gt, wk.Vt, wk.rotcount, wk.eigw, wk.perm, wk.einfo, MOPT_DEFAULT_SWEEPS, warm);
}

// Default: reduction route from n = 32 on (the Jacobi sweeps are cheaper below that and need no deflation logic);
// MOPT_EIGH_ALGO=jacobi / tridiag forces one route for every size (tridiag needs n >= 3).
int mopt_eigh_algo(int n) {
    static int algo = -2;
    if (algo == -2) {
        const char *e = getenv("MOPT_EIGH_ALGO");
        algo = !e ? -1 : (e[0] == 't' || e[0] == '1') ? 1 : 0;
    }
    if (n < 3) return 0;
    // smallest n of the reduction route: 16 since the end of round 2 (single-compute-unit tridiagonalisation + one-wave QL: n = 21
    // 0.48 -> 0.38 ms per RSIRFO.run, n = 27 0.56 -> 0.46 against the block Jacobi's 7 sweeps of 40 us; MOPT_EIGH_MIN=32 restores)
    static const int nmin = getenv("MOPT_EIGH_MIN") ? atoi(getenv("MOPT_EIGH_MIN")) : 16;
    return algo >= 0 ? algo : (n >= nmin ? 1 : 0);
}
