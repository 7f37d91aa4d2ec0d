// Batched RSIRFO.run pipeline on device buffers (multioptpy/Optimizer/rsirfo.py:285-490).
#include <cstdlib>
#include "pipeline.h"

RsirfoWork::~RsirfoWork() {
    // (the sentinel bookkeeping of k_tri_early is per address: memory that goes back to the allocator is not "known clean" any more)
    if (ctx && block && ctx->early_clean == (const void *)tri.D) ctx->early_clean = nullptr;
    if (ctx && block && ctx->defl.key == (const void *)tri.D) { ctx->defl.key = nullptr; ctx->defl.skip_left = 0; ctx->defl.dense_streak = 0; }
    if (block) hipFree(block);
}

// ---- debug mode: poisoned workspaces with guard gaps (mopt_debug_poison) -----------------------------------
#include <atomic>
static std::atomic<int> g_poison{-1};
bool mopt_debug_poison_on() {
    int v = g_poison.load();
    if (v < 0) { const char *e = getenv("MOPT_DEBUG_POISON"); v = (e && atoi(e) != 0) ? 1 : 0; g_poison.store(v); }
    return v != 0;
}
extern "C" int mopt_debug_poison(int enable) { g_poison.store(enable ? 1 : 0); return MOPT_OK; }

int mopt_check_gaps(mopt_ctx *ctx, const void *block, const std::vector<std::pair<size_t, size_t>> &gaps, const std::vector<const char *> &names, const char *what) {
    if (gaps.empty() || !block) return MOPT_OK;
    MOPT_HIP_TRY(ctx, hipDeviceSynchronize());
    std::vector<unsigned char> h(MOPT_CANARY_GAP + 256);
    for (size_t k = 0; k < gaps.size(); ++k) {
        MOPT_HIP_TRY(ctx, hipMemcpy(h.data(), (const char *)block + gaps[k].first, gaps[k].second, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < gaps[k].second; ++i)
            if (h[i] != MOPT_CANARY_BYTE) {
                ctx->err = std::string(what) + ": guard gap behind array '" + names[k] + "' damaged at byte " + std::to_string(i) + " of the gap (out-of-bounds write)";
                return MOPT_E_STATE;
            }
    }
    return MOPT_OK;
}

int RsirfoWork::check_canaries() { return mopt_check_gaps(ctx, block, gaps, gap_names, "RSIRFO workspace"); }

int RsirfoWork::alloc(mopt_ctx *c, int n_, int batch_, bool own_hessian, bool with_bias, bool with_inplace) {
    ctx = c; n = n_; batch = batch_; owns_H = own_hessian;
    const size_t nn = (size_t)batch * n * n, nv = (size_t)batch * n;
    const bool poison = mopt_debug_poison_on();
    const size_t gap = poison ? MOPT_CANARY_GAP : 0;
    auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    struct Item { void **p; size_t bytes; const char *name; size_t off; };
    std::vector<Item> items;
    double *unused0 = nullptr, *unused1 = nullptr;           // (two slots the layout has always carried)
    auto want = [&](auto **pp, size_t bytes, const char *name) { items.push_back({(void **)pp, bytes, name, 0}); };
    if (own_hessian) want(&H, nn * 8, "H");
    if (with_bias) want(&Bias, nn * 8, "Bias");
    if (with_inplace) want(&Hinpl, nn * 8, "Hinpl");
    want(&A, nn * 8, "A"); want(&Gt, nn * 8, "Gt"); want(&Vt, nn * 8, "Vt");
    want(&x, nv * 8, "x"); want(&bg, nv * 8, "bg"); want(&g, nv * 8, "g"); want(&px, nv * 8, "px"); want(&pg, nv * 8, "pg");
    want(&s, nv * 8, "s"); want(&y, nv * 8, "y"); want(&u, nv * 8, "u"); want(&Hy, nv * 8, "Hy"); want(&yd, nv * 8, "yd"); want(&Av, nv * 8, "Av");
    want(&gp, nv * 8, "gp"); want(&eigw, nv * 8, "eigw"); want(&gt_all, nv * 8, "gt_all"); want(&coefout, nv * 8, "coefout"); want(&step, nv * 8, "step");
    want(&unused0, nv * 8, "unused0");
    want(&Q, nv * 6 * 8, "Q"); want(&W6, nv * 6 * 8, "W6");
    want(&perm, nv * 4, "perm");
    want(&be, (size_t)batch * 8, "be"); want(&sigma, (size_t)batch * 8, "sigma"); want(&unused1, (size_t)batch * 8, "unused1");
    want(&info, (size_t)batch * 8 * 8, "info");
    want(&einfo, (size_t)batch * 2 * 4, "einfo");
    want(&rotcount, (size_t)batch * (mopt::JAC_MAX_SWEEPS + 1) * 4, "rotcount");
    want(&rules, (size_t)batch * 4, "rules"); want(&do_update, (size_t)batch * 4, "do_update");
    want(&coef, batch * sizeof(mopt::UpdateCoef), "coef");
    want(&params, batch * sizeof(mopt_rsirfo_params), "params");
    want(&states, batch * sizeof(mopt_rsirfo_state), "states");
    want(&tri.D, nv * 8, "tri.D"); want(&tri.E, nv * 8, "tri.E"); want(&tri.TAU, nv * 8, "tri.TAU"); want(&tri.WT, nv * 8, "tri.WT");
    want(&tri.onenrm, nv * 8, "tri.onenrm");
    want(&tri.bstart, nv * 4, "tri.bstart"); want(&tri.bend, nv * 4, "tri.bend"); want(&tri.cstart, 2 * nv * 4, "tri.cstart");     // tight + loose cluster tables
    want(&tri.ncl, (size_t)2 * batch * 4, "tri.ncl"); want(&tri.fail, (size_t)batch * 4, "tri.fail");
    want(&tri.Z2, (size_t)batch * ((size_t)n * n + MOPT_TRI_T_DOUBLES(n)) * 8, "tri.Z2");     // Loewdin scratch + T factors
    want(&sticky, 4, "sticky");
    want(&tri.defl, (size_t)batch * 4, "tri.defl");
    size_t tot = gap;
    for (auto &it : items) { it.off = tot; tot += al(it.bytes) + gap; }
    if (hipMalloc(&block, tot) != hipSuccess) { c->err = "hipMalloc failed (RSIRFO workspace)"; block = nullptr; return MOPT_E_NOMEM; }
    total_bytes = tot;
    for (auto &it : items) *it.p = (char *)block + it.off;
    gaps.clear(); gap_names.clear();
    if (poison) {
        // arrays: 0xFF bytes; gaps (and the alignment slack in front of them): the canary byte
        if (hipMemsetAsync(block, MOPT_CANARY_BYTE, tot, c->stream) != hipSuccess) { c->err = "hipMemset failed (poison)"; return MOPT_E_DEVICE; }
        for (auto &it : items)
            if (hipMemsetAsync((char *)block + it.off, 0xFF, it.bytes, c->stream) != hipSuccess) { c->err = "hipMemset failed (poison)"; return MOPT_E_DEVICE; }
        gaps.emplace_back(0, gap); gap_names.push_back("(start of block)");
        for (auto &it : items) { gaps.emplace_back(it.off + it.bytes, al(it.bytes) - it.bytes + gap); gap_names.push_back(it.name); }      // (alignment slack included)
    }
    if (hipMemsetAsync(sticky, 0, 4, c->stream) != hipSuccess || hipMemsetAsync(tri.defl, 0, (size_t)batch * 4, c->stream) != hipSuccess) { c->err = "hipMemset failed (workspace)"; return MOPT_E_DEVICE; }
    return MOPT_OK;
}

int mopt_rsirfo_pipeline(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, bool any_update, bool any_flowchart,
                         double *d_Hinpl, double *d_Hsym, bool warm_eigh, const mopt::FusedBand *fb) {
    int rc;
    const mopt::UpdateCoef *coef = nullptr;
    // small geometries: the whole step in one workgroup and one launch (k_small.hip)
    if (mopt_small_eligible(n)) return mopt_launch_small_rsirfo(ctx, wk, n, batch, any_update, any_flowchart, d_Hinpl, d_Hsym, fb);
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

int mopt_rsirfo_retry_timeouts(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, const double *d_Hsym, bool have_coef, int *nretried) {
    if (nretried) *nretried = 0;
    std::vector<int> einfo((size_t)2 * batch);
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(einfo.data(), wk.einfo, einfo.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t nn = (size_t)n * n;
    const bool saved = ctx->tri_no_persist;
    int rc = MOPT_OK, cnt = 0;
    for (int b = 0; b < batch && rc == MOPT_OK; ++b) {
        if (!(einfo[2 * b + 1] & 8)) continue;
        RsirfoWork v = wk;                                   // a view of image b (never owns the block)
        v.block = nullptr; v.gaps.clear(); v.gap_names.clear(); v.batch = 1;
        if (v.H) v.H += b * nn;
        if (v.Bias) v.Bias += b * nn;
        if (v.Hinpl) v.Hinpl += b * nn;
        v.A += b * nn; v.Gt += b * nn; v.Vt += b * nn;
        double **vecs[] = {&v.x, &v.bg, &v.g, &v.px, &v.pg, &v.s, &v.y, &v.u, &v.Hy, &v.yd, &v.Av, &v.gp, &v.eigw, &v.gt_all, &v.coefout, &v.step};
        for (double **pp : vecs) if (*pp) *pp += (size_t)b * n;
        v.Q += (size_t)b * 6 * n; v.W6 += (size_t)b * 6 * n;
        v.perm += (size_t)b * n; v.be += b; v.sigma += b; v.info += (size_t)b * 8; v.einfo += 2 * b;
        v.rotcount += (size_t)b * (mopt::JAC_MAX_SWEEPS + 1); v.rules += b; v.do_update += b;
        v.coef += b; v.params += b; v.states += b;
        v.tri.D += (size_t)b * n; v.tri.E += (size_t)b * n; v.tri.TAU += (size_t)b * n; v.tri.WT += (size_t)b * n; v.tri.onenrm += (size_t)b * n;
        v.tri.bstart += (size_t)b * n; v.tri.bend += (size_t)b * n; v.tri.cstart += (size_t)2 * b * n; v.tri.ncl += 2 * b; v.tri.fail += b;
        v.tri.Z2 += (size_t)b * (nn + MOPT_TRI_T_DOUBLES(n)); v.tri.defl += b;
        ctx->tri_no_persist = true;
        rc = mopt_launch_project_hess(ctx, n, 1, d_Hsym + b * nn, v.Bias, v.x, v.Q, v.W6, v.A, /*basis_ready=*/true);
        if (rc == MOPT_OK) rc = mopt_eigh_auto(ctx, v, n, 1, false, /*preserve_A=*/false, nullptr);
        if (rc == MOPT_OK) rc = mopt_launch_rfo(ctx, n, 1, v.params, v.states, v.eigw, v.perm, v.einfo, v.Vt, v.gp, v.bg, v.be, have_coef ? v.coef : nullptr,
                                                v.gt_all, v.coefout, v.step);
        ++cnt;
    }
    ctx->n_retries += cnt;
    ctx->tri_no_persist = saved;
    if (rc != MOPT_OK) return rc;
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (nretried) *nretried = cnt;
    return MOPT_OK;
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
                                        wk.tri.bend, wk.tri.cstart, wk.tri.ncl, wk.tri.fail, wk.eigw, wk.perm, wk.einfo, wk.tri.Z2, hook, wk.sticky, wk.tri.defl, wk.defl_hint);
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
