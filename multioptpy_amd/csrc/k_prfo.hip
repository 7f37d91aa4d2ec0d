// EnhancedRSPRFO.run (multioptpy/Optimizer/rsprfo.py:713-886) on the device, single geometry.
//   k_prfo_prev    _process_previous_step (:908-962): reduction ratio (:434-472), trust radius (:474-512), rejection counter
//   k_prfo_scalar  one wave: eigenvalue shifting (:287-355), mode following (:964-1071), micro cycles with the
//                  augmented-Hessian solves replaced by the secular equation of the arrowhead matrix (:514-711, 1097-1168),
//                  gradient-based scaling (:357-399), adaptive trust radius (:401-432), predicted energy change (:1170-1188)
// The eigen-solve, gradient projection, Hessian update and back-transformation reuse the RSIRFO kernels.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pipeline.h"
#include "prfo_body.hpp"

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
    prfo_prev_logic(p, st, v[0], v[1], v[2], B_e, pre_move_norm);
    state[0] = st;
}

// One wave (prfo_body.hpp).  dynamic LDS: 7 * n doubles + 4 * n ints
__global__ void __launch_bounds__(64) k_prfo_scalar(int n, const mopt_rsprfo_params *__restrict__ params, mopt_rsprfo_state *__restrict__ state,
                                                    const double *__restrict__ W, const int *__restrict__ perm, const int *__restrict__ einfo,
                                                    const double *__restrict__ gt_all, const double *__restrict__ ov_all, int have_ts,
                                                    double *__restrict__ coef_out, int *__restrict__ ts_row_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    prfo_scalar_body<false>(n, params, state, W, perm, einfo, gt_all, ov_all, have_ts, coef_out, ts_row_out, sm);
}

__global__ void k_norm1(int n, const double *__restrict__ v, double *__restrict__ out) {
    __shared__ double red[16];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += v[i] * v[i];
    a = block_sum1(a, red);
    if (threadIdx.x == 0) out[0] = sqrt(a);
}

__global__ void k_set_gnorm(mopt_rsprfo_state *st, const double *gn) { st[0].gradient_norm = gn[0]; }
// outcome of the checked update (rsprfo.py:1236-1250) without a host round trip: flag = 0 no update applied, 1 accepted (the bound
// ||candidate||_inf / 1.0625 <= 1e6 holds, so max |eig| <= 1e6 does too), 2 the spectrum itself has to be looked at (the caller repeats
// the call through the path that does)
// (sigma NaN / Inf: `<= 1e6` is false -> flag 2 -> the general path computes the spectrum and applies the reference's own test, which accepts a
// non-finite candidate because `NaN > 1e6` is False, rsprfo.py:1246-1255 with numpy 2.2)
__global__ void k_prfo_accept(const UpdateCoef *__restrict__ coef, const double *__restrict__ sigma, int *__restrict__ flag) {
    flag[0] = coef[0].applied ? ((sigma[0] / 1.0625 <= 1e6) ? 1 : 2) : 0;
}
__global__ void __launch_bounds__(256) k_copy_if(size_t count, const int *__restrict__ flag, int want, const double *__restrict__ src, double *__restrict__ dst) {
    if (flag[0] != want) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
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

int mopt_ws_get_small(mopt_ctx *ctx, int n, RsirfoWork **wk_out);      // mopt_abi.hip

// Small geometries: one launch (k_small_rsprfo), inputs and results through the context's pinned span, completion counter instead of a
// stream synchronisation (as rsirfo_run_small, mopt_abi.hip).  *fallback: the kernel asked for the general path (status 2).
static int rsprfo_run_small(mopt_ctx *ctx, int n, const mopt_rsprfo_params *params, mopt_rsprfo_state *state, const double *hessian,
                            const double *bias_hessian, const double *geom, const double *B_g, const double *pre_geom, const double *pre_B_g, double B_e,
                            const double *prev_gradient, const double *prev_move_vector, double pre_move_norm, const double *ts_vec, double *step,
                            double *H_out, double *ts_vec_out, bool *fallback) {
    *fallback = false;
    hipSetDevice(ctx->device);
    RsirfoWork *wkp = nullptr;
    PTRY(mopt_ws_get_small(ctx, n, &wkp));
    const size_t nn = (size_t)n * n * 8, nv = (size_t)n * 8;
    auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_x = take(nv), o_bg = take(nv), o_px = take(nv), o_pg = take(nv), o_prevg = take(nv), o_prevs = take(nv), o_ts = take(nv),
                 o_par = take(sizeof(mopt_rsprfo_params)), o_st = take(sizeof(mopt_rsprfo_state)), o_step = take(nv), o_tsout = take(nv), o_flags = take(64),
                 o_H = take(nn), o_B = bias_hessian ? take(nn) : 0, o_Hout = take(nn), o_done = take(64);
    char *pin = (char *)mopt_pin_get(ctx, off);
    if (!pin) { ctx->err = "hipHostMalloc failed (staging buffer)"; return MOPT_E_NOMEM; }
    char *dpin = nullptr;
    MOPT_HIP_TRY(ctx, hipHostGetDevicePointer((void **)&dpin, pin, 0));
    mopt_rsprfo_state st = *state;
    st.last_ratio = NAN; st.last_actual = NAN; st.step_accepted = 1; st.hess_updated = 0;
    const bool do_prev = st.has_prev && prev_gradient && prev_move_vector;
    const bool do_upd = pre_geom && pre_B_g;
    const int have_ts = (ts_vec != nullptr && st.ts_mode_idx >= 0) ? 1 : 0;
    std::memcpy(pin + o_x, geom, nv); std::memcpy(pin + o_bg, B_g, nv);
    if (do_upd) { std::memcpy(pin + o_px, pre_geom, nv); std::memcpy(pin + o_pg, pre_B_g, nv); }
    if (do_prev) { std::memcpy(pin + o_prevg, prev_gradient, nv); std::memcpy(pin + o_prevs, prev_move_vector, nv); }
    if (have_ts) std::memcpy(pin + o_ts, ts_vec, nv);
    std::memcpy(pin + o_par, params, sizeof(mopt_rsprfo_params));
    std::memcpy(pin + o_st, &st, sizeof(st));
    std::memcpy(pin + o_H, hessian, nn);
    if (bias_hessian) std::memcpy(pin + o_B, bias_hessian, nn);
    int *flags = (int *)(pin + o_flags);
    flags[0] = 0; flags[1] = -1; flags[2] = 0; flags[3] = -1;
    volatile unsigned long long *done = (volatile unsigned long long *)(pin + o_done);
    *done = 0ull;
    SmallPrfoIO io;
    io.n = n; io.has_prev_step = do_prev ? 1 : 0; io.do_update = do_upd ? 1 : 0; io.have_ts = have_ts;
    io.B_e = B_e; io.pre_move_norm = pre_move_norm;
    io.params = (const mopt_rsprfo_params *)(dpin + o_par); io.state = (mopt_rsprfo_state *)(dpin + o_st);
    io.x = (const double *)(dpin + o_x); io.bg = (const double *)(dpin + o_bg); io.px = (const double *)(dpin + o_px); io.pg = (const double *)(dpin + o_pg);
    io.prevg = (const double *)(dpin + o_prevg); io.prevs = (const double *)(dpin + o_prevs); io.ts = (const double *)(dpin + o_ts);
    io.Hin = (const double *)(dpin + o_H); io.Bias = bias_hessian ? (const double *)(dpin + o_B) : nullptr;
    io.Hout = (double *)(dpin + o_Hout); io.step = (double *)(dpin + o_step); io.tsout = ts_vec_out ? (double *)(dpin + o_tsout) : nullptr;
    io.flags = (int *)(dpin + o_flags); io.done = (unsigned long long *)(dpin + o_done);
    PTRY(mopt_launch_small_rsprfo(ctx, *wkp, io));
    {
        static const bool no_poll = getenv("MOPT_SMALL_NOPOLL") != nullptr;
        bool seen = false;
        if (!no_poll) {
            const auto t0 = std::chrono::steady_clock::now();
            for (long it = 0;; ++it) {
                if (__atomic_load_n((const unsigned long long *)done, __ATOMIC_ACQUIRE) >= 1ull) { seen = true; break; }
                if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
        }
        if (!seen) MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (flags[3] == 2) { *fallback = true; return MOPT_OK; }
    if (flags[3] != 0) { ctx->err = "k_small_rsprfo did not finish"; return MOPT_E_DEVICE; }
    std::memcpy(step, pin + o_step, nv);
    std::memcpy((void *)&st, pin + o_st, sizeof(st));
    const bool updated = flags[0] != 0;
    if (ts_vec_out && flags[1] >= 0) std::memcpy(ts_vec_out, pin + o_tsout, nv);
    if (updated) std::memcpy(H_out, pin + o_Hout, nn);
    else if (H_out != hessian) std::memcpy(H_out, hessian, nn);
    st.hess_updated = updated ? 1 : 0;
    *state = st;
    return MOPT_OK;
}


int mopt_ws_get(mopt_ctx *ctx, int n, bool bias, size_t io_bytes, RsirfoWork **wk_out, void **io_out);      // mopt_abi.hip

// The general path without its per-call allocations and without the synchronisation in the middle (round 3): cached workspace, the small
// inputs through the pinned span in one copy, the accept / reject decision of the checked update taken on the device (k_prfo_accept); only the
// rare case that needs the candidate's spectrum (flag 2) goes back to rsprfo_run_once.  n = 300: 2.07 -> see DESIGN section 6.
static int rsprfo_run_fast(mopt_ctx *ctx, int n, const mopt_rsprfo_params *params, mopt_rsprfo_state *state, const double *hessian,
                           const double *bias_hessian, const double *geom, const double *B_g, const double *pre_geom, const double *pre_B_g, double B_e,
                           const double *prev_gradient, const double *prev_move_vector, double pre_move_norm, const double *ts_vec, double *step,
                           double *H_out, double *ts_vec_out, bool *timed_out, bool *fallback) {
    *timed_out = false; *fallback = false;
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)n * n * 8, nv = (size_t)n * 8;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // device span: [x | bg | px | pg | prevg | prevs | ts | par | st | rule | one]  then  [ov | tsrow | gn | flag]  then out [step | st | tsrow | bad | flag | tsout]
    size_t off = 0;
    auto take = [&](size_t b) { const size_t o = off; off += al(b); return o; };
    const size_t o_x = take(nv), o_bg = take(nv), o_px = take(nv), o_pg = take(nv), o_prevg = take(nv), o_prevs = take(nv), o_ts = take(nv),
                 o_par = take(sizeof(mopt_rsprfo_params)), o_st = take(sizeof(mopt_rsprfo_state)), o_rule = take(8), o_one = take(8), in_bytes = off,
                 o_ov = take(nv), o_tsrow = take(16), o_gn = take(16), o_flag = take(16);
    RsirfoWork *wkp = nullptr;
    void *io = nullptr;
    PTRY(mopt_ws_get(ctx, n, bias_hessian != nullptr, off, &wkp, &io));
    RsirfoWork &wk = *wkp;
    char *dio = (char *)io;
    const bool stage_H = nn <= ((size_t)1 << 20);
    const size_t p_out = al(in_bytes), p_st = p_out + al(nv), p_row = p_st + al(sizeof(mopt_rsprfo_state)), p_bad = p_row + 256, p_flag = p_bad + 256,
                 p_ts = p_flag + 256, p_H = p_ts + al(nv);
    char *pin = (char *)mopt_pin_get(ctx, p_H + (stage_H ? nn : 0));
    if (!pin) { ctx->err = "hipHostMalloc failed (staging buffer)"; return MOPT_E_NOMEM; }
    mopt_rsprfo_state st = *state;
    st.last_ratio = NAN; st.last_actual = NAN; st.step_accepted = 1; st.hess_updated = 0;
    const bool do_prev = st.has_prev && prev_gradient && prev_move_vector, do_upd = pre_geom && pre_B_g;
    const int have_ts = (ts_vec != nullptr && st.ts_mode_idx >= 0) ? 1 : 0, rule = params->rule, one = 1;
    std::memcpy(pin + o_x, geom, nv); std::memcpy(pin + o_bg, B_g, nv);
    if (do_upd) { std::memcpy(pin + o_px, pre_geom, nv); std::memcpy(pin + o_pg, pre_B_g, nv); }
    if (do_prev) { std::memcpy(pin + o_prevg, prev_gradient, nv); std::memcpy(pin + o_prevs, prev_move_vector, nv); }
    if (have_ts) std::memcpy(pin + o_ts, ts_vec, nv);
    std::memcpy(pin + o_par, params, sizeof(mopt_rsprfo_params)); std::memcpy(pin + o_st, &st, sizeof(st));
    std::memcpy(pin + o_rule, &rule, sizeof(int)); std::memcpy(pin + o_one, &one, sizeof(int));
    PH2D(dio, pin, in_bytes);
    PH2D(wk.H, hessian, nn);
    if (bias_hessian) PH2D(wk.Bias, bias_hessian, nn);
    double *dx = (double *)(dio + o_x), *dbg = (double *)(dio + o_bg), *dpx = (double *)(dio + o_px), *dpg = (double *)(dio + o_pg),
           *dprevg = (double *)(dio + o_prevg), *dprevs = (double *)(dio + o_prevs), *dts = (double *)(dio + o_ts), *dov = (double *)(dio + o_ov),
           *dgn = (double *)(dio + o_gn);
    mopt_rsprfo_params *dpar = (mopt_rsprfo_params *)(dio + o_par);
    mopt_rsprfo_state *dst = (mopt_rsprfo_state *)(dio + o_st);
    int *drule = (int *)(dio + o_rule), *done_ = (int *)(dio + o_one), *dtsrow = (int *)(dio + o_tsrow), *dflag = (int *)(dio + o_flag);
    MOPT_HIP_TRY(ctx, hipMemsetAsync(dflag, 0, sizeof(int), ctx->stream));
    // ---- 1. previous step
    if (do_prev) {
        PTRY(mopt_launch_project_hess(ctx, n, 1, wk.H, wk.Bias, dx, wk.Q, wk.W6, wk.A));
        PTRY(mopt_launch_matvec(ctx, n, 1, wk.A, dprevs, wk.u));
        hipLaunchKernelGGL(k_prfo_prev, dim3(1), dim3(256), 0, ctx->stream, n, dpar, dst, dprevg, dprevs, wk.u, B_e, pre_move_norm);
    }
    // ---- 2. checked update: candidate in wk.A, accepted into wk.H on the device
    if (do_upd) {
        PTRY(mopt_launch_hess_update(ctx, n, 1, drule, done_, rule == MOPT_RULE_FLOWCHART, dx, dpx, dbg, dpg, wk.s, wk.y, wk.u, wk.Hy, wk.yd, wk.Av, wk.coef,
                                     wk.info, wk.H, nullptr, wk.A, false, /*guard_sy=*/false));
        PTRY(mopt_launch_norm_inf(ctx, n, 1, wk.A, wk.sigma));
        hipLaunchKernelGGL(k_prfo_accept, dim3(1), dim3(1), 0, ctx->stream, (const UpdateCoef *)wk.coef, (const double *)wk.sigma, dflag);
        int blocks = (int)(((size_t)n * n + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_copy_if, dim3(blocks), dim3(256), 0, ctx->stream, (size_t)n * n, (const int *)dflag, 1, (const double *)wk.A, wk.H);
    }
    // ---- 3. projected gradient and its norm; 4. eigh(H + bias); 5. scalar stage, step, tracked vector
    PTRY(mopt_launch_project_grad(ctx, n, 1, dbg, dx, wk.gp, /*variant=*/1));
    hipLaunchKernelGGL(k_norm1, dim3(1), dim3(256), 0, ctx->stream, n, wk.gp, dgn);
    hipLaunchKernelGGL(k_set_gnorm, dim3(1), dim3(1), 0, ctx->stream, dst, dgn);
    PTRY(mopt_launch_sym_add(ctx, n, 1, wk.H, wk.Bias, wk.A));
    PTRY(mopt_eigh_auto(ctx, wk, n, 1, false));
    PTRY(mopt_launch_vtg(ctx, n, 1, wk.Vt, wk.gp, wk.einfo, wk.gt_all));
    if (have_ts) PTRY(mopt_launch_vtg(ctx, n, 1, wk.Vt, dts, wk.einfo, dov));
    const size_t lds = (size_t)7 * n * 8 + (size_t)4 * n * 4 + 16;
    if (!g_prfo_attr) {
        MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_prfo_scalar, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
        g_prfo_attr = 1;
    }
    hipLaunchKernelGGL(k_prfo_scalar, dim3(1), dim3(64), lds, ctx->stream, n, (const mopt_rsprfo_params *)dpar, dst, wk.eigw, wk.perm, wk.einfo, wk.gt_all,
                       (const double *)dov, have_ts, wk.coefout, dtsrow);
    PTRY(mopt_launch_backtransform(ctx, n, 1, wk.Vt, wk.coefout, wk.einfo, wk.step));
    PD2H(pin + p_out, wk.step, nv);
    PD2H(pin + p_st, dst, sizeof(st));
    PD2H(pin + p_row, dtsrow, sizeof(int));
    PD2H(pin + p_bad, wk.einfo + 1, sizeof(int));
    PD2H(pin + p_flag, dflag, sizeof(int));
    if (ts_vec_out) {
        hipLaunchKernelGGL(k_prfo_tsrow, dim3(1), dim3(256), 0, ctx->stream, n, wk.Vt, (const int *)dtsrow, wk.einfo, dov);
        PD2H(pin + p_ts, dov, nv);
    }
    if (do_upd) PD2H(stage_H ? (void *)(pin + p_H) : (void *)H_out, wk.H, nn);      // (unchanged H when the update was not applied: same bits as the caller's)
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    MOPT_HIP_TRY(ctx, hipGetLastError());
    int ts_row = -1, bad = 0, flag = 0;
    std::memcpy(&flag, pin + p_flag, sizeof(int));
    if (flag == 2) { *fallback = true; return MOPT_OK; }     // (H_out may have received the caller's own matrix back: harmless)
    std::memcpy(step, pin + p_out, nv);
    std::memcpy((void *)&st, pin + p_st, sizeof(st));
    std::memcpy(&ts_row, pin + p_row, sizeof(int));
    std::memcpy(&bad, pin + p_bad, sizeof(int));
    const bool updated = flag == 1;
    if (ts_vec_out && ts_row >= 0) std::memcpy(ts_vec_out, pin + p_ts, nv);
    if (do_upd && stage_H && updated) std::memcpy(H_out, pin + p_H, nn);
    if (!updated && H_out != hessian) std::memcpy(H_out, hessian, nn);
    st.hess_updated = updated ? 1 : 0;
    *state = st;
    *timed_out = (bad & 8) != 0;
    return MOPT_OK;
}

extern "C" int mopt_rsprfo_run(mopt_ctx *ctx, int n, const mopt_rsprfo_params *params, mopt_rsprfo_state *state,
                               const double *hessian, const double *bias_hessian, const double *geom, const double *B_g,
                               const double *pre_geom, const double *pre_B_g, double B_e, const double *prev_gradient,
                               const double *prev_move_vector, double pre_move_norm, const double *ts_vec, double *step,
                               double *H_out, double *ts_vec_out) {
    if (!ctx || !state) return MOPT_E_ARG;
    // A hand-off time-out of the persistent tridiagonalisation (GPU shared with a long-running kernel) is not a numerical failure: the
    // call is repeated once from the caller's inputs with one launch per Householder step (H_out may alias `hessian`: keep a copy).
    if (ctx && params && state && geom && B_g && step && H_out && hessian && mopt_small_eligible(n) && !mopt_debug_poison_on() &&
        params->rule >= 0 && params->rule < MOPT_RULE_COUNT) {
        bool fallback = false;
        const int rc = rsprfo_run_small(ctx, n, params, state, hessian, bias_hessian, geom, B_g, pre_geom, pre_B_g, B_e, prev_gradient, prev_move_vector,
                                        pre_move_norm, ts_vec, step, H_out, ts_vec_out, &fallback);
        if (rc != MOPT_OK || !fallback) return rc;           // (fallback: nothing of the caller's was touched; the general path below starts over)
    }
    const mopt_rsprfo_state st0 = *state;
    std::vector<double> hkeep;
    if (H_out == hessian && hessian && n > 0) hkeep.assign(hessian, hessian + (size_t)n * n);
    bool timed_out = false, force_np = false;
    static const bool no_fast = getenv("MOPT_PRFO_FAST") && atoi(getenv("MOPT_PRFO_FAST")) == 0;
    if (!no_fast && ctx && params && geom && B_g && step && H_out && hessian && n >= 3 && n % 3 == 0 && n <= MOPT_MAX_N && !mopt_debug_poison_on() &&
        params->rule >= 0 && params->rule < MOPT_RULE_COUNT) {
        bool fallback = false;
        const int rcf = rsprfo_run_fast(ctx, n, params, state, hessian, bias_hessian, geom, B_g, pre_geom, pre_B_g, B_e, prev_gradient, prev_move_vector,
                                        pre_move_norm, ts_vec, step, H_out, ts_vec_out, &timed_out, &fallback);
        if (rcf != MOPT_OK) return rcf;
        if (!fallback && !timed_out) return MOPT_OK;
        // (spectrum check needed, or a hand-off time-out: start over from the caller's inputs on the path that handles them)
        *state = st0;
        if (!hkeep.empty() && H_out == hessian) std::memcpy(H_out, hkeep.data(), hkeep.size() * sizeof(double));
        force_np = timed_out;
        timed_out = false;
    }
    const bool saved_np = ctx->tri_no_persist;
    if (force_np) ctx->tri_no_persist = true;
    int rc = rsprfo_run_once(ctx, n, params, state, hessian, bias_hessian, geom, B_g, pre_geom, pre_B_g, B_e, prev_gradient, prev_move_vector,
                             pre_move_norm, ts_vec, step, H_out, ts_vec_out, &timed_out);
    ctx->tri_no_persist = saved_np;
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
    const size_t lds = (size_t)7 * n * 8 + (size_t)4 * n * 4 + 16;
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
