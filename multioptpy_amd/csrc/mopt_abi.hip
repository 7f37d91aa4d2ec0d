// C ABI of libmopt_hip.so (see include/mopt_hip.h): context, workspaces and the host-pointer entry points.
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <new>

#include "common.hpp"
#include "logic.h"
#include <algorithm>
#include "pipeline.h"

using namespace mopt;

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
extern "C" int mopt_abi_version(void) { return MOPT_ABI_VERSION; }

extern "C" int mopt_ctx_create(int device, mopt_ctx **out) {
    if (!out) return MOPT_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MOPT_E_DEVICE;
    mopt_ctx *c = new (std::nothrow) mopt_ctx();
    if (!c) return MOPT_E_NOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return MOPT_E_DEVICE; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MOPT_E_DEVICE; }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork3, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork4, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join4, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side3, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork5, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join5, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork6, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join6, hipEventDisableTiming) != hipSuccess) {
        if (c->ev_fork6) hipEventDestroy(c->ev_fork6);
        if (c->ev_join5) hipEventDestroy(c->ev_join5);
        if (c->ev_fork5) hipEventDestroy(c->ev_fork5);
        if (c->side3) hipStreamDestroy(c->side3);
        if (c->ev_join4) hipEventDestroy(c->ev_join4);
        if (c->ev_join3) hipEventDestroy(c->ev_join3);
        if (c->ev_fork4) hipEventDestroy(c->ev_fork4);
        if (c->side2) hipStreamDestroy(c->side2);
        if (c->ev_join2) hipEventDestroy(c->ev_join2);
        if (c->ev_fork3) hipEventDestroy(c->ev_fork3);
        if (c->ev_fork) hipEventDestroy(c->ev_fork);
        if (c->ev_join) hipEventDestroy(c->ev_join);
        if (c->ev_fork2) hipEventDestroy(c->ev_fork2);
        if (c->side) hipStreamDestroy(c->side);
        hipStreamDestroy(c->stream);
        delete c;
        return MOPT_E_DEVICE;
    }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) c->lds_limit = v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->num_cu = v;
    *out = c;
    return MOPT_OK;
}

// Pinned host staging per context (grow-only; freed by mopt_ctx_destroy): the small inputs / outputs of the single-geometry entry
// points travel through it in one copy per direction.
namespace {
struct PinBuf { void *p = nullptr; size_t bytes = 0; };
std::map<mopt_ctx *, PinBuf> g_pin;
std::mutex g_pin_mu;
void *pin_get_impl(mopt_ctx *ctx, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    PinBuf &b = g_pin[ctx];
    if (b.bytes < bytes) {
        if (b.p) hipHostFree(b.p);
        b.p = nullptr; b.bytes = 0;
        const size_t want = bytes < (1u << 16) ? (1u << 16) : bytes;
        if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) { b.p = nullptr; return nullptr; }
        b.bytes = want;
    }
    return b.p;
}
void *pin_get(mopt_ctx *ctx, size_t bytes) { return pin_get_impl(ctx, bytes); }
void pin_release(mopt_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = g_pin.find(ctx);
    if (it != g_pin.end()) { if (it->second.p) hipHostFree(it->second.p); g_pin.erase(it); }
}
}  // namespace

// (for the other host entry points of the library: k_prfo.hip)
void *mopt_pin_get(mopt_ctx *ctx, size_t bytes) { return pin_get_impl(ctx, bytes); }


void mopt_ws_release(mopt_ctx *ctx);
void mopt_mf_release(mopt_ctx *ctx);

extern "C" int mopt_ctx_destroy(mopt_ctx *ctx) {
    if (!ctx) return MOPT_E_ARG;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->timers) for (auto &e : kv.second.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (ctx->arena) hipFree(ctx->arena);
    hipStreamSynchronize(ctx->side);
    hipEventDestroy(ctx->ev_fork); hipEventDestroy(ctx->ev_join); hipEventDestroy(ctx->ev_fork2); hipEventDestroy(ctx->ev_join2);
    hipEventDestroy(ctx->ev_fork3); hipEventDestroy(ctx->ev_join3);
    hipStreamSynchronize(ctx->side2);
    hipEventDestroy(ctx->ev_fork4); hipEventDestroy(ctx->ev_join4);
    hipStreamSynchronize(ctx->side3);
    hipEventDestroy(ctx->ev_fork5); hipEventDestroy(ctx->ev_join5);
    hipEventDestroy(ctx->ev_fork6); hipEventDestroy(ctx->ev_join6);
    mopt_mf_release(ctx);
    mopt_ws_release(ctx);
    if (ctx->early_buf) hipFree(ctx->early_buf);
    if (ctx->defl.ev) { hipEventSynchronize(ctx->defl.ev); hipEventDestroy(ctx->defl.ev); }
    if (ctx->defl.pin) hipHostFree(ctx->defl.pin);
    pin_release(ctx);
    hipStreamDestroy(ctx->side3);
    hipStreamDestroy(ctx->side2);
    hipStreamDestroy(ctx->side);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return MOPT_OK;
}

namespace { __global__ void __launch_bounds__(256) k_debug_occupy(long long ticks, double *sink) {
    extern __shared__ double hog[];
    const long long t0 = wall_clock64();
    double a = 0.0;
    while (wall_clock64() - t0 < ticks) { hog[threadIdx.x] = a; a += hog[(threadIdx.x + 1) & 255] * 1e-30 + 1.0; __builtin_amdgcn_s_sleep(8); }
    if (a < 0.0) sink[0] = a;
} }

// Test hook: keep `blocks` workgroups with `lds_bytes` of LDS each resident for about `ms` milliseconds on a stream of their own
// (a co-tenant for the persistent kernels); returns at once.
extern "C" int mopt_debug_occupy(mopt_ctx *ctx, double ms, int lds_bytes, int blocks) {
    if (!ctx || blocks < 1 || lds_bytes < 2048) return MOPT_E_ARG;
    hipSetDevice(ctx->device);
    static hipStream_t st = nullptr;
    if (!st) MOPT_HIP_TRY(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_debug_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
    hipLaunchKernelGGL(k_debug_occupy, dim3(blocks), dim3(256), (size_t)lds_bytes, st, (long long)(ms * 1e5), (double *)nullptr);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

extern "C" const char *mopt_last_error(const mopt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int mopt_ctx_sync(mopt_ctx *ctx) {
    if (!ctx) return MOPT_E_ARG;
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // (the helper streams are joined into the main stream by events wherever a result is needed; what may still be in flight on them
    // is bookkeeping -- the tail of an idle divide-and-conquer chain, the read-back of the deflation outcome)
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->side));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->side2));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->side3));
    return MOPT_OK;
}

extern "C" int mopt_ctx_timing(mopt_ctx *ctx, int enable) {
    if (!ctx) return MOPT_E_ARG;
    hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->timers) for (auto &e : kv.second.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    ctx->timers.clear();
    ctx->timing = enable != 0;
    return MOPT_OK;
}

extern "C" int mopt_ctx_timing_read(mopt_ctx *ctx, const char *family, double *ms, int64_t *launches) {
    if (!ctx || !family) return MOPT_E_ARG;
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (std::string(family) == "timeline") {
        // developer aid (MOPT timing on): every timed scope of the last ~2 steps in start order -- start and length in us relative to
        // the first of them, and the idle time since the end of the previous scope on the list -- to stderr
        struct Rec { std::string name; float start, len; };
        std::vector<Rec> recs;
        hipEvent_t base = nullptr;
        for (auto &kv : ctx->timers) if (!kv.second.ev.empty()) { base = kv.second.ev.front().first; break; }
        if (base) {
            for (auto &kv : ctx->timers)
                for (auto &e : kv.second.ev) {
                    float a = 0.f, l = 0.f;
                    if (hipEventElapsedTime(&a, base, e.first) == hipSuccess && hipEventElapsedTime(&l, e.first, e.second) == hipSuccess) recs.push_back({kv.first, a * 1e3f, l * 1e3f});
                }
            std::sort(recs.begin(), recs.end(), [](const Rec &x, const Rec &y) { return x.start < y.start; });
            const char *all = getenv("MOPT_TIMELINE_ALL");
            const size_t from = (all || recs.size() <= 40) ? 0 : recs.size() - 40;
            float prev_end = recs.empty() ? 0.f : recs[from].start;
            for (size_t i = from; i < recs.size(); ++i) {
                fprintf(stderr, "[timeline] %9.1f us  %-16s %8.1f us   (gap %6.1f)\n", recs[i].start - recs[from].start, recs[i].name.c_str(), recs[i].len, recs[i].start - prev_end);
                if (recs[i].start + recs[i].len > prev_end) prev_end = recs[i].start + recs[i].len;
            }
        }
        if (ms) *ms = 0.0;
        if (launches) *launches = 0;
        return MOPT_OK;
    }
    if (std::string(family) == "tri_retries") {              // (not a timer: images solved again after a hand-off time-out, since the context exists)
        if (ms) *ms = 0.0;
        if (launches) *launches = ctx->n_retries;
        return MOPT_OK;
    }
    auto it = ctx->timers.find(family);
    double tot = 0.0; int64_t nl = 0;
    if (it != ctx->timers.end()) {
        for (auto &e : it->second.ev) { float t = 0.f; if (hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) tot += t; }
        nl = it->second.launches;
    }
    if (ms) *ms = tot;
    if (launches) *launches = nl;
    return MOPT_OK;
}

// ------------------------------------------------------------------------------------------
// small RAII device buffer for the host-pointer entry points
// ------------------------------------------------------------------------------------------
namespace {
struct DBuf {
    void *p = nullptr;
    ~DBuf() { if (p) hipFree(p); }
    int alloc(mopt_ctx *ctx, size_t bytes) {
        if (bytes == 0) bytes = 8;
        if (hipMalloc(&p, bytes) != hipSuccess) { ctx->err = "hipMalloc failed"; p = nullptr; return MOPT_E_NOMEM; }
        return MOPT_OK;
    }
    template <class T> T *as() { return (T *)p; }
};
#define TRY(x) do { int rc__ = (x); if (rc__ != MOPT_OK) return rc__; } while (0)
#define H2D(ctx, dst, src, bytes) MOPT_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (ctx)->stream))
#define D2H(ctx, dst, src, bytes) MOPT_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (ctx)->stream))
}  // namespace

static bool bad_dims(mopt_ctx *ctx, int n, int batch) {
    if (!ctx) return true;
    if (n < 3 || n % 3 != 0 || batch < 1 || n > MOPT_MAX_N) { ctx->err = "bad dimensions (need n = 3N, 3 <= n <= MOPT_MAX_N, batch >= 1)"; return true; }
    return false;
}

// ------------------------------------------------------------------------------------------
// building blocks
// ------------------------------------------------------------------------------------------
extern "C" int mopt_hess_update(mopt_ctx *ctx, int rule, int n, int batch, const double *H, const double *s, const double *y,
                                double *H_inplace, double *H_sym, double *info) {
    if (bad_dims(ctx, n, batch) || !H || !s || !y || !H_sym || rule < 0 || rule >= MOPT_RULE_COUNT) return MOPT_E_ARG;
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)batch * n * n * sizeof(double), nv = (size_t)batch * n * sizeof(double);
    DBuf dH, dHi, dHs, ds, dy, du, dHy, dyd, dA, dcoef, dinfo, drules;
    TRY(dH.alloc(ctx, nn)); TRY(dHs.alloc(ctx, nn)); if (H_inplace) TRY(dHi.alloc(ctx, nn));
    TRY(ds.alloc(ctx, nv)); TRY(dy.alloc(ctx, nv)); TRY(du.alloc(ctx, nv)); TRY(dHy.alloc(ctx, nv)); TRY(dyd.alloc(ctx, nv)); TRY(dA.alloc(ctx, nv));
    TRY(dcoef.alloc(ctx, batch * sizeof(UpdateCoef))); TRY(dinfo.alloc(ctx, (size_t)batch * 8 * sizeof(double)));
    TRY(drules.alloc(ctx, batch * sizeof(int)));
    std::vector<int> rules(batch, rule);
    H2D(ctx, dH.p, H, nn); H2D(ctx, ds.p, s, nv); H2D(ctx, dy.p, y, nv); H2D(ctx, drules.p, rules.data(), batch * sizeof(int));
    TRY(mopt_launch_hess_update(ctx, n, batch, drules.as<int>(), nullptr, rule == MOPT_RULE_FLOWCHART, nullptr, nullptr, nullptr, nullptr,
                                ds.as<double>(), dy.as<double>(), du.as<double>(), dHy.as<double>(), dyd.as<double>(), dA.as<double>(),
                                dcoef.as<UpdateCoef>(), dinfo.as<double>(), dH.as<double>(), H_inplace ? dHi.as<double>() : nullptr,
                                dHs.as<double>(), true));
    D2H(ctx, H_sym, dHs.p, nn);
    if (H_inplace) D2H(ctx, H_inplace, dHi.p, nn);
    if (info) D2H(ctx, info, dinfo.p, (size_t)batch * 8 * sizeof(double));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

namespace { __global__ void __launch_bounds__(256) k_delta(size_t total, const double *__restrict__ Hn, const double *__restrict__ H, double *__restrict__ dH) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) dH[i] = Hn[i] - H[i];
} }

// delta-H of one update rule WITHOUT RSIRFO.update_hessian's guards: what ModelHessianUpdate.<rule>_hessian_update returns
// (hessian_update.py:163-433) -- the form the IRC integrators call (IRC/lqa.py:286, IRC/hpc.py:421).
extern "C" int mopt_hess_delta(mopt_ctx *ctx, int rule, int n, int batch, const double *H, const double *s, const double *y, double *dH) {
    if (bad_dims(ctx, n, batch) || !H || !s || !y || !dH || rule < 0 || rule >= MOPT_RULE_COUNT) return MOPT_E_ARG;
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)batch * n * n * sizeof(double), nv = (size_t)batch * n * sizeof(double);
    DBuf dHb, dHi, dHs, ds, dy, du, dHy, dyd, dA, dcoef, dinfo, drules;
    TRY(dHb.alloc(ctx, nn)); TRY(dHs.alloc(ctx, nn)); TRY(dHi.alloc(ctx, nn));
    TRY(ds.alloc(ctx, nv)); TRY(dy.alloc(ctx, nv)); TRY(du.alloc(ctx, nv)); TRY(dHy.alloc(ctx, nv)); TRY(dyd.alloc(ctx, nv)); TRY(dA.alloc(ctx, nv));
    TRY(dcoef.alloc(ctx, batch * sizeof(UpdateCoef))); TRY(dinfo.alloc(ctx, (size_t)batch * 8 * sizeof(double)));
    TRY(drules.alloc(ctx, batch * sizeof(int)));
    std::vector<int> rules(batch, rule);
    H2D(ctx, dHb.p, H, nn); H2D(ctx, ds.p, s, nv); H2D(ctx, dy.p, y, nv); H2D(ctx, drules.p, rules.data(), batch * sizeof(int));
    TRY(mopt_launch_hess_update(ctx, n, batch, drules.as<int>(), nullptr, rule == MOPT_RULE_FLOWCHART, nullptr, nullptr, nullptr, nullptr,
                                ds.as<double>(), dy.as<double>(), du.as<double>(), dHy.as<double>(), dyd.as<double>(), dA.as<double>(),
                                dcoef.as<UpdateCoef>(), dinfo.as<double>(), dHb.as<double>(), dHi.as<double>(), dHs.as<double>(), true, /*guard_sy=*/false));
    const size_t total = (size_t)batch * n * n;
    hipLaunchKernelGGL(k_delta, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, total, dHi.as<double>(), dHb.as<double>(), dHs.as<double>());
    D2H(ctx, dH, dHs.p, nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_project_trrot_hess(mopt_ctx *ctx, int n, int batch, const double *H, const double *X, double *Hp) {
    if (bad_dims(ctx, n, batch) || !H || !X || !Hp) return MOPT_E_ARG;
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)batch * n * n * sizeof(double), nv = (size_t)batch * n * sizeof(double);
    DBuf dH, dA, dX, dQ, dW;
    TRY(dH.alloc(ctx, nn)); TRY(dA.alloc(ctx, nn)); TRY(dX.alloc(ctx, nv)); TRY(dQ.alloc(ctx, 6 * nv)); TRY(dW.alloc(ctx, 6 * nv));
    H2D(ctx, dH.p, H, nn); H2D(ctx, dX.p, X, nv);
    TRY(mopt_launch_project_hess(ctx, n, batch, dH.as<double>(), nullptr, dX.as<double>(), dQ.as<double>(), dW.as<double>(), dA.as<double>()));
    D2H(ctx, Hp, dA.p, nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_afir(mopt_ctx *ctx, int N, const double *X, const double *cov, int n1, const int *frag1, int n2, const int *frag2, double alpha,
                         double *E, double *grad, double *H) {
    if (!ctx) return MOPT_E_ARG;
    if (bad_dims(ctx, 3 * N, 1) || N < 2 || n1 < 1 || n2 < 1 || !X || !cov || !frag1 || !frag2 || !E || !grad || !H) return MOPT_E_ARG;
    for (int a = 0; a < n1; ++a) if (frag1[a] < 0 || frag1[a] >= N) { ctx->err = "mopt_afir: fragment atom out of range (0-based indices)"; return MOPT_E_ARG; }
    for (int b = 0; b < n2; ++b) {
        if (frag2[b] < 0 || frag2[b] >= N) { ctx->err = "mopt_afir: fragment atom out of range (0-based indices)"; return MOPT_E_ARG; }
        for (int a = 0; a < n1; ++a) if (frag1[a] == frag2[b]) { ctx->err = "mopt_afir: the fragments share an atom"; return MOPT_E_ARG; }
    }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    DBuf dX, dC, dF1, dF2, dW, dE, dG, dH;
    TRY(dX.alloc(ctx, (size_t)n * 8)); TRY(dC.alloc(ctx, (size_t)N * 8)); TRY(dF1.alloc(ctx, (size_t)n1 * 4)); TRY(dF2.alloc(ctx, (size_t)n2 * 4));
    TRY(dW.alloc(ctx, (size_t)(2 + 6 * N) * 8)); TRY(dE.alloc(ctx, 8)); TRY(dG.alloc(ctx, (size_t)n * 8)); TRY(dH.alloc(ctx, (size_t)n * n * 8));
    H2D(ctx, dX.p, X, (size_t)n * 8); H2D(ctx, dC.p, cov, (size_t)N * 8); H2D(ctx, dF1.p, frag1, (size_t)n1 * 4); H2D(ctx, dF2.p, frag2, (size_t)n2 * 4);
    TRY(mopt_launch_afir(ctx, N, dX.as<double>(), dC.as<double>(), n1, dF1.as<int>(), n2, dF2.as<int>(), alpha, dW.as<double>(), dE.as<double>(),
                         dG.as<double>(), dH.as<double>()));
    D2H(ctx, E, dE.p, 8); D2H(ctx, grad, dG.p, (size_t)n * 8); D2H(ctx, H, dH.p, (size_t)n * n * 8);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_model_hessian_ex(mopt_ctx *ctx, int kind, int flags, int N, int batch, const double *X, const double *par, const double *d3, const double *G,
                                     int project, double *H) {
    if (!ctx) return MOPT_E_ARG;
    if (kind < MOPT_MODEL_FISCHERD3OLD || kind > MOPT_MODEL_LINDH2007D3) { ctx->err = "mopt_model_hessian: unknown model"; return MOPT_E_ARG; }
    if (bad_dims(ctx, 3 * N, batch) || N < 1 || !X || !par || !d3 || !H) return MOPT_E_ARG;
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t nn = (size_t)batch * n * n * sizeof(double), nv = (size_t)batch * n * sizeof(double);
    DBuf dH, dA, dX, dQ, dW, dP, dG;
    TRY(dA.alloc(ctx, nn)); TRY(dX.alloc(ctx, nv)); TRY(dP.alloc(ctx, (size_t)6 * N * 8));
    H2D(ctx, dX.p, X, nv); H2D(ctx, dP.p, par, (size_t)6 * N * 8);
    const bool lindh = (kind == MOPT_MODEL_LINDH2007D3), clip = (flags & MOPT_MODEL_FLAG_CLIP) != 0;
    RsirfoWork wk;                                           // eigen-solver workspace: only the Lindh model and the clip modifier need one
    if (lindh || clip) TRY(wk.alloc(ctx, n, batch, false, false, false));
    if (lindh) {
        if (G) { TRY(dG.alloc(ctx, nv)); H2D(ctx, dG.p, G, nv); }
        TRY(mopt_lindh2007d3_full(ctx, N, batch, dX.as<double>(), dP.as<double>(), d3, G ? dG.as<double>() : nullptr, wk, dA.as<double>()));
    } else {
        TRY(dH.alloc(ctx, nn)); TRY(dQ.alloc(ctx, 6 * nv)); TRY(dW.alloc(ctx, 6 * nv));
        TRY(mopt_launch_model_hessian(ctx, kind, N, batch, dX.as<double>(), dP.as<double>(), d3, dH.as<double>(), dQ.as<double>(), dW.as<double>(), dA.as<double>(), project != 0));
    }
    if (clip) TRY(mopt_model_hessian_clip(ctx, n, batch, wk, dA.as<double>()));
    D2H(ctx, H, dA.p, nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (lindh || clip) {
        std::vector<int> einfo((size_t)2 * batch);
        MOPT_HIP_TRY(ctx, hipMemcpy(einfo.data(), wk.einfo, einfo.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int b = 0; b < batch; ++b) if (einfo[2 * b + 1] & 4) { ctx->err = "mopt_model_hessian: the eigen-solver failed"; return MOPT_E_NUMERIC; }
    }
    return MOPT_OK;
}

extern "C" int mopt_model_hessian(mopt_ctx *ctx, int kind, int N, int batch, const double *X, const double *par, const double *d3, int project, double *H) {
    return mopt_model_hessian_ex(ctx, kind, 0, N, batch, X, par, d3, nullptr, project, H);
}

extern "C" int mopt_project_trrot_grad(mopt_ctx *ctx, int n, int batch, const double *g, const double *X, double *gp) {
    if (bad_dims(ctx, n, batch) || !g || !X || !gp) return MOPT_E_ARG;
    hipSetDevice(ctx->device);
    const size_t nv = (size_t)batch * n * sizeof(double);
    DBuf dg, dX, dgp;
    TRY(dg.alloc(ctx, nv)); TRY(dX.alloc(ctx, nv)); TRY(dgp.alloc(ctx, nv));
    H2D(ctx, dg.p, g, nv); H2D(ctx, dX.p, X, nv);
    TRY(mopt_launch_project_grad(ctx, n, batch, dg.as<double>(), dX.as<double>(), dgp.as<double>()));
    D2H(ctx, gp, dgp.p, nv);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_eigh(mopt_ctx *ctx, int n, int batch, const double *A, double *W, double *V, int *sweeps_out) {
    if (!ctx) return MOPT_E_ARG;
    if (n < 1 || batch < 1 || n > MOPT_MAX_N || !A || !W || !V) { ctx->err = "mopt_eigh: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)batch * n * n * sizeof(double), nv = (size_t)batch * n * sizeof(double);
    DBuf dA, dG, dVt, dV, dsig, drot, dW, dperm, dinfo, ddefl;
    TRY(ddefl.alloc(ctx, (size_t)batch * sizeof(int)));
    ctx->last_defl.clear();
    TRY(dA.alloc(ctx, nn)); TRY(dG.alloc(ctx, nn + (size_t)batch * MOPT_TRI_T_DOUBLES(n) * sizeof(double))); TRY(dVt.alloc(ctx, nn)); TRY(dV.alloc(ctx, nn));
    TRY(dsig.alloc(ctx, batch * sizeof(double))); TRY(drot.alloc(ctx, (size_t)batch * (JAC_MAX_SWEEPS + 1) * sizeof(int)));
    TRY(dW.alloc(ctx, nv)); TRY(dperm.alloc(ctx, (size_t)batch * n * sizeof(int))); TRY(dinfo.alloc(ctx, (size_t)batch * 2 * sizeof(int)));
    std::vector<int> info(2 * batch);
    // Two attempts at most: when the persistent tridiagonalisation gave up on a hand-off (its workgroups were not all resident in
    // time: the GPU is shared with a long-running kernel), the matrices are solved again with one launch per Householder step,
    // which needs no co-residency -- a busy GPU is not a numerical failure.
    for (int attempt = 0; attempt < 2; ++attempt) {
        H2D(ctx, dA.p, A, nn);
        MOPT_HIP_TRY(ctx, hipMemsetAsync(ddefl.p, 0, (size_t)batch * sizeof(int), ctx->stream));
        if (mopt_eigh_algo(n) == 1) {
            DBuf dsc, dsi;
            const size_t nvi = (size_t)batch * n;
            TRY(dsc.alloc(ctx, 5 * nvi * 8)); TRY(dsi.alloc(ctx, (4 * nvi + 3 * batch) * 4));
            double *sc = dsc.as<double>();
            int *si = dsi.as<int>();
            const bool saved = ctx->tri_no_persist;
            if (attempt == 1) ctx->tri_no_persist = true;
            const int rc = mopt_launch_eigh_tridiag(ctx, n, batch, dA.as<double>(), dVt.as<double>(), sc, sc + nvi, sc + 2 * nvi, sc + 3 * nvi, sc + 4 * nvi,
                                                    si, si + nvi, si + 2 * nvi, si + 4 * nvi, si + 4 * nvi + 2 * batch, dW.as<double>(), dperm.as<int>(),
                                                    dinfo.as<int>(), dG.as<double>(), nullptr, nullptr, ddefl.as<int>(), -1);
            ctx->tri_no_persist = saved;
            ctx->last_defl.assign(batch, 0);
            if (rc == MOPT_OK) D2H(ctx, ctx->last_defl.data(), ddefl.p, (size_t)batch * sizeof(int));
            ctx->early_clean = nullptr;                      // (the scratch of this call is freed below: k_tri_early's sentinel bookkeeping is per address)
            if (rc != MOPT_OK) return rc;
            MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        } else
        TRY(mopt_launch_eigh(ctx, n, batch, dA.as<double>(), dsig.as<double>(), dG.as<double>(), dVt.as<double>(), drot.as<int>(),
                             dW.as<double>(), dperm.as<int>(), dinfo.as<int>(), MOPT_DEFAULT_SWEEPS, false));
        D2H(ctx, info.data(), dinfo.p, (size_t)batch * 2 * sizeof(int));
        MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        bool timed_out = false;
        for (int b = 0; b < batch; ++b) timed_out = timed_out || (info[2 * b + 1] & 8);
        if (!timed_out) break;
    }
    TRY(mopt_launch_eig_gather(ctx, n, batch, dVt.as<double>(), dperm.as<int>(), dV.as<double>()));
    D2H(ctx, W, dW.p, nv); D2H(ctx, V, dV.p, nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (sweeps_out) for (int b = 0; b < batch; ++b) sweeps_out[b] = info[2 * b];
    for (int b = 0; b < batch; ++b)
        if (info[2 * b + 1] & 4) {
            ctx->err = (info[2 * b + 1] & 8) ? "mopt_eigh: the workgroup hand-off of the tridiagonalisation timed out (GPU shared with a long-running kernel)"
                                             : "mopt_eigh: the reduction-route eigen-solver failed (inverse iteration / secular equation did not converge)";
            return MOPT_E_NUMERIC;
        }
    return MOPT_OK;
}

// test aid: where the tridiagonalisation of each matrix of the last mopt_eigh call on this context stopped early (row index; 0 = it ran
// to the end or took a route without the test).  Returns the number of entries written, or a negative status.
extern "C" int mopt_debug_last_deflation(mopt_ctx *ctx, int *out, int cap) {
    if (!ctx || !out || cap < 0) return MOPT_E_ARG;
    const int m = (int)ctx->last_defl.size() < cap ? (int)ctx->last_defl.size() : cap;
    for (int i = 0; i < m; ++i) out[i] = ctx->last_defl[i];
    return m;
}

extern "C" int mopt_dgemm(mopt_ctx *ctx, int n, int batch, const double *A, const double *B, double *C) {
    if (!ctx) return MOPT_E_ARG;
    if (n < 1 || batch < 1 || !A || !B || !C) { ctx->err = "mopt_dgemm: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)batch * n * n * sizeof(double);
    DBuf dA, dB, dC;
    TRY(dA.alloc(ctx, nn)); TRY(dB.alloc(ctx, nn)); TRY(dC.alloc(ctx, nn));
    H2D(ctx, dA.p, A, nn); H2D(ctx, dB.p, B, nn);
    TRY(mopt_launch_dgemm(ctx, n, batch, dA.as<double>(), dB.as<double>(), nullptr, dC.as<double>()));
    D2H(ctx, C, dC.p, nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_bneb_force(mopt_ctx *ctx, int I, int N, const double *X, const double *E, const double *G, int optimize_num,
                               int apply_ci_neb, double *F, double *tau) {
    if (!ctx) return MOPT_E_ARG;
    if (I < 2 || N < 1 || !X || !E || !G || !F || !tau) { ctx->err = "mopt_bneb_force: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nv = (size_t)I * N * 3 * sizeof(double);
    DBuf dX, dE, dG, dF, dT;
    TRY(dX.alloc(ctx, nv)); TRY(dE.alloc(ctx, I * sizeof(double))); TRY(dG.alloc(ctx, nv)); TRY(dF.alloc(ctx, nv)); TRY(dT.alloc(ctx, nv));
    H2D(ctx, dX.p, X, nv); H2D(ctx, dE.p, E, I * sizeof(double)); H2D(ctx, dG.p, G, nv);
    TRY(mopt_launch_bneb_force(ctx, I, N, 0, I, dX.as<double>(), dE.as<double>(), dG.as<double>(), optimize_num > apply_ci_neb ? 1 : 0,
                               dF.as<double>(), dT.as<double>()));
    D2H(ctx, F, dF.p, nv); D2H(ctx, tau, dT.p, nv);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_neb_force(mopt_ctx *ctx, int variant, int I, int N, const double *X, const double *E, const double *G,
                              int optimize_num, int apply_ci_neb, double *F, double *tau) {
    if (!ctx) return MOPT_E_ARG;
    if (variant == MOPT_NEB_BNEB) {
        if (!tau) { ctx->err = "mopt_neb_force: BNEB needs a tau buffer"; return MOPT_E_ARG; }
        return mopt_bneb_force(ctx, I, N, X, E, G, optimize_num, apply_ci_neb, F, tau);
    }
    if (variant < 0 || variant > MOPT_NEB_DNEB || I < 2 || N < 1 || !X || !E || !G || !F) { ctx->err = "mopt_neb_force: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nv = (size_t)I * N * 3 * sizeof(double);
    DBuf dX, dE, dG, dF;
    TRY(dX.alloc(ctx, nv)); TRY(dE.alloc(ctx, I * sizeof(double))); TRY(dG.alloc(ctx, nv)); TRY(dF.alloc(ctx, nv));
    H2D(ctx, dX.p, X, nv); H2D(ctx, dE.p, E, I * sizeof(double)); H2D(ctx, dG.p, G, nv);
    // CI switch: BNEB-type classes test `optimize_num > APPLY_CI_NEB`, DNEB `APPLY_CI_NEB < optimize_num` (the same)
    TRY(mopt_launch_neb_force_var(ctx, variant, I, N, 0, I, dX.as<double>(), dE.as<double>(), dG.as<double>(),
                                  optimize_num > apply_ci_neb ? 1 : 0, dF.as<double>()));
    D2H(ctx, F, dF.p, nv);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

// ------------------------------------------------------------------------------------------
// RSIRFO.run on host buffers
// ------------------------------------------------------------------------------------------
// Workspaces of the single-geometry entry points are kept per context and shape (a hipMalloc + hipFree pair per call -- the free
// synchronises the device -- was a quarter of a 0.4 ms call at n = 33); at most four shapes, least recently used goes first.  The
// debug mode (mopt_debug_poison) takes a fresh, poisoned and guarded workspace for every call instead.
namespace {
struct WsEntry { int n = 0, batch = 0, flags = 0; RsirfoWork *wk = nullptr; void *io = nullptr; size_t io_bytes = 0; unsigned long long stamp = 0; };
std::map<mopt_ctx *, std::vector<WsEntry>> g_ws;
std::mutex g_ws_mu;
unsigned long long g_ws_clock = 0;
void ws_drop(WsEntry &e) { delete e.wk; if (e.io) hipFree(e.io); e = WsEntry(); }
}  // namespace
void mopt_ws_release(mopt_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws.find(ctx);
    if (it == g_ws.end()) return;
    for (auto &e : it->second) ws_drop(e);
    g_ws.erase(it);
}
// the workspace of shape (n, batch, own H / bias / in-place copy) and an I/O span of io_bytes; both stay owned by the cache
static int ws_get(mopt_ctx *ctx, int n, int batch, bool own_h, bool bias, bool inplace, size_t io_bytes, RsirfoWork **wk_out, void **io_out) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto &v = g_ws[ctx];
    const int flags = (own_h ? 1 : 0) | (bias ? 2 : 0) | (inplace ? 4 : 0);
    WsEntry *hit = nullptr;
    for (auto &e : v) if (e.wk && e.n == n && e.batch == batch && e.flags == flags) hit = &e;
    if (!hit) {
        if (v.size() >= 4) {
            size_t old = 0;
            for (size_t i = 1; i < v.size(); ++i) if (v[i].stamp < v[old].stamp) old = i;
            hipStreamSynchronize(ctx->stream);
            ws_drop(v[old]);
            v.erase(v.begin() + (long)old);
        }
        WsEntry e;
        e.n = n; e.batch = batch; e.flags = flags;
        e.wk = new (std::nothrow) RsirfoWork();
        if (!e.wk) return MOPT_E_NOMEM;
        const int rc = e.wk->alloc(ctx, n, batch, own_h, bias, inplace);
        if (rc != MOPT_OK) { delete e.wk; return rc; }
        v.push_back(e);
        hit = &v.back();
    }
    if (hit->io_bytes < io_bytes) {
        if (hit->io) { hipStreamSynchronize(ctx->stream); hipFree(hit->io); }
        hit->io = nullptr; hit->io_bytes = 0;
        if (hipMalloc(&hit->io, io_bytes) != hipSuccess) { ctx->err = "hipMalloc failed"; return MOPT_E_NOMEM; }
        hit->io_bytes = io_bytes;
    }
    hit->stamp = ++g_ws_clock;
    *wk_out = hit->wk; *io_out = hit->io;
    return MOPT_OK;
}


// ---- MF_RSIRFO in two halves (include/mopt_hip.h) ---------------------------------------------------------------------------------
namespace {
// The workspace between the two halves belongs to the pending record, not to the shape cache above: a single-geometry call on the same
// context between begin and finish (same shape: it would overwrite eigw / Vt / perm; another shape: it may evict the entry) cannot
// touch it.  It is kept from call to call while the shape stays the same.
struct MfPending { RsirfoWork *own = nullptr; int n = 0, flags = 0; bool any_upd = false; bool valid = false; };
std::map<mopt_ctx *, MfPending> g_mf;
std::mutex g_mf_mu;
void mf_drop(MfPending &p) { if (p.own) delete p.own; p = MfPending(); }
}  // namespace
void mopt_mf_release(mopt_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_mf_mu);
    auto it = g_mf.find(ctx);
    if (it != g_mf.end()) { mf_drop(it->second); g_mf.erase(it); }
}

extern "C" int mopt_mf_rsirfo_begin(mopt_ctx *ctx, int n, const mopt_rsirfo_params *params, const mopt_rsirfo_state *state, const double *hessian,
                                    const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom,
                                    const double *pre_g, double B_e, double *W_out, double *V_out, double *H_inplace, double *H_sym, int *hess_updated) {
    if (bad_dims(ctx, n, 1)) return MOPT_E_ARG;
    if (!params || !state || !geom || !B_g || !W_out || !V_out) { ctx->err = "mopt_mf_rsirfo_begin: null argument"; return MOPT_E_ARG; }
    if (!hessian) { ctx->err = "Hessian matrix must be set"; return MOPT_E_STATE; }
    if (params->rule < 0 || params->rule >= MOPT_RULE_COUNT) { ctx->err = "unknown update rule"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    std::lock_guard<std::mutex> lk(g_mf_mu);                // held for the whole call: the record is not shared between threads half-way
    MfPending *pend = &g_mf[ctx];
    pend->valid = false;
    const int wflags = (bias_hessian ? 2 : 0) | (H_inplace ? 4 : 0);
    if (pend->own && (pend->n != n || pend->flags != wflags || mopt_debug_poison_on())) {
        hipStreamSynchronize(ctx->stream);
        mf_drop(*pend);
    }
    if (!pend->own) {
        pend->own = new (std::nothrow) RsirfoWork();
        if (!pend->own) return MOPT_E_NOMEM;
        const int rc = pend->own->alloc(ctx, n, 1, true, bias_hessian != nullptr, H_inplace != nullptr);
        if (rc != MOPT_OK) { mf_drop(*pend); return rc; }
        pend->n = n; pend->flags = wflags;
    }
    RsirfoWork &wk = *pend->own;
    const size_t nn = (size_t)n * n * sizeof(double), nv = (size_t)n * sizeof(double);
    const int upd = (pre_geom && pre_g && state->has_prev) ? 1 : 0, rule = params->rule;
    H2D(ctx, wk.x, geom, nv); H2D(ctx, wk.bg, B_g, nv); H2D(ctx, wk.g, g ? g : B_g, nv);
    if (upd) { H2D(ctx, wk.px, pre_geom, nv); H2D(ctx, wk.pg, pre_g, nv); }
    H2D(ctx, wk.be, &B_e, sizeof(double));
    H2D(ctx, wk.params, params, sizeof(mopt_rsirfo_params));
    H2D(ctx, wk.states, state, sizeof(mopt_rsirfo_state));
    H2D(ctx, wk.rules, &rule, sizeof(int)); H2D(ctx, wk.do_update, &upd, sizeof(int));
    H2D(ctx, wk.H, hessian, nn);
    if (bias_hessian) H2D(ctx, wk.Bias, bias_hessian, nn);
    if (upd) TRY(mopt_launch_hess_update(ctx, n, 1, wk.rules, wk.do_update, rule == MOPT_RULE_FLOWCHART, wk.x, wk.px, wk.g, wk.pg, wk.s, wk.y, wk.u, wk.Hy, wk.yd,
                                         wk.Av, wk.coef, wk.info, wk.H, wk.Hinpl, wk.H, false));
    std::vector<double> vt((size_t)n * n), w(n);
    std::vector<int> perm(n);
    int einfo[2] = {0, 0};
    mopt::UpdateCoef hc;
    hc.applied = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        TRY(mopt_launch_project_hess(ctx, n, 1, wk.H, wk.Bias, wk.x, wk.Q, wk.W6, wk.A));
        TRY(mopt_eigh_auto(ctx, wk, n, 1, false, /*preserve_A=*/false, nullptr));
        D2H(ctx, einfo, wk.einfo, 2 * sizeof(int));
        MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (!(einfo[1] & 8) || attempt == 1) break;
        ctx->tri_no_persist = true;                          // hand-off time-out (shared GPU): once more with one launch per Householder step
        ctx->n_retries += 1;
    }
    ctx->tri_no_persist = false;
    if (einfo[1] & 4) { ctx->err = "mopt_mf_rsirfo_begin: the eigen-solver failed"; return MOPT_E_NUMERIC; }
    D2H(ctx, w.data(), wk.eigw, nv); D2H(ctx, perm.data(), wk.perm, n * sizeof(int)); D2H(ctx, vt.data(), wk.Vt, nn);
    if (upd) D2H(ctx, &hc, wk.coef, sizeof(hc));
    if (H_sym) D2H(ctx, H_sym, wk.H, nn);
    if (H_inplace) D2H(ctx, H_inplace, (upd ? wk.Hinpl : wk.H), nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const bool ident = (einfo[1] & 1) != 0;                  // non-finite spectrum: the consumers continue with (w = 1, V = I), rsirfo.py:365-372
    for (int k = 0; k < n; ++k) {
        W_out[k] = ident ? 1.0 : w[k];
        const int r = ident ? k : perm[k];
        for (int i = 0; i < n; ++i) V_out[(size_t)k * n + i] = ident ? (i == k ? 1.0 : 0.0) : vt[(size_t)r * n + i];
    }
    if (hess_updated) *hess_updated = upd ? hc.applied : 0;
    pend->any_upd = upd != 0; pend->valid = true;
    return MOPT_OK;
}

extern "C" int mopt_mf_rsirfo_finish(mopt_ctx *ctx, int n, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const int *targets, int ntargets,
                                     double *step) {
    if (bad_dims(ctx, n, 1)) return MOPT_E_ARG;
    if (!params || !state || !step || ntargets < 0 || ntargets > 8 || (ntargets > 0 && !targets)) { ctx->err = "mopt_mf_rsirfo_finish: bad argument"; return MOPT_E_ARG; }
    std::lock_guard<std::mutex> lk(g_mf_mu);
    MfPending *pend = &g_mf[ctx];
    if (!pend->valid || pend->n != n || !pend->own) { ctx->err = "mopt_mf_rsirfo_finish without a matching mopt_mf_rsirfo_begin"; return MOPT_E_STATE; }
    hipSetDevice(ctx->device);
    RsirfoWork &wk = *pend->own;
    int tg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < ntargets; ++t) tg[t] = targets[t];
    H2D(ctx, wk.params, params, sizeof(mopt_rsirfo_params));
    H2D(ctx, wk.states, state, sizeof(mopt_rsirfo_state));
    H2D(ctx, wk.rotcount, tg, sizeof(tg));                  // (the Jacobi route's rotation counters: free behind the solve)
    TRY(mopt_launch_rfo(ctx, n, 1, wk.params, wk.states, wk.eigw, wk.perm, wk.einfo, wk.Vt, /*gradient: unprojected*/ wk.bg, wk.bg, wk.be,
                        pend->any_upd ? wk.coef : nullptr, wk.gt_all, wk.coefout, wk.step, wk.rotcount, ntargets));
    mopt_rsirfo_state st;
    std::vector<double> sv(n);
    D2H(ctx, &st, wk.states, sizeof(st)); D2H(ctx, sv.data(), wk.step, (size_t)n * sizeof(double));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    pend->valid = false;
    int rc = MOPT_OK;
    if (mopt_debug_poison_on()) { rc = wk.check_canaries(); mf_drop(*pend); }
    if (rc != MOPT_OK) return rc;
    if (st.jacobi_sweeps < 0) { ctx->err = "mopt_mf_rsirfo_finish: the eigen-solver failed"; return MOPT_E_NUMERIC; }
    *state = st;
    std::memcpy(step, sv.data(), (size_t)n * sizeof(double));
    return MOPT_OK;
}

// (for k_prfo.hip: the cached workspace of a single small geometry)
int mopt_ws_get(mopt_ctx *ctx, int n, bool bias, size_t io_bytes, RsirfoWork **wk_out, void **io_out) {
    return ws_get(ctx, n, 1, true, bias, false, io_bytes, wk_out, io_out);
}
int mopt_ws_get_small(mopt_ctx *ctx, int n, RsirfoWork **wk_out) {
    void *io = nullptr;
    return ws_get(ctx, n, 1, true, false, false, 256, wk_out, &io);
}

// Small geometries (k_small.hip): the kernel reads its inputs from, and writes its results to, one pinned host span it can address, and
// the host waits for the kernel's completion counter in that span instead of for the stream -- no copy command and no stream wake-up
// on the path (five copies and the synchronisation were 40 us of a 195 us call at n = 21).  Span: [inputs | states | step | H | bias |
// Hsym | Hinpl | counter].
static int rsirfo_run_small(mopt_ctx *ctx, int n, int batch, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const double *hessian,
                            const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom,
                            const double *pre_g, const double *B_e, double *step, double *H_inplace, double *H_sym, const std::vector<int> &rules,
                            const std::vector<int> &upd, bool any_upd, bool any_flow) {
    const size_t nn = (size_t)batch * n * n * sizeof(double), nv = (size_t)batch * n * sizeof(double);
    auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    RsirfoWork *wkp = nullptr;
    void *io_p = nullptr;
    TRY(ws_get(ctx, n, batch, true, false, false, 256, &wkp, &io_p));
    RsirfoWork &wk = *wkp;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_x = take(nv), o_bg = take(nv), o_g = take(nv), o_px = take(nv), o_pg = take(nv), o_be = take(batch * sizeof(double)),
                 o_par = take(batch * sizeof(mopt_rsirfo_params)), o_rul = take(batch * sizeof(int)), o_upd = take(batch * sizeof(int)),
                 o_st = take(batch * sizeof(mopt_rsirfo_state)), o_step = take(nv), o_H = take(nn), o_B = bias_hessian ? take(nn) : 0,
                 o_Hs = take(nn), o_Hp = H_inplace ? take(nn) : 0, o_done = take(64);
    char *pin = (char *)pin_get(ctx, off);
    if (!pin) { ctx->err = "hipHostMalloc failed (staging buffer)"; return MOPT_E_NOMEM; }
    char *dpin = nullptr;
    MOPT_HIP_TRY(ctx, hipHostGetDevicePointer((void **)&dpin, pin, 0));
    const bool have_prev = pre_geom && pre_g;
    std::memcpy(pin + o_x, geom, nv); std::memcpy(pin + o_bg, B_g, nv); std::memcpy(pin + o_g, g ? g : B_g, nv);
    if (have_prev) { std::memcpy(pin + o_px, pre_geom, nv); std::memcpy(pin + o_pg, pre_g, nv); }
    std::memcpy(pin + o_be, B_e, batch * sizeof(double));
    std::memcpy(pin + o_par, params, batch * sizeof(mopt_rsirfo_params));
    std::memcpy(pin + o_rul, rules.data(), batch * sizeof(int));
    std::memcpy(pin + o_upd, upd.data(), batch * sizeof(int));
    std::memcpy(pin + o_st, state, batch * sizeof(mopt_rsirfo_state));
    std::memcpy(pin + o_H, hessian, nn);
    if (bias_hessian) std::memcpy(pin + o_B, bias_hessian, nn);
    volatile unsigned long long *done = (volatile unsigned long long *)(pin + o_done);
    *done = 0ull;
    SmallStepIO io;
    io.params = (const mopt_rsirfo_params *)(dpin + o_par); io.states = (mopt_rsirfo_state *)(dpin + o_st);
    io.x = (const double *)(dpin + o_x); io.bg = (const double *)(dpin + o_bg); io.g = (const double *)(dpin + o_g);
    io.px = (const double *)(dpin + o_px); io.pg = (const double *)(dpin + o_pg); io.be = (const double *)(dpin + o_be);
    io.rules = (const int *)(dpin + o_rul); io.do_update = (const int *)(dpin + o_upd);
    io.Hin = (const double *)(dpin + o_H); io.Bias = bias_hessian ? (const double *)(dpin + o_B) : nullptr;
    io.Hsym = (double *)(dpin + o_Hs); io.Hinpl = H_inplace ? (double *)(dpin + o_Hp) : nullptr;
    io.step = (double *)(dpin + o_step); io.done = (unsigned long long *)(dpin + o_done);
    TRY(mopt_launch_small_rsirfo_io(ctx, wk, n, batch, any_upd, any_flow, io));
    {
        // wait for the counter (bounded: a kernel that never finishes would otherwise hang the caller without a message)
        static const bool no_poll = getenv("MOPT_SMALL_NOPOLL") != nullptr;
        bool seen = false;
        if (!no_poll) {
            const auto t0 = std::chrono::steady_clock::now();
            for (long it = 0;; ++it) {
                if (__atomic_load_n((const unsigned long long *)done, __ATOMIC_ACQUIRE) >= (unsigned long long)batch) { seen = true; break; }
                if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
        }
        if (!seen) MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    // a failed image is looked for BEFORE anything is copied out: on MOPT_E_NUMERIC the caller's state, step and Hessians stay as they were
    {
        const mopt_rsirfo_state *ps = (const mopt_rsirfo_state *)(pin + o_st);
        for (int b = 0; b < batch; ++b)
            if (ps[b].jacobi_sweeps < 0) { ctx->err = "mopt_rsirfo_run: the eigen-solver failed"; return MOPT_E_NUMERIC; }
    }
    std::memcpy(state, pin + o_st, batch * sizeof(mopt_rsirfo_state));
    std::memcpy(step, pin + o_step, nv);
    if (H_sym) std::memcpy(H_sym, pin + o_Hs, nn);
    if (H_inplace) std::memcpy(H_inplace, pin + o_Hp, nn);
    return MOPT_OK;
}

extern "C" int mopt_rsirfo_run(mopt_ctx *ctx, int n, int batch, const mopt_rsirfo_params *params, mopt_rsirfo_state *state,
                               const double *hessian, const double *bias_hessian, const double *geom, const double *B_g,
                               const double *g, const double *pre_geom, const double *pre_g, const double *B_e, double *step,
                               double *H_inplace, double *H_sym) {
    if (bad_dims(ctx, n, batch)) return MOPT_E_ARG;
    if (!params || !state || !geom || !B_g || !B_e || !step) { ctx->err = "mopt_rsirfo_run: null argument"; return MOPT_E_ARG; }
    if (!hessian) { ctx->err = "Hessian matrix must be set before running optimization"; return MOPT_E_STATE; }
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)batch * n * n * sizeof(double), nv = (size_t)batch * n * sizeof(double);
    auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    const bool have_prev = pre_geom && pre_g;
    std::vector<int> rules(batch), upd(batch);
    bool any_flow = false, any_upd = false;
    for (int b = 0; b < batch; ++b) {
        rules[b] = params[b].rule;
        if (rules[b] < 0 || rules[b] >= MOPT_RULE_COUNT) { ctx->err = "unknown update rule"; return MOPT_E_ARG; }
        upd[b] = (have_prev && state[b].has_prev) ? 1 : 0;       // rsirfo.py:308
        any_flow |= (rules[b] == MOPT_RULE_FLOWCHART);
        any_upd |= upd[b] != 0;
    }
    if (mopt_small_eligible(n) && !mopt_debug_poison_on() && (size_t)batch * n * n * sizeof(double) <= ((size_t)4 << 20))
        return rsirfo_run_small(ctx, n, batch, params, state, hessian, bias_hessian, geom, B_g, g, pre_geom, pre_g, B_e, step, H_inplace, H_sym, rules, upd,
                                any_upd, any_flow);
    const size_t io_tot = 6 * al(nv) + al(batch * sizeof(double)) + al(batch * sizeof(mopt_rsirfo_params)) + 2 * al(batch * sizeof(int)) +
                          al(batch * sizeof(mopt_rsirfo_state));
    // workspace + I/O span: cached per shape, or fresh (poisoned, guarded) in the debug mode
    RsirfoWork fresh;
    DBuf fresh_io;
    RsirfoWork *wkp = &fresh;
    void *io_p = nullptr;
    if (mopt_debug_poison_on()) {
        TRY(fresh.alloc(ctx, n, batch, /*own_hessian=*/true, bias_hessian != nullptr, H_inplace != nullptr));
        TRY(fresh_io.alloc(ctx, io_tot));
        io_p = fresh_io.p;
    } else {
        TRY(ws_get(ctx, n, batch, true, bias_hessian != nullptr, H_inplace != nullptr, io_tot, &wkp, &io_p));
    }
    RsirfoWork &wk = *wkp;
    // (the cached object keeps its own array pointers: the fields redirected into the I/O span below are restored on every exit)
    struct Restore { RsirfoWork &w; double *x, *bg, *g, *px, *pg, *be, *step; mopt_rsirfo_params *params; int *rules, *do_update; mopt_rsirfo_state *states;
                     ~Restore() { w.x = x; w.bg = bg; w.g = g; w.px = px; w.pg = pg; w.be = be; w.step = step; w.params = params; w.rules = rules; w.do_update = do_update; w.states = states; } }
        restore{wk, wk.x, wk.bg, wk.g, wk.px, wk.pg, wk.be, wk.step, wk.params, wk.rules, wk.do_update, wk.states};
    // The small inputs travel as ONE copy from a pinned staging buffer into one device span, the outputs [states | step] and -- up to
    // 1 MB each -- the Hessians come back through it as well, so no copy of the call blocks on pageable memory before the final
    // synchronisation (twelve small pageable copies in and four out were 130 us of a 0.5 ms call).  The span replaces the workspace's
    // own arrays of the same names for this call (the workspace layout itself is shared with the band and stays as it is):
    //   in:  [x | bg | g | px | pg | be | params | rules | do_update | states]      out: [states | step]
    {
        char *q = (char *)io_p;
        auto take = [&](size_t bytes) { void *r = q; q += al(bytes); return r; };
        wk.x = (double *)take(nv); wk.bg = (double *)take(nv); wk.g = (double *)take(nv); wk.px = (double *)take(nv); wk.pg = (double *)take(nv);
        wk.be = (double *)take(batch * sizeof(double));
        wk.params = (mopt_rsirfo_params *)take(batch * sizeof(mopt_rsirfo_params));
        wk.rules = (int *)take(batch * sizeof(int)); wk.do_update = (int *)take(batch * sizeof(int));
        wk.states = (mopt_rsirfo_state *)take(batch * sizeof(mopt_rsirfo_state));
        wk.step = (double *)take(nv);
    }
    const size_t in_bytes = (size_t)((char *)wk.step - (char *)wk.x), out_bytes = (size_t)((char *)wk.step - (char *)wk.states) + nv;
    const bool stage_H = nn <= ((size_t)1 << 20);
    const size_t hs_off = (in_bytes + out_bytes + 255) & ~(size_t)255, hi_off = hs_off + (stage_H ? ((nn + 255) & ~(size_t)255) : 0);
    char *pin = (char *)pin_get(ctx, hi_off + (stage_H ? nn : 0));
    if (!pin) { ctx->err = "hipHostMalloc failed (staging buffer)"; return MOPT_E_NOMEM; }
    auto put = [&](const void *dev_field, const void *src, size_t bytes) { std::memcpy(pin + ((const char *)dev_field - (const char *)wk.x), src, bytes); };
    put(wk.x, geom, nv); put(wk.bg, B_g, nv); put(wk.g, g ? g : B_g, nv);
    if (have_prev) { put(wk.px, pre_geom, nv); put(wk.pg, pre_g, nv); }
    put(wk.be, B_e, batch * sizeof(double));
    put(wk.params, params, batch * sizeof(mopt_rsirfo_params));
    put(wk.states, state, batch * sizeof(mopt_rsirfo_state));
    put(wk.rules, rules.data(), batch * sizeof(int));
    put(wk.do_update, upd.data(), batch * sizeof(int));
    H2D(ctx, wk.x, pin, in_bytes);
    H2D(ctx, wk.H, hessian, nn);
    if (bias_hessian) H2D(ctx, wk.Bias, bias_hessian, nn);
    TRY(mopt_rsirfo_pipeline(ctx, wk, n, batch, any_upd, any_flow, wk.Hinpl, wk.H /* symmetrised in place */));
    char *pout = pin + in_bytes;
    D2H(ctx, pout, wk.states, out_bytes);
    if (H_sym) D2H(ctx, stage_H ? (void *)(pin + hs_off) : (void *)H_sym, wk.H, nn);
    if (H_inplace) D2H(ctx, stage_H ? (void *)(pin + hi_off) : (void *)H_inplace, any_upd ? wk.Hinpl : wk.H, nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    {
        // a hand-off time-out of the persistent tridiagonalisation (jacobi_sweeps = -2: the GPU is shared with a long-running kernel) is
        // not a numerical failure: those geometries are solved again with one launch per Householder step
        bool timed_out = false;
        for (int b = 0; b < batch; ++b) timed_out = timed_out || ((const mopt_rsirfo_state *)pout)[b].jacobi_sweeps == -2;
        if (timed_out) {
            int nre = 0;
            TRY(mopt_rsirfo_retry_timeouts(ctx, wk, n, batch, wk.H, any_upd, &nre));
            D2H(ctx, pout, wk.states, out_bytes);
            MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    if (mopt_debug_poison_on()) TRY(wk.check_canaries());
    std::memcpy(state, pout, batch * sizeof(mopt_rsirfo_state));
    std::memcpy(step, pout + ((char *)wk.step - (char *)wk.states), nv);
    if (stage_H && H_sym) std::memcpy(H_sym, pin + hs_off, nn);
    if (stage_H && H_inplace) std::memcpy(H_inplace, pin + hi_off, nn);
    for (int b = 0; b < batch; ++b)
        if (state[b].jacobi_sweeps < 0) { ctx->err = "mopt_rsirfo_run: the eigen-solver failed"; return MOPT_E_NUMERIC; }
    return MOPT_OK;
}
