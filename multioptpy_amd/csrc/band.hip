// Device-resident NEB band: RFOOptimizer.optimize (multioptpy/Optimizer/rfo_neb.py:104-208) with the per-image
// Hessians and RSIRFO state kept in HBM between steps (the reference round-trips tmp_hessian_{i}.npy, :18-25,175).
//   k_gamma     calculate_gamma (pathopt_bneb_force.py:161-222): quintic fit through (E, g.t) of the triplet
//   k_rank1     H += gamma t t^T (rfo_neb.py:72)
//   k_neb_post  step limit (rfo_neb.py:76-83) -> TR_calc (trust_radius_neb.py:17-98) -> FIRE (fire_neb.py:38-92)
//               -> blend and Bohr->Angstrom (rfo_neb.py:193-206)
#include <cstdlib>
#include <cstring>
#include <new>

#include "pipeline.h"
#include "neb_tr.hpp"
#include "neb_gamma.hpp"

namespace mopt {

constexpr double BOHR2ANG = 0.52917721067;     // neb.py:201

// band inputs of a step handed over in device memory: X, G (I n doubles each) and E (I doubles)
__global__ void __launch_bounds__(256) k_band_take(int nd, int I, const double *__restrict__ sX, const double *__restrict__ sG,
                                                   const double *__restrict__ sE, double *__restrict__ X, double *__restrict__ G,
                                                   double *__restrict__ E) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nd) { X[i] = sX[i]; G[i] = sG[i]; }
    if (i < I) E[i] = sE[i];
}

// the same for an image-sharded band: only the owned images arrive (count * n doubles, count energies), stored at their
// global position
__global__ void __launch_bounds__(256) k_band_take_shard(int nd, int count, int first, int n, const double *__restrict__ sX,
                                                         const double *__restrict__ sG, const double *__restrict__ sE,
                                                         double *__restrict__ X, double *__restrict__ G, double *__restrict__ E) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nd) { X[(size_t)first * n + i] = sX[i]; G[(size_t)first * n + i] = sG[i]; }
    if (i < count) E[first + i] = sE[i];
}

// partial sum of FIRE's band-wide switch over the owned images: out[0] = sum_{own} pF . F   (one workgroup: fixed order)
__global__ void __launch_bounds__(1024) k_band_dot(int nd, size_t off, const double *__restrict__ pF, const double *__restrict__ F, double *__restrict__ out) {
    __shared__ double red[16];
    double a = 0.0;
    for (int k = threadIdx.x; k < nd; k += blockDim.x) a += pF[off + k] * F[off + k];
    a = block_sum1(a, red);
    if (threadIdx.x == 0) out[0] = a;
}

// grid = owned images; gamma[local] (0 for end images)
__global__ void __launch_bounds__(256) k_gamma(int I, int n, int first, const double *__restrict__ X, const double *__restrict__ E,
                                               const double *__restrict__ G, const double *__restrict__ tau,
                                               double *__restrict__ gamma) {
    __shared__ double red[5 * 16];
    const int loc = blockIdx.x, i = first + loc;
    if (i == 0 || i == I - 1) { if (threadIdx.x == 0) gamma[loc] = 0.0; return; }
    const double *xm = X + (size_t)(i - 1) * n, *x0 = X + (size_t)i * n, *xp = X + (size_t)(i + 1) * n;
    const double *gm = G + (size_t)(i - 1) * n, *g0 = G + (size_t)i * n, *gp = G + (size_t)(i + 1) * n;
    const double *t = tau + (size_t)i * n;
    double v[5];
    ayala_partial_sums(n, xm, x0, xp, gm, g0, gp, t, v);
    block_sum<5>(v, red);
    if (threadIdx.x == 0) gamma[loc] = ayala_gamma(v[0], v[1], v[2], v[3], v[4], E[i - 1], E[i], E[i + 1]);
}

// H[loc] += gamma[loc] * t t^T      grid (ceil(n/8), owned), block 256: 8 rows per block, 32 lanes per row
__global__ void __launch_bounds__(256) k_rank1(int n, int first, const double *__restrict__ gamma, const double *__restrict__ tau,
                                               double *__restrict__ H) {
    const int loc = blockIdx.y;
    const double gam = gamma[loc];
    if (gam == 0.0) return;
    const double *t = tau + (size_t)(first + loc) * n;
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (r >= n) return;
    const double tr = gam * t[r];
    double *row = H + ((size_t)loc * n + r) * n;
    for (int c = threadIdx.x & 31; c < n; c += 32) row[c] += tr * t[c];
}

// grid = owned images, block 128, dynamic LDS: 2 * n doubles
__global__ void __launch_bounds__(512) k_neb_post(int I, int N, int first, int optimize_num, int have_prev, double ratio, int fix_init,
                                                  int fix_end, const double *__restrict__ X, const double *__restrict__ G,
                                                  const double *__restrict__ F, const double *__restrict__ pF,
                                                  const double *__restrict__ rfo_step, double *__restrict__ rfo_delta_out,
                                                  double *__restrict__ newX, const double *__restrict__ dots, int ndots) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[6 * 16];
    const int n = 3 * N;
    double *dl = sm;            // RFO delta (after the step limit)
    double *fd = sm + n;        // FIRE total_delta
    const int loc = blockIdx.x, i = first + loc;
    const bool is_end = (i == 0 || i == I - 1);
    const double *x0 = X + (size_t)i * n;
    const double *g0 = G + (size_t)i * n;
    const double *f0 = F + (size_t)i * n;
    const double *rs = rfo_step + (size_t)loc * n;
    // ---- FIRE global switch: sum over ALL images of pre_velocity . force (fire_neb.py:60-65), and |rfo step|^2 for the step limit
    // (rfo_neb.py:76-83) -- independent sums, one reduction round (round 4)
    double dotp = 0.0;
    double v2[2] = {0.0, 0.0};
    const bool own_dot = (optimize_num != 0 && have_prev && !dots);
    if (own_dot) for (size_t k = threadIdx.x; k < (size_t)I * n; k += blockDim.x) v2[0] += pF[k] * F[k];
    for (int k = threadIdx.x; k < n; k += blockDim.x) v2[1] += rs[k] * rs[k];
    block_sum<2>(v2, red);
    if (optimize_num != 0 && have_prev) {
        if (dots) {
            // image-sharded band: every rank contributed the sum over its own images (k_band_dot), exchanged by comm.hip;
            // summed here in rank order on every rank -- the same number everywhere, whatever the transport
            for (int r = 0; r < ndots; ++r) dotp += dots[r];
        } else dotp = v2[0];
    }
    const bool keep = (optimize_num > 0 && dotp > 0.0 && have_prev);
    // a fresh FIREOptimizer every step (rfo_neb.py:187): dt = 0.5, a = 0.1, n_reset = 0 (never > N_accelerate)
    const double fa = 0.1;
    const double dt = keep ? 0.5 : 0.5 * 0.5;
    {
        const double nrm = sqrt(v2[1]);
        const double lim = is_end ? 0.2 : 0.1;
        for (int k = threadIdx.x; k < n; k += blockDim.x) dl[k] = (nrm > 1e-8) ? rs[k] / nrm * fmin(lim, nrm) : rs[k];
    }
    // ---- FIRE velocity mixing per atom and total_delta
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        const double fx = f0[3 * a], fy = f0[3 * a + 1], fz = f0[3 * a + 2];
        const double fn = sqrt(fx * fx + fy * fy + fz * fz);     // total_velocity == total_force here (neb.py:552,1159-1162)
        double vx = fx, vy = fy, vz = fz;
        if (fn > 1e-10) {
            const double vn = fn;
            vx = (1.0 - fa) * fx + fa * (vn / fn) * fx;
            vy = (1.0 - fa) * fy + fa * (vn / fn) * fy;
            vz = (1.0 - fa) * fz + fa * (vn / fn) * fz;
        }
        if (!keep) { vx *= 0.0; vy *= 0.0; vz *= 0.0; }
        double tv[3] = {vx + dt * fx, vy + dt * fy, vz + dt * fz};
        for (int c = 0; c < 3; ++c) {
            const int k = 3 * a + c;
            fd[k] = (optimize_num != 0 && have_prev) ? dt * (tv[c] + pF[(size_t)i * n + k]) : dt * tv[c];
        }
    }
    __syncthreads();
    double sc_rfo, sc_fire;
    if (is_end) {
        double w[6] = {0, 0, 0, 0, 0, 0};
        for (int k = threadIdx.x; k < n; k += blockDim.x) { w[0] += dl[k] * dl[k]; w[1] += fd[k] * fd[k]; }
        block_sum<6>(w, red);
        const bool fixed = (i == 0) ? (fix_init != 0) : (fix_end != 0);
        sc_rfo = tr_scale_edge(sqrt(w[0]), fixed);
        sc_fire = tr_scale_edge(sqrt(w[1]), fixed);
    } else {
        const double *xm = X + (size_t)(i - 1) * n, *xp = X + (size_t)(i + 1) * n;
        // distances and the dots needed by both TR_calc calls
        double w[6] = {0, 0, 0, 0, 0, 0};
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const double dm = xm[k] - x0[k], dp = xp[k] - x0[k];
            w[0] += dm * dm; w[1] += dp * dp;
            w[2] += dl[k] * dl[k]; w[3] += fd[k] * fd[k];
            w[4] += g0[k] * g0[k]; w[5] += f0[k] * f0[k];
        }
        block_sum<6>(w, red);
        const double d1 = sqrt(w[0]), d2 = sqrt(w[1]);
        const double dnr = sqrt(w[2]), dnf = sqrt(w[3]), gnr = sqrt(w[4]), fnr = sqrt(w[5]);
        double u[6] = {0, 0, 0, 0, 0, 0};
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const double v1 = (xm[k] - x0[k]) / (d1 + 1e-15), v2 = (xp[k] - x0[k]) / (d2 + 1e-15);
            const double ndr = dl[k] / dnr, ndf = fd[k] / dnf;
            u[0] += v1 * ndr; u[1] += v2 * ndr; u[2] += g0[k] * dl[k];
            u[3] += v1 * ndf; u[4] += v2 * ndf; u[5] += f0[k] * fd[k];
        }
        block_sum<6>(u, red);
        // RFO: TR_calc(geometry, total_force_list = the biased GRADIENTS, rfo_delta)   (rfo_neb.py:181-184)
        sc_rfo = tr_scale_interior(dnr, d1 / 2.0, d2 / 2.0, u[0], u[1], u[2] / (gnr * dnr));
        // FIRE: TR_calc(geometry, projected NEB force, total_delta)                     (fire_neb.py:84-86)
        sc_fire = tr_scale_interior(dnf, d1 / 2.0, d2 / 2.0, u[3], u[4], u[5] / (fnr * dnf));
    }
    double *out = newX + (size_t)loc * n;
    double *rd = rfo_delta_out + (size_t)loc * n;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double rmv = (sc_rfo == 1.0) ? dl[k] : dl[k] * sc_rfo;
        const double fmv = (sc_fire == 1.0) ? fd[k] : fd[k] * sc_fire;
        // FIRE returns Angstrom; the blend converts back: tmp/bohr2ang - geometry (rfo_neb.py:193-194)
        const double fire_move = ((x0[k] + fmv) * BOHR2ANG) / BOHR2ANG - x0[k];
        const double mv = is_end ? -1.0 * rmv : (1.0 - ratio) * fire_move - ratio * rmv;
        out[k] = (x0[k] + mv) * BOHR2ANG;
        rd[k] = rmv;
    }
}

}  // namespace mopt

#include "band.h"

using namespace mopt;

#define BTRY(x) do { int rc__ = (x); if (rc__ != MOPT_OK) return rc__; } while (0)

static int band_launch_post(mopt_band *b);
static int band_check_failures(mopt_band *b, int *bits_out);

int mopt_band_alloc_internal(mopt_band *b) {
    mopt_ctx *ctx = b->ctx;
    if (b->alloc_done) return MOPT_OK;
    const size_t nv = (size_t)b->I * b->n * 8, cv = (size_t)b->count * b->n * 8;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // (debug mode, pipeline.h: guard gaps between the arrays; the arrays themselves start from zero in either mode -- slots of images a
    // shard neither owns nor receives are never read, and the whole band reads previous-step arrays only behind have_prev)
    const bool poison = mopt_debug_poison_on();
    const size_t gap = poison ? MOPT_CANARY_GAP : 0;
    struct Item { double **p; size_t bytes; const char *name; size_t off; };
    double *spareE = nullptr;
    Item items[] = {{&b->X, nv, "X", 0}, {&b->G, nv, "G", 0}, {&b->pX, nv, "pX", 0}, {&b->pG, nv, "pG", 0}, {&b->F, nv, "F", 0}, {&b->pF, nv, "pF", 0},
                    {&b->tau, nv, "tau", 0}, {&b->E, (size_t)b->I * 8, "E", 0}, {&spareE, (size_t)b->I * 8, "spareE", 0},
                    {&b->rfo_delta, cv, "rfo_delta", 0}, {&b->newX, cv, "newX", 0}, {&b->gamma, (size_t)b->count * 8, "gamma", 0}};
    size_t tot = gap;
    for (auto &it : items) { it.off = tot; tot += al(it.bytes) + gap; }
    MOPT_HIP_TRY(ctx, hipMalloc(&b->block, tot));
    MOPT_HIP_TRY(ctx, hipMemsetAsync(b->block, poison ? MOPT_CANARY_BYTE : 0, tot, ctx->stream));
    b->gaps.clear(); b->gap_names.clear();
    if (poison) { b->gaps.emplace_back(0, gap); b->gap_names.push_back("(start of block)"); }
    for (auto &it : items) {
        *it.p = (double *)((char *)b->block + it.off);
        if (poison) {
            MOPT_HIP_TRY(ctx, hipMemsetAsync(*it.p, 0, it.bytes, ctx->stream));
            b->gaps.emplace_back(it.off + it.bytes, al(it.bytes) - it.bytes + gap); b->gap_names.push_back(it.name);
        }
    }
    BTRY(b->wk.alloc(ctx, b->n, b->count, true, false, false));
    MOPT_HIP_TRY(ctx, hipMalloc(&b->dots, MOPT_MAX_RANKS * sizeof(double)));
    MOPT_HIP_TRY(ctx, hipMemsetAsync(b->dots, 0, MOPT_MAX_RANKS * sizeof(double), ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamCreateWithFlags(&b->cst, hipStreamNonBlocking));
    MOPT_HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming));
    MOPT_HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_out, hipEventDisableTiming));
    MOPT_HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_dot, hipEventDisableTiming));
    MOPT_HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_dot_done, hipEventDisableTiming));
    // default Hessians: identity (rfo_neb.py:25)
    std::vector<double> eye((size_t)b->n * b->n, 0.0);
    for (int i = 0; i < b->n; ++i) eye[(size_t)i * b->n + i] = 1.0;
    for (int k = 0; k < b->count; ++k)
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->wk.H + (size_t)k * b->n * b->n, eye.data(), eye.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemsetAsync(b->wk.be, 0, (size_t)b->count * 8, ctx->stream));     // OPT.run(..., 0.0, 0.0, ...) (rfo_neb.py:166)
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    b->H_sym.assign((size_t)b->count, 1);
    b->alloc_done = true;
    return MOPT_OK;
}

extern "C" int mopt_band_create(mopt_ctx *ctx, int I, int N, mopt_band **out) {
    if (!ctx || !out) return MOPT_E_ARG;
    *out = nullptr;
    if (I < 3 || N < 1 || 3 * N > MOPT_MAX_N) { ctx->err = "mopt_band_create: need I >= 3 images and 1 <= N <= MOPT_MAX_N/3 atoms"; return MOPT_E_ARG; }
    mopt_band *b = new (std::nothrow) mopt_band();
    if (!b) return MOPT_E_NOMEM;
    b->ctx = ctx; b->I = I; b->N = N; b->n = 3 * N; b->first = 0; b->count = I;
    *out = b;
    return MOPT_OK;
}

extern "C" int mopt_band_destroy(mopt_band *b) {
    if (!b) return MOPT_E_ARG;
    // (a band that is destroyed before its communicator leaves the group first: mopt_comm_destroy must not find a dangling pointer)
    if (b->comm) { mopt_comm_detach_band(b->comm, b); b->comm = nullptr; }
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    if (b->cst) { hipStreamSynchronize(b->cst); hipStreamDestroy(b->cst); }
    if (b->ev_in) hipEventDestroy(b->ev_in);
    if (b->ev_out) hipEventDestroy(b->ev_out);
    if (b->ev_dot) hipEventDestroy(b->ev_dot);
    if (b->ev_dot_done) hipEventDestroy(b->ev_dot_done);
    if (b->dots) hipFree(b->dots);
    if (b->block) hipFree(b->block);
    delete b;
    return MOPT_OK;
}

extern "C" int mopt_band_set_shard(mopt_band *b, int first, int count) {
    if (!b) return MOPT_E_ARG;
    if (b->alloc_done) { b->ctx->err = "mopt_band_set_shard must precede the first Hessian/upload call"; return MOPT_E_STATE; }
    if (first < 0 || count < 1 || first + count > b->I) { b->ctx->err = "bad shard"; return MOPT_E_ARG; }
    b->first = first; b->count = count;
    return MOPT_OK;
}

extern "C" int mopt_band_configure(mopt_band *b, double ratio, int fix_init_edge, int fix_end_edge, int apply_ts_opt, int apply_ci_neb) {
    if (!b) return MOPT_E_ARG;
    if (const char *e = getenv("MOPT_EIGH_WARM")) b->warm_enabled = atoi(e);
    b->ratio = ratio; b->fix_init = fix_init_edge; b->fix_end = fix_end_edge; b->apply_ts_opt = apply_ts_opt; b->apply_ci = apply_ci_neb;
    return MOPT_OK;
}

static int band_local(mopt_band *b, int image) {
    if (image < b->first || image >= b->first + b->count) { b->ctx->err = "image not owned by this shard"; return -1; }
    return image - b->first;
}

extern "C" int mopt_band_set_hessian(mopt_band *b, int image, const double *H) {
    if (!b) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    BTRY(mopt_band_alloc_internal(b));
    const int loc = band_local(b, image);
    if (loc < 0) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    const size_t nn = (size_t)b->n * b->n;
    if (H) {
        // The kernels of the update (u = H s stands for s^T H as well) assume a symmetric matrix; the reference would carry an unsymmetric
        // one as it is until its first update (rsirfo.py:1368-1372).  An upload that is not bitwise symmetric is therefore replaced by
        // 0.5 (H + H^T) here: results then differ from the reference's by O(asymmetry) -- as they did with the matrix kept as it was.
        bool sym = true;
        for (int i = 0; i < b->n && sym; ++i)
            for (int j = 0; j < i; ++j)
                if (std::memcmp(&H[(size_t)i * b->n + j], &H[(size_t)j * b->n + i], 8) != 0) { sym = false; break; }
        if (sym) {
            MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->wk.H + loc * nn, H, nn * 8, hipMemcpyHostToDevice, ctx->stream));
        } else {
            std::vector<double> hs(nn);
            for (int i = 0; i < b->n; ++i)
                for (int j = 0; j <= i; ++j) { const double v = 0.5 * (H[(size_t)i * b->n + j] + H[(size_t)j * b->n + i]); hs[(size_t)i * b->n + j] = v; hs[(size_t)j * b->n + i] = v; }
            MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->wk.H + loc * nn, hs.data(), nn * 8, hipMemcpyHostToDevice, ctx->stream));
            MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        b->H_sym[loc] = 1;
        b->dense_hessians = true;
    } else {
        b->H_sym[loc] = 1;
        std::vector<double> eye(nn, 0.0);
        for (int i = 0; i < b->n; ++i) eye[(size_t)i * b->n + i] = 1.0;
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->wk.H + loc * nn, eye.data(), nn * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

namespace mopt { __global__ void __launch_bounds__(256) k_band_add_sym(int n, const double *__restrict__ B, double *__restrict__ H) {
    // H += 0.5 (B + B^T), element (i, j) and (j, i) through the same operations: a bitwise symmetric H stays bitwise symmetric
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * n) return;
    const int i = (int)(e / n), j = (int)(e - (size_t)i * n);
    const double a = B[(size_t)i * n + j], c = B[(size_t)j * n + i];
    H[e] += (i <= j) ? 0.5 * (a + c) : 0.5 * (c + a);
} }

// H_image += B (symmetrised): the bias Hessian of nebmain's _apply_bias_potential (neb.py:806-832 adds it to tmp_hessian_i.npy every step)
extern "C" int mopt_band_add_hessian(mopt_band *b, int image, const double *B) {
    if (!b || !B) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    BTRY(mopt_band_alloc_internal(b));
    const int loc = band_local(b, image);
    if (loc < 0) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    const size_t nn = (size_t)b->n * b->n;
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->wk.Gt, B, nn * 8, hipMemcpyHostToDevice, ctx->stream));      // (Gt: scratch of the eigen-solver, free between steps)
    hipLaunchKernelGGL(mopt::k_band_add_sym, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, b->n, (const double *)b->wk.Gt, b->wk.H + loc * nn);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    b->dense_hessians = true;
    b->have_vecs = false;
    return MOPT_OK;
}

extern "C" int mopt_band_get_hessian(mopt_band *b, int image, double *H) {
    if (!b || !H) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    BTRY(mopt_band_alloc_internal(b));
    const int loc = band_local(b, image);
    if (loc < 0) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    const size_t nn = (size_t)b->n * b->n;
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(H, b->wk.H + loc * nn, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_band_get_state(mopt_band *b, int image, mopt_rsirfo_state *out) {
    if (!b || !out) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    BTRY(mopt_band_alloc_internal(b));
    const int loc = band_local(b, image);
    if (loc < 0) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    BTRY(band_check_failures(b, nullptr));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(out, b->wk.states + loc, sizeof(mopt_rsirfo_state), hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (out->jacobi_sweeps < 0) { ctx->err = "mopt_band_get_state: the eigen-solver failed for this image"; return MOPT_E_NUMERIC; }
    return MOPT_OK;
}

extern "C" int mopt_band_upload(mopt_band *b, const double *X, const double *E, const double *G) {
    if (!b || !X || !E || !G) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    BTRY(mopt_band_alloc_internal(b));
    mopt_ctx *ctx = b->ctx;
    const size_t nv = (size_t)b->I * b->n * 8;
    if (b->uploaded) {
        // the inputs of the previous step become pre_geom / pre_gradient / pre_velocity (neb.py:594-600)
        std::swap(b->X, b->pX); std::swap(b->G, b->pG); std::swap(b->F, b->pF);
    }
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->X, X, nv, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->G, G, nv, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->E, E, (size_t)b->I * 8, hipMemcpyHostToDevice, ctx->stream));
    b->hE.assign(E, E + b->I);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    b->uploaded = true;
    return MOPT_OK;
}

static void band_default_params(mopt_rsirfo_params &p, bool is_end, bool neb_mode) {
    // RSIRFO(method=..., saddle_order=0, trust_radius=...) defaults (rsirfo.py:26-91; rfo_neb.py:119-121)
    p.alpha0 = 1.0; p.alpha_max = 1000.0; p.alpha_step_max = 10.0; p.small_eigval_thresh = 1e-6;
    p.trust_radius_min = 0.01; p.trust_radius_max = 0.5;
    p.good_step_threshold = 0.75; p.poor_step_threshold = 0.25;
    p.trust_radius_increase_factor = 1.2; p.trust_radius_decrease_factor = 0.5;
    p.energy_change_threshold = 1e-6; p.gradient_norm_threshold = 1e-4; p.step_norm_tolerance = 1e-3;
    p.adaptive_trust_gradient_norm_threshold = 1e-2; p.max_curvature_factor = 2.5; p.negative_curvature_safety = 0.8;
    p.level_shift_value = 1e-5; p.condition_number_threshold = 1e8;
    p.max_micro_cycles = 40; p.saddle_order = 0;
    p.rule = is_end ? MOPT_RULE_BLOCK_FSB : MOPT_RULE_BLOCK_BOFILL;
    p.use_adaptive_trust_radius = 1; p.use_level_shift = 0; p.auto_level_shift = 1;
    p.neb_mode = neb_mode ? 1 : 0; p.reserved = 0;
}

// One band step in two halves, so that an image-sharded band can exchange FIRE's partial sums between them (comm.hip):
//   phase A: optimizer (re-)creation, BNEB force of the owned images, the partial sum of  v_prev . F  over the owned images
//   phase B: Ayala correction, one RSIRFO.run per owned image, step limit + TR_calc + FIRE + blend
int mopt_band_step_phase_a(mopt_band *b, int optimize_num) {
    if (!b) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    if (!b->uploaded) { ctx->err = "mopt_band_step: no inputs uploaded"; return MOPT_E_STATE; }
    hipSetDevice(ctx->device);
    const int I = b->I, N = b->N, n = b->n, first = b->first, count = b->count;
    RsirfoWork &wk = b->wk;
    const bool sharded = (b->comm != nullptr);
    if (sharded && !b->halo_done) { ctx->err = "mopt_band_step: image-sharded band without mopt_halo_exchange for this step's inputs"; return MOPT_E_STATE; }
    if (sharded) MOPT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, b->ev_out, 0));      // the neighbours' images and all energies have arrived
    if (optimize_num == 0 || !b->opts_init) {
        // rfo_neb.py:112-127: fresh optimizer instances; interior images switch to NEB mode unless they are
        // energy maxima and apply_ts_opt is set
        if (sharded) {                                       // the energies of the whole band were gathered on the device
            b->hE.resize(I);
            MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->hE.data(), b->E, (size_t)I * 8, hipMemcpyDeviceToHost, ctx->stream));
            MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        std::vector<mopt_rsirfo_params> ps(count);
        std::vector<mopt_rsirfo_state> ss(count);
        std::vector<int> rules(count);
        for (int k = 0; k < count; ++k) {
            const int i = first + k;
            const bool is_end = (i == 0 || i == I - 1);
            const bool is_max = (!is_end) && b->hE[i] > b->hE[i - 1] && b->hE[i] > b->hE[i + 1];
            band_default_params(ps[k], is_end, !is_end && !(is_max && b->apply_ts_opt));
            std::memset((void *)&ss[k], 0, sizeof(mopt_rsirfo_state));
            ss[k].trust_radius = is_end ? 0.5 : 0.2;
            ss[k].prev_eigvec_size = -1;
            rules[k] = ps[k].rule;
        }
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(wk.params, ps.data(), count * sizeof(mopt_rsirfo_params), hipMemcpyHostToDevice, ctx->stream));
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(wk.states, ss.data(), count * sizeof(mopt_rsirfo_state), hipMemcpyHostToDevice, ctx->stream));
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(wk.rules, rules.data(), count * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        b->opts_init = true;
        b->have_prev = false;
        b->have_vecs = false;
    }
    const bool have_prev = b->have_prev && optimize_num != 0;
    // fused update + projection (k_fused.hip) when every owned Hessian is known to be bitwise symmetric; MOPT_BAND_FUSED=0: unfused chain.
    // Its first kernel needs the TR/rot basis of the new geometries: forked onto the side stream here, ahead of the force kernel.
    {
        static const bool fused_env = !(getenv("MOPT_BAND_FUSED") && atoi(getenv("MOPT_BAND_FUSED")) == 0);
        bool fused = fused_env;
        for (char c : b->H_sym) fused = fused && c;
        b->fused_step = fused;
        if (fused && !mopt_small_eligible(n)) BTRY(mopt_launch_trrot_basis_side(ctx, n, count, b->X + (size_t)first * n, wk.Q));      // (small images: k_small.hip forms it itself)
    }
    // (a fill node, not a copy from a host buffer: the step enqueues without waiting for the stream; the fused chain takes the flag as a
    // kernel argument instead)
    if (!b->fused_step) MOPT_HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)wk.do_update, have_prev ? 1 : 0, (size_t)count, ctx->stream));
    // 1. NEB force.  Whole band on one device: all images (FIRE's switch sums over all of F).  Image-sharded: the owned images;
    //    the switch is assembled from per-rank partial sums.
    if (!sharded) {
        // (gamma of the owned images comes out of the same launch when this band owns them all: gamma[] is indexed by the local image)
        BTRY(mopt_launch_bneb_force(ctx, I, N, 0, I, b->X, b->E, b->G, optimize_num > b->apply_ci ? 1 : 0, b->F, b->tau, (first == 0 && count == I) ? b->gamma : nullptr));
    } else {
        BTRY(mopt_launch_bneb_force(ctx, I, N, first, count, b->X, b->E, b->G, optimize_num > b->apply_ci ? 1 : 0, b->F, b->tau));
        if (have_prev) {
            mopt_scope sc(ctx, "halo", 1);
            hipLaunchKernelGGL(k_band_dot, dim3(1), dim3(1024), 0, ctx->stream, count * n, (size_t)first * n, b->pF, b->F, b->dots + mopt_comm_rank(b->comm));
        }
        MOPT_HIP_TRY(ctx, hipEventRecord(b->ev_dot, ctx->stream));
    }
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

// step limit, TR_calc, FIRE, blend of the last step (idempotent: reads the step's inputs and wk.step, writes rfo_delta and newX)
static int band_launch_post(mopt_band *b) {
    mopt_ctx *ctx = b->ctx;
    const bool sharded = (b->comm != nullptr);
    mopt_scope sc(ctx, "neb_post", 1);
    hipLaunchKernelGGL(k_neb_post, dim3(b->count), dim3(512), (size_t)2 * b->n * 8, ctx->stream, b->I, b->N, b->first, b->last_optimize_num,
                       b->last_have_prev ? 1 : 0, b->ratio, b->fix_init, b->fix_end, b->X, b->G, b->F, b->pF, b->wk.step, b->rfo_delta, b->newX,
                       sharded ? (const double *)b->dots : (const double *)nullptr, sharded ? mopt_comm_nranks(b->comm) : 0);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

// Failure words of the steps since the host last looked.  A hand-off time-out (bit 8) of the LAST step is repaired here: the flagged images
// are solved again with one launch per Householder step and the tail of the step is repeated (the optimizer state of a flagged image was
// left untouched by k_rfo_scalar).  A failure of an EARLIER step that nobody looked at cannot be repaired any more (its inputs are gone)
// and is reported.  bits_out (may be null): the sticky bits found.
static int band_check_failures(mopt_band *b, int *bits_out) {
    mopt_ctx *ctx = b->ctx;
    if (bits_out) *bits_out = 0;
    if (!b->stepped) return MOPT_OK;
    int sticky = 0;
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(&sticky, b->wk.sticky, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int nsteps = b->steps_since_check;
    b->steps_since_check = 0;
    if (bits_out) *bits_out = sticky;
    if (!sticky) return MOPT_OK;
    MOPT_HIP_TRY(ctx, hipMemsetAsync(b->wk.sticky, 0, sizeof(int), ctx->stream));
    if ((sticky & 8) && nsteps <= 1) {
        RsirfoWork &wk = b->wk;
        const int n = b->n, first = b->first;
        double *sx = wk.x, *sbg = wk.bg, *sg = wk.g, *spx = wk.px, *spg = wk.pg;
        wk.x = b->X + (size_t)first * n; wk.bg = b->G + (size_t)first * n; wk.g = wk.bg;
        wk.px = b->pX + (size_t)first * n; wk.pg = b->pG + (size_t)first * n;
        int nre = 0;
        int rc = mopt_rsirfo_retry_timeouts(ctx, wk, n, b->count, wk.H, b->last_have_coef, &nre);
        wk.x = sx; wk.bg = sbg; wk.g = sg; wk.px = spx; wk.pg = spg;
        if (rc != MOPT_OK) return rc;
        BTRY(band_launch_post(b));
        int again = 0;
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(&again, b->wk.sticky, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        MOPT_HIP_TRY(ctx, hipMemsetAsync(b->wk.sticky, 0, sizeof(int), ctx->stream));
        if (bits_out) *bits_out = again;
        if (!again) return MOPT_OK;
        sticky = again;
    }
    ctx->err = (sticky & 8) ? "band step: the workgroup hand-off of the tridiagonalisation timed out in a step that can no longer be repeated (GPU shared with a long-running kernel)"
                            : "band step: the eigen-solver failed for an image of this shard";
    return MOPT_E_NUMERIC;
}

// test hook of the debug mode: writes one byte into the guard gap behind the band's X array
extern "C" int mopt_debug_damage_gap(mopt_band *b) {
    if (!b || b->gaps.size() < 2) return MOPT_E_STATE;
    hipSetDevice(b->ctx->device);
    MOPT_HIP_TRY(b->ctx, hipMemset((char *)b->block + b->gaps[1].first + 3, 0, 1));
    return MOPT_OK;
}

extern "C" int mopt_band_failures(mopt_band *b, int *bits) {
    if (!b) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    return band_check_failures(b, bits);
}

int mopt_band_step_phase_b(mopt_band *b, int optimize_num) {
    mopt_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    const int I = b->I, N = b->N, n = b->n, first = b->first, count = b->count;
    RsirfoWork &wk = b->wk;
    const bool sharded = (b->comm != nullptr);
    const bool have_prev = b->have_prev && optimize_num != 0;
    // 2. Ayala curvature correction of the owned interior images
    const bool fused = b->fused_step;
    {
        mopt_scope sc(ctx, "neb_ayala", fused ? 1 : 2);
        // (since round 3 the force kernel of phase a leaves gamma behind: MOPT_BAND_GAMMA_KERNEL=1 restores the separate launch)
        // bands that own only a part of the images keep the launch (the force kernel above runs over all images or before the halo is complete)
        static const bool own_kernel = getenv("MOPT_BAND_GAMMA_KERNEL") && atoi(getenv("MOPT_BAND_GAMMA_KERNEL")) == 1;
        if (own_kernel || sharded || !(first == 0 && count == I)) hipLaunchKernelGGL(k_gamma, dim3(count), dim3(128) /* the force kernel's block: same summation order, bitwise the same gamma */, 0, ctx->stream, I, n, first, b->X, b->E, b->G, b->tau, b->gamma);
        if (!fused) hipLaunchKernelGGL(k_rank1, dim3((n + 7) / 8, count), dim3(256), 0, ctx->stream, n, first, b->gamma, b->tau, wk.H);
    }
    const mopt::FusedBand fb = {b->gamma, b->tau, first, have_prev ? 1 : 0, false};
    // 3. per-image RSIRFO.run with B_g = g = biased gradient, energies 0.0 (rfo_neb.py:159-166)
    double *sx = wk.x, *sbg = wk.bg, *sg = wk.g, *spx = wk.px, *spg = wk.pg;
    wk.x = b->X + (size_t)first * n; wk.bg = b->G + (size_t)first * n; wk.g = wk.bg;
    wk.px = b->pX + (size_t)first * n; wk.pg = b->pG + (size_t)first * n;
    const bool warm = b->warm_enabled && b->have_vecs && b->warm_run < 64;
    wk.defl_hint = b->dense_hessians ? 0 : 1;
    int rc = mopt_rsirfo_pipeline(ctx, wk, n, count, have_prev, false, nullptr, wk.H, warm, fused ? &fb : nullptr);
    b->warm_run = warm ? b->warm_run + 1 : 0;
    b->have_vecs = true;
    b->stepped = true;
    wk.x = sx; wk.bg = sbg; wk.g = sg; wk.px = spx; wk.pg = spg;
    if (rc != MOPT_OK) return rc;
    b->last_optimize_num = optimize_num; b->last_have_prev = have_prev; b->last_have_coef = fused || have_prev;
    ++b->steps_since_check;
    // 4.-7. step limit, TR_calc, FIRE, blend
    if (sharded) MOPT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, b->ev_dot_done, 0));      // every rank's partial sum has arrived
    BTRY(band_launch_post(b));
    b->have_prev = true;
    b->halo_done = false;
    return MOPT_OK;
}

extern "C" int mopt_band_step(mopt_band *b, int optimize_num) {
    if (!b) return MOPT_E_ARG;
    if (b->comm) { mopt_band *one[1] = {b}; return mopt_band_step_multi(1, one, optimize_num); }
    BTRY(mopt_band_step_phase_a(b, optimize_num));
    return mopt_band_step_phase_b(b, optimize_num);
}

// Model Hessians of the owned images from the uploaded geometries, written straight into the band's device-resident Hessians
// (the reference computes ApproxHessian().main per image every -mfc steps and passes it through tmp_hessian_i.npy, neb.py:515-523).
extern "C" int mopt_band_model_hessian_ex(mopt_band *b, int kind, int flags, const double *par, const double *d3) {
    if (!b || !par || !d3) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    if (kind < MOPT_MODEL_FISCHERD3OLD || kind > MOPT_MODEL_LINDH2007D3) { ctx->err = "mopt_band_model_hessian: unknown model"; return MOPT_E_ARG; }
    if (!b->uploaded) { ctx->err = "mopt_band_model_hessian: no geometries uploaded"; return MOPT_E_STATE; }
    hipSetDevice(ctx->device);
    double *d_par = nullptr;
    MOPT_HIP_TRY(ctx, hipMalloc((void **)&d_par, (size_t)6 * b->N * 8));
    hipError_t e = hipMemcpyAsync(d_par, par, (size_t)6 * b->N * 8, hipMemcpyHostToDevice, ctx->stream);
    int rc = MOPT_E_DEVICE;
    const double *Xown = b->X + (size_t)b->first * b->n;
    if (e == hipSuccess) {
        if (kind == MOPT_MODEL_LINDH2007D3)                  // (its eigenvalue scale takes the gradient of each image: the ones last uploaded)
            rc = mopt_lindh2007d3_full(ctx, b->N, b->count, Xown, d_par, d3, b->G + (size_t)b->first * b->n, b->wk, b->wk.H);
        else
            rc = mopt_launch_model_hessian(ctx, kind, b->N, b->count, Xown, d_par, d3, b->wk.Gt, b->wk.Q, b->wk.W6, b->wk.H, true);
    }
    if (rc == MOPT_OK && (flags & MOPT_MODEL_FLAG_CLIP)) rc = mopt_model_hessian_clip(ctx, b->n, b->count, b->wk, b->wk.H);
    // the projection's rank-12 correction (and V f(L) V^T) is symmetric only up to rounding: one in-place 0.5 (H + H^T) makes the resident
    // matrices bitwise symmetric -- what an upload of the same matrices would store, see mopt_band_set_hessian
    if (rc == MOPT_OK) rc = mopt_launch_sym_add(ctx, b->n, b->count, b->wk.H, nullptr, b->wk.H);
    hipStreamSynchronize(ctx->stream);
    hipFree(d_par);
    b->have_vecs = false;
    b->dense_hessians = true;
    b->H_sym.assign((size_t)b->count, 1);
    return rc;
}

extern "C" int mopt_band_model_hessian(mopt_band *b, int kind, const double *par, const double *d3) { return mopt_band_model_hessian_ex(b, kind, 0, par, d3); }

// Inputs of the owned images only (image-sharded band): X_own, G_own: count x N x 3, E_own: count.  The neighbours' boundary images
// and the other ranks' energies arrive through mopt_halo_exchange.
extern "C" int mopt_band_upload_shard(mopt_band *b, const double *X_own, const double *E_own, const double *G_own) {
    if (!b || !X_own || !E_own || !G_own) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    BTRY(mopt_band_alloc_internal(b));
    mopt_ctx *ctx = b->ctx;
    const size_t cv = (size_t)b->count * b->n * 8, off = (size_t)b->first * b->n;
    if (b->uploaded) { std::swap(b->X, b->pX); std::swap(b->G, b->pG); std::swap(b->F, b->pF); }
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->X + off, X_own, cv, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->G + off, G_own, cv, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(b->E + b->first, E_own, (size_t)b->count * 8, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    b->uploaded = true;
    b->halo_done = false;
    return MOPT_OK;
}

extern "C" int mopt_band_upload_shard_dev(mopt_band *b, const double *dX_own, const double *dE_own, const double *dG_own) {
    if (!b || !dX_own || !dE_own || !dG_own) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    BTRY(mopt_band_alloc_internal(b));
    mopt_ctx *ctx = b->ctx;
    if (b->uploaded) { std::swap(b->X, b->pX); std::swap(b->G, b->pG); std::swap(b->F, b->pF); }
    const int nd = b->count * b->n;
    hipLaunchKernelGGL(mopt::k_band_take_shard, dim3((nd + 255) / 256), dim3(256), 0, ctx->stream, nd, b->count, b->first, b->n, dX_own, dG_own, dE_own,
                       b->X, b->G, b->E);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    b->uploaded = true;
    b->halo_done = false;
    return MOPT_OK;
}

extern "C" int mopt_band_download(mopt_band *b, double *newX_angstrom, double *F, double *rfo_step) {
    if (!b) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    const size_t cv = (size_t)b->count * b->n * 8;
    BTRY(band_check_failures(b, nullptr));                   // (repairs a hand-off time-out of the last step, reports anything else)
    if (newX_angstrom) MOPT_HIP_TRY(ctx, hipMemcpyAsync(newX_angstrom, b->newX, cv, hipMemcpyDeviceToHost, ctx->stream));
    if (F) MOPT_HIP_TRY(ctx, hipMemcpyAsync(F, b->F, (size_t)b->I * b->n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (rfo_step) MOPT_HIP_TRY(ctx, hipMemcpyAsync(rfo_step, b->rfo_delta, cv, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int> einfo((size_t)2 * b->count, 0);
    if (b->stepped) MOPT_HIP_TRY(ctx, hipMemcpyAsync(einfo.data(), b->wk.einfo, einfo.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < b->count; ++i)
        if (einfo[2 * i + 1] & 4) { ctx->err = "mopt_band_download: the eigen-solver failed for an image of this shard"; return MOPT_E_NUMERIC; }
    if (!b->gaps.empty()) {                                  // debug mode: guard gaps of the band's arrays and of its RSIRFO workspace
        BTRY(mopt_check_gaps(ctx, b->block, b->gaps, b->gap_names, "band arrays"));
        BTRY(b->wk.check_canaries());
    }
    return MOPT_OK;
}

extern "C" int mopt_band_upload_dev(mopt_band *b, const double *dX, const double *dE, const double *dG, const double *E_host) {
    if (!b || !dX || !dE || !dG) return MOPT_E_ARG;
    hipSetDevice(b->ctx->device);
    BTRY(mopt_band_alloc_internal(b));
    mopt_ctx *ctx = b->ctx;
    const size_t nv = (size_t)b->I * b->n * 8;
    if (b->uploaded) { std::swap(b->X, b->pX); std::swap(b->G, b->pG); std::swap(b->F, b->pF); }
    // one small kernel instead of three copy nodes (5-6 us each on this stream)
    {
        const int nd = b->I * b->n;
        hipLaunchKernelGGL(mopt::k_band_take, dim3((nd + 255) / 256), dim3(256), 0, ctx->stream, nd, b->I, dX, dG, dE, b->X, b->G, b->E);
        MOPT_HIP_TRY(ctx, hipGetLastError());
    }
    (void)nv;
    if (E_host) b->hE.assign(E_host, E_host + b->I);
    b->uploaded = true;
    return MOPT_OK;
}

extern "C" int mopt_band_download_dev(mopt_band *b, double *d_newX) {
    if (!b || !d_newX) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(d_newX, b->newX, (size_t)b->count * b->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return MOPT_OK;
}
