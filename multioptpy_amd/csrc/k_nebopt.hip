// Standalone NEB optimizers of nebmain's force-only branch, one workgroup per image:
//   mopt_neb_tr_calc   TR_NEB.TR_calc                      (multioptpy/Optimizer/trust_radius_neb.py:17-98)
//   mopt_neb_fire      FIREOptimizer.optimize              (multioptpy/Optimizer/fire_neb.py:38-92)
//   mopt_neb_sd        SteepestDescentOptimizer.optimize   (multioptpy/Optimizer/gradientdescent_neb.py:22-51)
// The fused band step (band.hip, k_neb_post) contains the same TR/FIRE arithmetic for the RFO blend.
#include <cstring>

#include "pipeline.h"
#include "neb_tr.hpp"

namespace mopt {

// move[i] = TR_calc(X, F, delta)[i]
__global__ void __launch_bounds__(128) k_tr_calc(int I, int n, int fix_init, int fix_end, const double *__restrict__ X,
                                                 const double *__restrict__ F, const double *__restrict__ delta,
                                                 double *__restrict__ move) {
    __shared__ double red[6 * 16];
    const int i = blockIdx.x;
    const double *x0 = X + (size_t)i * n, *f0 = F + (size_t)i * n, *dl = delta + (size_t)i * n;
    double *mv = move + (size_t)i * n;
    double sc;
    if (i == 0 || i == I - 1) {
        double a = 0.0;
        for (int k = threadIdx.x; k < n; k += blockDim.x) a += dl[k] * dl[k];
        a = block_sum1(a, red);
        sc = tr_scale_edge(sqrt(a), (i == 0) ? (fix_init != 0) : (fix_end != 0));
    } else {
        const double *xm = X + (size_t)(i - 1) * n, *xp = X + (size_t)(i + 1) * n;
        double w[4] = {0, 0, 0, 0};
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const double dm = xm[k] - x0[k], dp = xp[k] - x0[k];
            w[0] += dm * dm; w[1] += dp * dp; w[2] += dl[k] * dl[k]; w[3] += f0[k] * f0[k];
        }
        block_sum<4>(w, red);
        const double d1 = sqrt(w[0]), d2 = sqrt(w[1]), dn = sqrt(w[2]), fn = sqrt(w[3]);
        double u[3] = {0, 0, 0};
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const double v1 = (xm[k] - x0[k]) / (d1 + 1e-15), v2 = (xp[k] - x0[k]) / (d2 + 1e-15);
            const double nd = dl[k] / dn;
            u[0] += v1 * nd; u[1] += v2 * nd; u[2] += f0[k] * dl[k];
        }
        __syncthreads();
        block_sum<3>(u, red);
        sc = tr_scale_interior(dn, d1 / 2.0, d2 / 2.0, u[0], u[1], u[2] / (fn * dn));
    }
    for (int k = threadIdx.x; k < n; k += blockDim.x) mv[k] = (sc == 1.0) ? dl[k] : dl[k] * sc;
}

// sum over the whole band of pre_velocity . force (fire_neb.py:60-65); one workgroup
__global__ void __launch_bounds__(1024) k_band_dot(size_t len, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (size_t k = threadIdx.x; k < len; k += blockDim.x) s += a[k] * b[k];
    s = block_sum1(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

// FIRE velocity mixing per atom and total_delta (fire_neb.py:43-83) with the already updated a, dt
__global__ void __launch_bounds__(128) k_fire_delta(int N, double fa, double dt, int keep, int have_prev, const double *__restrict__ F,
                                                    const double *__restrict__ V, const double *__restrict__ preV,
                                                    double *__restrict__ delta) {
    const int i = blockIdx.x;
    const size_t off = (size_t)i * 3 * N;
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        const double *f = F + off + 3 * a, *v = V + off + 3 * a;
        const double fn = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
        const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int c = 0; c < 3; ++c) {
            double vm = v[c];
            if (fn > 1e-10) vm = (1.0 - fa) * v[c] + fa * (vn / fn) * f[c];
            if (!keep) vm *= 0.0;
            const double tv = vm + dt * f[c];
            delta[off + 3 * a + c] = have_prev ? dt * (tv + preV[off + 3 * a + c]) : dt * tv;
        }
    }
}

// steepest descent with the half-neighbour-distance clip (gradientdescent_neb.py:22-51); writes the move vector
__global__ void __launch_bounds__(128) k_sd_move(int I, int n, int fix_init, int fix_end, const double *__restrict__ X,
                                                 const double *__restrict__ F, double *__restrict__ move) {
    __shared__ double red[3 * 16];
    const int i = blockIdx.x;
    const double *x0 = X + (size_t)i * n, *f0 = F + (size_t)i * n;
    double *mv = move + (size_t)i * n;
    double sc = 1.0;
    if (i == 0) sc = fix_init ? 0.0 : 1.0;
    else if (i == I - 1) sc = fix_end ? 0.0 : 1.0;
    else {
        const double *xm = X + (size_t)(i - 1) * n, *xp = X + (size_t)(i + 1) * n;
        double w[3] = {0, 0, 0};
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const double dm = x0[k] - xm[k], dp = x0[k] - xp[k], d = 0.5 * f0[k];
            w[0] += dm * dm; w[1] += dp * dp; w[2] += d * d;
        }
        block_sum<3>(w, red);
        const double r1 = sqrt(w[0]) / 2.0, r2 = sqrt(w[1]) / 2.0, dn = sqrt(w[2]);
        if (dn > r1) sc = r1 / dn; else if (dn > r2) sc = r2 / dn;
    }
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double d = 0.5 * f0[k];
        mv[k] = (sc == 1.0) ? d : ((sc == 0.0) ? d * 0.0 : d * sc);
    }
}

__global__ void k_geom_add(size_t len, double scale, const double *__restrict__ X, const double *__restrict__ move, double *__restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < len) out[k] = (X[k] + move[k]) * scale;
}

}  // namespace mopt

using namespace mopt;

namespace {
struct NBuf {
    void *p = nullptr;
    ~NBuf() { if (p) hipFree(p); }
    int alloc(mopt_ctx *ctx, size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { ctx->err = "hipMalloc failed"; p = nullptr; return MOPT_E_NOMEM; }
        return MOPT_OK;
    }
    double *d() { return (double *)p; }
};
#define NTRY(x) do { int rc__ = (x); if (rc__ != MOPT_OK) return rc__; } while (0)
#define NH2D(dst, src, bytes) MOPT_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream))
#define ND2H(dst, src, bytes) MOPT_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream))
}  // namespace

extern "C" int mopt_neb_tr_calc(mopt_ctx *ctx, int I, int N, const double *X, const double *F, const double *delta, int fix_init,
                                int fix_end, double *move) {
    if (!ctx) return MOPT_E_ARG;
    if (I < 2 || N < 1 || !X || !F || !delta || !move) { ctx->err = "mopt_neb_tr_calc: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t nv = (size_t)I * n * 8;
    NBuf dX, dF, dD, dM;
    NTRY(dX.alloc(ctx, nv)); NTRY(dF.alloc(ctx, nv)); NTRY(dD.alloc(ctx, nv)); NTRY(dM.alloc(ctx, nv));
    NH2D(dX.p, X, nv); NH2D(dF.p, F, nv); NH2D(dD.p, delta, nv);
    hipLaunchKernelGGL(k_tr_calc, dim3(I), dim3(128), 0, ctx->stream, I, n, fix_init, fix_end, dX.d(), dF.d(), dD.d(), dM.d());
    MOPT_HIP_TRY(ctx, hipGetLastError());
    ND2H(move, dM.p, nv);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_neb_fire(mopt_ctx *ctx, int I, int N, const double *X, const double *F, const double *vel, const double *pre_vel,
                             int optimize_num, int fix_init, int fix_end, const double *params, double *state, double bohr2ang,
                             double *new_geometry_ang, double *dot_out) {
    if (!ctx) return MOPT_E_ARG;
    if (I < 2 || N < 1 || !X || !F || !vel || !params || !state || !new_geometry_ang) { ctx->err = "mopt_neb_fire: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t len = (size_t)I * n, nv = len * 8;
    NBuf dX, dF, dV, dP, dD, dM, dO, dS;
    NTRY(dX.alloc(ctx, nv)); NTRY(dF.alloc(ctx, nv)); NTRY(dV.alloc(ctx, nv)); NTRY(dP.alloc(ctx, nv)); NTRY(dD.alloc(ctx, nv));
    NTRY(dM.alloc(ctx, nv)); NTRY(dO.alloc(ctx, nv)); NTRY(dS.alloc(ctx, 8));
    NH2D(dX.p, X, nv); NH2D(dF.p, F, nv); NH2D(dV.p, vel, nv);
    const bool have_prev = (optimize_num != 0 && pre_vel != nullptr);      // `len(pre_total_velocity) > 1`
    double dot = 0.0;
    if (have_prev) {
        NH2D(dP.p, pre_vel, nv);
        hipLaunchKernelGGL(k_band_dot, dim3(1), dim3(1024), 0, ctx->stream, len, dP.d(), dF.d(), dS.d());
        ND2H(&dot, dS.p, 8);
        MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    // scalar state machine of fire_neb.py:67-75; params = {N_accelerate, f_inc, f_decelerate, a_start, dt_max}
    double dt = state[0], a = state[1];
    int n_reset = (int)state[2];
    const bool keep = (optimize_num > 0 && dot > 0.0 && have_prev);
    if (keep) {
        if (n_reset > (int)params[0]) { dt = (dt * params[1] < params[4]) ? dt * params[1] : params[4]; a *= params[1]; }
        n_reset += 1;
    } else {
        a = params[3];
        dt *= params[2];
        n_reset = 0;
    }
    state[0] = dt; state[1] = a; state[2] = (double)n_reset;
    if (dot_out) *dot_out = dot;
    hipLaunchKernelGGL(k_fire_delta, dim3(I), dim3(128), 0, ctx->stream, N, a, dt, keep ? 1 : 0, have_prev ? 1 : 0, dF.d(), dV.d(), dP.d(), dD.d());
    hipLaunchKernelGGL(k_tr_calc, dim3(I), dim3(128), 0, ctx->stream, I, n, fix_init, fix_end, dX.d(), dF.d(), dD.d(), dM.d());
    hipLaunchKernelGGL(k_geom_add, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, len, bohr2ang, dX.d(), dM.d(), dO.d());
    MOPT_HIP_TRY(ctx, hipGetLastError());
    ND2H(new_geometry_ang, dO.p, nv);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_neb_sd(mopt_ctx *ctx, int I, int N, const double *X, const double *F, int fix_init, int fix_end, double bohr2ang,
                           double *new_geometry_ang) {
    if (!ctx) return MOPT_E_ARG;
    if (I < 2 || N < 1 || !X || !F || !new_geometry_ang) { ctx->err = "mopt_neb_sd: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t len = (size_t)I * n, nv = len * 8;
    NBuf dX, dF, dM, dO;
    NTRY(dX.alloc(ctx, nv)); NTRY(dF.alloc(ctx, nv)); NTRY(dM.alloc(ctx, nv)); NTRY(dO.alloc(ctx, nv));
    NH2D(dX.p, X, nv); NH2D(dF.p, F, nv);
    hipLaunchKernelGGL(k_sd_move, dim3(I), dim3(128), 0, ctx->stream, I, n, fix_init, fix_end, dX.d(), dF.d(), dM.d());
    hipLaunchKernelGGL(k_geom_add, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, len, bohr2ang, dX.d(), dM.d(), dO.d());
    MOPT_HIP_TRY(ctx, hipGetLastError());
    ND2H(new_geometry_ang, dO.p, nv);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}
