// The `sr` modifier of ApproxHessian.main (multioptpy/ModelHessian/approx_hessian.py:100-102 -> ModelHessian/shortrange.py:7-346) and the
// `clip` modifier as an operation of its own (approx_hessian.py:104-110), so that the reference's order ts -> sr -> clip can be kept for
// every combination of modifiers.
//
// ShortRangeCorrectionHessian.main: for every NON-bonded atom pair (BondConnectivity: distance > 1.1 x the sum of the covalent radii,
// Utils/bond_connectivity.py:24-40) within 15 bohr the second derivatives of q_i q_j c_x s V(r), V(r) = (1 - erf(omega r)) / r, as a
// radial pair term:   block = q_i q_j c_x s [ V'' u u^T + (V' / r)(1 - u u^T) ],   H_ii += block, H_jj += block, H_ij -= block  (:215-284),
// then the TR/rot projection of the CORRECTION alone (:303), base + correction, symmetrised (:305-307), NaN -> 0 (:339).
// Charges q = 0.2 (mean electronegativity - electronegativity) come from the host (:147-186, a table look-up).
//   k_sr_assemble   one workgroup per atom: its off-diagonal blocks, then its diagonal block as the sum over the partners in ascending
//                   order (the order in which the reference's pair loop reaches them) -- no atomics, deterministic
//   k_sr_add        out = 0.5 ((H + C) + (H + C)^T), nan_to_num
#include <cfloat>

#include "pipeline.h"

namespace mopt {

// prm: omega, cx_sr, scaling_factor, cut-off (bohr)
__global__ void __launch_bounds__(128) k_sr_assemble(int N, const double *__restrict__ X_, const double *__restrict__ cov, const double *__restrict__ chg,
                                                     double omega, double cx, double scal, double cutoff, double *__restrict__ C_) {
    const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x, n = 3 * N;
    const double *x = X_ + (size_t)b * n;
    double *C = C_ + (size_t)b * n * n;
    const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2], ci = cov[i], qi = chg[i];
    const double rsp = 1.7724538509055159;                  // sqrt(pi)
    for (int j = tid; j < N; j += blockDim.x) {
        if (j == i) continue;
        // (the reference's pair loop runs over i < j with r_vec = x_j - x_i: the sign of u cancels in u u^T)
        const double dx = x[3 * j] - xi, dy = x[3 * j + 1] - yi, dz = x[3 * j + 2] - zi;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        const bool bonded = r <= (ci + cov[j]) * 1.1;
        double blk[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (!bonded && !(r > cutoff)) {
            const double u[3] = {dx / r, dy / r, dz / r};
            const double qf = qi * chg[j] * cx * scal;
            double d1, d2;
            if (r < 1e-10) { d1 = -2.0 * omega * omega * omega / (3.0 * rsp); d2 = 0.0; }      // shortrange.py:100-102, 131-133
            else {
                const double ef = erf(omega * r), ex = exp(-(omega * r) * (omega * r));
                d1 = 2.0 * omega * ex / (rsp * r) + (ef - 1.0) / (r * r);                       // :104-113
                const double exf = ex / rsp;
                d2 = 2.0 * (2.0 * ef - 1.0) / (r * r * r) + 4.0 * omega * exf / (r * r) + 2.0 * (omega * omega * omega) * exf;      // :135-145
            }
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) {
                    const double ro = u[a] * u[c];
                    blk[3 * a + c] = qf * (d2 * ro + d1 / r * ((a == c ? 1.0 : 0.0) - ro));       // :246-248
                }
        }
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) C[(size_t)(3 * i + a) * n + 3 * j + c] = 0.0 - blk[3 * a + c];
    }
    __syncthreads();
    if (tid < 9) {
        const int a = tid / 3, c = tid % 3;
        double s = 0.0;
        for (int j = 0; j < N; ++j) if (j != i) s += 0.0 - C[(size_t)(3 * i + a) * n + 3 * j + c];
        C[(size_t)(3 * i + a) * n + 3 * i + c] = s;
    }
}

__global__ void __launch_bounds__(256) k_sr_add(int n, const double *__restrict__ H_, const double *__restrict__ A_, double *__restrict__ out_) {
    const int b = blockIdx.y;
    const size_t nn = (size_t)n * n, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nn) return;
    const int i = (int)(e / n), j = (int)(e - (size_t)i * n);
    const double *H = H_ + b * nn, *A = A_ + b * nn;
    double v = 0.5 * ((H[e] + A[e]) + (H[(size_t)j * n + i] + A[(size_t)j * n + i]));
    if (v != v) v = 0.0;                                     // np.nan_to_num(nan = 0.0): infinities become the largest finite numbers
    else if (v > DBL_MAX) v = DBL_MAX;
    else if (v < -DBL_MAX) v = -DBL_MAX;
    out_[b * nn + e] = v;
}

}  // namespace mopt

namespace {
struct SrBuf {
    void *p = nullptr;
    ~SrBuf() { if (p) hipFree(p); }
    int alloc(mopt_ctx *ctx, size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { ctx->err = "hipMalloc failed"; p = nullptr; return MOPT_E_NOMEM; }
        return MOPT_OK;
    }
    template <class T> T *as() { return (T *)p; }
};
}  // namespace
#define STRY(x) do { int rc__ = (x); if (rc__ != MOPT_OK) return rc__; } while (0)

// X: batch x N x 3 (bohr); cov: N covalent radii (bohr); chg: N charges; prm: omega, cx_sr, scaling factor, cut-off; H: batch x 3N x 3N, in and out.
// Host pointers.
extern "C" int mopt_model_hessian_sr(mopt_ctx *ctx, int N, int batch, const double *X, const double *cov, const double *chg, const double *prm, double *H) {
    using namespace mopt;
    if (!ctx) return MOPT_E_ARG;
    if (N < 1 || 3 * N > MOPT_MAX_N || batch < 1 || !X || !cov || !chg || !prm || !H) { ctx->err = "mopt_model_hessian_sr: bad argument"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t nn = (size_t)batch * n * n * sizeof(double), nv = (size_t)batch * n * sizeof(double);
    SrBuf dX, dP, dC, dA, dH, dQ, dW;
    STRY(dX.alloc(ctx, nv)); STRY(dP.alloc(ctx, (size_t)2 * N * 8)); STRY(dC.alloc(ctx, nn)); STRY(dA.alloc(ctx, nn)); STRY(dH.alloc(ctx, nn));
    STRY(dQ.alloc(ctx, 6 * nv)); STRY(dW.alloc(ctx, 6 * nv));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(dX.p, X, nv, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(dP.p, cov, (size_t)N * 8, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(dP.as<double>() + N, chg, (size_t)N * 8, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(dH.p, H, nn, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_sr_assemble, dim3(N, batch), dim3(128), 0, ctx->stream, N, (const double *)dX.p, (const double *)dP.p, (const double *)dP.as<double>() + N,
                       prm[0], prm[1], prm[2], prm[3], dC.as<double>());
    // the correction alone is projected (shortrange.py:303), with the projection of the optimizer path (Utils/calc_tools.py:249-316)
    STRY(mopt_launch_project_hess(ctx, n, batch, dC.as<double>(), nullptr, dX.as<double>(), dQ.as<double>(), dW.as<double>(), dA.as<double>()));
    hipLaunchKernelGGL(k_sr_add, dim3((unsigned)(((size_t)n * n + 255) / 256), batch), dim3(256), 0, ctx->stream, n, (const double *)dH.p, (const double *)dA.p, dC.as<double>());
    MOPT_HIP_TRY(ctx, hipGetLastError());
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(H, dC.p, nn, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

// H <- V smooth(lambda) V^T (approx_hessian.py:104-120) for batch n x n host matrices, in place.
extern "C" int mopt_hessian_clip(mopt_ctx *ctx, int n, int batch, double *H) {
    if (!ctx) return MOPT_E_ARG;
    if (n < 1 || n > MOPT_MAX_N || batch < 1 || !H) { ctx->err = "mopt_hessian_clip: bad argument"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)batch * n * n * sizeof(double);
    SrBuf dA;
    STRY(dA.alloc(ctx, nn));
    RsirfoWork wk;
    STRY(wk.alloc(ctx, n, batch, false, false, false));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(dA.p, H, nn, hipMemcpyHostToDevice, ctx->stream));
    STRY(mopt_model_hessian_clip(ctx, n, batch, wk, dA.as<double>()));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(H, dA.p, nn, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> einfo((size_t)2 * batch);
    MOPT_HIP_TRY(ctx, hipMemcpy(einfo.data(), wk.einfo, einfo.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; ++b) if (einfo[2 * b + 1] & 4) { ctx->err = "mopt_hessian_clip: the eigen-solver failed"; return MOPT_E_NUMERIC; }
    return MOPT_OK;
}
