// Harmonic distance restraints ("keep potential") on the device: energy, gradient and Hessian of
//   E = sum_p 1/2 k_p (r_p - r0_p)^2,   r_p = |x_i - x_j|  (clamped from below at 1e-12 like the reference's torch.clamp)
// StructKeepPotential.calc_energy of the reference (multioptpy/Potential/keep_potential.py:21-58, one pair per object), whose gradient and
// Hessian the reference obtains by torch.func.jacrev / torch.func.hessian on the CPU every optimizer step (Potential/potential.py:127-152).
// Written out:  g_i = k (r - r0) u,  H_ii = H_jj = k [u u^T + (1 - r0 / r)(I - u u^T)],  H_ij = -H_ii,  u = (x_i - x_j) / r.
// A pair may instead carry the anharmonic (Morse-type) restraint of anharmonic_keep_potential.py:13-27,
//   E = D (1 - exp(-a (r - r0)))^2,  a = sqrt(k / (2 D)),  zero for D = 0,
// as a radial potential V(r):  g_i = V' u,  H_ii = V'' u u^T + (V' / r)(I - u u^T).
// The Hessian is n x n with at most 36 non-zero entries per pair: one launch zeroes it, one workgroup adds the pairs in the order
// given (a barrier between two pairs: they may share an atom), so the sum is deterministic.
#include "pipeline.h"

namespace mopt {

__global__ void __launch_bounds__(256) k_keep_zero(size_t total, double *__restrict__ H, int n, double *__restrict__ g, double *__restrict__ E) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) H[i] = 0.0;
    if (i < (size_t)n) g[i] = 0.0;
    if (i == 0) E[0] = 0.0;
}

// one workgroup of 64 threads; pairs: [npairs][2] 0-based atoms; kr: [npairs][3] = (k, r0 in bohr, well depth D; D < 0: harmonic)
__global__ void __launch_bounds__(64) k_keep_pairs(int N, const double *__restrict__ x, int npairs, const int *__restrict__ pairs, const double *__restrict__ kr,
                                                   double *__restrict__ E, double *__restrict__ g, double *__restrict__ H) {
    const int t = threadIdx.x, n = 3 * N;
    for (int p = 0; p < npairs; ++p) {
        const int i = pairs[2 * p], j = pairs[2 * p + 1];
        const double k = kr[3 * p], r0 = kr[3 * p + 1], D = kr[3 * p + 2];
        const bool harmonic = D < 0.0;
        const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
        const double rr = sqrt(dx * dx + dy * dy + dz * dz);
        const bool clamped = harmonic && !(rr > 1e-12);      // (torch.clamp(min = 1e-12) of the harmonic form: constant below the bound, zero derivatives)
        const double r = clamped ? 1e-12 : rr;
        const double u[3] = {dx / r, dy / r, dz / r};
        double V, V1, V2;                                    // the radial potential and its first two derivatives at r
        if (harmonic) { V = 0.5 * k * (r - r0) * (r - r0); V1 = k * (r - r0); V2 = k; }
        else if (D == 0.0) { V = 0.0; V1 = 0.0; V2 = 0.0; }  // (anharmonic_keep_potential.py:24-25: no well, no restraint)
        else {
            const double a_ = sqrt(k / (2.0 * D)), ex = exp(-a_ * (r - r0));
            V = D * (1.0 - ex) * (1.0 - ex); V1 = 2.0 * D * a_ * ex * (1.0 - ex); V2 = 2.0 * D * a_ * a_ * ex * (2.0 * ex - 1.0);
        }
        if (t == 0) E[0] += V;
        if (!clamped) {
            if (t < 6) { const int a = t % 3; const double gv = V1 * u[a]; if (t < 3) g[3 * i + a] += gv; else g[3 * j + a] -= gv; }
            if (t < 36) {
                const int a = t / 6, b = t % 6;              // entry (a, b) of the 6 x 6 block over (i, j)
                const int aa = a % 3, bb = b % 3;
                const double P = u[aa] * u[bb], dl = (aa == bb ? 1.0 : 0.0);
                double h = harmonic ? k * (P + (1.0 - r0 / r) * (dl - P)) : V2 * P + (V1 / r) * (dl - P);
                if ((a < 3) != (b < 3)) h = -h;
                const int row = (a < 3 ? 3 * i : 3 * j) + aa, col = (b < 3 ? 3 * i : 3 * j) + bb;
                H[(size_t)row * n + col] += h;
            }
        }
        __syncthreads();
    }
}

}  // namespace mopt

// X: N x 3 (bohr); pairs: npairs x 2 (0-based, i != j); kr: npairs x 3 = (spring constant in hartree / bohr^2, r0 in bohr, well depth D
// in hartree: D < 0 selects the harmonic form, D >= 0 the anharmonic one).
// Out: E (1), grad (N x 3), H (3N x 3N).  Host pointers.
extern "C" int mopt_keep_pot(mopt_ctx *ctx, int N, const double *X, int npairs, const int *pairs, const double *kr, double *E, double *grad, double *H) {
    using namespace mopt;
    if (!ctx) return MOPT_E_ARG;
    if (N < 1 || 3 * N > MOPT_MAX_N || npairs < 0 || !X || (npairs > 0 && (!pairs || !kr)) || !E || !grad || !H) { ctx->err = "mopt_keep_pot: bad argument"; return MOPT_E_ARG; }
    for (int p = 0; p < npairs; ++p)
        if (pairs[2 * p] < 0 || pairs[2 * p] >= N || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= N || pairs[2 * p] == pairs[2 * p + 1]) {
            ctx->err = "mopt_keep_pot: atom pair out of range (0-based indices, two different atoms)";
            return MOPT_E_ARG;
        }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t nn = (size_t)n * n;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_g = al(nn * 8), o_E = o_g + al((size_t)n * 8), o_x = o_E + 256, o_p = o_x + al((size_t)n * 8), o_k = o_p + al((size_t)(npairs + 1) * 8);
    char *d = nullptr;
    MOPT_HIP_TRY(ctx, hipMalloc((void **)&d, o_k + al((size_t)(npairs + 1) * 24)));
    struct Free { char *p; ~Free() { if (p) hipFree(p); } } guard{d};
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_x, X, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    if (npairs > 0) {
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_p, pairs, (size_t)npairs * 8, hipMemcpyHostToDevice, ctx->stream));
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_k, kr, (size_t)npairs * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    {
        mopt_scope sc(ctx, "bias_keep", 2);
        hipLaunchKernelGGL(k_keep_zero, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, nn, (double *)d, n, (double *)(d + o_g), (double *)(d + o_E));
        if (npairs > 0)
            hipLaunchKernelGGL(k_keep_pairs, dim3(1), dim3(64), 0, ctx->stream, N, (const double *)(d + o_x), npairs, (const int *)(d + o_p), (const double *)(d + o_k),
                               (double *)(d + o_E), (double *)(d + o_g), (double *)d);
    }
    MOPT_HIP_TRY(ctx, hipGetLastError());
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(E, d + o_E, 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(grad, d + o_g, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(H, d, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}
