// NEB band kernels (one workgroup per image).
//   k_bneb_force   CaluculationBNEB.calc_force (multioptpy/MEP/pathopt_bneb_force.py:33-117).  The reference
//                  builds an N x 3N B-matrix and pseudo-inverts the 3N x 3N matrix B^T B with a full SVD
//                  (Coordinate/redundant_coordinate.py:377-440); B^T B is block diagonal with 3x3 blocks u_a u_a^T,
//                  so the projection is exactly per atom:  t_a = -(g_a . u_a) w_a u_a,
//                  u_a = (x'_a - x_a)/(|x'_a - x_a| + 1e-15), w_a = 1/|u_a|^2 if |u_a|^2 > 1e-6 else |u_a|^2.
//   k_ayala        calculate_gamma (pathopt_bneb_force.py:161-222) + H += gamma t t^T (rfo_neb.py:43-74)
//   k_neb_post     step limit (rfo_neb.py:76-83), TR_calc (trust_radius_neb.py:17-98), FIRE (fire_neb.py:38-92),
//                  blend (rfo_neb.py:197-206)
#include "common.cuh"

namespace mopt {

__device__ __forceinline__ bool is_local_max(const double *E, int I, int j) {
    return j >= 1 && j <= I - 2 && E[j] > E[j - 1] && E[j] > E[j + 1];
}

__device__ __forceinline__ void proj_atom(const double *xa, const double *xb, const double *g, int a, double coefw, double *t) {
    const double dx = xb[3 * a] - xa[3 * a], dy = xb[3 * a + 1] - xa[3 * a + 1], dz = xb[3 * a + 2] - xa[3 * a + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz) + 1e-15;
    const double ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
    const double s = ux * ux + uy * uy + uz * uz;
    const double w = (s > 1e-6) ? 1.0 / s : s;
    const double d = -(g[3 * a] * ux + g[3 * a + 1] * uy + g[3 * a + 2] * uz) * w * coefw;
    t[0] += d * ux; t[1] += d * uy; t[2] += d * uz;
}

// X, G: [Iglob][N][3]; E: [Iglob]; images [first, first+count) are computed (image-sharded band).
__global__ void __launch_bounds__(128) k_bneb_force(int I, int N, int first, const double *__restrict__ X,
                                                    const double *__restrict__ E, const double *__restrict__ G,
                                                    int ci_active, double *__restrict__ F, double *__restrict__ tau) {
    const int i = first + blockIdx.x;
    const size_t off = (size_t)i * N * 3;
    const double *g = G + off;
    double *f = F + off, *tg = tau + off;
    if (i == 0 || i == I - 1) {
        for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) { f[k] = -1.0 * g[k]; tg[k] = 0.0; }
        return;
    }
    const double *xm = X + off - (size_t)N * 3, *x0 = X + off, *xp = X + off + (size_t)N * 3;
    const double e0 = E[i - 1], e1 = E[i], e2 = E[i + 1];
    int mode;            // 0 rising, 1 falling, 2 extremum
    double wp = 0.0, wm = 0.0;
    if (e0 < e1 && e1 < e2) { mode = 0; wp = 1.0; }
    else if (e0 > e1 && e1 > e2) { mode = 1; wm = 1.0; }
    else {
        mode = 2;
        const double d1 = fabs(e2 - e1), d0 = fabs(e1 - e0);
        const double mx = fmax(d1, d0), mn = fmin(d1, d0);
        const double a = mx / (mx + mn + 1e-8), bb = mn / (mx + mn + 1e-8);
        if (e0 < e2) { wp = a; wm = bb; } else { wp = bb; wm = a; }
    }
    // CI-NEB switches (pathopt_bneb_force.py:50-55; the maxima list carries two appended zeros)
    const bool near_max = is_local_max(E, I, i + 1) || is_local_max(E, I, i - 1) || (i - 1 == 0);
    const bool c1 = ci_active && near_max && (i != 1 && i != I - 2);
    const bool c2 = !c1 && ci_active && is_local_max(E, I, i) && (i != 1 || i != I - 2);
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        double t[3] = {0.0, 0.0, 0.0};
        if (mode != 1) proj_atom(x0, xp, g, a, wp, t);
        if (mode != 0) { double t2[3] = {0.0, 0.0, 0.0}; proj_atom(xm, x0, g, a, wm, t2); t[0] += t2[0]; t[1] += t2[1]; t[2] += t2[2]; }
        for (int c = 0; c < 3; ++c) {
            double force = g[3 * a + c] + t[c];
            if (c1) force *= 0.001;
            else if (c2) force = -2.0 * t[c];
            f[3 * a + c] = -1.0 * force;
            tg[3 * a + c] = t[c];
        }
    }
}

}  // namespace mopt

int mopt_launch_bneb_force(mopt_ctx *ctx, int I, int N, int first, int count, const double *d_X, const double *d_E,
                           const double *d_G, int ci_active, double *d_F, double *d_tau) {
    using namespace mopt;
    mopt_scope sc(ctx, "neb_force", 1);
    hipLaunchKernelGGL(k_bneb_force, dim3(count), dim3(128), 0, ctx->stream, I, N, first, d_X, d_E, d_G, ci_active, d_F, d_tau);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
