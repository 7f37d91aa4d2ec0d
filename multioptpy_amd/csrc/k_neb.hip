// NEB band kernels (one workgroup per image).
//   k_bneb_force   CaluculationBNEB.calc_force (multioptpy/MEP/pathopt_bneb_force.py:33-117).  The reference
//                  builds an N x 3N B-matrix and pseudo-inverts the 3N x 3N matrix B^T B with a full SVD
//                  (Coordinate/redundant_coordinate.py:377-440); B^T B is block diagonal with 3x3 blocks u_a u_a^T,
//                  so the projection is exactly per atom:  t_a = -(g_a . u_a) w_a u_a,
//                  u_a = (x'_a - x_a)/(|x'_a - x_a| + 1e-15), w_a = 1/|u_a|^2 if |u_a|^2 > 1e-6 else |u_a|^2.
//   k_ayala        calculate_gamma (pathopt_bneb_force.py:161-222) + H += gamma t t^T (rfo_neb.py:43-74)
//   k_neb_post     step limit (rfo_neb.py:76-83), TR_calc (trust_radius_neb.py:17-98), FIRE (fire_neb.py:38-92),
//                  blend (rfo_neb.py:197-206)
#include "common.hpp"
#include "neb_gamma.hpp"

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
                                                    int ci_active, double *__restrict__ F, double *__restrict__ tau, double *__restrict__ gamma) {
    // gamma != null: the Ayala curvature of the image (gamma[blockIdx.x]; 0 for the end images) is left behind as well -- the band step needs it
    // right after the force and a launch of its own cost 8 us of a 0.2 ms step (16 x 14)
    __shared__ double red[5 * 16];
    const int i = first + blockIdx.x;
    const size_t off = (size_t)i * N * 3;
    const double *g = G + off;
    double *f = F + off, *tg = tau + off;
    if (i == 0 || i == I - 1) {
        for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) { f[k] = -1.0 * g[k]; tg[k] = 0.0; }
        if (gamma && threadIdx.x == 0) gamma[blockIdx.x] = 0.0;
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
    if (gamma) {
        __syncthreads();                                      // (the tangent written above, read back by other threads of the workgroup)
        const double *gm = G + off - (size_t)N * 3, *gp = G + off + (size_t)N * 3;
        double v[5];
        ayala_partial_sums(3 * N, xm, x0, xp, gm, g, gp, tg, v);
        block_sum<5>(v, red);
        if (threadIdx.x == 0) gamma[blockIdx.x] = ayala_gamma(v[0], v[1], v[2], v[3], v[4], e0, e1, e2);
    }
}

// ---- spring / tangent variants ----------------------------------------------------------------
// variant 1: CaluculationBNEB3 (pathopt_bneb_force.py:305-356)   BNEB projection + next-nearest-neighbour springs, k = 0.05
// variant 2: CaluculationEWBNEB (pathopt_ewbneb_force.py:21-117) BNEB projection (+CI switches) + energy-weighted springs
// variant 3: CaluculationDNEB (pathopt_dneb_force.py:20-130)     per-atom tangents, spring k = 0.01, "doubly nudged" term
__device__ __forceinline__ void unit3(const double *a, const double *b, int at, double *u) {   // (b - a)/|b - a| or 0
    const double dx = b[3 * at] - a[3 * at], dy = b[3 * at + 1] - a[3 * at + 1], dz = b[3 * at + 2] - a[3 * at + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    if (nrm != 0.0) { u[0] = dx / nrm; u[1] = dy / nrm; u[2] = dz / nrm; } else { u[0] = u[1] = u[2] = 0.0; }
}

__global__ void __launch_bounds__(128) k_neb_force_var(int variant, int I, int N, int first, const double *__restrict__ X,
                                                       const double *__restrict__ E, const double *__restrict__ G,
                                                       int ci_active, double *__restrict__ F) {
    __shared__ double red[4 * 16];
    const int i = first + blockIdx.x;
    const int n = 3 * N;
    const size_t off = (size_t)i * n;
    const double *g = G + off;
    double *f = F + off;
    if (i == 0 || i == I - 1) {
        for (int k = threadIdx.x; k < n; k += blockDim.x) f[k] = -1.0 * g[k];
        return;
    }
    const double *xm = X + off - n, *x0 = X + off, *xp = X + off + n;
    const double e0 = E[i - 1], e1 = E[i], e2 = E[i + 1];
    if (variant == 3) {
        // ---- DNEB
        const double dmax = fmax(e2 - e1, e0 - e1), dmin = fmin(e2 - e1, e0 - e1);
        const int mode = (e0 < e1 && e1 < e2) ? 0 : ((e0 > e1 && e1 > e2) ? 1 : 2);
        // CI: `energy_list[i] == energy_list[maxlist[0]]` with the list consumed in image order (:83-88)
        bool ci_hit = false;
        if (ci_active) {
            int ptr = 0;      // index into maxima + [0, 0]
            int nmax = 0;
            for (int j = 1; j < I - 1; ++j) if (is_local_max(E, I, j)) ++nmax;
            for (int ii = 1; ii <= i; ++ii) {
                if (ptr >= nmax + 2) break;
                int target = 0, seen = 0;
                if (ptr < nmax) { for (int j = 1; j < I - 1; ++j) if (is_local_max(E, I, j)) { if (seen == ptr) { target = j; break; } ++seen; } }
                const bool hit = (E[ii] == E[target]);
                if (ii == i) ci_hit = hit;
                if (hit) ++ptr;
            }
        }
        double s2[2] = {0.0, 0.0};      // nansum, count
        for (int a = threadIdx.x; a < N; a += blockDim.x) {
            double tau[3] = {0, 0, 0}, tp[3], tm[3];
            if (mode == 0) unit3(x0, xp, a, tau);
            else if (mode == 1) unit3(xm, x0, a, tau);
            else {
                unit3(xm, x0, a, tm);
                unit3(x0, xp, a, tp);
                if (e0 > e2) {
                    double v[3];
                    for (int c = 0; c < 3; ++c) v[c] = tp[c] * dmin + tm[c] * dmax;
                    const bool any = (v[0] != 0.0) || (v[1] != 0.0) || (v[2] != 0.0);
                    const double nn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                    for (int c = 0; c < 3; ++c) tau[c] = any ? v[c] / nn : 0.0;
                } else {
                    double v[3], o[3];
                    for (int c = 0; c < 3; ++c) { v[c] = tp[c] * dmax + tm[c] * dmin; o[c] = tp[c] * dmin + tm[c] * dmax; }
                    const double nn = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
                    for (int c = 0; c < 3; ++c) tau[c] = (nn != 0.0) ? v[c] / nn : 0.0;
                }
            }
            const double gx = g[3 * a], gy = g[3 * a + 1], gz = g[3 * a + 2];
            const double gt = gx * tau[0] + gy * tau[1] + gz * tau[2];
            double tot[3];
            if (ci_hit) {
                const double gg[3] = {gx, gy, gz};
                for (int c = 0; c < 3; ++c) tot[c] = (-1.0) * 0.01 * (gg[c] - 2.0 * gt * tau[c]);
            } else {
                const double dpx = xp[3 * a] - x0[3 * a], dpy = xp[3 * a + 1] - x0[3 * a + 1], dpz = xp[3 * a + 2] - x0[3 * a + 2];
                const double dmx = x0[3 * a] - xm[3 * a], dmy = x0[3 * a + 1] - xm[3 * a + 1], dmz = x0[3 * a + 2] - xm[3 * a + 2];
                const double cc = 0.01 * sqrt(dpx * dpx + dpy * dpy + dpz * dpz) + (-1.0) * 0.01 * sqrt(dmx * dmx + dmy * dmy + dmz * dmz);
                const double gg[3] = {gx, gy, gz};
                double fperp[3], w[3];
                for (int c = 0; c < 3; ++c) { fperp[c] = gg[c] - gt * tau[c]; w[c] = cc - (cc * tau[c]) * tau[c]; }
                const double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                const double swf = (2.0 / 3.141592653589793) * atan(nw != 0.0 ? 1.0 : 0.0);
                const double wf = w[0] * fperp[0] + w[1] * fperp[1] + w[2] * fperp[2];
                for (int c = 0; c < 3; ++c) tot[c] = (-1.0) * fperp[c] - cc * tau[c] - swf * (w[c] - wf * fperp[c]);
            }
            for (int c = 0; c < 3; ++c) { f[3 * a + c] = tot[c]; if (tot[c] == tot[c]) { s2[0] += tot[c]; s2[1] += 1.0; } }
        }
        block_sum<2>(s2, red);
        const double mm = s2[0] / s2[1];
        if (mm > 10.0) {
            __syncthreads();
            for (int k = threadIdx.x; k < n; k += blockDim.x) f[k] = f[k] / mm;
        }
        return;
    }
    // ---- BNEB-type projection (variants 1, 2)
    double wp = 0.0, wm = 0.0;
    int mode;
    if (e0 < e1 && e1 < e2) { mode = 0; wp = 1.0; }
    else if (e0 > e1 && e1 > e2) { mode = 1; wm = 1.0; }
    else {
        mode = 2;
        const double d1 = fabs(e2 - e1), d0 = fabs(e1 - e0);
        const double mx = fmax(d1, d0), mn = fmin(d1, d0);
        const double a = mx / (mx + mn + 1e-8), bb = mn / (mx + mn + 1e-8);
        if (e0 < e2) { wp = a; wm = bb; } else { wp = bb; wm = a; }
    }
    bool c1 = false, c2 = false;
    if (variant == 2) {
        const bool near_max = is_local_max(E, I, i + 1) || is_local_max(E, I, i - 1) || (i - 1 == 0);
        c1 = ci_active && near_max && (i != 1 && i != I - 2);
        c2 = !c1 && ci_active && is_local_max(E, I, i) && (i != 1 || i != I - 2);
    }
    // whole-image distances
    double v4[4] = {0, 0, 0, 0};
    const bool has2 = (variant == 1) && (i > 1 && i < I - 2);
    const double *xmm = has2 ? X + off - 2 * (size_t)n : xm, *xpp = has2 ? X + off + 2 * (size_t)n : xp;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double d12 = xm[k] - x0[k], d23 = x0[k] - xp[k];
        v4[0] += d12 * d12; v4[1] += d23 * d23;
        if (has2) { const double d01 = xmm[k] - xm[k], d34 = xp[k] - xpp[k]; v4[2] += d01 * d01; v4[3] += d34 * d34; }
    }
    block_sum<4>(v4, red);
    const double n12 = sqrt(v4[0]), n23 = sqrt(v4[1]), n01 = sqrt(v4[2]), n34 = sqrt(v4[3]);
    double ki = 0.0, kim = 0.0;
    if (variant == 2) {
        double max_ene = E[0];
        for (int j = 1; j < I; ++j) max_ene = fmax(max_ene, E[j]);
        const double ref = (E[0] > E[I - 1]) ? E[I - 1] : E[0];
        double den = max_ene - ref;
        if (den < 1e-8) den = 1e-8;
        for (int t = 0; t < 2; ++t) {
            const int j = i - t;
            double kk = 0.0;
            if (j != 0 && j != I - 1) {
                const double alpha = (E[j] - ref) / den;
                if (E[j] < ref) kk = 0.0001;
                else if (E[j] > ref) kk = 0.005 * (1.0 - alpha) + alpha * 0.0001;
            }
            if (t == 0) ki = kk; else kim = kk;
        }
    }
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        double t[3] = {0.0, 0.0, 0.0};
        if (mode != 1) proj_atom(x0, xp, g, a, wp, t);
        if (mode != 0) { double t2[3] = {0.0, 0.0, 0.0}; proj_atom(xm, x0, g, a, wm, t2); t[0] += t2[0]; t[1] += t2[1]; t[2] += t2[2]; }
        for (int c = 0; c < 3; ++c) {
            const int k = 3 * a + c;
            double force = g[k] + t[c];
            if (c1) force *= 0.001;
            else if (c2) force = -2.0 * t[c];
            if (variant == 2) {
                const double fwd = xp[k] - x0[k], bwd = x0[k] - xm[k];
                const double uf = (n23 > 1e-8) ? fwd / n23 : 0.0, ub = (n12 > 1e-8) ? bwd / n12 : 0.0;
                force = force + (ki * uf - kim * ub);
                f[k] = -1.0 * force;
            } else {
                double spring = 0.0 * force;
                if (has2) {
                    const double d12 = xm[k] - x0[k], d23 = x0[k] - xp[k], d34 = xp[k] - xpp[k];
                    const double f1 = 0.05 * (n12 - n01) * d12 / (n12 + 1e-15);
                    const double f2 = 0.05 * (n23 - n12) * (-1.0 * d12 / (n12 + 1e-15) - d23 / (n23 + 1e-15));
                    const double f3 = 0.05 * (n34 - n23) * d34 / (n34 + 1e-15);
                    spring = f1 + f2 + f3;
                }
                f[k] = -1.0 * force - 1.0 * spring;
            }
        }
    }
}

}  // namespace mopt

int mopt_launch_neb_force_var(mopt_ctx *ctx, int variant, int I, int N, int first, int count, const double *d_X, const double *d_E,
                              const double *d_G, int ci_active, double *d_F) {
    using namespace mopt;
    mopt_scope sc(ctx, "neb_force", 1);
    hipLaunchKernelGGL(k_neb_force_var, dim3(count), dim3(128), 0, ctx->stream, variant, I, N, first, d_X, d_E, d_G, ci_active, d_F);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

int mopt_launch_bneb_force(mopt_ctx *ctx, int I, int N, int first, int count, const double *d_X, const double *d_E,
                           const double *d_G, int ci_active, double *d_F, double *d_tau, double *d_gamma) {
    using namespace mopt;
    mopt_scope sc(ctx, "neb_force", 1);
    hipLaunchKernelGGL(k_bneb_force, dim3(count), dim3(128), 0, ctx->stream, I, N, first, d_X, d_E, d_G, ci_active, d_F, d_tau, d_gamma);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
