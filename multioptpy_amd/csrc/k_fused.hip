// Band step, fused form of everything between the NEB force and the eigen-solver: the Ayala term, the
// quasi-Newton update and the TR/rot projection are all low-rank corrections of the resident Hessian H0, so H crosses the memory side
// four times (one read for the products, one read + two writes for the result) instead of ten, in three launches instead of eight:
//
//   unfused chain (band.hip, k_update.hip, k_project.hip)                      fused (this file)
//   k_rank1          H += gamma t t^T                       (r, w)
//   k_diff2          s = x - x_prev, y = g - g_prev
//   k_matvec         u = H s                                (r)               k_fused_mv7    u0 = H0 s, W0 = H0 Q^T              (r)
//   k_update_coef    dots, rule coefficients, y', a                           k_fused_coef   s, y, u = u0 + gamma t (t.s), dots,
//   k_apply_update   H = sym(H + dH)                        (r, w)                           coefficients, y', a, W = W0 + low rank,
//   k_sym_add        A = sym(H + bias)                      (r, w)                           W' = W - 1/2 Q^T sym(Q W)
//   k_proj_w         W = A Q^T                              (r)               k_fused_final  H = (H0 + gamma t t^T) + dH,          (r, w, w)
//   k_proj_fix       W' = W - 1/2 Q^T (Q W)                                                  A = H - Q^T W'^T - W' Q
//   k_proj_apply     A -= Q^T W'^T + W' Q                   (r, w)
//
// Reference: rfo_neb.py:63-68,147-156 (Ayala), rsirfo.py:1316-1372 + hessian_update.py (update), calc_tools.py:249-316 (projection).
// Precondition (checked by the caller, band.hip): H0 is BITWISE symmetric, there is no bias Hessian, no in-place copy is asked for and
// no image uses the flowchart rule (it needs H y).  Then H0 + gamma t t^T + dH is symmetric term by term (every product below is
// written commutatively), the reference's 0.5 (X + X^T) returns X itself and the result is again bitwise symmetric -- the fused form
// reproduces the unfused one up to the association of the rank-1 term and of u, W (differences of a few 1e-16 relative).
#include "pipeline.h"

namespace mopt {

// u0[b][i] = sum_j H0[b][i][j] (x[b][j] - px[b][j]);  W0[b][i][k] = sum_j H0[b][i][j] Q[b][k][j].  One wave per row.
__global__ void __launch_bounds__(256) k_fused_mv7(int n, const double *__restrict__ H, const double *__restrict__ x, const double *__restrict__ px,
                                                   const double *__restrict__ Q, double *__restrict__ u0, double *__restrict__ W0) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wid;
    if (r >= n) return;
    const double *row = H + ((size_t)b * n + r) * n;
    const double *xb = x + (size_t)b * n, *pb = px + (size_t)b * n, *q = Q + (size_t)b * 6 * n;
    double au = 0.0, a[6] = {0, 0, 0, 0, 0, 0};
    for (int j = lane; j < n; j += WAVE) {
        const double h = row[j];
        au = fma(h, xb[j] - pb[j], au);
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] = fma(h, q[k * n + j], a[k]);
    }
    // seven sums in two four-value reductions: 16-lane row 0, 1, 2, 3 of the result holds argument 0, 2, 1, 3
    const double f0 = wave_sum4(au, a[1], a[0], a[2]);       // rows: u0, w0, w1, w2
    const double f1 = wave_sum4(a[3], a[5], a[4], 0.0);      // rows: w3, w4, w5, -
    if ((lane & 15) == 0) {
        const int rw = lane >> 4;
        if (rw == 0) u0[(size_t)b * n + r] = f0; else W0[((size_t)b * n + r) * 6 + (rw - 1)] = f0;
        if (rw < 3) W0[((size_t)b * n + r) * 6 + 3 + rw] = f1;
    }
}

// One workgroup per image, blockDim = n rounded up to whole waves (<= 512: the 33 + 30 and 36 + 30 running sums need more than the 128
// registers of a 1024-thread workgroup): every vector of the image is staged in LDS by ONE round of
// global loads (all in flight together); the four dependent phases after that -- dots, coefficients, W, W' -- touch LDS only and the
// results leave in one round of stores.  Dynamic LDS: 18 n doubles (t, s, y, u, y', a: 6 n; Q: 6 n; W: 6 n -- W0 is taken into W, x - px etc.
// are formed on the way in) + the reduction scratch.
__global__ void __launch_bounds__(512) k_fused_coef(int n, int guard_sy, int first, const int *__restrict__ rules, int upd_all,
                                                     const double *__restrict__ x_, const double *__restrict__ px_, const double *__restrict__ g_,
                                                     const double *__restrict__ pg_, const double *__restrict__ gamma_, const double *__restrict__ tau_,
                                                     const double *__restrict__ Q_, const double *__restrict__ u0_, const double *__restrict__ W0_,
                                                     double *__restrict__ s_, double *__restrict__ y_, double *__restrict__ u_, double *__restrict__ yd_,
                                                     double *__restrict__ A_, UpdateCoef *__restrict__ coef, double *__restrict__ info, double *__restrict__ W_) {
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    __shared__ double red[36 * 16];
    __shared__ double M[6][6];
    __shared__ UpdateCoef sc;
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t o = (size_t)b * n;
    double *lt = fsm, *ls = lt + n, *ly = ls + n, *lu = ly + n, *lyd = lu + n, *la = lyd + n, *lq = la + n, *lw = lq + 6 * n;
    const double gam = gamma_[b];
    const int rule = rules[b];
    const bool upd = upd_all != 0;
    {
        const double *x = x_ + o, *px = px_ + o, *g = g_ + o, *pg = pg_ + o, *u0 = u0_ + o, *q = Q_ + (size_t)b * 6 * n;
        const double *t = tau_ + (size_t)(first + b) * n, *w0 = W0_ + (size_t)b * n * 6;
        for (int i = tid; i < n; i += blockDim.x) {
            lt[i] = t[i];
            ls[i] = upd ? x[i] - px[i] : 0.0;
            ly[i] = upd ? g[i] - pg[i] : 0.0;
            lu[i] = upd ? u0[i] : 0.0;
        }
        for (int e = tid; e < 6 * n; e += blockDim.x) { lq[e] = q[e]; lw[e] = w0[e]; }
    }
    __syncthreads();
    double qt[6], qy[6], qu[6], qa[6], qs[6];
    UpdateCoef c; c.c_yy = c.c_uu = c.c_AA = c.c_As = c.c_ss = 0.0; c.applied = 0; c.block_sym = 0; c.mix = 0.0; c.branch = 0;
    if (upd) {
        // ---- ss, sy, yy, s.u0, t.s
        double v[5] = {0, 0, 0, 0, 0};
        for (int i = tid; i < n; i += blockDim.x) {
            const double si = ls[i], yi = ly[i];
            v[0] += si * si; v[1] += si * yi; v[2] += yi * yi; v[3] += si * lu[i]; v[4] += lt[i] * si;
        }
        block_sum_x4<5>(v, red);
        const double ts = v[4];
        UpdateDots d;
        d.ss = v[0]; d.sy = v[1]; d.yy = v[2]; d.su = v[3] + gam * ts * ts;      // s.(H0 + gamma t t^T) s
        double theta = 1.0;
        bool damp = false;
        if (rule_is_dd(rule)) {
            damp = d.sy < 0.2 * d.ss;
            theta = dd_theta(d.ss, d.sy, rule_is_block(rule) ? 1e-12 : 1e-10);
        }
        // ---- u, y', a; s.y', a.a, a.s and the 30 products of the six basis vectors with t, y', u, a, s
        double z[33];
#pragma unroll
        for (int e = 0; e < 33; ++e) z[e] = 0.0;
        const double gts = gam * ts;
        for (int i = tid; i < n; i += blockDim.x) {
            const double si = ls[i], yi = ly[i], ti = lt[i];
            const double ui = fma(gts, ti, lu[i]);
            const double ydi = damp ? theta * yi + (1.0 - theta) * si : yi;
            const double ai = ydi - ui;
            lu[i] = ui; lyd[i] = ydi; la[i] = ai;
            z[0] += si * ydi; z[1] += ai * ai; z[2] += ai * si;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double qk = lq[k * n + i];
                z[3 + k] += qk * ti; z[9 + k] += qk * ydi; z[15 + k] += qk * ui; z[21 + k] += qk * ai; z[27 + k] += qk * si;
            }
        }
        block_sum_x4<33>(z, red);
        d.syd = z[0]; d.AA = z[1]; d.As = z[2]; d.zz = 0.0; d.zs = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { qt[k] = z[3 + k]; qy[k] = z[9 + k]; qu[k] = z[15 + k]; qa[k] = z[21 + k]; qs[k] = z[27 + k]; }
        if (tid == 0) {
            c = update_coefficients(rule, d, guard_sy != 0);
            sc = c;
            coef[b] = c;
            if (info) {
                double *oo = info + (size_t)b * 8;
                oo[0] = c.applied; oo[1] = d.sy; oo[2] = d.ss; oo[3] = d.yy; oo[4] = d.su; oo[5] = c.mix; oo[6] = c.branch; oo[7] = 0.0;
            }
        }
        __syncthreads();
        c = sc;
        // (the vectors of the update leave now: k_fused_final reads them)
        double *s = s_ + o, *y = y_ + o, *u = u_ + o, *yd = yd_ + o, *av = A_ + o;
        for (int i = tid; i < n; i += blockDim.x) { s[i] = ls[i]; y[i] = ly[i]; u[i] = lu[i]; yd[i] = lyd[i]; av[i] = la[i]; }
    } else {
        double z[6] = {0, 0, 0, 0, 0, 0};
        for (int i = tid; i < n; i += blockDim.x) {
            const double ti = lt[i];
#pragma unroll
            for (int k = 0; k < 6; ++k) z[k] += lq[k * n + i] * ti;
        }
        block_sum_x4<6>(z, red);
#pragma unroll
        for (int k = 0; k < 6; ++k) { qt[k] = z[k]; qy[k] = qu[k] = qa[k] = qs[k] = 0.0; }
        if (tid == 0) {
            coef[b] = c;
            if (info) for (int k = 0; k < 8; ++k) info[(size_t)b * 8 + k] = 0.0;
        }
    }
    // ---- W = (H0 + gamma t t^T + dH) Q^T, then M = Q W
    const bool ap = upd && c.applied;
    double m[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) m[e] = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
        const double ti = lt[i];
        double ydi = 0.0, ui = 0.0, ai = 0.0, si = 0.0;
        if (ap) { ydi = lyd[i]; ui = lu[i]; ai = la[i]; si = ls[i]; }
        double wl[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            double wv = lw[i * 6 + l] + gam * (ti * qt[l]);
            if (ap) wv += c.c_yy * ydi * qy[l] + c.c_uu * ui * qu[l] + c.c_AA * ai * qa[l] + c.c_As * (ai * qs[l] + si * qa[l]) + c.c_ss * si * qs[l];
            wl[l] = wv;
            lw[i * 6 + l] = wv;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double qk = lq[k * n + i];
#pragma unroll
            for (int l = 0; l < 6; ++l) m[k * 6 + l] += qk * wl[l];
        }
    }
    block_sum_x4<36>(m, red);
    if (tid < 36) {                                          // (from the per-wave partial sums: m[tid] would push the array to scratch memory)
        double a = 0.0;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) a += red[tid * 16 + wv];
        M[tid / 6][tid % 6] = a;
    }
    __syncthreads();
    // ---- W' = W - 1/2 Q^T sym(M)    (k_proj_fix)
    double *w = W_ + (size_t)b * n * 6;
    for (int e = tid; e < 6 * n; e += blockDim.x) {
        const int i = e / 6, l = e - 6 * i;
        double tt = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) tt += lq[k * n + i] * 0.5 * (M[k][l] + M[l][k]);
        w[e] = lw[e] - 0.5 * tt;
    }
}

// a x b rounded on its own: the empty asm statement hides the product from the optimiser, which would otherwise fold it into an FMA with
// the following addition (HIP compiles with -ffp-contract=fast; __dmul_rn is a plain multiplication there)
__device__ __forceinline__ double add_r(double a, double b) { return a + b; }
__device__ __forceinline__ double mul_r(double a, double b) { double p = a * b; asm volatile("" : "+v"(p)); return p; }

// 32 x 32 tile, every element on its own: H_ij = (H0_ij + gamma (t_i t_j)) + dH_ij, A_ij = H_ij - sum_k (q_ki W'_jk + W'_ik q_kj)
__global__ void __launch_bounds__(256) k_fused_final(int n, int first, double *__restrict__ H, double *__restrict__ A, const double *__restrict__ gamma_,
                                                     const double *__restrict__ tau_, const UpdateCoef *__restrict__ coef,
                                                     const double *__restrict__ yd_, const double *__restrict__ u_, const double *__restrict__ A_v,
                                                     const double *__restrict__ s_, const double *__restrict__ Q_, const double *__restrict__ W_,
                                                     int *__restrict__ fail_, unsigned long long *__restrict__ xch_, size_t xch_stride) {
    // (what the eigen-solver needs zeroed before its persistent kernel starts -- failure flags, exchange area at the head of each slot of its
    // eigenvector buffer -- is zeroed here instead of by two fill kernels in front of it; EighRfoHook::pre_cleared, pipeline.h)
    if (blockIdx.x == 0 && fail_) {
        if (threadIdx.x == 0) fail_[blockIdx.y] = 0;
        unsigned long long *xp = xch_ + (size_t)blockIdx.y * xch_stride;
        for (int i = threadIdx.x; i < 8 * n + 64; i += 256) xp[i] = 0ull;
    }
    __shared__ double vr[17][32], vc[17][32];                // t, y', u, a, s, q[6], w'[6] of the tile's rows and of its columns
    const int T = (n + 31) / 32;
    const int bi = blockIdx.x / T, bj = blockIdx.x % T, b = blockIdx.y;
    const UpdateCoef c = coef[b];
    const double gam = gamma_[b];
    const size_t o = (size_t)b * n;
    const double *q = Q_ + (size_t)b * 6 * n, *w = W_ + (size_t)b * n * 6;
    const double *vecs[5] = {tau_ + (size_t)(first + b) * n, yd_ + o, u_ + o, A_v + o, s_ + o};
    for (int e = threadIdx.x; e < 17 * 32; e += 256) {
        const int k = e >> 5, tt = e & 31;
        const int i = bi * 32 + tt, j = bj * 32 + tt;
        double a = 0.0, bb = 0.0;
        if (k < 5) {
            if (k == 0 || c.applied) { if (i < n) a = vecs[k][i]; if (j < n) bb = vecs[k][j]; }
        } else if (k < 11) {
            if (i < n) a = q[(k - 5) * n + i];
            if (j < n) bb = q[(k - 5) * n + j];
        } else {
            if (i < n) a = w[i * 6 + (k - 11)];
            if (j < n) bb = w[j * 6 + (k - 11)];
        }
        vr[k][tt] = a; vc[k][tt] = bb;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double *Hb = H + (size_t)b * n * n, *Ab = A + (size_t)b * n * n;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int r = ty + 8 * kk, i = bi * 32 + r, j = bj * 32 + tx;
        if (i < n && j < n) {
            // every product and sum is rounded on its own (no contraction into FMAs): element (j, i) then runs through the same operations on
            // the same numbers -- a x b and b x a, p + q and q + p -- and the result is bitwise symmetric
            double h = add_r(Hb[(size_t)i * n + j], mul_r(gam, mul_r(vr[0][r], vc[0][tx])));
            if (c.applied) {
                const double yi = vr[1][r], ui = vr[2][r], ai = vr[3][r], si = vr[4][r];
                const double yj = vc[1][tx], uj = vc[2][tx], aj = vc[3][tx], sj = vc[4][tx];
                double dl = mul_r(c.c_yy, mul_r(yi, yj));
                dl = add_r(dl, mul_r(c.c_uu, mul_r(ui, uj)));
                dl = add_r(dl, mul_r(c.c_AA, mul_r(ai, aj)));
                dl = add_r(dl, mul_r(c.c_As, add_r(mul_r(ai, sj), mul_r(si, aj))));
                dl = add_r(dl, mul_r(c.c_ss, mul_r(si, sj)));
                h = add_r(h, dl);
            }
            double p = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) p = add_r(p, add_r(mul_r(vr[5 + k][r], vc[11 + k][tx]), mul_r(vr[11 + k][r], vc[5 + k][tx])));
            Hb[(size_t)i * n + j] = h;
            Ab[(size_t)i * n + j] = h - p;
        }
    }
}

}  // namespace mopt

// Fused update + projection of the band path.  d_tau: the band's tangents [I][n]; d_gamma: [batch].  On return wk.H holds the updated
// Hessians and wk.A the projected matrices for the eigen-solver; wk.s, y, u, yd, Av, coef, info, W6 are filled as the unfused kernels fill them.
int mopt_launch_fused_update_project(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, const mopt::FusedBand &fb) {
    using namespace mopt;
    {
        mopt_scope sc(ctx, "hess_update", 2);
        MOPT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join3, 0));      // the TR/rot basis (side stream, forked by the band ahead of the force kernel)
        hipLaunchKernelGGL(k_fused_mv7, dim3((n + 3) / 4, batch), dim3(256), 0, ctx->stream, n, wk.H, wk.x, wk.px, wk.Q, wk.Hy, wk.Gt);
        const size_t lds = (size_t)18 * n * 8;
        if (lds > 60000) MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_fused_coef, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nthr = ((n + 63) / 64) * 64;
        if (nthr > 512) nthr = 512;
        hipLaunchKernelGGL(k_fused_coef, dim3(batch), dim3(nthr), lds, ctx->stream, n, 1, fb.first, wk.rules, fb.upd_all, wk.x, wk.px, wk.g, wk.pg,
                           fb.gamma, fb.tau, wk.Q, wk.Hy, wk.Gt, wk.s, wk.y, wk.u, wk.yd, wk.Av, wk.coef, wk.info, wk.W6);
    }
    {
        mopt_scope sc(ctx, "project", 1);
        const int T = (n + 31) / 32;
        hipLaunchKernelGGL(k_fused_final, dim3(T * T, batch), dim3(256), 0, ctx->stream, n, fb.first, wk.H, wk.A, fb.gamma, fb.tau, wk.coef, wk.yd, wk.u,
                           wk.Av, wk.s, wk.Q, wk.W6, fb.clear_for_eigh ? wk.tri.fail : (int *)nullptr, (unsigned long long *)wk.Vt, (size_t)n * n);
    }
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
