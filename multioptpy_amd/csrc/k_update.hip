// Quasi-Newton Hessian update kernels (batched over images).
//   k_diff2          s = x - x_prev, y = g - g_prev
//   k_matvec         u = H s (and Hy = H y for the flowchart rule): one wave per row, coalesced row reads
//   k_update_coef    dots -> rule coefficients (logic.h update_coefficients), damped y', A = y' - u
//   k_apply_update   H += dH (rank <= 5 symmetric), then 0.5 (X + X^T): 32x32 tile pairs through LDS
// Reference: multioptpy/Optimizer/rsirfo.py:1316-1372, hessian_update.py, block_hessian_update.py.
#include "common.hpp"
#include "logic.h"

namespace mopt {

__global__ void k_diff2(int n, const double *__restrict__ x, const double *__restrict__ px,
                        const double *__restrict__ g, const double *__restrict__ pg,
                        double *__restrict__ s, double *__restrict__ y) {
    const size_t b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        s[b * n + i] = x[b * n + i] - px[b * n + i];
        y[b * n + i] = g[b * n + i] - pg[b * n + i];
    }
}

// out1[b][r] = sum_j H[b][r][j] v1[b][j]   (out2 with v2 when NV == 2)
template <int NV>
__global__ void __launch_bounds__(256) k_matvec(int n, const double *__restrict__ H, const double *__restrict__ v1,
                                                const double *__restrict__ v2, double *__restrict__ out1,
                                                double *__restrict__ out2) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wid;
    if (r >= n) return;
    const double *row = H + ((size_t)b * n + r) * n;
    const double *a1 = v1 + (size_t)b * n;
    const double *a2 = (NV == 2) ? v2 + (size_t)b * n : nullptr;
    double s1 = 0.0, s2 = 0.0;
    for (int j = lane; j < n; j += WAVE) {
        double h = row[j];
        s1 += h * a1[j];
        if (NV == 2) s2 += h * a2[j];
    }
    s1 = wave_sum(s1);
    if (NV == 2) s2 = wave_sum(s2);
    if (lane == 0) {
        out1[(size_t)b * n + r] = s1;
        if (NV == 2) out2[(size_t)b * n + r] = s2;
    }
}

// One workgroup per image.  rule[b] per image (band: end images and interior images differ).
__global__ void __launch_bounds__(256) k_update_coef(int n, int guard_sy, const int *__restrict__ rules, const int *__restrict__ do_update,
                                                     const double *__restrict__ s_, const double *__restrict__ y_,
                                                     const double *__restrict__ u_, const double *__restrict__ Hy_,
                                                     double *__restrict__ yd_, double *__restrict__ A_,
                                                     UpdateCoef *__restrict__ coef, double *__restrict__ info) {
    __shared__ double red[5 * 16];
    const int b = blockIdx.x;
    const int rule = rules[b];
    const double *s = s_ + (size_t)b * n, *y = y_ + (size_t)b * n, *u = u_ + (size_t)b * n;
    const double *Hy = Hy_ ? Hy_ + (size_t)b * n : nullptr;
    double *yd = yd_ + (size_t)b * n, *A = A_ + (size_t)b * n;
    if (do_update && !do_update[b]) {
        if (threadIdx.x == 0) {
            UpdateCoef c; c.c_yy = c.c_uu = c.c_AA = c.c_As = c.c_ss = 0.0; c.applied = 0;
            c.block_sym = 0; c.mix = 0.0; c.branch = 0;
            coef[b] = c;
            if (info) for (int k = 0; k < 8; ++k) info[b * 8 + k] = 0.0;
        }
        return;
    }
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double si = s[i], yi = y[i], ui = u[i];
        v[0] += si * si; v[1] += si * yi; v[2] += yi * yi; v[3] += si * ui;
    }
    block_sum<5>(v, red);
    UpdateDots d;
    d.ss = v[0]; d.sy = v[1]; d.yy = v[2]; d.su = v[3];
    double theta = 1.0;
    bool damp = false;
    if (rule_is_dd(rule)) {
        damp = d.sy < 0.2 * d.ss;
        theta = dd_theta(d.ss, d.sy, rule_is_block(rule) ? 1e-12 : 1e-10);
    }
    double w[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double si = s[i], yi = y[i], ui = u[i];
        double ydi = damp ? theta * yi + (1.0 - theta) * si : yi;
        double ai = ydi - ui;
        yd[i] = ydi; A[i] = ai;
        w[0] += si * ydi; w[1] += ai * ai; w[2] += ai * si;
        if (Hy) { double zi = yi - Hy[i]; w[3] += zi * zi; w[4] += zi * si; }
    }
    block_sum<5>(w, red);
    d.syd = w[0]; d.AA = w[1]; d.As = w[2]; d.zz = w[3]; d.zs = w[4];
    if (threadIdx.x == 0) {
        UpdateCoef c = update_coefficients(rule, d, guard_sy != 0);
        coef[b] = c;
        if (info) {
            double *o = info + (size_t)b * 8;
            o[0] = c.applied; o[1] = d.sy; o[2] = d.ss; o[3] = d.yy; o[4] = d.su; o[5] = c.mix; o[6] = c.branch; o[7] = 0.0;
        }
    }
}

// Tile pair (bi <= bj) of 32x32.  Hin may alias Hsym (band path, in place).  Hinpl may be null.
// copy_if_skipped: when the update was not applied and outputs differ from the input, copy H through unchanged
// (the reference leaves self.hessian untouched, not even symmetrised, when it skips: rsirfo.py:1326-1335).
__global__ void __launch_bounds__(256) k_apply_update(int n, const double *Hin, double *Hinpl, double *Hsym,
                                                      const UpdateCoef *__restrict__ coef, const double *__restrict__ yd_,
                                                      const double *__restrict__ u_, const double *__restrict__ A_,
                                                      const double *__restrict__ s_) {
    __shared__ double L[32][33], S[32][33], T2[32][33];
    __shared__ double vr[4][32], vc[4][32];
    const int T = (n + 31) / 32;
    const int bi = blockIdx.x / T, bj = blockIdx.x % T;
    if (bi > bj) return;
    const int b = blockIdx.y;
    const UpdateCoef c = coef[b];
    const double *H = Hin + (size_t)b * n * n;
    double *Hp = Hinpl ? Hinpl + (size_t)b * n * n : nullptr;
    double *Hs = Hsym + (size_t)b * n * n;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    if (!c.applied) {
        if (Hs == H && (Hp == nullptr || Hp == H)) return;
        for (int k = 0; k < 4; ++k) {
            int r = ty + 8 * k;
            int i = bi * 32 + r, j = bj * 32 + tx;
            if (i < n && j < n) { double h = H[(size_t)i * n + j]; if (Hs != H) Hs[(size_t)i * n + j] = h; if (Hp && Hp != H) Hp[(size_t)i * n + j] = h; }
            if (bi != bj) {
                int i2 = bj * 32 + r, j2 = bi * 32 + tx;
                if (i2 < n && j2 < n) { double h = H[(size_t)i2 * n + j2]; if (Hs != H) Hs[(size_t)i2 * n + j2] = h; if (Hp && Hp != H) Hp[(size_t)i2 * n + j2] = h; }
            }
        }
        return;
    }
    const double *vecs[4] = {yd_ + (size_t)b * n, u_ + (size_t)b * n, A_ + (size_t)b * n, s_ + (size_t)b * n};
    if (threadIdx.x < 128) {
        int k = threadIdx.x >> 5, t = threadIdx.x & 31;
        int i = bi * 32 + t, j = bj * 32 + t;
        vr[k][t] = (i < n) ? vecs[k][i] : 0.0;
        vc[k][t] = (j < n) ? vecs[k][j] : 0.0;
    }
    // lower tile (rows of block bj, cols of block bi) -> LDS
    for (int k = 0; k < 4; ++k) {
        int r = ty + 8 * k;
        int i2 = bj * 32 + r, j2 = bi * 32 + tx;
        L[r][tx] = (i2 < n && j2 < n) ? H[(size_t)i2 * n + j2] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        int r = ty + 8 * k;
        int i = bi * 32 + r, j = bj * 32 + tx;
        double hij = (i < n && j < n) ? H[(size_t)i * n + j] : 0.0;
        double hji = L[tx][r];
        double yi = vr[0][r], ui = vr[1][r], ai = vr[2][r], si = vr[3][r];
        double yj = vc[0][tx], uj = vc[1][tx], aj = vc[2][tx], sj = vc[3][tx];
        double dlt = c.c_yy * yi * yj + c.c_uu * ui * uj + c.c_AA * ai * aj + c.c_As * (ai * sj + si * aj) + c.c_ss * si * sj;
        double t1 = hij + dlt, t2 = hji + dlt;
        double sym = 0.5 * (t1 + t2);
        if (c.block_sym) { t1 = sym; t2 = sym; }
        S[r][tx] = sym;
        T2[r][tx] = t2;
        if (i < n && j < n) {
            Hs[(size_t)i * n + j] = sym;
            if (Hp) Hp[(size_t)i * n + j] = t1;
        }
    }
    if (bi != bj) {
        __syncthreads();
        for (int k = 0; k < 4; ++k) {
            int r = ty + 8 * k;
            int i2 = bj * 32 + r, j2 = bi * 32 + tx;
            if (i2 < n && j2 < n) {
                Hs[(size_t)i2 * n + j2] = S[tx][r];
                if (Hp) Hp[(size_t)i2 * n + j2] = T2[tx][r];
            }
        }
    }
}

}  // namespace mopt

// ---- host launchers ---------------------------------------------------------------------
int mopt_launch_hess_update(mopt_ctx *ctx, int n, int batch, const int *d_rules, const int *d_do_update, bool any_flowchart,
                            const double *d_x, const double *d_px, const double *d_g, const double *d_pg,
                            double *d_s, double *d_y, double *d_u, double *d_Hy, double *d_yd, double *d_A,
                            mopt::UpdateCoef *d_coef, double *d_info, const double *d_H, double *d_Hinpl, double *d_Hsym,
                            bool have_sy, bool guard_sy) {
    using namespace mopt;
    mopt_scope sc(ctx, "hess_update", 4);
    if (!have_sy) {
        dim3 g((n + 255) / 256, batch);
        hipLaunchKernelGGL(k_diff2, g, dim3(256), 0, ctx->stream, n, d_x, d_px, d_g, d_pg, d_s, d_y);
    }
    dim3 gm((n + 3) / 4, batch);
    if (any_flowchart)
        hipLaunchKernelGGL(k_matvec<2>, gm, dim3(256), 0, ctx->stream, n, d_H, d_s, d_y, d_u, d_Hy);
    else
        hipLaunchKernelGGL(k_matvec<1>, gm, dim3(256), 0, ctx->stream, n, d_H, d_s, (const double *)nullptr, d_u, (double *)nullptr);
    hipLaunchKernelGGL(k_update_coef, dim3(batch), dim3(256), 0, ctx->stream, n, guard_sy ? 1 : 0, d_rules, d_do_update, d_s, d_y, d_u,
                       any_flowchart ? d_Hy : (const double *)nullptr, d_yd, d_A, d_coef, d_info);
    const int T = (n + 31) / 32;
    hipLaunchKernelGGL(k_apply_update, dim3(T * T, batch), dim3(256), 0, ctx->stream, n, d_H, d_Hinpl, d_Hsym, d_coef,
                       d_yd, d_u, d_A, d_s);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

int mopt_launch_matvec(mopt_ctx *ctx, int n, int batch, const double *d_H, const double *d_v, double *d_out) {
    using namespace mopt;
    hipLaunchKernelGGL(k_matvec<1>, dim3((n + 3) / 4, batch), dim3(256), 0, ctx->stream, n, d_H, d_v, (const double *)nullptr, d_out, (double *)nullptr);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
