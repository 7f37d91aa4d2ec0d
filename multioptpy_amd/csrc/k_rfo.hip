// RS-I-RFO step in the eigenbasis (batched over images).
//   k_vtg            gt_all[i] = v_i . g_proj           one wave per eigenvector (vector-major Vt: coalesced)
//   k_rfo_scalar     one wave per image: level-shift bookkeeping, trust radius, image function, 1e-6 filter,
//                    secular equation + alpha search (logic.h rs_step), predicted energy, state update
//   k_backtransform  step = - sum_i coef[i] v_i         (run() returns -move, rsirfo.py:490)
// Reference: multioptpy/Optimizer/rsirfo.py:285-490, 492-657, 924-1313.
#include "common.hpp"
#include "logic.h"
#include "rfo_scalar_body.hpp"

namespace mopt {

// (gfb: the vector the identity fallback takes -- the projected gradient itself when gp is Q^T g and Vt holds the eigenvectors of the
// tridiagonal form, see pipeline.hip; null: gp)
__global__ void __launch_bounds__(256) k_vtg(int n, const double *__restrict__ Vt, const double *__restrict__ gp,
                                             const int *__restrict__ einfo, double *__restrict__ gt_all, const double *__restrict__ gfb) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wid;
    if (i >= n) return;
    const double *g = gp + (size_t)b * n;
    if (einfo[2 * b + 1] & 1) {           // eigensolver produced non-finite values: identity fallback (rsirfo.py:365-369)
        if (lane == 0) gt_all[(size_t)b * n + i] = gfb ? gfb[(size_t)b * n + i] : g[i];
        return;
    }
    const double *v = Vt + ((size_t)b * n + i) * n;
    double d = 0.0;
    for (int j = lane; j < n; j += WAVE) d += v[j] * g[j];
    d = wave_sum(d);
    if (lane == 0) gt_all[(size_t)b * n + i] = d;
}

// Three waves per image (rfo_scalar_body.hpp).  dynamic LDS: 15 * n doubles + 3 * n ints
__global__ void __launch_bounds__(192) k_rfo_scalar(int n, const mopt_rsirfo_params *__restrict__ params,
                                                   mopt_rsirfo_state *__restrict__ states, const double *__restrict__ W,
                                                   const int *__restrict__ perm, const int *__restrict__ einfo,
                                                   const double *__restrict__ gt_all_, const double *__restrict__ Bg,
                                                   const double *__restrict__ Be, const UpdateCoef *__restrict__ coef,
                                                   double *__restrict__ coef_out_, int dbg, const int *__restrict__ mf, int nmf) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    rfo_scalar_body<0>(n, (int)blockIdx.x, params, states, W, perm, einfo, gt_all_, Bg, Be, coef, coef_out_, dbg, sm, mf, nmf);
}

// out[b][j] = - sum_i coef[b][i] Vt[b][i][j]     grid (ceil(n/64), batch), block 1024: 16 waves share the rows of a
// 64-column strip (four independent loads in flight per wave: the kernel is a latency chain of row reads, not a bandwidth user)
__global__ void __launch_bounds__(1024) k_backtransform(int n, const double *__restrict__ Vt, const double *__restrict__ coef,
                                                        const int *__restrict__ einfo, double *__restrict__ out) {
    __shared__ double part[16][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const double *c = coef + (size_t)b * n;
    if (einfo[2 * b + 1] & 1) {
        if (wid == 0 && j < n) out[(size_t)b * n + j] = -c[j];
        return;
    }
    const double *V = Vt + (size_t)b * n * n;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (j < n) {
        int i = wid;
        for (; i + 48 < n; i += 64) {
            const double c0 = c[i], c1 = c[i + 16], c2 = c[i + 32], c3 = c[i + 48];
            const double v0 = V[(size_t)i * n + j], v1 = V[(size_t)(i + 16) * n + j], v2 = V[(size_t)(i + 32) * n + j], v3 = V[(size_t)(i + 48) * n + j];
            a0 = fma(c0, c0 != 0.0 ? v0 : 0.0, a0); a1 = fma(c1, c1 != 0.0 ? v1 : 0.0, a1);      // (a zero coefficient drops its vector whatever it holds)
            a2 = fma(c2, c2 != 0.0 ? v2 : 0.0, a2); a3 = fma(c3, c3 != 0.0 ? v3 : 0.0, a3);
        }
        for (; i < n; i += 16) { const double ci = c[i]; a0 = fma(ci, ci != 0.0 ? V[(size_t)i * n + j] : 0.0, a0); }
    }
    part[wid][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wid == 0 && j < n) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][lane];
        out[(size_t)b * n + j] = -t;
    }
}

}  // namespace mopt


int mopt_launch_rfo_scalar(mopt_ctx *ctx, hipStream_t st, int n, int batch, const mopt_rsirfo_params *d_params, mopt_rsirfo_state *d_states,
                           const double *d_W, const int *d_perm, const int *d_einfo, const double *d_gt_all, const double *d_Bg, const double *d_Be,
                           const mopt::UpdateCoef *d_coef, double *d_coefout, const int *d_mf, int nmf) {
    using namespace mopt;
    static const int dbg = getenv("MOPT_RFO_DBG") ? 1 : 0;
    const size_t lds = (size_t)15 * n * sizeof(double) + (size_t)3 * n * sizeof(int) + 16;
    if (lds > 48000) MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_rfo_scalar, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
    hipLaunchKernelGGL(k_rfo_scalar, dim3(batch), dim3(192), lds, st, n, d_params, d_states, d_W, d_perm, d_einfo, d_gt_all, d_Bg, d_Be, d_coef, d_coefout, dbg, d_mf, nmf);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

int mopt_launch_rfo(mopt_ctx *ctx, int n, int batch, const mopt_rsirfo_params *d_params, mopt_rsirfo_state *d_states,
                    const double *d_W, const int *d_perm, const int *d_einfo, const double *d_Vt, const double *d_gp,
                    const double *d_Bg, const double *d_Be, const mopt::UpdateCoef *d_coef, double *d_gt_all,
                    double *d_coefout, double *d_step, const int *d_mf, int nmf) {
    using namespace mopt;
    mopt_scope sc(ctx, "rfo", 3);
    hipLaunchKernelGGL(k_vtg, dim3((n + 3) / 4, batch), dim3(256), 0, ctx->stream, n, d_Vt, d_gp, d_einfo, d_gt_all, (const double *)nullptr);
    { const int rc = mopt_launch_rfo_scalar(ctx, ctx->stream, n, batch, d_params, d_states, d_W, d_perm, d_einfo, d_gt_all, d_Bg, d_Be, d_coef, d_coefout, d_mf, nmf); if (rc != MOPT_OK) return rc; }
    hipLaunchKernelGGL(k_backtransform, dim3((n + 63) / 64, batch), dim3(1024), 0, ctx->stream, n, d_Vt, d_coefout, d_einfo, d_step);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

int mopt_launch_vtg(mopt_ctx *ctx, int n, int batch, const double *d_Vt, const double *d_vec, const int *d_einfo, double *d_out, const double *d_fallback,
                    hipStream_t st) {
    using namespace mopt;
    hipLaunchKernelGGL(k_vtg, dim3((n + 3) / 4, batch), dim3(256), 0, st ? st : ctx->stream, n, d_Vt, d_vec, d_einfo, d_out, d_fallback);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

int mopt_launch_backtransform(mopt_ctx *ctx, int n, int batch, const double *d_Vt, const double *d_coef, const int *d_einfo, double *d_out) {
    using namespace mopt;
    hipLaunchKernelGGL(k_backtransform, dim3((n + 63) / 64, batch), dim3(1024), 0, ctx->stream, n, d_Vt, d_coef, d_einfo, d_out);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
