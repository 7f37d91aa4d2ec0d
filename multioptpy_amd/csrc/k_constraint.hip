// Constraint projection of the Hessian and the constrained RS-I-RFO step.
//   mopt_constraint_project_hess   ProjectOutConstrain.calc_project_out_hess (multioptpy/Constraint/constraint_condition.py:657-754):
//       H_eff = H - P H - H P + P H P + k_hard P_hard + k_soft P_soft,   P = P_hard = sum q q^T over the orthonormal basis of the
//       current constraint rows, P_soft over the orthogonalised residuals of the stored bases.  The reference forms three dense n^3
//       products; with c = rank(P) << n the same matrix is H minus a rank-2c correction:  W1 = H Q, W2 = H^T Q, M = Q^T W1 (c x c).
//   mopt_crsirfo_run               CRSIRFO.run (multioptpy/Optimizer/crsirfo.py:48-182) behind the SHAKE-like adjustment and the
//       null-space basis U (both O(c n) / one LAPACK call on c x n in the host class): Hessian update, H_sub = U^T (H + bias) U on the
//       FP64 matrix cores, symmetric eigen-solve of the (n - c) x (n - c) matrix, RS-I-RFO step in the subspace (the kernels of
//       RSIRFO.run: same trust-radius, image-function and secular logic), step = U step_sub.  No TR/rot projection, as in the reference.
#include <cstring>

#include "pipeline.h"

namespace mopt {

// W1[i][a] = sum_j H[i][j] Q[a][j]      grid n, block 256 (one workgroup per row, coalesced along j)
__global__ void __launch_bounds__(256) k_cp_w1(int n, int c, const double *__restrict__ H, const double *__restrict__ Q, double *__restrict__ W1) {
    __shared__ double red[16];
    const int i = blockIdx.x;
    for (int a = 0; a < c; ++a) {
        double s = 0.0;
        for (int j = threadIdx.x; j < n; j += blockDim.x) s += H[(size_t)i * n + j] * Q[(size_t)a * n + j];
        s = block_sum1(s, red);
        if (threadIdx.x == 0) W1[(size_t)i * c + a] = s;
    }
}
// W2[i][a] = sum_j H[j][i] Q[a][j]      thread per column i (coalesced along i), fixed summation order
__global__ void __launch_bounds__(256) k_cp_w2(int n, int c, const double *__restrict__ H, const double *__restrict__ Q, double *__restrict__ W2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int a = 0; a < c; ++a) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += H[(size_t)j * n + i] * Q[(size_t)a * n + j];
        W2[(size_t)i * c + a] = s;
    }
}
// M[a][b] = sum_i Q[a][i] W1[i][b]       one workgroup
__global__ void __launch_bounds__(256) k_cp_m(int n, int c, const double *__restrict__ Q, const double *__restrict__ W1, double *__restrict__ M) {
    __shared__ double red[16];
    for (int a = 0; a < c; ++a)
        for (int b = 0; b < c; ++b) {
            double s = 0.0;
            for (int i = threadIdx.x; i < n; i += blockDim.x) s += Q[(size_t)a * n + i] * W1[(size_t)i * c + b];
            s = block_sum1(s, red);
            if (threadIdx.x == 0) M[a * c + b] = s;
        }
}
// grid (ceil(n / 256), n): element (i = blockIdx.y, j)
__global__ void __launch_bounds__(256) k_cp_apply(int n, int c, int cs, const double *__restrict__ H, const double *__restrict__ Q, const double *__restrict__ S,
                                                  const double *__restrict__ W1, const double *__restrict__ W2, const double *__restrict__ M, double kh, double ks,
                                                  double *__restrict__ out) {
    const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double ph = 0.0, hp = 0.0, php = 0.0, pp = 0.0, sp = 0.0;
    for (int a = 0; a < c; ++a) {
        const double qai = Q[(size_t)a * n + i], qaj = Q[(size_t)a * n + j];
        ph += qai * W2[(size_t)j * c + a];
        hp += W1[(size_t)i * c + a] * qaj;
        pp += qai * qaj;
        double t = 0.0;
        for (int b = 0; b < c; ++b) t += M[a * c + b] * Q[(size_t)b * n + j];
        php += qai * t;
    }
    for (int a = 0; a < cs; ++a) sp += S[(size_t)a * n + i] * S[(size_t)a * n + j];
    out[(size_t)i * n + j] = H[(size_t)i * n + j] - ph - hp + php + kh * pp + ks * sp;
}

// leading m x m block of an n x n matrix, symmetrised
__global__ void __launch_bounds__(256) k_lead_block_sym(int n, int m, const double *__restrict__ src, double *__restrict__ dst) {
    const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (i < m && j < m) dst[(size_t)i * m + j] = 0.5 * (src[(size_t)i * n + j] + src[(size_t)j * n + i]);
}
__global__ void __launch_bounds__(256) k_pad_copy(int m, int n, const double *__restrict__ src, double *__restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (i < m) ? src[i] : 0.0;
}


// ---- null-space basis of the constraints on the device (round 3) -------------------------------------------------------------------
// Input: r orthonormal rows spanning the constraints' row space (Y: r x n, overwritten).  k_rowspace_qr: Householder QR (dgeqr2
// conventions) of the n x r matrix whose column k is row k of Y -- reflector k stays in Y[k][k + 1 ..] (v[k] = 1 implied), tau[k].
// k_nullspace_fill: column j of U = H_0 ... H_{r-1} e_{r+j} (the last n - r columns of the orthogonal factor), written into the padded
// n x n array Up (U in its first n - r columns) -- one thread per column, the reflectors applied in place; k_transpose_sq makes Up^T.
// One workgroup / a few dozen: O(n r^2) and O(n (n - r) r) flops, nothing that crosses PCIe but the r x n rows.
__global__ void __launch_bounds__(256) k_rowspace_qr(int n, int r, double *__restrict__ Y, double *__restrict__ tau) {
    __shared__ double red[16];
    __shared__ double s_tau;
    for (int k = 0; k < r; ++k) {
        double *col = Y + (size_t)k * n;
        double nn = 0.0;
        for (int i = k + 1 + threadIdx.x; i < n; i += blockDim.x) nn += col[i] * col[i];
        nn = block_sum1(nn, red);
        const double xnorm = sqrt(nn), alpha = col[k];
        __syncthreads();
        double t = 0.0;
        if (xnorm != 0.0) {
            const double beta = -copysign(hypot(alpha, xnorm), alpha);
            t = (beta - alpha) / beta;
            const double scal = 1.0 / (alpha - beta);
            for (int i = k + 1 + threadIdx.x; i < n; i += blockDim.x) col[i] *= scal;
        }
        if (threadIdx.x == 0) { s_tau = t; tau[k] = t; }
        __syncthreads();
        if (t != 0.0) {
            for (int c = k + 1; c < r; ++c) {
                double *cc = Y + (size_t)c * n;
                double d = 0.0;
                for (int i = k + threadIdx.x; i < n; i += blockDim.x) d += (i == k ? 1.0 : col[i]) * cc[i];
                d = block_sum1(d, red);
                for (int i = k + threadIdx.x; i < n; i += blockDim.x) cc[i] -= t * (i == k ? 1.0 : col[i]) * d;
                __syncthreads();
            }
        }
    }
}
__global__ void __launch_bounds__(256) k_nullspace_fill(int n, int r, const double *__restrict__ Y, const double *__restrict__ tau, double *__restrict__ Up) {
    const int j = blockIdx.x * 256 + threadIdx.x, m = n - r;
    if (j >= m) return;
    for (int i = 0; i < n; ++i) Up[(size_t)i * n + j] = (i == r + j) ? 1.0 : 0.0;
    for (int k = r - 1; k >= 0; --k) {
        const double t = tau[k];
        if (t == 0.0) continue;
        const double *v = Y + (size_t)k * n;
        double d = Up[(size_t)k * n + j];
        for (int i = k + 1; i < n; ++i) d += v[i] * Up[(size_t)i * n + j];
        d *= t;
        Up[(size_t)k * n + j] -= d;
        for (int i = k + 1; i < n; ++i) Up[(size_t)i * n + j] -= d * v[i];
    }
}
__global__ void __launch_bounds__(256) k_transpose_sq(int n, const double *__restrict__ A, double *__restrict__ At) {
    __shared__ double T[16][17];
    const int bx = blockIdx.x * 16, by = blockIdx.y * 16, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (by + ty < n && bx + tx < n) T[ty][tx] = A[(size_t)(by + ty) * n + bx + tx];
    __syncthreads();
    if (bx + ty < n && by + tx < n) At[(size_t)(bx + ty) * n + by + tx] = T[tx][ty];
}

}  // namespace mopt

namespace {
struct DB {
    void *p = nullptr;
    ~DB() { if (p) hipFree(p); }
    int alloc(mopt_ctx *ctx, size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { ctx->err = "hipMalloc failed"; p = nullptr; return MOPT_E_NOMEM; } return MOPT_OK; }
    template <class T> T *as() { return (T *)p; }
};
}
#define CTRY_(x) do { int rc__ = (x); if (rc__ != MOPT_OK) return rc__; } while (0)
#define CH2D(ctx, d, s, b) MOPT_HIP_TRY(ctx, hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, (ctx)->stream))
#define CD2H(ctx, d, s, b) MOPT_HIP_TRY(ctx, hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, (ctx)->stream))

// Q: c x n orthonormal rows (hard wall), S: cs x n (soft wall, may be empty).  All host pointers.
extern "C" int mopt_constraint_project_hess(mopt_ctx *ctx, int n, const double *H, int c, const double *Q, int cs, const double *S, double k_hard,
                                            double k_soft, double *H_out) {
    using namespace mopt;
    if (!ctx) return MOPT_E_ARG;
    if (n < 1 || n > MOPT_MAX_N || c < 1 || c > 64 || cs < 0 || cs > 256 || !H || !Q || (cs > 0 && !S) || !H_out) { ctx->err = "mopt_constraint_project_hess: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)n * n * 8;
    DB dH, dO, dQ, dS, dW1, dW2, dM;
    CTRY_(dH.alloc(ctx, nn)); CTRY_(dO.alloc(ctx, nn)); CTRY_(dQ.alloc(ctx, (size_t)c * n * 8)); CTRY_(dS.alloc(ctx, (size_t)(cs > 0 ? cs : 1) * n * 8));
    CTRY_(dW1.alloc(ctx, (size_t)n * c * 8)); CTRY_(dW2.alloc(ctx, (size_t)n * c * 8)); CTRY_(dM.alloc(ctx, (size_t)c * c * 8));
    CH2D(ctx, dH.p, H, nn); CH2D(ctx, dQ.p, Q, (size_t)c * n * 8);
    if (cs > 0) CH2D(ctx, dS.p, S, (size_t)cs * n * 8);
    {
        mopt_scope sc(ctx, "constraint", 4);
        hipLaunchKernelGGL(k_cp_w1, dim3(n), dim3(256), 0, ctx->stream, n, c, dH.as<double>(), dQ.as<double>(), dW1.as<double>());
        hipLaunchKernelGGL(k_cp_w2, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, c, dH.as<double>(), dQ.as<double>(), dW2.as<double>());
        hipLaunchKernelGGL(k_cp_m, dim3(1), dim3(256), 0, ctx->stream, n, c, dQ.as<double>(), dW1.as<double>(), dM.as<double>());
        hipLaunchKernelGGL(k_cp_apply, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, n, c, cs, dH.as<double>(), dQ.as<double>(), dS.as<double>(),
                           dW1.as<double>(), dW2.as<double>(), dM.as<double>(), k_hard, k_soft, dO.as<double>());
    }
    CD2H(ctx, H_out, dO.p, nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

// U: n x m (row-major, orthonormal columns: the null space of the constraints).  Everything else as mopt_rsirfo_run with batch 1;
// B_g is the full-space gradient (after the caller's gradient transport, if any).
static int crsirfo_run_once(mopt_ctx *ctx, int n, int m, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const double *hessian,
                            const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom,
                            const double *pre_g, const double *B_e, const double *U, double *step, double *H_inplace, double *H_sym, const double *Qrows = nullptr);

extern "C" int mopt_crsirfo_run(mopt_ctx *ctx, int n, int m, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const double *hessian,
                                const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom,
                                const double *pre_g, const double *B_e, const double *U, double *step, double *H_inplace, double *H_sym) {
    if (!ctx || !state) return MOPT_E_ARG;
    // (a hand-off time-out -- jacobi_sweeps = -2, k_rfo_scalar -- is repeated once with one launch per Householder step; the outputs
    // may alias the input Hessian: keep a copy)
    const mopt_rsirfo_state st0 = *state;
    std::vector<double> hkeep;
    if (hessian && n > 0 && (H_inplace == hessian || H_sym == hessian)) hkeep.assign(hessian, hessian + (size_t)n * n);
    int rc = crsirfo_run_once(ctx, n, m, params, state, hessian, bias_hessian, geom, B_g, g, pre_geom, pre_g, B_e, U, step, H_inplace, H_sym);
    if (rc == MOPT_E_NUMERIC && state->jacobi_sweeps == -2) {
        *state = st0;
        const bool saved = ctx->tri_no_persist;
        ctx->tri_no_persist = true;
        rc = crsirfo_run_once(ctx, n, m, params, state, hkeep.empty() ? hessian : hkeep.data(), bias_hessian, geom, B_g, g, pre_geom, pre_g, B_e, U, step,
                              H_inplace, H_sym);
        ctx->tri_no_persist = saved;
    }
    return rc;
}


// The same with the null-space basis formed on the device: Qrows = r orthonormal rows (r x n) spanning the row space of the constraints
// (r <= 64); the step lives in their orthogonal complement (m = n - r columns).
extern "C" int mopt_crsirfo_run_rowspace(mopt_ctx *ctx, int n, int r, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const double *hessian,
                                         const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom,
                                         const double *pre_g, const double *B_e, const double *Qrows, double *step, double *H_inplace, double *H_sym) {
    if (!ctx || !state) return MOPT_E_ARG;
    if (r < 0 || r > 64 || r >= n || (r > 0 && !Qrows)) { ctx->err = "mopt_crsirfo_run_rowspace: bad row space"; return MOPT_E_ARG; }
    static const double dummy = 0.0;
    const double *Q = r > 0 ? Qrows : &dummy;
    const mopt_rsirfo_state st0 = *state;
    std::vector<double> hkeep;
    if (hessian && n > 0 && (H_inplace == hessian || H_sym == hessian)) hkeep.assign(hessian, hessian + (size_t)n * n);
    int rc = crsirfo_run_once(ctx, n, n - r, params, state, hessian, bias_hessian, geom, B_g, g, pre_geom, pre_g, B_e, nullptr, step, H_inplace, H_sym, Q);
    if (rc == MOPT_E_NUMERIC && state->jacobi_sweeps == -2) {
        *state = st0;
        const bool saved = ctx->tri_no_persist;
        ctx->tri_no_persist = true;
        rc = crsirfo_run_once(ctx, n, n - r, params, state, hkeep.empty() ? hessian : hkeep.data(), bias_hessian, geom, B_g, g, pre_geom, pre_g, B_e, nullptr,
                              step, H_inplace, H_sym, Q);
        ctx->tri_no_persist = saved;
    }
    return rc;
}

static int crsirfo_run_once(mopt_ctx *ctx, int n, int m, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const double *hessian,
                            const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom,
                            const double *pre_g, const double *B_e, const double *U, double *step, double *H_inplace, double *H_sym, const double *Qrows) {
    using namespace mopt;
    if (!ctx) return MOPT_E_ARG;
    if (n < 1 || n > MOPT_MAX_N || m < 1 || m > n || !params || !state || !geom || !B_g || !B_e || (!U && !Qrows) || !step) { ctx->err = "mopt_crsirfo_run: bad arguments"; return MOPT_E_ARG; }
    if (!hessian) { ctx->err = "Hessian matrix must be set before running optimization"; return MOPT_E_STATE; }
    hipSetDevice(ctx->device);
    RsirfoWork wk, ws;
    CTRY_(wk.alloc(ctx, n, 1, true, bias_hessian != nullptr, H_inplace != nullptr));
    CTRY_(ws.alloc(ctx, m, 1, false, false, false));
    const size_t nn = (size_t)n * n * 8, nv = (size_t)n * 8;
    CH2D(ctx, wk.H, hessian, nn);
    if (bias_hessian) CH2D(ctx, wk.Bias, bias_hessian, nn);
    CH2D(ctx, wk.x, geom, nv); CH2D(ctx, wk.bg, B_g, nv); CH2D(ctx, wk.g, g ? g : B_g, nv);
    const bool have_prev = pre_geom && pre_g;
    if (have_prev) { CH2D(ctx, wk.px, pre_geom, nv); CH2D(ctx, wk.pg, pre_g, nv); }
    const int rule = params->rule;
    if (rule < 0 || rule >= MOPT_RULE_COUNT) { ctx->err = "unknown update rule"; return MOPT_E_ARG; }
    const int upd = (have_prev && state->has_prev) ? 1 : 0;
    CH2D(ctx, wk.rules, &rule, sizeof(int)); CH2D(ctx, wk.do_update, &upd, sizeof(int));
    CH2D(ctx, ws.params, params, sizeof(mopt_rsirfo_params)); CH2D(ctx, ws.states, state, sizeof(mopt_rsirfo_state));
    CH2D(ctx, ws.be, B_e, 8);
    // padded basis: Upad (n x n, U in its first m columns) and its transpose
    DB dU, dUt, dvec, dY;
    CTRY_(dU.alloc(ctx, nn)); CTRY_(dUt.alloc(ctx, nn)); CTRY_(dvec.alloc(ctx, 2 * nv));
    std::vector<double> up, upt;
    if (Qrows) {
        // from the r = n - m orthonormal rows of the constraints' row space, on the device: only r x n numbers cross PCIe
        const int r = n - m;
        CTRY_(dY.alloc(ctx, (size_t)(r > 0 ? r : 1) * n * 8 + 64 * 8));
        double *Y = dY.as<double>(), *tauv = Y + (size_t)(r > 0 ? r : 1) * n;
        if (r > 0) CH2D(ctx, Y, Qrows, (size_t)r * n * 8);
        MOPT_HIP_TRY(ctx, hipMemsetAsync(dU.p, 0, nn, ctx->stream));
        mopt_scope sc(ctx, "constraint", 3);
        if (r > 0) hipLaunchKernelGGL(k_rowspace_qr, dim3(1), dim3(256), 0, ctx->stream, n, r, Y, tauv);
        hipLaunchKernelGGL(k_nullspace_fill, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, n, r, (const double *)Y, (const double *)tauv, dU.as<double>());
        hipLaunchKernelGGL(k_transpose_sq, dim3((n + 15) / 16, (n + 15) / 16), dim3(256), 0, ctx->stream, n, (const double *)dU.as<double>(), dUt.as<double>());
    } else {
        up.assign((size_t)n * n, 0.0); upt.assign((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) for (int k = 0; k < m; ++k) { const double v = U[(size_t)i * m + k]; up[(size_t)i * n + k] = v; upt[(size_t)k * n + i] = v; }
        CH2D(ctx, dU.p, up.data(), nn); CH2D(ctx, dUt.p, upt.data(), nn);
    }
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // (the staging vectors go out of use below)
    const UpdateCoef *coef = nullptr;
    if (upd) {
        CTRY_(mopt_launch_hess_update(ctx, n, 1, wk.rules, wk.do_update, rule == MOPT_RULE_FLOWCHART, wk.x, wk.px, wk.g, wk.pg, wk.s, wk.y, wk.u, wk.Hy,
                                      wk.yd, wk.Av, wk.coef, wk.info, wk.H, wk.Hinpl, wk.H, false));
        coef = wk.coef;
    }
    CTRY_(mopt_launch_sym_add(ctx, n, 1, wk.H, wk.Bias, wk.A));                              // A = sym(H + bias)
    CTRY_(mopt_launch_dgemm(ctx, n, 1, wk.A, dU.as<double>(), nullptr, wk.Gt));             // A U
    CTRY_(mopt_launch_dgemm(ctx, n, 1, dUt.as<double>(), wk.Gt, nullptr, wk.Vt));           // U^T A U (leading m x m block)
    hipLaunchKernelGGL(k_lead_block_sym, dim3((m + 255) / 256, m), dim3(256), 0, ctx->stream, n, m, wk.Vt, ws.A);
    double *tmp = dvec.as<double>(), *spad = tmp + n;
    CTRY_(mopt_launch_matvec(ctx, n, 1, dUt.as<double>(), wk.bg, tmp));                     // U^T g (first m entries)
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(ws.gp, tmp, (size_t)m * 8, hipMemcpyDeviceToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(ws.bg, tmp, (size_t)m * 8, hipMemcpyDeviceToDevice, ctx->stream));
    CTRY_(mopt_eigh_auto(ctx, ws, m, 1, false, /*preserve_A=*/false));
    CTRY_(mopt_launch_rfo(ctx, m, 1, ws.params, ws.states, ws.eigw, ws.perm, ws.einfo, ws.Vt, ws.gp, ws.bg, ws.be, coef, ws.gt_all, ws.coefout, ws.step));
    hipLaunchKernelGGL(k_pad_copy, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, m, n, ws.step, spad);
    CTRY_(mopt_launch_matvec(ctx, n, 1, dU.as<double>(), spad, wk.step));                   // step = U step_sub
    CD2H(ctx, step, wk.step, nv);
    CD2H(ctx, state, ws.states, sizeof(mopt_rsirfo_state));
    if (H_sym) CD2H(ctx, H_sym, wk.H, nn);
    if (H_inplace) CD2H(ctx, H_inplace, upd ? wk.Hinpl : wk.H, nn);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    MOPT_HIP_TRY(ctx, hipGetLastError());
    if (mopt_debug_poison_on()) { CTRY_(wk.check_canaries()); CTRY_(ws.check_canaries()); }
    if (state->jacobi_sweeps < 0) { ctx->err = "mopt_crsirfo_run: the eigen-solver failed"; return MOPT_E_NUMERIC; }
    return MOPT_OK;
}
