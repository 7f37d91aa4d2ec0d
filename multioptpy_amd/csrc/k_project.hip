// Translation / rotation projection kernels (batched over images).
//   k_trrot_basis   6 TR/rot vectors about the geometric centre, classical Gram-Schmidt (drop |w| <= 1e-10)
//                   (multioptpy/Utils/calc_tools.py:250-293)
//   k_sym_add       A = 0.5 ((H + B) + (H + B)^T)
//   k_proj_w        W = A Q^T            (n x 6; one wave per row)
//   k_proj_fix      W' = W - 0.5 Q^T (Q W)
//   k_proj_apply    A -= Q^T W'^T + W' Q  == sym(P^T A P), P = I - Q^T Q   (rank-12 instead of two n^3 GEMMs)
//   k_proj_grad     g - Q Q^T g with Q from a LAPACK-convention Householder QR of the (n,6) matrix
//                   (multioptpy/Optimizer/rsirfo.py:128-190)
#include "common.hpp"

namespace mopt {

// Builds the six raw vectors for image b into dst[6][ld] (LDS or global), from x (N atoms).
__device__ inline void build_trrot(int n, const double *x, double cx, double cy, double cz, double *dst, int ld) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int a = i / 3, c = i % 3;
        const double X = x[3 * a] - cx, Y = x[3 * a + 1] - cy, Z = x[3 * a + 2] - cz;
        dst[0 * ld + i] = (c == 0) ? 1.0 : 0.0;
        dst[1 * ld + i] = (c == 1) ? 1.0 : 0.0;
        dst[2 * ld + i] = (c == 2) ? 1.0 : 0.0;
        dst[3 * ld + i] = (c == 0) ? 0.0 : (c == 1 ? -Z : Y);      // (0, -z, y)
        dst[4 * ld + i] = (c == 0) ? Z : (c == 1 ? 0.0 : -X);      // (z, 0, -x)
        dst[5 * ld + i] = (c == 0) ? -Y : (c == 1 ? X : 0.0);      // (-y, x, 0)
    }
}

__device__ inline void centre_of(int n, const double *x, double *red, double &cx, double &cy, double &cz) {
    double v[3] = {0, 0, 0};
    const int N = n / 3;
    for (int a = threadIdx.x; a < N; a += blockDim.x) { v[0] += x[3 * a]; v[1] += x[3 * a + 1]; v[2] += x[3 * a + 2]; }
    block_sum<3>(v, red);
    cx = v[0] / N; cy = v[1] / N; cz = v[2] / N;
}

// Q out: [b][6][n] (rows beyond the kept count are zero).  dynamic LDS: 12 * n doubles.
__global__ void __launch_bounds__(256) k_trrot_basis(int n, const double *__restrict__ X, double *__restrict__ Q) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[9 * 16];
    double *raw = sm;            // [6][n]
    double *bas = sm + 6 * n;    // [6][n]
    const int b = blockIdx.x;
    const double *x = X + (size_t)b * n;
    double cx, cy, cz;
    centre_of(n, x, red, cx, cy, cz);
    build_trrot(n, x, cx, cy, cz, raw, n);
    __syncthreads();
    // Classical Gram-Schmidt in the order of the reference (calc_tools.py:262-290), with the reductions that cannot change a bit taken out
    // (round 4; the kernel stands between the upload and the fused update on the critical path of a band step: 13 block reductions -> 7):
    //  * the three translation vectors have disjoint supports: their mutual projections are exact zeros and |t|^2 = N exactly, so they
    //    are normalised directly -- the same numbers the general loop produces;
    //  * the projections of the three rotation vectors on the translations depend on nothing computed in between: one reduction of nine sums.
    int nb = 3;
    {
        const double nrm_t = sqrt((double)(n / 3));
        for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) bas[i] = raw[i] / nrm_t;
    }
    __syncthreads();
    double dA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double rv = raw[(3 + v) * n + i];
#pragma unroll
            for (int k = 0; k < 3; ++k) dA[3 * v + k] += rv * bas[k * n + i];
        }
    }
    block_sum<9>(dA, red);
    for (int v = 3; v < 6; ++v) {
        const int ne = nb - 3;                               // rotation vectors kept so far (uniform)
        double dx[2] = {0, 0};
        if (ne > 0) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const double rv = raw[v * n + i];
                for (int k = 0; k < ne; ++k) dx[k] += rv * bas[(3 + k) * n + i];
            }
            block_sum<2>(dx, red);
        }
        double nn = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            double wv = raw[v * n + i];
#pragma unroll
            for (int k = 0; k < 3; ++k) wv -= dA[3 * (v - 3) + k] * bas[k * n + i];
            for (int k = 0; k < ne; ++k) wv -= dx[k] * bas[(3 + k) * n + i];
            bas[nb * n + i] = wv;       // tentative
            nn += wv * wv;
        }
        nn = block_sum1(nn, red);
        double nrm = sqrt(nn);
        if (nrm > 1e-10) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) bas[nb * n + i] /= nrm;
            ++nb;
        }
        __syncthreads();
    }
    double *q = Q + (size_t)b * 6 * n;
    for (int i = threadIdx.x; i < 6 * n; i += blockDim.x) q[i] = (i / n < nb) ? bas[i] : 0.0;
}

// A = 0.5 ((H+B) + (H+B)^T); tile pairs as in k_apply_update.  Bias may be null.
// A may be H itself (in place: a tile pair is read completely before it is written).
__global__ void __launch_bounds__(256) k_sym_add(int n, const double *H, const double *__restrict__ Bias, double *A) {
    __shared__ double L[32][33], S[32][33];
    const int T = (n + 31) / 32;
    const int bi = blockIdx.x / T, bj = blockIdx.x % T;
    if (bi > bj) return;
    const size_t off = (size_t)blockIdx.y * n * n;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = 0; k < 4; ++k) {
        int r = ty + 8 * k, i2 = bj * 32 + r, j2 = bi * 32 + tx;
        double v = 0.0;
        if (i2 < n && j2 < n) { v = H[off + (size_t)i2 * n + j2]; if (Bias) v += Bias[off + (size_t)i2 * n + j2]; }
        L[r][tx] = v;
    }
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        int r = ty + 8 * k, i = bi * 32 + r, j = bj * 32 + tx;
        double v = 0.0;
        if (i < n && j < n) { v = H[off + (size_t)i * n + j]; if (Bias) v += Bias[off + (size_t)i * n + j]; }
        double sym = 0.5 * (v + L[tx][r]);
        S[r][tx] = sym;
        if (i < n && j < n) A[off + (size_t)i * n + j] = sym;
    }
    if (bi != bj) {
        __syncthreads();
        for (int k = 0; k < 4; ++k) {
            int r = ty + 8 * k, i2 = bj * 32 + r, j2 = bi * 32 + tx;
            if (i2 < n && j2 < n) A[off + (size_t)i2 * n + j2] = S[tx][r];
        }
    }
}

// W[b][i][k] = sum_j A[b][i][j] Q[b][k][j]
__global__ void __launch_bounds__(256) k_proj_w(int n, const double *__restrict__ A, const double *__restrict__ Q,
                                                double *__restrict__ W) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wid;
    if (r >= n) return;
    const double *row = A + ((size_t)b * n + r) * n;
    const double *q = Q + (size_t)b * 6 * n;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int j = lane; j < n; j += WAVE) {
        double a = row[j];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += a * q[k * n + j];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) W[((size_t)b * n + r) * 6 + k] = acc[k];
    }
}

// M = Q W (6x6), W' = W - 0.5 Q^T M.  One workgroup per image.
__global__ void __launch_bounds__(256) k_proj_fix(int n, const double *__restrict__ Q, double *__restrict__ W) {
    __shared__ double red[36 * 16];
    __shared__ double M[6][6];
    const int b = blockIdx.x;
    const double *q = Q + (size_t)b * 6 * n;
    double *w = W + (size_t)b * n * 6;
    {
        // M = Q W: all 36 dot products in one pass and one block reduction (six passes with a barrier each before)
        double d[36];
#pragma unroll
        for (int e = 0; e < 36; ++e) d[e] = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            double wl[6];
#pragma unroll
            for (int l = 0; l < 6; ++l) wl[l] = w[i * 6 + l];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double qk = q[k * n + i];
#pragma unroll
                for (int l = 0; l < 6; ++l) d[k * 6 + l] += qk * wl[l];
            }
        }
        block_sum<36>(d, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int e = 0; e < 36; ++e) M[e / 6][e % 6] = d[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) t += q[k * n + i] * 0.5 * (M[k][l] + M[l][k]);
            w[i * 6 + l] -= 0.5 * t;
        }
    }
}

// A_ij -= sum_k (q_ki W'_jk + W'_ik q_kj)
__global__ void __launch_bounds__(256) k_proj_apply(int n, double *__restrict__ A, const double *__restrict__ Q,
                                                    const double *__restrict__ W) {
    __shared__ double qr[6][32], qc[6][32], wr[6][32], wc[6][32];
    const int T = (n + 31) / 32;
    const int bi = blockIdx.x / T, bj = blockIdx.x % T;
    const int b = blockIdx.y;
    const double *q = Q + (size_t)b * 6 * n;
    const double *w = W + (size_t)b * n * 6;
    if (threadIdx.x < 192) {
        int k = threadIdx.x >> 5, t = threadIdx.x & 31;
        int i = bi * 32 + t, j = bj * 32 + t;
        qr[k][t] = (i < n) ? q[k * n + i] : 0.0;
        qc[k][t] = (j < n) ? q[k * n + j] : 0.0;
        wr[k][t] = (i < n) ? w[i * 6 + k] : 0.0;
        wc[k][t] = (j < n) ? w[j * 6 + k] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double *Ab = A + (size_t)b * n * n;
    for (int kk = 0; kk < 4; ++kk) {
        int r = ty + 8 * kk, i = bi * 32 + r, j = bj * 32 + tx;
        if (i < n && j < n) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) t += qr[k][r] * wc[k][tx] + wr[k][r] * qc[k][tx];
            Ab[(size_t)i * n + j] -= t;
        }
    }
}

// Householder QR (dgeqr2 conventions) of the n x kc matrix, applied to g.  dynamic LDS: 7 * n doubles.
__global__ void __launch_bounds__(256) k_proj_grad(int n, int variant, const double *__restrict__ G, const double *__restrict__ X,
                                                   double *__restrict__ GP) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[6 * 16];
    __shared__ double taus[6];
    __shared__ double rdiag[6];
    double *Acol = sm;          // [6][n] column-major
    double *t = sm + 6 * n;     // [n]
    const int b = blockIdx.x;
    const double *x = X + (size_t)b * n;
    const double *g = G + (size_t)b * n;
    if (variant == 1 && n < 9) {          // rsprfo.py:245-246: fewer than 3 atoms -> gradient returned unprojected
        for (int i = threadIdx.x; i < n; i += blockDim.x) GP[(size_t)b * n + i] = g[i];
        return;
    }
    double cx, cy, cz;
    centre_of(n, x, red, cx, cy, cz);
    build_trrot(n, x, cx, cy, cz, Acol, n);
    for (int i = threadIdx.x; i < n; i += blockDim.x) t[i] = g[i];
    __syncthreads();
    const int kc = n < 6 ? n : 6;
    for (int j = 0; j < kc; ++j) {
        double *col = Acol + j * n;
        double nn = 0.0;
        for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) nn += col[i] * col[i];
        nn = block_sum1(nn, red);
        const double xnorm = sqrt(nn);
        const double alpha = col[j];
        double tau = 0.0;
        __syncthreads();
        if (xnorm != 0.0) {
            const double beta = -copysign(hypot(alpha, xnorm), alpha);
            tau = (beta - alpha) / beta;
            const double scal = 1.0 / (alpha - beta);
            for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) col[i] *= scal;
            if (threadIdx.x == 0) col[j] = 1.0;     // v[0] = 1 (R's diagonal is not needed afterwards)
        } else {
            if (threadIdx.x == 0) col[j] = 1.0;
        }
        if (threadIdx.x == 0) { taus[j] = tau; rdiag[j] = (xnorm != 0.0) ? -copysign(hypot(alpha, xnorm), alpha) : alpha; }
        __syncthreads();
        if (tau != 0.0) {
            // apply H_j = I - tau v v^T to the trailing columns and to t
            double d[6] = {0, 0, 0, 0, 0, 0};
            for (int i = j + threadIdx.x; i < n; i += blockDim.x) {
                double vi = col[i];
                for (int c = j + 1; c < kc; ++c) d[c - j - 1] += vi * Acol[c * n + i];
                d[5] += vi * t[i];
            }
            block_sum<6>(d, red);
            for (int i = j + threadIdx.x; i < n; i += blockDim.x) {
                double vi = col[i];
                for (int c = j + 1; c < kc; ++c) Acol[c * n + i] -= tau * vi * d[c - j - 1];
                t[i] -= tau * vi * d[5];
            }
            __syncthreads();
        }
    }
    // t = [Q^T g ; rest] -> keep the first kc entries, then Q t' = H_0 ... H_{kc-1} [t'; 0]
    for (int i = kc + threadIdx.x; i < n; i += blockDim.x) t[i] = 0.0;
    if (variant == 1 && threadIdx.x < kc && !(fabs(rdiag[threadIdx.x]) > 1e-10)) t[threadIdx.x] = 0.0;   // Q[:, valid_cols] (rsprfo.py:275-277)
    __syncthreads();
    for (int j = kc - 1; j >= 0; --j) {
        const double tau = taus[j];
        if (tau != 0.0) {
            double *col = Acol + j * n;
            double d = 0.0;
            for (int i = j + threadIdx.x; i < n; i += blockDim.x) d += col[i] * t[i];
            d = block_sum1(d, red);
            for (int i = j + threadIdx.x; i < n; i += blockDim.x) t[i] -= tau * col[i] * d;
            __syncthreads();
        }
    }
    double *gp = GP + (size_t)b * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) gp[i] = g[i] - t[i];
}

}  // namespace mopt

// basis_ready: the TR/rot basis Q was already requested on the side stream (mopt_launch_trrot_basis_side); wait for it
// in front of the first kernel that reads it
int mopt_launch_project_hess(mopt_ctx *ctx, int n, int batch, const double *d_H, const double *d_Bias, const double *d_X,
                             double *d_Q, double *d_W, double *d_A, bool basis_ready) {
    using namespace mopt;
    mopt_scope sc(ctx, "project", basis_ready ? 4 : 5);
    if (!basis_ready) hipLaunchKernelGGL(k_trrot_basis, dim3(batch), dim3(256), (size_t)12 * n * sizeof(double), ctx->stream, n, d_X, d_Q);
    const int T = (n + 31) / 32;
    hipLaunchKernelGGL(k_sym_add, dim3(T * T, batch), dim3(256), 0, ctx->stream, n, d_H, d_Bias, d_A);
    if (basis_ready) MOPT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join3, 0));
    hipLaunchKernelGGL(k_proj_w, dim3((n + 3) / 4, batch), dim3(256), 0, ctx->stream, n, d_A, d_Q, d_W);
    hipLaunchKernelGGL(k_proj_fix, dim3(batch), dim3(256), 0, ctx->stream, n, d_Q, d_W);
    hipLaunchKernelGGL(k_proj_apply, dim3(T * T, batch), dim3(256), 0, ctx->stream, n, d_A, d_Q, d_W);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

// The basis depends on the geometry only: fork from the main stream, build it on the side stream (followed there by the
// gradient projection, see below) and let the main stream run the Hessian update meanwhile.
int mopt_launch_trrot_basis_side(mopt_ctx *ctx, int n, int batch, const double *d_X, double *d_Q) {
    using namespace mopt;
    MOPT_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork3, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork3, 0));
    hipLaunchKernelGGL(k_trrot_basis, dim3(batch), dim3(256), (size_t)12 * n * sizeof(double), ctx->side, n, d_X, d_Q);
    MOPT_HIP_TRY(ctx, hipEventRecord(ctx->ev_join3, ctx->side));
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

int mopt_launch_sym_add(mopt_ctx *ctx, int n, int batch, const double *d_H, const double *d_Bias, double *d_A) {
    using namespace mopt;
    const int T = (n + 31) / 32;
    hipLaunchKernelGGL(k_sym_add, dim3(T * T, batch), dim3(256), 0, ctx->stream, n, d_H, d_Bias, d_A);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

// side_stream: the projected gradient depends on g and x only and is not needed before the RFO stage, so the pipeline runs
// it on the helper stream next to the Hessian chain (fork here, join in front of k_vtg); it is then not part of a timed family.
int mopt_launch_project_grad(mopt_ctx *ctx, int n, int batch, const double *d_g, const double *d_X, double *d_gp, int variant, bool side_stream) {
    using namespace mopt;
    if (side_stream) {
        MOPT_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork2, ctx->stream));
        MOPT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork2, 0));
        hipLaunchKernelGGL(k_proj_grad, dim3(batch), dim3(256), (size_t)7 * n * sizeof(double), ctx->side, n, variant, d_g, d_X, d_gp);
        MOPT_HIP_TRY(ctx, hipEventRecord(ctx->ev_join2, ctx->side));
        MOPT_HIP_TRY(ctx, hipGetLastError());
        return MOPT_OK;
    }
    mopt_scope sc(ctx, "project", 1);
    hipLaunchKernelGGL(k_proj_grad, dim3(batch), dim3(256), (size_t)7 * n * sizeof(double), ctx->stream, n, variant, d_g, d_X, d_gp);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
