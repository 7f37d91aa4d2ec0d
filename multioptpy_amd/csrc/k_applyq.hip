// y <- Q^T y for ONE vector per matrix, Q = H_0 H_1 ... H_{n-3} the reflectors left in the rows of A by the tridiagonalisation
// (k_tridiag.hip / k_tritail.hip), taken 16 at a time in compact WY form with the T factors of k_tri_larft (k_backtrans.hip):
//   Q = B_0 B_1 ... B_{NB-1},  B_beta = I - V T V^T   =>   Q^T y = B_{NB-1}^T ... B_0^T y,   B^T y = y - V (T^T (V^T y)).
// Why: the RS-I-RFO stage needs g~ = V^T g with V = Q Z (rsirfo.py:606-657).  Z^T (Q^T g) is the same vector and needs the eigenvectors
// Z of the TRIDIAGONAL form only, so the scalar stage of the step (secular equation, trust radius: one wave per image, 68 us at
// 32 x 300) runs NEXT TO the back-transformation V = Q Z instead of behind it; this kernel itself runs on a helper stream next to
// the eigenvalue / eigenvector kernels of the tridiagonal form.
// One workgroup of 16 waves per matrix; the block of 16 reflectors and its T factor are staged through LDS, the next block's rows are
// in flight in registers while the current one is applied (three LDS barriers per block).  n <= 576 (two staged blocks of 16 x n doubles in LDS).
#include "pipeline.h"

namespace mopt {

constexpr int AQ_CH = 9;

__global__ void __launch_bounds__(1024) k_apply_qt(int n, int NB, const double *__restrict__ A_, const double *__restrict__ T_,
                                                   const double *__restrict__ in_, double *__restrict__ out_) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int s_nz[2];                                  // the staged block has a non-zero T factor (a block of trivial reflectors is the identity)
    const int ldv = n | 1;
    double *y = sm, *Vs = y + n, *Ts = Vs + (size_t)2 * 16 * ldv, *w = Ts + 2 * 256, *tt = w + 16;      // y[n], V[2][16][ldv], T[2][256], w[16], t[16]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, c = tid >> 6;
    const int chn = (n + 63) >> 6;
    const double *A = A_ + (size_t)b * n * n;
    for (int i = tid; i < n; i += 1024) y[i] = in_[(size_t)b * n + i];
    double r[AQ_CH], tv = 0.0;
    auto fetch = [&](int beta) {
        const int k = 16 * beta + c;
#pragma unroll
        for (int q = 0; q < AQ_CH; ++q) {
            const int i = lane + 64 * q;
            double v = 0.0;
            if (q < chn && beta < NB && k <= n - 3) { if (i == k + 1) v = 1.0; else if (i > k + 1 && i < n) v = A[(size_t)k * n + i]; }
            r[q] = v;
        }
        tv = (tid < 256 && beta < NB) ? T_[((size_t)b * NB + beta) * 256 + tid] : 0.0;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < AQ_CH; ++q) { const int i = lane + 64 * q; if (q < chn && i < n) Vs[((size_t)buf * 16 + c) * ldv + i] = r[q]; }
        if (tid < 256) { Ts[buf * 256 + tid] = tv; if (tv != 0.0) s_nz[buf] = 1; }
    };
    if (tid == 0) { s_nz[0] = 0; s_nz[1] = 0; }
    __syncthreads();
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int beta = 0; beta < NB; ++beta) {
        fetch(beta + 1);                                     // in flight during the three phases below
        if (tid == 0) s_nz[buf ^ 1] = 0;                     // (set again by the stage below, behind a barrier)
        if (!s_nz[buf]) {                                    // (uniform) T = 0: B_beta = I
            lds_barrier();
            stage(buf ^ 1);
            lds_barrier();
            buf ^= 1;
            continue;
        }
        const double *V = Vs + (size_t)buf * 16 * ldv, *T = Ts + buf * 256;
        {   // w_c = v_c . y
            const double *vc = V + (size_t)c * ldv;
            double d = 0.0;
            for (int i = 16 * beta + c + 1 + lane; i < n; i += 64) d = fma(vc[i], y[i], d);
            d = wave_sum(d);
            if (lane == 0) w[c] = d;
        }
        lds_barrier();
        if (tid < 16) {                                      // t = T^T w  (T upper triangular, row-major)
            double a = 0.0;
            for (int j = 0; j <= tid; ++j) a = fma(T[j * 16 + tid], w[j], a);
            tt[tid] = a;
        }
        lds_barrier();
        for (int i = 16 * beta + 1 + tid; i < n; i += 1024) {
            double a = y[i];
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) a = fma(-V[(size_t)cc * ldv + i], tt[cc], a);
            y[i] = a;
        }
        stage(buf ^ 1);
        lds_barrier();
        buf ^= 1;
    }
    for (int i = tid; i < n; i += 1024) out_[(size_t)b * n + i] = y[i];
}

}  // namespace mopt

bool mopt_apply_qt_supported(int n) { return n >= 3 && n <= 64 * mopt::AQ_CH; }

// d_out[b] = Q_b^T d_in[b]  (d_T: the T factors of mopt_launch_bt_larft, NB = ceil((n - 2) / 16) per matrix)
int mopt_launch_apply_qt(mopt_ctx *ctx, hipStream_t st, int n, int batch, const double *d_A, const double *d_T, const double *d_in, double *d_out) {
    using namespace mopt;
    if (!mopt_apply_qt_supported(n)) { ctx->err = "k_apply_qt: n out of range"; return MOPT_E_ARG; }
    const int NB = (n - 2 + 15) / 16;
    const size_t lds = ((size_t)n + (size_t)2 * 16 * (n | 1) + 2 * 256 + 32) * 8;
    if (lds > 60000) MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_apply_qt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_apply_qt, dim3(batch), dim3(1024), lds, st, n, NB, d_A, d_T, d_in, d_out);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
