// Batched FP64 GEMM on the matrix cores (v_mfma_f64_16x16x4_f64), used to warm-start the eigen-solver:
//   C[b] = A[b] * B[b] + addscale[b] * A[b]        (all n x n, row-major)
// With A = Vt (previous eigenvectors, vector-major), B = projected Hessian (symmetric), addscale = sigma this is
// Gt = Vt * (B + sigma I): the images of the old eigenvectors under the shifted matrix.
// Workgroup = 4 waves = 64 x 64 tile of C, each wave 32 x 32 (2 x 2 MFMA tiles), K staged through LDS in chunks of 16.
// f64 MFMA operand layout (cdna_hip_programming.md section 3): A: lane l holds A[l&15][l>>4]; B: B[l>>4][l&15];
// C/D: 4 doubles per lane, col = l&15, row = (l>>4) + 4*reg.
#include "pipeline.h"

namespace mopt {

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_dgemm_nn(int n, const double *__restrict__ A_, const double *__restrict__ B_,
                                                  const double *__restrict__ addscale, double *__restrict__ C_) {
    __shared__ double As[64][17];
    __shared__ double Bs[16][65];
    const int b = blockIdx.z;
    const double *A = A_ + (size_t)b * n * n, *B = B_ + (size_t)b * n * n;
    double *C = C_ + (size_t)b * n * n;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = (wid >> 1) * 32, wc = (wid & 1) * 32;
    double4_t acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < n; k0 += 16) {
        // A tile 64 x 16: thread t -> row t>>2, cols (t&3)*4 .. +3
        {
            const int r = tid >> 2, c = (tid & 3) * 4;
            const int gr = row0 + r;
            for (int q = 0; q < 4; ++q) {
                const int gc = k0 + c + q;
                As[r][c + q] = (gr < n && gc < n) ? A[(size_t)gr * n + gc] : 0.0;
            }
        }
        // B tile 16 x 64: thread t -> row t>>4, cols (t&15)*4 .. +3
        {
            const int r = tid >> 4, c = (tid & 15) * 4;
            const int gr = k0 + r;
            for (int q = 0; q < 4; ++q) {
                const int gc = col0 + c + q;
                Bs[r][c + q] = (gr < n && gc < n) ? B[(size_t)gr * n + gc] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            double a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[wr + i * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = Bs[kk + (lane >> 4)][wc + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const double sc = addscale ? addscale[b] : 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = row0 + wr + i * 16 + (lane >> 4) + 4 * r;
                const int gc = col0 + wc + j * 16 + (lane & 15);
                if (gr < n && gc < n) {
                    double v = acc[i][j][r];
                    if (addscale) v += sc * A[(size_t)gr * n + gc];
                    C[(size_t)gr * n + gc] = v;
                }
            }
}

}  // namespace mopt

int mopt_launch_dgemm(mopt_ctx *ctx, int n, int batch, const double *d_A, const double *d_B, const double *d_addscale, double *d_C) {
    using namespace mopt;
    const int T = (n + 63) / 64;
    hipLaunchKernelGGL(k_dgemm_nn, dim3(T, T, batch), dim3(256), 0, ctx->stream, n, d_A, d_B, d_addscale, d_C);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
