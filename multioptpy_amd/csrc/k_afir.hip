// AFIR bias potential on the device: energy, gradient and Hessian of
//   E = alpha A / B,  A = sum_ij omega_ij r_ij,  B = sum_ij omega_ij,  omega_ij = ((R_i + R_j) / r_ij)^6,  i in fragment 1, j in fragment 2
// (multioptpy/Potential/AFIR_potential.py:19-54).  The reference differentiates this with torch autograd on the CPU every step
// (Potential/potential.py:127-152: torch.func.jacrev + torch.func.hessian, the only other per-step O(n^2) producer next to the
// optimizer itself); here the derivatives are written out:
//   dE  = alpha (dA / B - A dB / B^2)
//   d2E = alpha (d2A / B - (dA dB^T + dB dA^T) / B^2 - A d2B / B^2 + 2 A dB dB^T / B^3)
// with pair terms f(r): a = c r^-5, omega = c r^-6, c = (R_i + R_j)^6 -- the radii are float32 values added in float32, as the
// reference's default-dtype torch.tensor does.
//   k_afir_pairs   one workgroup: A, B and the per-atom gradients dA, dB (each atom sums over its partners in list order)
//   k_afir_hess    one workgroup per atom row: the pair blocks of d2A, d2B and the four dense rank-1 terms, no atomics
#include "pipeline.h"

namespace mopt {

struct AfirPair { double da1, da2, dw1, dw2, r, ux, uy, uz; };

__device__ inline AfirPair afir_pair(const double *x, const double *cov, int i, int j) {
    AfirPair p;
    const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    const double rs = (double)((float)cov[i] + (float)cov[j]);
    const double c = (rs * rs * rs) * (rs * rs * rs);
    const double r2 = r * r, r4 = r2 * r2, r6 = r4 * r2, r7 = r6 * r, r8 = r4 * r4;
    p.r = r; p.ux = dx / r; p.uy = dy / r; p.uz = dz / r;
    p.da1 = -5.0 * c / r6; p.da2 = 30.0 * c / r7;       // a = c r^-5
    p.dw1 = -6.0 * c / r7; p.dw2 = 42.0 * c / r8;       // omega = c r^-6
    return p;
}

// grid 1, block 256.  f1[n1], f2[n2]: 0-based atoms.  out: sums[0] = A, sums[1] = B; gA[n], gB[n]
__global__ void __launch_bounds__(256) k_afir_pairs(int N, const double *__restrict__ x, const double *__restrict__ cov, int n1, const int *__restrict__ f1,
                                                    int n2, const int *__restrict__ f2, double *__restrict__ sums, double *__restrict__ gA,
                                                    double *__restrict__ gB) {
    __shared__ double red[2 * 16];
    const int tid = threadIdx.x;
    for (int k = tid; k < 3 * N; k += blockDim.x) { gA[k] = 0.0; gB[k] = 0.0; }
    __syncthreads();
    double v[2] = {0.0, 0.0};
    // fragment-1 atoms: sum over fragment 2 in list order (a thread per atom: fixed order, no atomics)
    for (int a = tid; a < n1; a += blockDim.x) {
        const int i = f1[a];
        double ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0};
        for (int b = 0; b < n2; ++b) {
            const int j = f2[b];
            const AfirPair p = afir_pair(x, cov, i, j);
            const double rs = (double)((float)cov[i] + (float)cov[j]);
            const double c = (rs * rs * rs) * (rs * rs * rs);
            const double r2 = p.r * p.r, r5 = r2 * r2 * p.r;
            v[0] += c / r5; v[1] += c / (r5 * p.r);
            ga[0] += p.da1 * p.ux; ga[1] += p.da1 * p.uy; ga[2] += p.da1 * p.uz;
            gb[0] += p.dw1 * p.ux; gb[1] += p.dw1 * p.uy; gb[2] += p.dw1 * p.uz;
        }
        for (int q = 0; q < 3; ++q) { gA[3 * i + q] += ga[q]; gB[3 * i + q] += gb[q]; }
    }
    __syncthreads();
    for (int b = tid; b < n2; b += blockDim.x) {
        const int j = f2[b];
        double ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0};
        for (int a = 0; a < n1; ++a) {
            const AfirPair p = afir_pair(x, cov, f1[a], j);
            ga[0] -= p.da1 * p.ux; ga[1] -= p.da1 * p.uy; ga[2] -= p.da1 * p.uz;
            gb[0] -= p.dw1 * p.ux; gb[1] -= p.dw1 * p.uy; gb[2] -= p.dw1 * p.uz;
        }
        for (int q = 0; q < 3; ++q) { gA[3 * j + q] += ga[q]; gB[3 * j + q] += gb[q]; }
    }
    block_sum<2>(v, red);
    if (tid == 0) { sums[0] = v[0]; sums[1] = v[1]; }
}

// grid N, block 128; dynamic LDS 18 N doubles (row blocks of d2A and d2B of atom m).  alpha by value.  Writes E (block 0), grad, hess.
__global__ void __launch_bounds__(128) k_afir_hess(int N, const double *__restrict__ x, const double *__restrict__ cov, int n1, const int *__restrict__ f1,
                                                   int n2, const int *__restrict__ f2, double alpha, const double *__restrict__ sums,
                                                   const double *__restrict__ gA, const double *__restrict__ gB, double *__restrict__ E,
                                                   double *__restrict__ grad, double *__restrict__ H) {
    extern __shared__ __attribute__((aligned(16))) double rowAB[];      // [2][3][3N]
    const int m = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, n = 3 * N;
    double *ra = rowAB, *rb = rowAB + 3 * n;
    for (int k = tid; k < 6 * n; k += nt) rowAB[k] = 0.0;
    __syncthreads();
    const double A = sums[0], B = sums[1];
    // pair blocks that involve atom m: thread 0 walks them in list order (few pairs per atom; keeps the sums deterministic)
    if (tid == 0) {
        for (int side = 0; side < 2; ++side) {
            const int *mine = side ? f2 : f1, *other = side ? f1 : f2;
            const int nm = side ? n2 : n1, no = side ? n1 : n2;
            for (int a = 0; a < nm; ++a) {
                if (mine[a] != m) continue;
                for (int b = 0; b < no; ++b) {
                    const int o = other[b];
                    const AfirPair p = side ? afir_pair(x, cov, o, m) : afir_pair(x, cov, m, o);
                    const double u[3] = {p.ux, p.uy, p.uz};
                    for (int q = 0; q < 3; ++q)
                        for (int s = 0; s < 3; ++s) {
                            const double P = u[q] * u[s], I = (q == s) ? 1.0 : 0.0;
                            const double ha = p.da2 * P + (p.da1 / p.r) * (I - P), hb = p.dw2 * P + (p.dw1 / p.r) * (I - P);
                            ra[q * n + 3 * m + s] += ha; ra[q * n + 3 * o + s] -= ha;
                            rb[q * n + 3 * m + s] += hb; rb[q * n + 3 * o + s] -= hb;
                        }
                }
            }
        }
    }
    __syncthreads();
    const double iB = 1.0 / B, iB2 = iB * iB, iB3 = iB2 * iB;
    for (int k = tid; k < 3 * n; k += nt) {
        const int q = k / n, col = k % n, row = 3 * m + q;
        const double h = ra[k] * iB - (gA[row] * gB[col] + gB[row] * gA[col]) * iB2 - A * rb[k] * iB2 + 2.0 * A * gB[row] * gB[col] * iB3;
        H[(size_t)row * n + col] = alpha * h;
    }
    if (tid < 3) { const int row = 3 * m + tid; grad[row] = alpha * (gA[row] * iB - A * gB[row] * iB2); }
    if (m == 0 && tid == 0) E[0] = alpha * A * iB;
}

}  // namespace mopt

// d_x [N][3], d_cov [N], d_f1 / d_f2: 0-based atom lists (device).  d_work: 2 + 6 N doubles.  Outputs on the device.
int mopt_launch_afir(mopt_ctx *ctx, int N, const double *d_x, const double *d_cov, int n1, const int *d_f1, int n2, const int *d_f2, double alpha,
                     double *d_work, double *d_E, double *d_grad, double *d_H) {
    using namespace mopt;
    mopt_scope sc(ctx, "bias_afir", 2);
    double *sums = d_work, *gA = d_work + 2, *gB = gA + 3 * N;
    hipLaunchKernelGGL(k_afir_pairs, dim3(1), dim3(256), 0, ctx->stream, N, d_x, d_cov, n1, d_f1, n2, d_f2, sums, gA, gB);
    const size_t lds = (size_t)18 * N * 8;
    if (lds > 48000) MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_afir_hess, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
    hipLaunchKernelGGL(k_afir_hess, dim3(N), dim3(128), lds, ctx->stream, N, d_x, d_cov, n1, d_f1, n2, d_f2, alpha, sums, gA, gB, d_E, d_grad, d_H);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
