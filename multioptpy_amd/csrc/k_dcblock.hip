// Divide and conquer for ONE unreduced block of the tridiagonal form inside ONE workgroup (round 3).
//
// Why: the blocks of 13 .. 72 rows are what the small problems of this path consist of -- the whole matrix of a 7- to 24-atom
// geometry (BASELINE's optmain configurations, the 16 x 14 band) and the leading block of an identity + low-rank Hessian once a NEB run
// is a dozen steps old.  The one-wave implicit QL (ql_block.hpp) needs ~0.85 m^2 dependent rotations of 0.27 us (m = 25: 0.14 ms, the
// longest kernel of a 16 x 14 band step), the launch-per-stage chain of k_dc.hip costs 4 + 5 L launches (0.14 ms of a 0.40 ms
// RSIRFO.run at n = 33).  Here the whole tree lives in LDS: leaves of at most 10 rows by the wave-level QL (all leaves side by side, one
// wave each), then the merges of Cuppen's method with Gu and Eisenstat's eigenvector formula, stage by stage behind workgroup barriers:
//   prep     z from the eigenvector rows at the cut, rank sort, deflation (the parallel layout of k_dc_prep; scalar walk: dc_logic.h)
//   secular  one root per 16-lane group, 64 groups at once (dc_logic.h dc_secular); the row d_i - lambda_j is written straight into S
//   zhat     Loewner weights, thread = pole
//   ufill    rows of the k x k eigenvector matrix, in place of S (position-permuted columns); deflated vectors copied
//   gemm     Xnext = U * Xcur, plain FMAs (at most 72^3: the matrix cores would idle on the staging)
// Same mathematics and the same scalar code as k_dc.hip (LAPACK's published dstedc family restated; host-side checks: tests/test_dc_hostsim.py).
// One workgroup of 1024 threads per block; grid (DCB_SLOTS, batch): workgroup (q, b) takes the q-th eligible block of matrix b.
#include <cstdlib>

#include "dc_block_body.hpp"

namespace mopt {

__global__ void __launch_bounds__(1024) k_dc_block(int n, int mmin, int mmax, const double *__restrict__ D_, const double *__restrict__ E_,
                                                   const int *__restrict__ bstart_, const int *__restrict__ bend_, double *__restrict__ WT_,
                                                   double *__restrict__ Zt_, int *__restrict__ fail_, const int *__restrict__ dcflag_,
                                                   const double *__restrict__ stage_, const int *__restrict__ early_, int dbg) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int s_blk;
    const int b = blockIdx.y, tid = threadIdx.x;
    if (dcflag_ && dcflag_[b]) return;                       // (the whole matrix goes through the launch-per-stage chain)
    const int *bstart = bstart_ + (size_t)b * n, *bend = bend_ + (size_t)b * n;
    // ---- the blockIdx.x-th eligible block of this matrix (block starts are the rows i with bstart[i] == i)
    if (tid == 0) s_blk = -1;
    __syncthreads();
    {
        // rank of an eligible block start = number of eligible starts in front of it; n <= 1500: two rows per thread at most
        for (int i = tid; i < n; i += blockDim.x) {
            const int t = bend[i];
            if (bstart[i] == i && t - i >= mmin && t - i <= mmax) {
                int r = 0;
                for (int j = 0; j < i;) { const int tj = bend[j]; if (tj <= j) break; r += (tj - j >= mmin && tj - j <= mmax) ? 1 : 0; j = tj; }
                if (r == (int)blockIdx.x) s_blk = i;
            }
        }
    }
    __syncthreads();
    const int s0 = s_blk;
    if (s0 < 0) return;
    const int m = bend[s0] - s0;
    double *Zt = Zt_ + (size_t)b * n * n;
    double *WT = WT_ + (size_t)b * n;
    // ---- the leading block may have been solved next to the tridiagonalisation already (k_early.hip)
    if (early_ && s0 == 0 && early_[b] == m && m <= 32) {
        const double *stage = stage_ + (size_t)b * MOPT_EARLY_STAGE;
        if (tid < m) WT[tid] = stage[tid];
        for (int e = tid; e < m * n; e += blockDim.x) {
            const int c = e / n, i = e - c * n;
            Zt[(size_t)c * n + i] = (i < m) ? stage[32 + c * 32 + i] : 0.0;
        }
        return;
    }
    dc_block_body(n, s0, m, b, D_ + (size_t)b * n + s0, E_ + (size_t)b * n + s0, WT, Zt, fail_ + b, sm, dbg);
}

}  // namespace mopt

int mopt_dc_block_max() { return mopt::DCB_MAX; }

// Every unreduced block with mmin <= rows <= mmax (<= DCB_MAX) of every matrix that is not flagged for the launch-per-stage chain.
int mopt_launch_dc_block(mopt_ctx *ctx, hipStream_t st, int n, int batch, int mmin, int mmax, const double *d_D, const double *d_E, const int *d_bstart,
                         const int *d_bend, double *d_WT, double *d_Zt, int *d_fail, const int *d_dcflag, const double *d_stage, const int *d_early) {
    using namespace mopt;
    if (mmax > DCB_MAX) mmax = DCB_MAX;
    if (mmin < 2 || mmin > mmax) return MOPT_OK;
    const int slots = n / mmin > 0 ? n / mmin : 1;           // a matrix cannot hold more blocks of at least mmin rows
    const int mcap = mmax < n ? mmax : n;
    // three m x m tables, ten m-vectors, rotations, candidate words, eight index arrays; the 16 leaf tables (40 KB) overlap X1 and S
    // (worst case over the block sizes 2 .. mcap: small arrays + X0 + max(X1 + S, leaf tables))
    const size_t small = (size_t)10 * mcap * 8 + (size_t)mcap * sizeof(DcRot) + 8 * 8 + (size_t)8 * mcap * 4 + 64;
    const size_t tabs = (size_t)3 * mcap * mcap * 8, leaf = ((size_t)mcap * mcap + (size_t)16 * 32 * DCB_LEAF) * 8;
    const size_t lds = small + (tabs > leaf ? tabs : leaf);
    MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dc_block, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
    if (lds > (size_t)ctx->lds_limit - 1024) { ctx->err = "k_dc_block: block too large for the LDS"; return MOPT_E_ARG; }
    hipLaunchKernelGGL(k_dc_block, dim3(slots, batch), dim3(1024), lds, st, n, mmin, mmax, d_D, d_E, d_bstart, d_bend, d_WT, d_Zt, d_fail, d_dcflag, d_stage,
                       d_early, getenv("MOPT_DCB_DBG") ? 1 : 0);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
