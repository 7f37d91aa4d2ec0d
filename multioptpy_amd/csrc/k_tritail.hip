// Householder tridiagonalisation of a trailing block of at most 192 rows inside ONE compute unit (no cross-workgroup hand-off).
//
// Reference operation: the `eigh` of RSIRFO.run (multioptpy/Optimizer/rsirfo.py:553-657 -> numpy.linalg.eigh -> LAPACK dsytrd).
//
// k_tri_persist3 (k_tridiag.hip) spreads a matrix over G >= 8 workgroups and pays one exchange through the XCD's L2 per Householder
// step (~2 us at n = 300, of which 0.36 us is arithmetic).  Once the trailing block has m <= 192 rows it fits the registers of one
// workgroup: 16 waves, wave w owns local rows w, w + 16, ... (TR slots), a lane holds columns lane + 64 q (CH chunks) of each of
// them, i.e. TR x CH <= 36 doubles.  A step then costs two workgroup barriers and the single-wave reflector instead of the exchange:
//   a. every wave forms w = tau p + c v in registers from the p = A v that the previous pass left in LDS (redundantly: no broadcast),
//   b. the wave that owns the pivot row updates it, forms the next reflector v' (one DPP reduction), leaves it in LDS and stores the
//      step's results (D, E, TAU, reflector in dsytrd layout) to global memory,
//   c. barrier; all waves: rank-2 update of the live rows with (v, w) fused with p_i = row_i . v' (row-wise product, both triangles
//      are kept current, as in k_tri_persist3); the row sums of FOUR rows are reduced together -- two half-wave / row swaps
//      (v_permlane32_swap, v_permlane16_swap) fold four accumulators into one register whose 16-lane rows then finish with four DPP
//      stages: 21 instructions per four rows instead of 4 x 18,
//   d. barrier.
// The kernel either reduces a whole matrix (K = -1, n <= 192) or takes over from k_tri_persist3 stopped behind pivot row K: rows
// K + 1 .. n - 1 of A current up to the reflector of row K - 1, the reflector of row K in A[K][K + 2 ..] (v[K + 1] = 1) and TAU[K].
// Output contract = k_tri_persist3's: reflector of pivot row lo in A[lo][lo + 2 ..], D, E (E[n - 1] = 0), TAU (TAU[n - 2] = TAU[n - 1] = 0).
// grid = batch, block = 1024, static LDS only.  Deterministic: fixed summation order, independent of the batch slot.

#include "pipeline.h"
#include "tri_tail_body.hpp"

namespace mopt {

template <int CH, int NW>
__global__ void __launch_bounds__(64 * NW) k_tri_tail(int n, int K, double *__restrict__ A_, double *__restrict__ D_, double *__restrict__ E_,
                                                      double *__restrict__ TAU_, long long *__restrict__ prof, const unsigned long long *__restrict__ defl,
                                                      size_t defl_stride) {
    // (the persistent kernel in front stopped early for this matrix: its trailing block is diagonal and D, E, TAU are complete)
    if (defl && defl[(size_t)blockIdx.x * defl_stride] != 0ull) return;
    extern __shared__ __attribute__((aligned(16))) double tsm[];
    tri_tail_body<CH, NW>(n, K, (int)blockIdx.x, A_, D_, E_, TAU_, prof, tsm);
}

}  // namespace mopt

// Tridiagonalises rows / columns K + 1 .. n - 1 of every matrix of the batch (K = -1: the whole matrix), n - K - 1 <= 192.
int mopt_tri_tail_max() { return 192; }
int mopt_launch_tri_tail(mopt_ctx *ctx, hipStream_t st, int n, int K, int batch, double *d_A, double *d_D, double *d_E, double *d_TAU,
                         long long *d_prof, const unsigned long long *d_defl, size_t defl_stride) {
    using namespace mopt;
    static const bool W8 = !(getenv("MOPT_TAIL_W8") && atoi(getenv("MOPT_TAIL_W8")) == 0);
    const int m = n - K - 1;
    if (m < 2 || m > 192 || K < -1) { ctx->err = "k_tri_tail: trailing block must have 2 .. 192 rows"; return MOPT_E_ARG; }
    const int ch = (m + 63) / 64;
    const size_t lds = ((size_t)5 * 64 * ch + (size_t)(m - 2) * (m - 1) / 2 + 2) * 8;
    const void *fn = ch == 1 ? (const void *)k_tri_tail<1, 16> : ch == 2 ? (const void *)k_tri_tail<2, 16> : W8 ? (const void *)k_tri_tail<3, 8> : (const void *)k_tri_tail<3, 16>;
    const int nthreads = (ch == 3 && W8) ? 512 : 1024;
    if (lds > 64 * 1024) MOPT_HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int n_ = n, K_ = K;
    void *args[] = {&n_, &K_, &d_A, &d_D, &d_E, &d_TAU, &d_prof, &d_defl, &defl_stride};
    MOPT_HIP_TRY(ctx, hipLaunchKernel(fn, dim3(batch), dim3(nthreads), args, lds, st));
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
