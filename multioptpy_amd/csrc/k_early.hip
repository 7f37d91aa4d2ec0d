// Early start for the leading block of the tridiagonal form.
//
// On the NEB path the Hessians are identity + low rank: their tridiagonal form is ONE unreduced leading block of 13-25 rows followed by
// rows that are decoupled one by one.  Its eigen-decomposition (implicit QL, one wave per matrix: ~m^2 dependent rotations of 0.27 us,
// 100-250 us) used to start behind the tridiagonalisation and was the longest thing between it and the back-transformation -- while
// the rows it needs are final after the first m + 1 of the n - 1 Householder steps.  k_tri_early is launched NEXT TO the persistent
// tridiagonalisation (k_tri_persist3 leaves more than half of every CU's registers and wave slots free): one wave per matrix polls
// D and E -- of eight candidate workgroups per matrix the one on the XCD of the matrix' tridiagonalisation group stays, so its
// agent-scope loads find the group's plain stores in that XCD's L2, exactly as the members' hand-off does; from another XCD they
// are not visible before the kernel ends.  (Write-through stores of D, E in k_tri_persist3 instead were
// tried: with them the kernel's hand-off timed out once in ~40 steps.)  Both arrays carry an all-ones sentinel beforehand; the wave
// finds the first split with the criterion of k_tri_bisect, runs ql_block32 and leaves eigenvalues and the m x m eigenvector table in
// a staging area.  k_tri_ql32 later takes the staged result for block 0 when its length matches (same data, same criterion: it always
// does) instead of iterating.  A wave that does not see a split within the first 33 rows, or no data within its spin limit, gives up
// (early[b] = 0) and everything runs as before.  Reference operation: numpy.linalg.eigh of RSIRFO.run (rsirfo.py:553-657).
#include "pipeline.h"
#include "ql_block.hpp"

namespace mopt {

constexpr double EARLY_DEPS = 2.220446049250313e-16, EARLY_DTINY = 2.2250738585072014e-308;
constexpr unsigned long long EARLY_SENTINEL = 0xFFFFFFFFFFFFFFFFull;

__global__ void __launch_bounds__(64) k_tri_early(int n, int batch, const double *D_, const double *E_, const unsigned long long *XCH_, size_t xch_stride,
                                                  double *__restrict__ stage_, int *__restrict__ early_, unsigned spin_limit) {
    __shared__ double Zs[32 * 32];
    const int lane = threadIdx.x;
    // Eight candidate workgroups per matrix, consecutive ids = one on every XCD: the one that sits on the XCD of the matrix'
    // tridiagonalisation group (published by the group's first member in its exchange area, k_tri_persist3) goes on, the others leave.
    const int b = (int)blockIdx.x >> 3;
    if (b >= batch) return;
    {
        const unsigned long long *XID = XCH_ + (size_t)b * xch_stride + (size_t)8 * n;
        const unsigned myx = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;      // HW_REG_XCC_ID[3:0]
        unsigned long long w = 0;
        for (unsigned spin = 0;; ++spin) {
            w = __hip_atomic_load(&XID[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == 0x5EEDu || spin >= spin_limit) break;
            __builtin_amdgcn_s_sleep(32);
        }
        if ((unsigned)(w >> 32) != 0x5EEDu || (unsigned)(w & 0xF) != myx) return;
    }
    const unsigned long long *D = (const unsigned long long *)(D_ + (size_t)b * n), *E = (const unsigned long long *)(E_ + (size_t)b * n);
    double *stage = stage_ + (size_t)b * MOPT_EARLY_STAGE;
    int m = 0;
    double d = 0.0, e = 0.0;
    const int rows = n < 34 ? n : 34;
    for (unsigned spin = 0;; ++spin) {
        unsigned long long du = EARLY_SENTINEL, eu = EARLY_SENTINEL;
        if (lane < rows) {
            du = __hip_atomic_load(&D[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            eu = __hip_atomic_load(&E[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool have = (du != EARLY_SENTINEL) && (eu != EARLY_SENTINEL);
        const unsigned long long hm = __ballot(have);
        const int p = (~hm == 0ull) ? 64 : __ffsll((long long)~hm) - 1;      // rows 0 .. p - 1 are there
        const double dv = __longlong_as_double((long long)du), ev = __longlong_as_double((long long)eu);
        const double dn = __shfl_down(dv, 1, WAVE);
        const bool cut = (lane + 1 < p) && (fabs(ev) <= 32.0 * EARLY_DEPS * (fabs(dv) + fabs(dn)) || fabs(ev) < EARLY_DTINY);
        const unsigned long long cm = __ballot(cut);
        if (cm) {
            const int first = __ffsll((long long)cm) - 1;     // the block is rows 0 .. first
            if (first <= 31) { m = first + 1; d = dv; e = (lane < first) ? ev : 0.0; }
            break;
        }
        if (p >= rows || spin >= spin_limit) break;           // no split within the rows watched, or no data: the usual path takes the block
        __builtin_amdgcn_s_sleep(32);
    }
    if (m < 2) return;                                      // (early_[b] stays 0)
    if (lane >= m) { d = 0.0; e = 0.0; }
    const double nrm = wave_max(fmax(fabs(d), fabs(e)));
    const double sc = (nrm > 0.0) ? nrm : 1.0;
    d /= sc; e /= sc;
    for (int c = 0; c < m; ++c) Zs[c * 32 + (lane & 31)] = ((lane & 31) == c) ? 1.0 : 0.0;
    __syncthreads();
    const bool ok = ql_block32(m, d, e, Zs, lane);
    __syncthreads();
    if (lane < m) stage[lane] = d * sc;
    for (int i = lane; i < 32 * m; i += WAVE) stage[32 + i] = Zs[i];
    if (lane == 0) early_[b] = ok ? m : 0;
}

}  // namespace mopt

// d_stage: batch * MOPT_EARLY_STAGE doubles, d_early: batch ints.  D and E must hold the all-ones sentinel wherever the tridiagonalisation
// has not stored yet.
int mopt_launch_tri_early(mopt_ctx *ctx, hipStream_t st, int n, int batch, const double *d_D, const double *d_E, const void *d_xch, size_t xch_stride_u64,
                          double *d_stage, int *d_early) {
    using namespace mopt;
    MOPT_HIP_TRY(ctx, hipMemsetAsync(d_early, 0, (size_t)batch * sizeof(int), st));
    hipLaunchKernelGGL(k_tri_early, dim3(8 * batch), dim3(64), 0, st, n, batch, d_D, d_E, (const unsigned long long *)d_xch, xch_stride_u64, d_stage, d_early, 2500u);      // (a few ms of polling at most: the rows it needs are there after ~100 us)
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
