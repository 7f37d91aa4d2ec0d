// Back-transformation V = Q Z of the reduction-route eigen-solver (the dormtr step of numpy.linalg.eigh, rsirfo.py:606): the
// eigenvectors of the tridiagonal form (rows of Zt) times the Householder reflectors left in the rows of A by the
// tridiagonalisation (k_tridiag.hip).  Its own translation unit: the unrolled MFMA instantiations compile for minutes.
//   k_tri_larft + k_tri_backtrans_mfma   compact-WY blocks of 16 reflectors on v_mfma_f64_16x16x4_f64, eigenvectors resident in
//                                        registers (n <= 608)
//   k_tri_backtrans                      vector-ALU form: 16 lanes per eigenvector, reflectors staged through LDS in blocks
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <cstdio>

#include "pipeline.h"

namespace mopt {

// ------------------------------------------------------------------------------------------------------------------
// V = Q Z.  grid (ceil(n/16), batch), block 256 = 4 waves x 4 vectors; each lane keeps NPER slices of its 4 vectors.
// V = Q Z.  Lane mapping: a wave carries FOUR eigenvectors, 16 lanes each; lane part p of a vector owns the components
// i = p + 16 u (u < NP16 = ceil(n / 16)).  The reflector dot product then needs only a 16-lane (one DPP row) reduction --
// four DPP steps shared by the four vectors of the wave -- instead of one full-wave reduction per vector, which was most of
// the instruction stream.  16 vectors per workgroup as before; the next reflector is fetched from L2 while the current one
// (in LDS) is applied.
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<0xB1, 0xF>(0.0, v);       // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xF>(0.0, v);       // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xF>(0.0, v);      // row_half_mirror
    v += dpp_f64<0x140, 0xF>(0.0, v);      // row_mirror: every lane of the 16-lane row holds the row's sum
    return v;
}

template <int NP16, int BK>
__global__ void __launch_bounds__(256) k_tri_backtrans(int n, const double *__restrict__ A_, const double *__restrict__ TAU_,
                                                       double *__restrict__ Zt_, const int *__restrict__ bend_, int dbg) {
    constexpr int LEN = 16 * NP16;                          // padded reflector length
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sv0 = sm, *sv1 = sm + (size_t)BK * LEN;         // two blocks of BK reflectors
    double *stau = sm + (size_t)2 * BK * LEN;               // [2][BK]
    const int b = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, tid = threadIdx.x;
    const int part = lane & 15, vsub = lane >> 4;
    const double *A = A_ + (size_t)b * n * n, *TAU = TAU_ + (size_t)b * n;
    double *Zt = Zt_ + (size_t)b * n * n;
    const int j = blockIdx.x * 16 + wid * 4 + vsub;         // eigenvector of this 16-lane group
    const long long c0 = (dbg & 32) ? clock64() : 0, w0 = (dbg & 32) ? wall_clock64() : 0;
    double z[NP16];
#pragma unroll
    for (int u = 0; u < NP16; ++u) {
        const int i = part + 16 * u;
        z[u] = (j < n && i < n) ? Zt[(size_t)j * n + i] : 0.0;
    }
    // Reflectors are staged BK at a time: the global loads of the next block (one L2/MALL latency, ~1 us) are in flight
    // while the current block is applied from LDS, and there is one barrier per block instead of one per reflector.
    constexpr int NLD = (BK * LEN + 255) / 256;
    // block kb covers reflectors k = kb, kb - 1, ..., kb - BK + 1 (slot c <-> k = kb - c)
    auto fetch = [&](int kb, double (&r)[NLD], double &tv) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int e = tid + 256 * m, c = e / LEN, i = e - c * LEN, k = kb - c;
            double v = 0.0;
            if (c < BK && k >= 0 && dbg != 2) { if (i == k + 1) v = 1.0; else if (i > k + 1 && i < n) v = A[(size_t)k * n + i]; }
            r[m] = v;
        }
        tv = (tid < BK && kb - tid >= 0) ? TAU[kb - tid] : 0.0;
    };
    auto stage = [&](int buf, const double (&r)[NLD], double tv) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) { const int e = tid + 256 * m; if (e < BK * LEN) (buf ? sv1 : sv0)[e] = r[m]; }
        if (tid < BK) stau[buf * BK + tid] = tv;
    };
    // reflector k acts on components k+1 .. n-1; applied in the order k = n-3 .. 0  (Q = H_0 H_1 ... H_{n-3}).
    // An eigenvector of the tridiagonal lives on its unreduced block [s, t) only, so reflectors with k + 1 >= t do not touch it:
    // the sweep starts at k = (largest block end among the 16 vectors of this workgroup) - 2.  Split spectra (the usual case on
    // this path) skip about half of the work; one full block skips nothing.
    int kfirst = n - 3;
    if (bend_) {
        int tmax = 0;
        for (int q = 0; q < 16; ++q) { const int jj = blockIdx.x * 16 + q; if (jj < n) { const int t = bend_[(size_t)b * n + jj]; tmax = t > tmax ? t : tmax; } }
        kfirst = (tmax - 2 < n - 3) ? tmax - 2 : n - 3;
    }
    double pre[NLD], ptau;
    fetch(kfirst, pre, ptau);
    stage(0, pre, ptau);
    __syncthreads();
    int buf = 0;
    for (int kb = kfirst; kb >= 0; kb -= BK) {
        fetch(kb - BK, pre, ptau);                             // in flight during the arithmetic below
        for (int c = 0; c < BK; ++c) {
            const int k = kb - c;
            if (k < 0) break;
            const double tau = stau[buf * BK + c];
            if (((dbg & 16) ? false : tau == 0.0) || dbg == 1) continue;
            const double *v = (buf ? sv1 : sv0) + (size_t)c * LEN;
            // all LDS reads first (pipelined), then the arithmetic from registers; components above the reflector's support
            // hold zeros in LDS, so no per-chunk predicate (a predicate puts an LDS wait in front of every FMA)
            double vv[NP16];
            if (dbg & 8) {
#pragma unroll
                for (int u = 0; u < NP16; ++u) vv[u] = 1e-3 * (u + part);
            } else {
#pragma unroll
                for (int u = 0; u < NP16; ++u) vv[u] = v[part + 16 * u];
            }
            double d4[4] = {0.0, 0.0, 0.0, 0.0};               // four independent chains: the FMA latency, not its issue rate, is the cost
#pragma unroll
            for (int u = 0; u < NP16; ++u) d4[u & 3] = fma(vv[u], z[u], d4[u & 3]);
            double dot = (d4[0] + d4[1]) + (d4[2] + d4[3]);
            dot = ((dbg & 4) ? dot : row16_sum(dot)) * tau;
#pragma unroll
            for (int u = 0; u < NP16; ++u) z[u] = fma(-dot, vv[u], z[u]);
        }
        stage(buf ^ 1, pre, ptau);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int u = 0; u < NP16; ++u) {
        const int i = part + 16 * u;
        if (j < n && i < n) Zt[(size_t)j * n + i] = z[u];
    }
    if ((dbg & 32) && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (b == 0 || b == gridDim.y - 1)) {
        const long long dc = clock64() - c0, dw = wall_clock64() - w0;
        printf("[bt_prof] wg (%d,%d): start %lld, %.1f us, shader clock %.0f MHz\n", (int)blockIdx.x, b, w0 % 100000000LL, dw * 0.01, 100.0 * dc / (dw ? dw : 1));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Blocked back-transformation on the FP64 matrix cores.  The reflectors are taken 16 at a time in compact WY form,
//   H_{k0} H_{k0+1} ... H_{k0+15} = I - V T V^T      (dlarft, forward / columnwise; V = n x 16, T upper triangular),
// so that for the eigenvectors z (rows of Zt)   z <- z - V (T (V^T z))   is three small matrix products, all of them
// shaped for v_mfma_f64_16x16x4_f64 (operand maps as in k_gemm.hip: A: lane l holds A[l&15][l>>4]; B: B[l>>4][l&15];
// C/D: col = l&15, row = (l>>4) + 4 reg).  A wave owns 16 eigenvectors for the whole sweep and keeps them in registers IN
// the B-operand layout: lane (i = l&15, h = l>>4) holds z_i[4 r + h], r = 0 .. 4 NT - 1.  Per block of 16 reflectors:
//   1.  W^T  = V^T Z^T :  acc[j][i] += V[4r+h][j] * z_i[4r+h]      one MFMA per r   (A from LDS, B = register r)
//   2.  W'^T = T  W^T  :  4 MFMAs; register m of the result of 1 holds rows j = 4m + h, exactly the k-slice MFMA m needs as B
//   3.  Z^T -= V W'^T  :  per 16-row tile tt 4 MFMAs with A = -V[16tt + (l&15)][4m + h], B = register m of 2, and the
//       accumulator = registers 4tt .. 4tt+3 of z: row (l>>4) + 4 reg of the tile is component 4 (4tt + reg) + h -- the layout z is kept in.
// The dot products, their reductions and the updates of the VALU kernel (2 x 19 FMAs + a 12-instruction DPP reduction per
// reflector and vector) become ~1 MFMA per 64 multiply-adds; no cross-lane instruction is left.

template <int... I, class F>
__device__ __forceinline__ void static_for_desc(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, (int)sizeof...(I) - 1 - I>{}), ...);         // f(N-1), f(N-2), ..., f(0)
}

// T factors: grid (NB, batch), 256 threads.  T[beta] is stored row-major 16 x 16 (zeros below the diagonal).
__global__ void __launch_bounds__(256) k_tri_larft(int n, int NB, const double *__restrict__ A_, const double *__restrict__ TAU_,
                                                   double *__restrict__ T_) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int ldv = n | 1;
    double *Vs = sm;                                         // [16][ldv]
    double *Gm = sm + (size_t)16 * ldv, *Tm = Gm + 16 * 17, *st = Tm + 16 * 17;     // Gram matrix, T, tau
    const int b = blockIdx.y, beta = blockIdx.x, tid = threadIdx.x, k0 = 16 * beta;
    const double *A = A_ + (size_t)b * n * n, *TAU = TAU_ + (size_t)b * n;
    {
        // a block of sixteen trivial reflectors (tau = 0) has T = 0: nothing to load or to form
        const double tv = (tid < 16 && k0 + tid <= n - 3) ? TAU[k0 + tid] : 0.0;
        if (!__syncthreads_or(tv != 0.0)) { T_[((size_t)b * NB + beta) * 256 + tid] = 0.0; return; }
    }
    for (int e = tid; e < 16 * n; e += 256) {
        const int c = e / n, i = e - c * n, k = k0 + c;
        double v = 0.0;
        if (k <= n - 3) { if (i == k + 1) v = 1.0; else if (i > k + 1) v = A[(size_t)k * n + i]; }
        Vs[c * ldv + i] = v;
    }
    if (tid < 16) st[tid] = (k0 + tid <= n - 3) ? TAU[k0 + tid] : 0.0;
    __syncthreads();
    {
        const int c1 = tid >> 4, c2 = tid & 15;
        double g = 0.0;
        if (c1 < c2) { const double *x = Vs + c1 * ldv, *y = Vs + c2 * ldv; for (int i = k0 + c2 + 1; i < n; ++i) g = fma(x[i], y[i], g); }
        Gm[c1 * 17 + c2] = g;
        Tm[c1 * 17 + c2] = 0.0;
    }
    __syncthreads();
    // T(0:c, c) = -tau_c T(0:c, 0:c) (V(:, 0:c)^T v_c),  T(c, c) = tau_c
    for (int c = 0; c < 16; ++c) {
        if (tid < c) {
            double a = 0.0;
            for (int j = tid; j < c; ++j) a = fma(Tm[tid * 17 + j], Gm[j * 17 + c], a);
            Tm[tid * 17 + c] = -st[c] * a;
        } else if (tid == c) Tm[c * 17 + c] = st[c];
        __syncthreads();
    }
    T_[((size_t)b * NB + beta) * 256 + tid] = Tm[(tid >> 4) * 17 + (tid & 15)];
}

template <int NT, int WPW>
__global__ void __launch_bounds__(64 * WPW) k_tri_backtrans_mfma(int n, int NB, const double *__restrict__ A_, const double *__restrict__ T_,
                                                            const double *Zt_, const int *__restrict__ bend_, double *Vout_, const double *__restrict__ TAU_) {
    // (Vout_ may be Zt_ itself: a wave reads its 16 vectors completely before it writes them)
    __shared__ int s_kact;
    constexpr int LEN = 16 * NT;                                           // padded vector length
    constexpr int LDV = LEN + ((2 - LEN % 32 + 32) % 32);                   // = 2 mod 32: the 16 x 2 operand reads of step 1 hit 32 different bank pairs
    constexpr int NTHR = 64 * WPW, NTL = 256 / NTHR;      // T entries per thread
    typedef double d4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Vs = sm;                                                        // [16][LDV]
    double *Ts = sm + (size_t)16 * LDV;                                     // [16][17]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lh = lane >> 4;
    const double *A = A_ + (size_t)b * n * n;
    const double *Tg = T_ + (size_t)b * NB * 256;
    const double *Zt = Zt_ + (size_t)b * n * n;
    double *Vout = Vout_ + (size_t)b * n * n;
    const int j0 = blockIdx.x * (16 * WPW) + wid * 16;                              // first eigenvector of this wave
    double z[4 * NT];
#pragma unroll
    for (int r = 0; r < 4 * NT; ++r) { const int k = 4 * r + lh; z[r] = (j0 + li < n && k < n) ? Zt[(size_t)(j0 + li) * n + k] : 0.0; }
    // an eigenvector lives on its unreduced block [s, t) of the tridiagonal: reflectors k >= t - 1 do not touch it
    int beta_first = NB - 1;
    if (bend_) {
        int tmax = 0;
        for (int q = 0; q < 16 * WPW; ++q) { const int jj = blockIdx.x * (16 * WPW) + q; if (jj < n) { const int t = bend_[(size_t)b * n + jj]; tmax = t > tmax ? t : tmax; } }
        const int kf = tmax - 2 < 0 ? 0 : tmax - 2;
        beta_first = (kf / 16 < NB - 1) ? kf / 16 : NB - 1;
    }
    // blocks behind the last non-trivial reflector are the identity (tau = 0: a tridiagonalisation that stopped early on a diagonal
    // trailing block, k_tri_persist3, or rows that needed no reflector): skipped as well
    if (TAU_) {
        if (tid == 0) s_kact = -1;
        __syncthreads();
        int last = -1;
        for (int i = tid; i <= n - 3; i += NTHR) if (TAU_[(size_t)b * n + i] != 0.0) last = i;
        if (last >= 0) atomicMax(&s_kact, last);
        __syncthreads();
        const int bl = s_kact < 0 ? -1 : s_kact / 16;
        beta_first = bl < beta_first ? bl : beta_first;
    }
    // Staging map (WPW = 4: 256 threads): thread t carries row c = t / 16 of the block and the columns t % 16 + 16 m, one per
    // 16-column tile m -- 128-byte row segments, addresses that differ by constants, and tiles below the block's first
    // reflector (structural zeros, never read) are not touched at all.
    static_assert(WPW == 4, "staging map assumes 256 threads");
    const int sc = tid >> 4, si = tid & 15;
    auto fetch = [&](auto bc, double (&pre)[NT], double (&pt)[NTL]) {
        constexpr int BETA = decltype(bc)::value;
        if constexpr (BETA >= 0) {
            const int k = 16 * BETA + sc;
            const double *row = A + (size_t)(k <= n - 3 ? k : 0) * n;
#pragma unroll
            for (int m = (BETA > 0 ? BETA : 0); m < NT; ++m) {
                const int i = si + 16 * m;
                const double a = row[i < n ? i : n - 1];
                pre[m] = (k <= n - 3 && i > k + 1 && i < n) ? a : ((k <= n - 3 && i == k + 1) ? 1.0 : 0.0);
            }
#pragma unroll
            for (int m = 0; m < NTL; ++m) pt[m] = Tg[(size_t)BETA * 256 + tid + NTHR * m];
        }
    };
    auto stage = [&](auto bc, const double (&pre)[NT], const double (&pt)[NTL]) {
        constexpr int BETA = decltype(bc)::value;
#pragma unroll
        for (int m = (BETA > 0 ? BETA : 0); m < NT; ++m) Vs[sc * LDV + si + 16 * m] = pre[m];
#pragma unroll
        for (int m = 0; m < NTL; ++m) { const int e = tid + NTHR * m; Ts[(e >> 4) * 17 + (e & 15)] = pt[m]; }
    };
    double pre[NT], pt[NTL];
    // the first block to run is found at run time: its fetch is the one non-overlapped load of the sweep
    static_for_desc(std::make_integer_sequence<int, NT>{}, [&](auto bc) { if (decltype(bc)::value == beta_first) fetch(bc, pre, pt); });
    // The sweep over the blocks is unrolled at compile time (block BETA touches components >= 16 BETA + 1 only, i.e. the
    // 16-row tiles tt >= BETA: half of the MFMAs of the rectangular form act on structural zeros), so every index into z
    // is static; blocks above beta_first are skipped by a uniform test.
    static_for_desc(std::make_integer_sequence<int, NT>{}, [&](auto bc) {
        constexpr int BETA = decltype(bc)::value;
        if (BETA <= beta_first) {
            __syncthreads();                                                // the previous block's operand reads are done
            stage(bc, pre, pt);
            __syncthreads();
            fetch(std::integral_constant<int, BETA - 1>{}, pre, pt);       // in flight during the arithmetic below
            // 1. W^T = V^T Z^T   (two accumulators: the MFMAs of one chain depend on each other)
            // (the LDS operands are requested one group of four MFMAs ahead: a read issued right in front of its MFMA leaves
            // the matrix pipe idle for the LDS latency)
            d4 w0 = {0.0, 0.0, 0.0, 0.0}, w1 = {0.0, 0.0, 0.0, 0.0};
#ifdef MOPT_BT_ABL_NOLDSOP
            const double *va = Vs + li, *vb = Vs + lh;       // (timing ablation: every operand read hits the same few words)
#define BT_IDX(x) 0
#else
            const double *va = Vs + li * LDV + lh;
            const double *vb = Vs + lh * LDV + li;
#define BT_IDX(x) (x)
#endif
            double tq[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) tq[m] = Ts[li * 17 + 4 * m + lh];
            {
                double ac[4], an[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ac[u] = va[BT_IDX(4 * (4 * BETA + u))];
#pragma unroll
                for (int tt = BETA; tt < NT; ++tt) {
                    if (tt + 1 < NT) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) an[u] = va[BT_IDX(4 * (4 * (tt + 1) + u))];
                    }
                    __builtin_amdgcn_sched_barrier(0);       // keep the reads ahead of the MFMAs (the scheduler sinks them to their uses)
                    w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[0], z[4 * tt], w0, 0, 0, 0);
                    w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[1], z[4 * tt + 1], w1, 0, 0, 0);
                    w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[2], z[4 * tt + 2], w0, 0, 0, 0);
                    w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[3], z[4 * tt + 3], w1, 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) ac[u] = an[u];
                }
            }
            w0 += w1;
            // 2. W'^T = T W^T
            d4 x = {0.0, 0.0, 0.0, 0.0};
            double bc[4], bn[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) bc[m] = vb[BT_IDX((size_t)4 * m * LDV + 16 * BETA)];
#pragma unroll
            for (int m = 0; m < 4; ++m) x = __builtin_amdgcn_mfma_f64_16x16x4f64(tq[m], w0[m], x, 0, 0, 0);
            // 3. Z^T -= V W'^T
#pragma unroll
            for (int tt = BETA; tt < NT; ++tt) {
                if (tt + 1 < NT) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) bn[m] = vb[BT_IDX((size_t)4 * m * LDV + 16 * (tt + 1))];
                }
                __builtin_amdgcn_sched_barrier(0);
                d4 c = {z[4 * tt], z[4 * tt + 1], z[4 * tt + 2], z[4 * tt + 3]};
#pragma unroll
                for (int m = 0; m < 4; ++m) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-bc[m], x[m], c, 0, 0, 0);
                z[4 * tt] = c[0]; z[4 * tt + 1] = c[1]; z[4 * tt + 2] = c[2]; z[4 * tt + 3] = c[3];
#pragma unroll
                for (int m = 0; m < 4; ++m) bc[m] = bn[m];
            }
        }
    });
#pragma unroll
    for (int r = 0; r < 4 * NT; ++r) { const int k = 4 * r + lh; if (j0 + li < n && k < n) Vout[(size_t)(j0 + li) * n + k] = z[r]; }
}

}  // namespace mopt

// eigenvectors of a wave live in registers, 4 * ceil(n / 16) doubles per lane (one wave per SIMD: 512 registers per lane)
bool mopt_bt_use_mfma(int n) {
    const int nt16 = (n + 15) / 16;
    return !(getenv("MOPT_BT_MFMA") && atoi(getenv("MOPT_BT_MFMA")) == 0) && nt16 <= 38 && n >= 16;
}

int mopt_launch_bt_larft(mopt_ctx *ctx, hipStream_t st, int n, int batch, const double *d_A, const double *d_TAU, double *d_T) {
    using namespace mopt;
    const int NB = (n - 2 + 15) / 16;
    const size_t lds = ((size_t)16 * (n | 1) + 2 * 16 * 17 + 16) * 8;
    if (lds > 48000) MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_tri_larft, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
    hipLaunchKernelGGL(k_tri_larft, dim3(NB, batch), dim3(256), lds, st, n, NB, d_A, d_TAU, d_T);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

// d_Vout: where V goes (null: in place; another buffer only with the blocked form)
int mopt_launch_backtrans(mopt_ctx *ctx, int n, int batch, const double *d_A, const double *d_TAU, const double *d_T, double *d_Vt, const int *d_bend, bool mfma,
                          double *d_Vout) {
    if (!d_Vout) d_Vout = d_Vt;
    if (!mfma && d_Vout != d_Vt) { ctx->err = "back-transformation: a separate output needs the blocked form"; return MOPT_E_ARG; }
    using namespace mopt;
    const int nt16 = (n + 15) / 16;
    mopt_scope sc(ctx, "eigh_backtrans", 1);
    if (mfma) {
        // blocked form on the matrix cores; the T factors were formed on the side stream
        const int NB = (n - 2 + 15) / 16;
        // 4 waves (16 eigenvectors each) per workgroup; measured with 2 / 1 waves (twice / four times the staging work per
        // thread, 320 / 608 workgroups): 0.58 / 1.11 ms against 0.148 ms
#define BTW(NT_, W_) do { constexpr int LEN_ = 16 * NT_, LDV_ = LEN_ + ((2 - LEN_ % 32 + 32) % 32); const size_t lds = ((size_t)16 * LDV_ + 16 * 17) * 8; \
        MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_tri_backtrans_mfma<NT_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024)); \
        hipLaunchKernelGGL((k_tri_backtrans_mfma<NT_, W_>), dim3((n + 16 * W_ - 1) / (16 * W_), batch), dim3(64 * W_), lds, ctx->stream, n, NB, d_A, d_T, (const double *)d_Vt, d_bend, d_Vout, d_TAU); } while (0)
#define BTM(NT_) BTW(NT_, 4)
        if (nt16 <= 4) BTM(4); else if (nt16 <= 8) BTM(8); else if (nt16 <= 12) BTM(12); else if (nt16 <= 16) BTM(16); else if (nt16 <= 19) BTM(19);
        else if (nt16 <= 24) BTM(24); else if (nt16 <= 32) BTM(32); else BTM(38);
#undef BTM
#undef BTW
    } else {
        const int np16 = (n + 15) / 16;
        const dim3 g((n + 15) / 16, batch);
        const int bt_dbg = getenv("MOPT_BT_DBG") ? atoi(getenv("MOPT_BT_DBG")) : 0;      // developer switch: 1 = no arithmetic, 2 = no reflector loads
        // BK reflectors per staged block, chosen so that the two blocks stay near 20 KB of LDS: with 39 KB (BK = 8 at n = 300) only
        // two workgroups fit a CU and the 608 workgroups of the headline band ran in two rounds (0.30 ms); BK = 4: one round, 0.24 ms
#define BT(NP, BK) do { const size_t lds = ((size_t)2 * BK * 16 * NP + 2 * BK) * 8; \
        hipLaunchKernelGGL((k_tri_backtrans<NP, BK>), g, dim3(256), lds, ctx->stream, n, d_A, d_TAU, d_Vt, d_bend, bt_dbg); } while (0)
        if (np16 <= 2) BT(2, 8); else if (np16 <= 4) BT(4, 8); else if (np16 <= 6) BT(6, 8); else if (np16 <= 8) BT(8, 8); else if (np16 <= 10) BT(10, 8);
        else if (np16 <= 12) BT(12, 6); else if (np16 <= 14) BT(14, 5); else if (np16 <= 16) BT(16, 5); else if (np16 <= 19) BT(19, 4); else if (np16 <= 24) BT(24, 3);
        else if (np16 <= 32) BT(32, 2); else if (np16 <= 40) BT(40, 2); else if (np16 <= 48) BT(48, 1); else if (np16 <= 64) BT(64, 1); else BT(96, 1);
#undef BT
    }
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
