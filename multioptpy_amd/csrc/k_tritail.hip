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

namespace mopt {

// 1 / sqrt(x) and 1 / x for normal x of moderate size: hardware estimate + two Newton steps (a few ulp)
__device__ __forceinline__ double tail_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(x * y), y, 1.0);
    return fma(0.5 * y, e, y);
}
__device__ __forceinline__ double tail_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// NW waves per workgroup (16: 128 registers per lane, 8: 256 registers per lane -- the 3 x 24 tile of m <= 192 with its operands)
template <int CH, int NW>
__global__ void __launch_bounds__(64 * NW) k_tri_tail(int n, int K, double *__restrict__ A_, double *__restrict__ D_, double *__restrict__ E_,
                                                      double *__restrict__ TAU_, long long *__restrict__ prof, const unsigned long long *__restrict__ defl,
                                                      size_t defl_stride) {
    // (the persistent kernel in front stopped early for this matrix: its trailing block is diagonal and D, E, TAU are complete)
    if (defl && defl[(size_t)blockIdx.x * defl_stride] != 0ull) return;
#ifdef MOPT_TAIL_PROF
    // developer build (tools/ubench/tail_main.hip): shader-clock stamps of wave 0 around the phases of a step, summed over the steps
    long long tp[6] = {0, 0, 0, 0, 0, 0}, tc0 = 0, tc1 = 0;
    int prof_l = -1;
#define TAIL_STAMP(k) do { tc1 = clock64(); tp[k] += tc1 - tc0; tc0 = tc1; \
        if (prof && prof_l == 5 && lane == 0 && blockIdx.x == 0) prof[(size_t)gridDim.x * 8 + (size_t)(tid >> 6) * 8 + (k) + 1] = tc1; } while (0)
#else
#define TAIL_STAMP(k) do { } while (0)
#endif
    constexpr int NS = 64 * CH, TR = NS / NW, SPC = 64 / NW;      // SPC slots of a wave per column chunk
    static_assert(TR % 4 == 0, "rows are reduced four at a time");
    // LDS: p and the current reflector (NS each), then the results of all steps -- D, E, TAU (NS each) and the reflectors packed as a
    // triangle (row l: columns l + 2 .. m - 1) -- which go to global memory in one coalesced flush behind the last step: a global
    // store inside the loop costs the owner wave its address arithmetic and every later scalar clock / wait on it
    extern __shared__ __attribute__((aligned(16))) double tsm[];
    double *sp = tsm, *sz = tsm + NS, *sD = tsm + 2 * NS, *sE = tsm + 3 * NS, *sT = tsm + 4 * NS, *sZ = tsm + 5 * NS;
    __shared__ double s_tau;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = K + 1, m = n - base;                     // local index c <-> global index base + c
    double *A = A_ + (size_t)b * n * n;
    double *D = D_ + (size_t)b * n, *E = E_ + (size_t)b * n, *TAU = TAU_ + (size_t)b * n;
    double areg[TR][CH];                                      // slot t: local row wid + NW t
#pragma unroll
    for (int t = 0; t < TR; ++t) {
        const int i = wid + NW * t;
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int c = lane + 64 * q; areg[t][q] = (i < m && c < m) ? A[(size_t)(base + i) * n + base + c] : 0.0; }
    }
    double vj[CH], wj[CH], zj[CH];
    double tau_z = (K >= 0) ? TAU[K] : 0.0;
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        const int c = lane + 64 * q;
        vj[q] = 0.0; wj[q] = 0.0;
        zj[q] = (K >= 0 && c < m) ? (c == 0 ? 1.0 : A[(size_t)K * n + base + c]) : 0.0;
    }
    for (int i = tid; i < NS; i += 64 * NW) { sp[i] = 0.0; sz[i] = 0.0; }
    if (tid == 0) s_tau = 0.0;
    __syncthreads();

    // Rows behind local row l.  UPD: rank-2 update with the pending (vj, wj); DOT: p_i = row_i . zj into sp.  Q0 = first live chunk.
    auto pass = [&](auto q0c, auto updc, auto dotc, int l) {
        constexpr int Q0 = decltype(q0c)::value;
        constexpr bool UPD = decltype(updc)::value, DOT = decltype(dotc)::value;
#pragma unroll
        for (int g = 0; g < TR / 4; ++g) {
            if (wid + NW * (4 * g + 3) <= l) continue;       // (wave-uniform) all four rows of the group are dead
            if (wid + NW * (4 * g) >= m) continue;           // ... or beyond the block
            double d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = 4 * g + u, i = wid + NW * t;
                d[u] = 0.0;
                if (i > l && i < m) {
                    if constexpr (UPD) {
                        // row operands v_i, w_i: column i sits in chunk t / SPC, lane wid + NW (t % SPC)
                        const double vi = -lane_bcast(vj[t / SPC], wid + NW * (t % SPC)), wi = -lane_bcast(wj[t / SPC], wid + NW * (t % SPC));
#pragma unroll
                        for (int q = Q0; q < CH; ++q) {
                            areg[t][q] = fma(vi, wj[q], areg[t][q]);
                            areg[t][q] = fma(wi, vj[q], areg[t][q]);
                        }
                    }
                    if constexpr (DOT) {
#pragma unroll
                        for (int q = Q0; q < CH; ++q) d[u] = fma(areg[t][q], zj[q], d[u]);
                    }
                }
            }
            if constexpr (DOT) {
#ifdef MOPT_TAIL_ABL_NORED
                const double f = d[0] + d[1] + d[2] + d[3];
#else
                const double f = wave_sum4(d[0], d[1], d[2], d[3]);
#endif
                const int r = lane >> 4, u = ((r & 1) << 1) | (r >> 1);      // row 0, 1, 2, 3 of the wave holds d0, d2, d1, d3
                const int i = wid + NW * (4 * g + u);
                if ((lane & 15) == 0) sp[i] = f;
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    if (K >= 0) {                                             // p = A v_K for the take-over (nothing pending)
        pass(std::integral_constant<int, 0>{}, F_{}, T_{}, -1);
        lds_barrier();
    }

    auto step = [&](auto q0c, int l) {
        constexpr int Q0 = decltype(q0c)::value;
        // ---- a. w = tau p + c v for the reflector whose products are in sp
#ifdef MOPT_TAIL_PROF
        tc0 = clock64();
        prof_l = l;
        if (prof && prof_l == 5 && lane == 0 && blockIdx.x == 0) prof[(size_t)gridDim.x * 8 + (size_t)(tid >> 6) * 8] = tc0;
#endif
        const double tau = tau_z;
        double pvl = 0.0, wl_l = 0.0, vl_l = 0.0;
#pragma unroll
        for (int q = Q0; q < CH; ++q) {
            const int c = lane + 64 * q;
            vj[q] = zj[q];
            const double pj = (c >= l && c < m) ? tau * sp[c] : 0.0;
            wj[q] = pj;
            pvl = fma(pj, vj[q], pvl);
        }
#ifdef MOPT_TAIL_ABL_NOA
        const double cc = -0.5 * tau * pvl;                  // (timing ablation: no reduction; results are garbage)
#else
        const double cc = -0.5 * tau * wave_sum(pvl);
#endif
#pragma unroll
        for (int q = Q0; q < CH; ++q) {
            const int c = lane + 64 * q;
            wj[q] = fma(cc, vj[q], wj[q]);
            wl_l = (c == l) ? wj[q] : wl_l;
            vl_l = (c == l) ? vj[q] : vl_l;
        }
        const bool owner = (wid == (l % NW));
        TAIL_STAMP(0);
        if (owner) {
            // ---- b. the owner of pivot row l: row after the pending update, next reflector
            const double w_l = lane_bcast(wl_l, l & 63), v_l = lane_bcast(vl_l, l & 63);
            const int to = l / NW;
            double rq[CH];
#pragma unroll
            for (int t = 0; t < TR; ++t)
                if (t == to) {
#pragma unroll
                    for (int q = Q0; q < CH; ++q) rq[q] = areg[t][q];
                }
            double xl = 0.0, al_l = 0.0, dg_l = 0.0;
#pragma unroll
            for (int q = Q0; q < CH; ++q) {
                const int c = lane + 64 * q;
                rq[q] -= v_l * wj[q] + w_l * vj[q];
                al_l = (c == l + 1) ? rq[q] : al_l;
                dg_l = (c == l) ? rq[q] : dg_l;
                rq[q] = (c >= l + 2) ? rq[q] : 0.0;
                xl = fma(rq[q], rq[q], xl);
            }
#ifdef MOPT_TAIL_ABL_NOB
            const double xn2 = xl;
#else
            const double xn2 = wave_sum(xl);
#endif
            const double alpha = lane_bcast(al_l, (l + 1) & 63);
            const double diag = lane_bcast(dg_l, l & 63);
            const int lo = base + l;
            const bool refl = (lo <= n - 3 && xn2 != 0.0);
            double betan = alpha, taun = 0.0, scal = 0.0;
#ifdef MOPT_TAIL_ABL_NOB
            if (false) {
#else
            if (refl) {
#endif
                const double x2 = fma(alpha, alpha, xn2);
                if (x2 > 1e-280 && x2 < 1e280) {              // beta, tau and the scale from one reciprocal square root and one reciprocal
                    const double rs = tail_rsqrt(x2), nrm = x2 * rs, aa = fabs(alpha);
                    betan = -copysign(nrm, alpha);
                    taun = fma(aa, rs, 1.0);                  // (beta - alpha) / beta = 1 + |alpha| / norm
                    scal = copysign(tail_rcp(aa + nrm), alpha);      // 1 / (alpha - beta)
                } else {
                    const double bt = -copysign(sqrt(x2), alpha);
                    betan = bt; taun = (bt - alpha) / bt; scal = 1.0 / (alpha - bt);
                }
            }
            double *zrow = sZ + (l * (m - 2) - (l * (l - 1)) / 2) - (l + 2);      // packed row l, indexed by the column
#pragma unroll
            for (int q = Q0; q < CH; ++q) {
                const int c = lane + 64 * q;
                const double vn = (c == l + 1) ? 1.0 : rq[q] * scal;
                sz[c] = vn;
                if (c >= l + 2 && c < m) zrow[c] = vn;
            }
            if (lane == 0) { s_tau = taun; sD[l] = diag; sE[l] = betan; sT[l] = taun; }
        }
        // every wave updates its rows while the reflector is being formed (the owner behind it).  The update and the products are
        // two sweeps on purpose: with `if (owner) fused else split` the compiler keeps two copies of the tile and spills.
        TAIL_STAMP(1);
#ifndef MOPT_TAIL_ABL_NOUPD
        pass(q0c, T_{}, F_{}, l);
#endif
        TAIL_STAMP(2);
        lds_barrier();
        TAIL_STAMP(3);
        // ---- c. products with the new reflector
        tau_z = s_tau;
#pragma unroll
        for (int q = Q0; q < CH; ++q) zj[q] = sz[lane + 64 * q];
#ifndef MOPT_TAIL_ABL_NODOT
        pass(q0c, F_{}, T_{}, l);
#endif
        TAIL_STAMP(4);
        lds_barrier();
        TAIL_STAMP(5);
    };

    // (three loops rather than a switch inside one: chunk q holds dead columns only from l = 64 (q + 1) on)
    int l = 0;
    for (; l <= m - 2 && l < 64; ++l) step(std::integral_constant<int, 0>{}, l);
    if constexpr (CH > 1) for (; l <= m - 2 && l < 128; ++l) step(std::integral_constant<int, 1>{}, l);
    if constexpr (CH > 2) for (; l <= m - 2; ++l) step(std::integral_constant<int, 2>{}, l);
    // D[n - 1]: the last row after the last update (the reflector of row n - 2 has tau = 0)
    {
        const int i = m - 1;
        if (wid == (i % NW)) {
#pragma unroll
            for (int t = 0; t < TR; ++t)
#pragma unroll
                for (int q = 0; q < CH; ++q)
                    if (t == (i / NW) && lane + 64 * q == i) { sD[i] = areg[t][q]; sE[i] = 0.0; sT[i] = 0.0; }
        }
    }
    __syncthreads();
    // flush: D, E, TAU of rows base .. n - 1 and the reflectors (dsytrd layout: row lo, columns lo + 2 ..)
    for (int i = tid; i < m; i += 64 * NW) { D[base + i] = sD[i]; E[base + i] = sE[i]; TAU[base + i] = sT[i]; }
    for (int r = wid; r <= m - 3; r += NW) {
        const double *zrow = sZ + (r * (m - 2) - (r * (r - 1)) / 2) - (r + 2);
        double *arow = A + (size_t)(base + r) * n + base;
        for (int c = r + 2 + lane; c < m; c += 64) arow[c] = zrow[c];
    }
#ifdef MOPT_TAIL_PROF
    if (prof && tid == 0) for (int k = 0; k < 6; ++k) prof[(size_t)b * 8 + k] = tp[k];
#endif
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
