// Householder tridiagonalisation of a block of at most 192 rows inside one workgroup: the body of k_tri_tail (k_tritail.hip, where the
// method is described), as a device function shared with k_small.hip.
#pragma once
#include <type_traits>
#include "common.hpp"

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
// The body is a device function so that the one-workgroup step kernel for small geometries (k_small.hip) can run it as one of its phases:
// tsm = the LDS region described below, b = the matrix of this workgroup.  Pointers are not marked __restrict__ here: inside k_small.hip
// the same workgroup has written A a phase earlier.
template <int CH, int NW>
__device__ __forceinline__ void tri_tail_body(int n, int K, int b, double *A_, double *D_, double *E_, double *TAU_, long long *prof, double *tsm) {
#ifdef MOPT_TAIL_PROF
    // developer build (tools/ubench/tail_main.hip): shader-clock stamps of wave 0 around the phases of a step, summed over the steps
    long long tp[6] = {0, 0, 0, 0, 0, 0}, tc0 = 0, tc1 = 0;
    int prof_l = -1;
#define TAIL_STAMP(k) do { tc1 = clock64(); tp[k] += tc1 - tc0; tc0 = tc1; \
        if (prof && prof_l == 5 && lane == 0 && b == 0) prof[(size_t)gridDim.x * 8 + (size_t)(tid >> 6) * 8 + (k) + 1] = tc1; } while (0)
#else
#define TAIL_STAMP(k) do { } while (0)
#endif
    constexpr int NS = 64 * CH, TR = NS / NW, SPC = 64 / NW;      // SPC slots of a wave per column chunk
    static_assert(TR % 4 == 0, "rows are reduced four at a time");
    // LDS: p and the current reflector (NS each), then the results of all steps -- D, E, TAU (NS each) and the reflectors packed as a
    // triangle (row l: columns l + 2 .. m - 1) -- which go to global memory in one coalesced flush behind the last step: a global
    // store inside the loop costs the owner wave its address arithmetic and every later scalar clock / wait on it
    double *sp = tsm, *sz = tsm + NS, *sD = tsm + 2 * NS, *sE = tsm + 3 * NS, *sT = tsm + 4 * NS, *sZ = tsm + 5 * NS;
    __shared__ double s_tau;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        if (prof && prof_l == 5 && lane == 0 && b == 0) prof[(size_t)gridDim.x * 8 + (size_t)(tid >> 6) * 8] = tc0;
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
