// Divide and conquer for one unreduced block of a symmetric tridiagonal matrix inside one workgroup: the body of k_dc_block
// (k_dcblock.hip, where the method is described), as a device function shared with k_small.hip.
#pragma once
#include "pipeline.h"
#include "dc_logic.h"
#include "ql_block.hpp"
#include "dc_red.hpp"

namespace mopt {

constexpr int DCB_LEAF = 10;       // largest leaf
constexpr int DCB_MAX = 72;        // largest block (three m x m tables + leaf tables in 160 KB of LDS)

// a / b with a hardware reciprocal estimate and two Newton steps (|b| a normal number), as dc_logic.h's dc_quot on the device
__device__ __forceinline__ double dcb_quot(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return a * r;
}
__device__ __forceinline__ int dcb_levels(int m) { int L = 0; while (((m + (1 << L) - 1) >> L) > DCB_LEAF) ++L; return L; }
__device__ __forceinline__ int dcb_bound(int m, int L, int i) { return (i * m) >> L; }

// One block of m rows (rows s0 .. s0 + m - 1 of an n x n problem): D, E point at the block's first entry, WT and Zt at the matrix (row j of
// Zt = eigenvector j, stride n), fail at the matrix' failure counter, sm at the LDS region sized as in mopt_launch_dc_block.  Called by
// all 1024 threads of the workgroup; b is only used for the developer print.
__device__ __forceinline__ void dc_block_body(int n, int s0, int m, int b, const double *D, const double *E, double *WT, double *Zt, int *fail, double *sm, int dbg) {
    __shared__ int s_bad, s_k[4], s_nk[4], s_ndf[4], s_nrot[4], s_k1[4], s_k2[4];      // (per merge of a level: at most four side by side)
    __shared__ double s_red[32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    long long tq[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = dbg ? wall_clock64() : 0;
#define DCB_STAMP(i) do { if (dbg) { const long long tn_ = wall_clock64(); tq[i] += tn_ - tlast; tlast = tn_; } } while (0)
    if (tid == 0) s_bad = 0;
    __syncthreads();
    // ---- LDS layout: the small arrays first, then the three m x m tables; the 16 leaf tables (40 KB) start at X1 and may run past S --
    // the launcher sizes the allocation for the largest block, and X1, S and what lies behind are free until the first merge
    const int mm2 = m * m;
    double *fd = sm;
    double *D2 = fd, *E2 = fd + m, *rho = fd + 2 * m, *d0 = fd + 3 * m, *d1 = fd + 4 * m, *dl = fd + 5 * m, *wv = fd + 6 * m, *wh = fd + 7 * m,
           *dd = fd + 8 * m, *zz = fd + 9 * m;
    DcRot *rot = (DcRot *)(fd + 10 * m);
    unsigned long long *cand = (unsigned long long *)(rot + m);      // [4 merges][2 words]
    int *order = (int *)(cand + 8), *surv = order + m, *nd = surv + m, *defl = nd + m, *coltyp = defl + m, *removed = coltyp + m, *pos = removed + m,
        *ndp = pos + m;
    double *X0 = (double *)(((uintptr_t)(ndp + m) + 15) & ~(uintptr_t)15);      // eigenvector tables (vector-major, row j = eigenvector j) ...
    double *X1 = X0 + mm2, *S = X0 + 2 * mm2;                                  // ... and the secular table
    // ---- scaled copy, torn at the leaf boundaries
    const int L = dcb_levels(m), nl = 1 << L;
    {
        double mx = 0.0;
        for (int i = tid; i < m; i += blockDim.x) { mx = fmax(mx, fabs(D[i])); if (i < m - 1) mx = fmax(mx, fabs(E[i])); }
        mx = wave_max(mx);
        if (lane == 0) s_red[wid] = mx;
        __syncthreads();
        mx = 0.0;
        for (int w = 0; w < 16; ++w) mx = fmax(mx, s_red[w]);
        __syncthreads();
        const double sc = (mx > 0.0) ? mx : 1.0;
        if (tid == 0) s_red[17] = sc;
        for (int i = tid; i < m; i += blockDim.x) {
            int l = (int)((((long long)i + 1) << L) / m);
            while (l > 0 && dcb_bound(m, L, l) > i) --l;
            while (l + 1 < nl && dcb_bound(m, L, l + 1) <= i) ++l;
            const int ls = dcb_bound(m, L, l), lt = dcb_bound(m, L, l + 1);
            double d = D[i] / sc;
            if (i == ls && ls > 0) d -= fabs(E[ls - 1] / sc);
            if (i == lt - 1 && lt < m) d -= fabs(E[lt - 1] / sc);
            D2[i] = d;
            E2[i] = (i == lt - 1) ? 0.0 : E[i] / sc;
            rho[i] = (i == ls && ls > 0) ? E[ls - 1] / sc : 0.0;
        }
        for (int e = tid; e < 2 * mm2; e += blockDim.x) X0[e] = 0.0;      // X0, X1
    }
    __syncthreads();
    DCB_STAMP(0);
    const double sc = s_red[17];
    // ---- leaves: wave w solves leaves w, w + 16, ... (tables of 32 x DCB_LEAF doubles per wave in the S region and behind it -- free until level 1)
    {
        double *Zs = X1 + (size_t)wid * 32 * DCB_LEAF;       // (16 tables of 2.5 KB, see the layout note)
        for (int l = wid; l < nl; l += 16) {
            const int ls = dcb_bound(m, L, l), lt = dcb_bound(m, L, l + 1), ml = lt - ls;
            double d = (lane < ml) ? D2[ls + lane] : 0.0;
            double e = (lane < ml - 1) ? E2[ls + lane] : 0.0;
            for (int c = 0; c < ml; ++c) Zs[c * 32 + (lane & 31)] = ((lane & 31) == c) ? 1.0 : 0.0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            bool ok = true;
            if (ml >= 2) ok = ql_block32(ml, d, e, Zs, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (!ok && lane == 0) s_bad = 1;
            if (lane < ml) d0[ls + lane] = d;
            for (int c = 0; c < ml; ++c) if (lane < ml) X0[(size_t)(ls + c) * m + ls + lane] = Zs[c * 32 + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    DCB_STAMP(1);
    // (the leaf tables overlapped X1: clear it again)
    for (int e = tid; e < mm2; e += blockDim.x) X1[e] = 0.0;
    __syncthreads();
    // ---- merges, level by level.  The merges of a level work on disjoint index ranges [s, t) of every array, so they run side by side: the
    // workgroup splits into nm = 4, 2, 1 thread groups (L <= 3: 72 rows in leaves of at most 10) that execute the same sequence of barriers.
    for (int lev = 1; lev <= L; ++lev) {
        double *Xc = ((lev - 1) & 1) ? X1 : X0, *Xn = (lev & 1) ? X1 : X0;
        const double *dc = ((lev - 1) & 1) ? d1 : d0;
        double *dn = (lev & 1) ? d1 : d0;
        const int nm = 1 << (L - lev);
        const int gsz = 1024 / nm, q = tid / gsz, gt = tid - q * gsz, gw = gt >> 6, nwv = gsz >> 6;      // group size (>= 256), merge, thread / wave inside the group
        const int s = dcb_bound(m, L, q << lev), c = dcb_bound(m, L, (q << lev) + (1 << (lev - 1))), t = dcb_bound(m, L, (q + 1) << lev);
        const int ms = t - s, m1 = c - s;
        double *ddq = dd + s, *zzq = zz + s, *dlq = dl + s, *wvq = wv + s, *whq = wh + s;
        int *orderq = order + s, *survq = surv + s, *ndq = nd + s, *deflq = defl + s, *coltypq = coltyp + s, *removedq = removed + s, *posq = pos + s, *ndpq = ndp + s;
        DcRot *rotq = rot + s;
        unsigned long long *candq = cand + 2 * q;
        double *Sq = S + (size_t)s * m;                      // (rows s .. t-1 of the m x m table, used with the merge's own row length ms)
        // ---------------- prep (first 256 threads of the group: the layout of k_dc_prep)
        const double rho0 = rho[c];
        const double sgn = (rho0 < 0.0) ? -0.70710678118654752440 : 0.70710678118654752440;
        double zmx = 0.0, dmx = 0.0;
        if (gt < 256) {
            for (int i = gt; i < ms; i += 256) {
                const double di = dc[s + i];
                const double zi = (i < m1) ? 0.70710678118654752440 * Xc[(size_t)(s + i) * m + (c - 1)] : sgn * Xc[(size_t)(s + i) * m + c];
                ddq[i] = di; zzq[i] = zi;
                zmx = fmax(zmx, fabs(zi)); dmx = fmax(dmx, fabs(di));
                coltypq[i] = (i < m1) ? 1 : 3; removedq[i] = 0;
                orderq[i] = i;
            }
            zmx = wave_max(zmx); dmx = wave_max(dmx);
            if (lane == 0) { s_red[q * 8 + gw] = zmx; s_red[q * 8 + 4 + gw] = dmx; }
            if (gt < 2) candq[gt] = 0ull;
        }
        __syncthreads();
        zmx = fmax(fmax(s_red[q * 8 + 0], s_red[q * 8 + 1]), fmax(s_red[q * 8 + 2], s_red[q * 8 + 3]));
        dmx = fmax(fmax(s_red[q * 8 + 4], s_red[q * 8 + 5]), fmax(s_red[q * 8 + 6], s_red[q * 8 + 7]));
        if (gt < 256) {
            for (int i = gt; i < ms; i += 256) {              // rank sort (ties by index)
                const double di = ddq[i];
                int r = 0;
                for (int j = 0; j < i; ++j) r += ddq[j] <= di;
                for (int j = i + 1; j < ms; ++j) r += ddq[j] < di;
                orderq[r] = i;
            }
        }
        __syncthreads();
        const double rho2 = 2.0 * fabs(rho0);
        const double tol = 8.0 * DC_EPS * fmax(dmx, zmx);
        const bool all_defl = (rho2 * zmx <= tol);
        if (gw == 0) {                                       // survivors in ascending order, tiny weights to defl[]
            int nk = 0, ndf = 0;
            const unsigned long long below = (1ull << lane) - 1ull;
            for (int j0 = 0; j0 < ms; j0 += 64) {
                const int j = j0 + lane;
                const int nj = (j < ms) ? orderq[j] : 0;
                const bool tiny = (j < ms) && (all_defl || rho2 * fabs(zzq[nj]) <= tol), keep = (j < ms) && !tiny;
                const unsigned long long mk = __ballot(keep), mt = __ballot(tiny);
                if (keep) survq[nk + __popcll(mk & below)] = nj;
                if (tiny) deflq[ndf + __popcll(mt & below)] = nj;
                nk += __popcll(mk); ndf += __popcll(mt);
            }
            if (lane == 0) { s_nk[q] = nk; s_ndf[q] = ndf; }
        }
        __syncthreads();
        const int nk = s_nk[q];
        if (gw < 2) {                                        // closeness test of neighbouring survivors (ms <= 72: two 64-chunks)
            const int i0 = 64 * gw, i = i0 + lane;
            if (i0 < nk) {
                bool cl = false;
                if (i >= 1 && i < nk) {
                    const int a = survq[i - 1], bq = survq[i];
                    const double za = zzq[a], zb = zzq[bq], tt = ddq[bq] - ddq[a];
                    cl = fabs(tt * za * zb) <= 2.0 * tol * (za * za + zb * zb);
                }
                const unsigned long long mk = __ballot(cl);
                if (lane == 0) candq[gw] = mk;
            }
        }
        __syncthreads();
        if (gt == 0) {
            int nrot = 0;
            dc_walk_candidates(nk, survq, candq, ddq, zzq, tol, coltypq, removedq, rotq, &nrot);
            s_nrot[q] = nrot;
        }
        __syncthreads();
        if (gw == 0) {                                       // final survivors nd[0 .. k), positions by type, the rest to defl[]
            const unsigned long long below = (1ull << lane) - 1ull;
            int k = 0, ndf = s_ndf[q], n1 = 0, n2 = 0;
            for (int i0 = 0; i0 < nk; i0 += 64) {
                const int i = i0 + lane;
                const bool in = i < nk, rem = in && removedq[i], keep = in && !rem;
                const int idx = in ? survq[i] : 0;
                const int ty = keep ? coltypq[idx] : 0;
                const unsigned long long mk = __ballot(keep), mr = __ballot(rem);
                if (keep) ndq[k + __popcll(mk & below)] = idx;
                if (rem) deflq[ndf + __popcll(mr & below)] = idx;
                k += __popcll(mk); ndf += __popcll(mr);
                n1 += __popcll(__ballot(ty == 1)); n2 += __popcll(__ballot(ty == 2));
            }
            int c1 = 0, c2 = 0, c3 = 0;
            for (int i0 = 0; i0 < k; i0 += 64) {
                const int i = i0 + lane;
                const int ty = (i < k) ? coltypq[ndq[i]] : 0;
                const unsigned long long m1_ = __ballot(ty == 1), m2_ = __ballot(ty == 2), m3_ = __ballot(ty == 3);
                if (ty == 1) posq[i] = c1 + __popcll(m1_ & below);
                if (ty == 2) posq[i] = n1 + c2 + __popcll(m2_ & below);
                if (ty == 3) posq[i] = n1 + n2 + c3 + __popcll(m3_ & below);
                c1 += __popcll(m1_); c2 += __popcll(m2_); c3 += __popcll(m3_);
            }
            if (lane == 0) { s_k[q] = k; s_k1[q] = n1; s_k2[q] = n2; }
        }
        __syncthreads();
        DCB_STAMP(2);
        const int k = s_k[q], nrot = s_nrot[q], k1 = s_k1[q], k2 = s_k2[q];
        // Givens rotations of the deflated pairs on the eigenvector rows (each thread owns its columns for the whole list)
        if (nrot > 0) {
            for (int x = s + gt; x < t; x += gsz) {
                for (int r = 0; r < nrot; ++r) {
                    const DcRot g = rotq[r];
                    double *xa = Xc + (size_t)(s + g.a) * m + x, *xb = Xc + (size_t)(s + g.b) * m + x;
                    const double va = *xa, vb = *xb;
                    *xa = g.c * va + g.s * vb;
                    *xb = g.c * vb - g.s * va;
                }
            }
        }
        for (int i = gt; i < ms; i += gsz) {
            if (i < k) { const int idx = ndq[i]; dlq[i] = ddq[idx]; wvq[i] = zzq[idx]; ndpq[posq[i]] = idx; }
            else { const int qd = i - k; dn[s + i] = ddq[deflq[qd]]; }
        }
        __syncthreads();
        DCB_STAMP(3);
        // ---------------- secular equation: one root per 16-lane group (8 lanes when the merge has more roots than 16-lane groups: one round
        // instead of two), the row d_i - lambda_j lands in S[j][.]
        {
            const int ng16 = gsz >> 4;
            if (k <= ng16) {
                const int g16 = gt >> 4;
                for (int j = g16; j < k; j += ng16) {
                    DcRow16Red red;
                    double lam;
                    const int rc = dc_secular(red, k, j, dlq, wvq, rho2, Sq + (size_t)j * ms, &lam);
                    red.sync();
                    if ((tid & 15) == 0) { dn[s + j] = lam; if (rc < 0) s_bad = 1; }
                }
            } else {
                const int g8 = gt >> 3, ng8 = gsz >> 3;
                for (int j = g8; j < k; j += ng8) {
                    DcRow8Red red;
                    double lam;
                    const int rc = dc_secular(red, k, j, dlq, wvq, rho2, Sq + (size_t)j * ms, &lam);
                    red.sync();
                    if ((tid & 7) == 0) { dn[s + j] = lam; if (rc < 0) s_bad = 1; }
                }
            }
        }
        __syncthreads();
        DCB_STAMP(4);
        // ---------------- Loewner weights: a 16-lane group per pole, the product over the roots split over its lanes
        {
            const int g16 = gt >> 4, ng16 = gsz >> 4, l16 = tid & 15;
            for (int i = g16; i < k; i += ng16) {
                const double di = dlq[i];
                double prod = 1.0;
                for (int j = l16; j < k; j += 16) { const double sji = Sq[(size_t)j * ms + i]; prod *= (j == i) ? -sji : dcb_quot(sji, di - dlq[j]); }
                prod *= dpp_f64<0xB1, 0xF>(1.0, prod);       // (products over the DPP row: quad permutes, half-row mirror, row mirror)
                prod *= dpp_f64<0x4E, 0xF>(1.0, prod);
                prod *= dpp_f64<0x141, 0xF>(1.0, prod);
                prod *= dpp_f64<0x140, 0xF>(1.0, prod);
                if (l16 == 0) whq[i] = copysign(sqrt(fabs(prod)), wvq[i]);
            }
        }
        __syncthreads();
        DCB_STAMP(5);
        // ---------------- rows of U in place of S (wave per row; all reads of a row precede its writes), deflated vectors copied
        for (int j = gw; j < ms; j += nwv) {
            if (j < k) {
                double *row = Sq + (size_t)j * ms;
                const double u0 = (lane < k) ? whq[lane] / row[lane] : 0.0, u1 = (lane + 64 < k) ? whq[lane + 64] / row[lane + 64] : 0.0;
                const double inv = 1.0 / sqrt(wave_sum(fma(u0, u0, u1 * u1)));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (lane < k) row[posq[lane]] = u0 * inv;
                if (lane + 64 < k) row[posq[lane + 64]] = u1 * inv;
            } else {
                const int src = deflq[j - k];
                for (int x = lane; x < ms; x += 64) Xn[(size_t)(s + j) * m + s + x] = Xc[(size_t)(s + src) * m + s + x];
            }
        }
        __syncthreads();
        DCB_STAMP(6);
        // ---------------- Xnext[s + j][s + x] = sum_ip U[j][ip] Xcur[s + ndp[ip]][s + x]     (j < k; upper columns skip lower-only vectors and v.v.)
        // a thread forms four neighbouring columns of one row: one U entry and one 32-byte row segment per multiply-add group
        {
            const int nx4 = (ms + 3) >> 2;
            for (int e = gt; e < k * nx4; e += gsz) {
                const int j = e / nx4, x0 = (e - j * nx4) << 2;
                // (a group of four columns may straddle the cut: take the wider range, the extra terms multiply exact zeros)
                const bool up = (x0 + 3 < m1), lowr = (x0 >= m1);
                const int ilo = lowr ? k1 : 0, ihi = up ? k1 + k2 : k;
                const double *ur = Sq + (size_t)j * ms;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
                for (int ip = ilo; ip < ihi; ++ip) {
                    const double u = ur[ip];
                    const double *xr = Xc + (size_t)(s + ndpq[ip]) * m + s + x0;
                    a0 = fma(u, xr[0], a0);
                    if (x0 + 1 < ms) a1 = fma(u, xr[1], a1);
                    if (x0 + 2 < ms) a2 = fma(u, xr[2], a2);
                    if (x0 + 3 < ms) a3 = fma(u, xr[3], a3);
                }
                double *o = Xn + (size_t)(s + j) * m + s + x0;
                o[0] = a0;
                if (x0 + 1 < ms) o[1] = a1;
                if (x0 + 2 < ms) o[2] = a2;
                if (x0 + 3 < ms) o[3] = a3;
            }
        }
        __syncthreads();
        DCB_STAMP(7);
    }
    // ---- results: eigenvalues back to the caller's scale, eigenvectors into rows s0 .. s0 + m - 1 of Zt (zero outside the block)
    const double *Xf = (L & 1) ? X1 : X0, *df = (L & 1) ? d1 : d0;
    if (tid < m) WT[s0 + tid] = df[tid] * sc;
    for (int e = tid; e < m * n; e += blockDim.x) {
        const int j = e / n, i = e - j * n;
        Zt[(size_t)(s0 + j) * n + i] = (i >= s0 && i < s0 + m) ? Xf[(size_t)j * m + (i - s0)] : 0.0;
    }
    if (tid == 0 && s_bad) atomicAdd(fail, 1);
    DCB_STAMP(8);
    if (dbg && tid == 0 && b == 0)
        printf("[dcb] m=%d L=%d: setup %.1f, leaves %.1f, prep %.1f, rot+scatter %.1f, secular %.1f, zhat %.1f, ufill %.1f, gemm %.1f, out %.1f us\n", m, L, tq[0] * 0.01, tq[1] * 0.01,
               tq[2] * 0.01, tq[3] * 0.01, tq[4] * 0.01, tq[5] * 0.01, tq[6] * 0.01, tq[7] * 0.01, tq[8] * 0.01);
#undef DCB_STAMP
}

}  // namespace mopt
