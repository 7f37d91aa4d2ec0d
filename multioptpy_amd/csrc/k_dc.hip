// Divide-and-conquer eigen-solver for the symmetric tridiagonal form (the part of numpy.linalg.eigh, rsirfo.py:606, behind the
// Householder reduction) -- used for every matrix whose tridiagonal does not fall apart into small blocks, i.e. for dense
// molecular Hessians (the reference enters the RFO-NEB path with exact or model Hessians: neb.py:794-804).  There bisection +
// inverse iteration is the wrong tool: dense spectra sit in loose clusters that need re-orthogonalisation and the six
// TR/rot zeros form a tight cluster that is walked in sequence (k_tri_invit 3.7 ms + k_tri_bisect 1.0 ms + k_tri_lowdin 0.8 ms
// per band step of the 32 x 100 band started from exact Hessians, profiles/r02a_*).  Here clusters deflate and orthogonality
// follows from the Loewner formula; all eigenvector work is a batched FP64 GEMM on the matrix cores.
//
// Fixed binary tree over the rows of the whole matrix (leaf <= 32 rows, solved by a wave-level implicit QL, k_dc_leaf); per
// level five launches for all merges of all matrices:
//   k_dc_prep     z, rank sort, deflation scan (dc_logic.h dc_deflate), Givens rotations of deflated pairs
//   k_dc_secular  one wave per root (dc_logic.h dc_secular): lambda_j and the row d_i - lambda_j
//   k_dc_zhat     Loewner weights, lane = pole, no reduction
//   k_dc_ufill    normalised eigenvectors of the k x k problem (rows of U), copies of the deflated eigenvectors
//   k_dc_gemm     Xnext = U * Xcur (row gather) on v_mfma_f64_16x16x4_f64; upper / lower halves skip the other half's columns
// Eigenvectors are kept vector-major (row j = eigenvector j, zero outside the merge's index range) and ping-pong between the
// caller's Zt and its scratch copy.  Scalar logic shared with the host-side tests: dc_logic.h.
#include <cstdlib>

#include "pipeline.h"
#include "dc_logic.h"
#include "ql_block.hpp"
#include "dc_red.hpp"
#include "dc_block_body.hpp"

namespace mopt {

typedef double dc_double4 __attribute__((ext_vector_type(4)));

struct DcPtrs {
    int *flag;                    // [B] 1 = this matrix is solved here
    double *scale;                // [B]
    double *D2, *E2, *rho;        // [B][n] scaled leaf tridiagonal, rho[c] = scaled off-diagonal torn at leaf boundary c
    int *bs2, *be2;               // [B][n] leaf table (bs2 = -1: matrix not handled here)
    double *d0, *d1;              // [B][n] eigenvalues of the current / next level
    double *dl, *wv, *wh, *rhom;  // [B][n] merge-local at offset s: poles, weights, Loewner weights; rhom[s] = 2 |rho|
    int *nd, *ndp, *dfl, *pos, *meta;   // [B][n] merge-local; meta[s] = k, meta[s+1] = upper-only count, meta[s+2] = mixed count
    double *S, *U;                // [B][n][n]
};

__device__ __forceinline__ double dc_rsqrt(double w) {      // 1/sqrt(w), w > 0 normal: hardware estimate + two Newton steps
    double y = __builtin_amdgcn_rsq(w);
    double e = fma(-(w * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(w * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}

__device__ __forceinline__ void dc_range(int n, int L, int lev, int q, int &s, int &c, int &t) {
    s = dc_bound(n, L, q << lev); c = dc_bound(n, L, (q << lev) + (1 << (lev - 1))); t = dc_bound(n, L, (q + 1) << lev);
}

// grid B, block 256.  For flagged matrices: scaled copy of (D, E) torn at the leaf boundaries, leaf table, and the block table
// of the caller (one block [0, n): the back-transformation then treats the eigenvectors as dense).
__global__ void __launch_bounds__(256) k_dc_setup(int n, int L, const double *__restrict__ D_, const double *__restrict__ E_, DcPtrs p,
                                                  int *__restrict__ bstart_, int *__restrict__ bend_) {
    __shared__ double red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    int *bs2 = p.bs2 + (size_t)b * n, *be2 = p.be2 + (size_t)b * n;
    if (!p.flag[b]) { for (int i = tid; i < n; i += blockDim.x) { bs2[i] = -1; be2[i] = -1; } return; }
    const double *D = D_ + (size_t)b * n, *E = E_ + (size_t)b * n;
    double mx = 0.0;
    for (int i = tid; i < n; i += blockDim.x) { mx = fmax(mx, fabs(D[i])); if (i < n - 1) mx = fmax(mx, fabs(E[i])); }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const double sc = (mx > 0.0) ? mx : 1.0;
    if (tid == 0) p.scale[b] = sc;
    double *D2 = p.D2 + (size_t)b * n, *E2 = p.E2 + (size_t)b * n, *rho = p.rho + (size_t)b * n;
    for (int i = tid; i < n; i += blockDim.x) {
        // leaf of row i: the largest l with lb(l) <= i
        int l = (int)((((long long)i + 1) << L) / n);
        while (l > 0 && dc_bound(n, L, l) > i) --l;
        while (l + 1 < (1 << L) && dc_bound(n, L, l + 1) <= i) ++l;
        const int s = dc_bound(n, L, l), t = dc_bound(n, L, l + 1);
        bs2[i] = s; be2[i] = t;
        double d = D[i] / sc;
        if (i == s && s > 0) d -= fabs(E[s - 1] / sc);
        if (i == t - 1 && t < n) d -= fabs(E[t - 1] / sc);
        D2[i] = d;
        E2[i] = (i == t - 1) ? 0.0 : E[i] / sc;
        rho[i] = (i == s && s > 0) ? E[s - 1] / sc : 0.0;
        bstart_[(size_t)b * n + i] = 0; bend_[(size_t)b * n + i] = n;
    }
}

// Leaves: implicit QL with Wilkinson shifts and accumulated rotations (the classical tql2 / dsteqr iteration), one wave per leaf of
// at most 32 rows.  d and e live in registers (lane i holds d_i, e_i; scalars are fetched with v_readlane), the rotation chain is
// computed redundantly by every lane, lane r owns row r of the eigenvector table in LDS (column-major, conflict-free).  A leaf
// of 19 rows takes ~2.5 QL steps per eigenvalue, ~500 rotations in all -- the one-sided Jacobi of k_tri_small needed ~200 us
// for the same leaves (LDS round trips and a barrier per round).
// grid (2^L, B), block 64.  Writes dv[s .. t) and rows s .. t-1 of X (zero outside the leaf).
__global__ void __launch_bounds__(64) k_dc_leaf(int n, int L, DcPtrs p, double *__restrict__ dv_, double *__restrict__ X_, int *__restrict__ fail_) {
    __shared__ double Zs[32 * 32];
    const int b = blockIdx.y, lane = threadIdx.x;
    if (!p.flag[b]) return;
    const int s = dc_bound(n, L, blockIdx.x), t = dc_bound(n, L, blockIdx.x + 1), m = t - s;
    const size_t o = (size_t)b * n + s;
    double d = (lane < m) ? p.D2[o + lane] : 0.0;
    double e = (lane < m - 1) ? p.E2[o + lane] : 0.0;
    for (int c = 0; c < m; ++c) Zs[c * 32 + (lane & 31)] = ((lane & 31) == c) ? 1.0 : 0.0;      // (lanes 32 .. 63 mirror 0 .. 31: same values)
    __syncthreads();
    const bool failed = !ql_block32(m, d, e, Zs, lane);
    __syncthreads();
    if (failed && lane == 0) atomicAdd(&fail_[b], 1);
    if (lane < m) dv_[o + lane] = d;
    for (int c = 0; c < m; ++c) {
        double *row = X_ + ((size_t)b * n + s + c) * n;
        for (int i = lane; i < n; i += WAVE) row[i] = (i >= s && i < t) ? Zs[c * 32 + (i - s)] : 0.0;
    }
}

// Leaves of up to DC_LEAF_WG rows by the divide and conquer inside one workgroup (dc_block_body.hpp: leaves of <= 10 rows by the one-wave
// QL side by side, two merge levels in LDS) -- 85 us for the 8 x 32 leaves of 37 / 38 rows of a 32 x 300 batch where the one-wave QL
// took 112 us for 18 / 19-row leaves, and the chain behind it has one level (five launches, 65 us) less.
// grid (2^L, B), block 1024, dynamic LDS as for k_dc_block with mcap = the largest leaf.  Writes dv[s .. t) and rows s .. t-1 of X (zero outside the leaf).
__global__ void __launch_bounds__(1024) k_dc_leaf_wg(int n, int L, DcPtrs p, double *__restrict__ dv_, double *__restrict__ X_, int *__restrict__ fail_) {
    extern __shared__ __attribute__((aligned(16))) double smleaf[];
    const int b = blockIdx.y;
    if (!p.flag[b]) return;
    const int s = dc_bound(n, L, blockIdx.x), t = dc_bound(n, L, blockIdx.x + 1), m = t - s;
    const size_t o = (size_t)b * n + s;
    dc_block_body(n, s, m, b, p.D2 + o, p.E2 + o, dv_ + (size_t)b * n, X_ + (size_t)b * n * n, fail_ + b, smleaf, 0);
}

// grid (n, B): zero row blockIdx.x of X for flagged matrices
__global__ void __launch_bounds__(256) k_dc_zero(int n, const int *__restrict__ flag, double *__restrict__ X) {
    const int b = blockIdx.y;
    if (!flag[b]) return;
    double *row = X + ((size_t)b * n + blockIdx.x) * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) row[i] = 0.0;
}

// grid (merges, B), block 256.  dynamic LDS: 4 m doubles + m rotations (24 bytes) + 7 m ints + (m / 64 + 2) words.
// Deflation in the kernels' layout (dc_logic.h: dc_deflate is the sequential scan this reproduces, checked on the host): tiny
// weights are dropped and the survivors compacted in parallel, every pair of neighbouring survivors gets the cheap closeness
// test in parallel, and one thread visits only the flagged pairs (dc_walk_candidates: the rotations, rare in dense spectra).
__global__ void __launch_bounds__(256) k_dc_prep(int n, int L, int lev, DcPtrs p, const double *__restrict__ dcur_, double *__restrict__ dnext_,
                                                 double *__restrict__ Xcur_) {
    extern __shared__ __attribute__((aligned(16))) double smd[];
    __shared__ int s_k, s_nk, s_ndf, s_nrot, s_k1, s_k2;
    __shared__ double s_red[8];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (!p.flag[b]) return;
    int s, c, t;
    dc_range(n, L, lev, blockIdx.x, s, c, t);
    const int m = t - s, m1 = c - s;
    double *d = smd, *z = d + m, *dl = z + m, *wv = dl + m;
    DcRot *rot = (DcRot *)(wv + m);                        // (8-byte aligned: 32 m bytes behind the 16-byte aligned base)
    unsigned long long *cand = (unsigned long long *)(rot + m);
    const int nwords = (m + 63) / 64 + 1;
    int *order = (int *)(cand + nwords), *surv = order + m, *nd = surv + m, *defl = nd + m, *coltyp = defl + m, *removed = coltyp + m, *pos = removed + m;
    const size_t o = (size_t)b * n + s;
    double *X = Xcur_ + (size_t)b * n * n;
    const double rho0 = p.rho[(size_t)b * n + c];
    const double sgn = (rho0 < 0.0) ? -0.70710678118654752440 : 0.70710678118654752440;
    double zmx = 0.0, dmx = 0.0;
    for (int i = tid; i < m; i += blockDim.x) {
        const double di = dcur_[o + i];
        const double zi = (i < m1) ? 0.70710678118654752440 * X[(size_t)(s + i) * n + (c - 1)] : sgn * X[(size_t)(s + i) * n + c];
        d[i] = di; z[i] = zi;
        zmx = fmax(zmx, fabs(zi)); dmx = fmax(dmx, fabs(di));
        coltyp[i] = (i < m1) ? 1 : 3; removed[i] = 0;
    }
    zmx = wave_max(zmx); dmx = wave_max(dmx);
    if (lane == 0) { s_red[wid] = zmx; s_red[4 + wid] = dmx; }
    for (int i = tid; i < nwords; i += blockDim.x) cand[i] = 0ull;
    __syncthreads();
    zmx = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    dmx = fmax(fmax(s_red[4], s_red[5]), fmax(s_red[6], s_red[7]));
    for (int i = tid; i < m; i += blockDim.x) order[i] = i;      // (a valid permutation even if NaNs make ranks collide)
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) {            // rank sort (ties by index)
        const double di = d[i];
        int r = 0;
        for (int j = 0; j < i; ++j) r += d[j] <= di;
        for (int j = i + 1; j < m; ++j) r += d[j] < di;
        order[r] = i;
    }
    __syncthreads();
    const double rho2 = 2.0 * fabs(rho0);
    const double tol = 8.0 * DC_EPS * fmax(dmx, zmx);
    const bool all_defl = (rho2 * zmx <= tol);
    // survivors in ascending order (wave 0, ballot compaction), the tiny weights to defl[]
    if (wid == 0) {
        int nk = 0, ndf = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int j0 = 0; j0 < m; j0 += 64) {
            const int j = j0 + lane;
            const int nj = (j < m) ? order[j] : 0;
            const bool tiny = (j < m) && (all_defl || rho2 * fabs(z[nj]) <= tol), keep = (j < m) && !tiny;
            const unsigned long long mk = __ballot(keep), mt = __ballot(tiny);
            if (keep) surv[nk + __popcll(mk & below)] = nj;
            if (tiny) defl[ndf + __popcll(mt & below)] = nj;
            nk += __popcll(mk); ndf += __popcll(mt);
        }
        if (lane == 0) { s_nk = nk; s_ndf = ndf; }
    }
    __syncthreads();
    const int nk = s_nk;
    // closeness test of neighbouring survivors, one bit per pair (wave w takes the 64-chunks w, w + 4, ...)
    for (int i0 = 64 * wid; i0 < nk; i0 += 256) {
        const int i = i0 + lane;
        bool cl = false;
        if (i >= 1 && i < nk) {
            const int a = surv[i - 1], bq = surv[i];
            const double za = z[a], zb = z[bq], tt = d[bq] - d[a];
            cl = fabs(tt * za * zb) <= 2.0 * tol * (za * za + zb * zb);
        }
        const unsigned long long mk = __ballot(cl);
        if (lane == 0) cand[i0 >> 6] = mk;
    }
    __syncthreads();
    if (tid == 0) {
        int nrot = 0;
        dc_walk_candidates(nk, surv, cand, d, z, tol, coltyp, removed, rot, &nrot);
        s_nrot = nrot;
    }
    __syncthreads();
    // final survivors nd[0 .. k) (ascending), their positions in the product by type (upper-only, mixed, lower-only), the rest to defl[]
    if (wid == 0) {
        const unsigned long long below = (1ull << lane) - 1ull;
        int k = 0, ndf = s_ndf, n1 = 0, n2 = 0;
        for (int i0 = 0; i0 < nk; i0 += 64) {
            const int i = i0 + lane;
            const bool in = i < nk, rem = in && removed[i], keep = in && !rem;
            const int idx = in ? surv[i] : 0;
            const int ty = keep ? coltyp[idx] : 0;
            const unsigned long long mk = __ballot(keep), mr = __ballot(rem);
            if (keep) nd[k + __popcll(mk & below)] = idx;
            if (rem) defl[ndf + __popcll(mr & below)] = idx;
            k += __popcll(mk); ndf += __popcll(mr);
            n1 += __popcll(__ballot(ty == 1)); n2 += __popcll(__ballot(ty == 2));
        }
        int c1 = 0, c2 = 0, c3 = 0;
        for (int i0 = 0; i0 < k; i0 += 64) {
            const int i = i0 + lane;
            const int ty = (i < k) ? coltyp[nd[i]] : 0;
            const unsigned long long m1_ = __ballot(ty == 1), m2_ = __ballot(ty == 2), m3_ = __ballot(ty == 3);
            if (ty == 1) pos[i] = c1 + __popcll(m1_ & below);
            if (ty == 2) pos[i] = n1 + c2 + __popcll(m2_ & below);
            if (ty == 3) pos[i] = n1 + n2 + c3 + __popcll(m3_ & below);
            c1 += __popcll(m1_); c2 += __popcll(m2_); c3 += __popcll(m3_);
        }
        if (lane == 0) { s_k = k; s_k1 = n1; s_k2 = n2; }
    }
    __syncthreads();
    const int k = s_k, nrot = s_nrot;
    // Givens rotations of the deflated pairs on the eigenvector rows (each thread owns its columns for the whole list)
    if (nrot > 0) {
        for (int x = s + tid; x < t; x += blockDim.x) {
            for (int r = 0; r < nrot; ++r) {
                const DcRot g = rot[r];
                double *xa = X + (size_t)(s + g.a) * n + x, *xb = X + (size_t)(s + g.b) * n + x;
                const double va = *xa, vb = *xb;
                *xa = g.c * va + g.s * vb;
                *xb = g.c * vb - g.s * va;
            }
        }
    }
    for (int i = tid; i < m; i += blockDim.x) {
        if (i < k) {
            const int idx = nd[i];
            p.dl[o + i] = d[idx]; p.wv[o + i] = z[idx]; p.nd[o + i] = idx; p.pos[o + i] = pos[i]; p.ndp[o + pos[i]] = idx;
        } else {
            const int q = i - k;
            p.dfl[o + q] = defl[q];
            dnext_[o + i] = d[defl[q]];
        }
    }
    if (tid == 0) { p.meta[o] = k; p.meta[o + 1] = s_k1; p.meta[o + 2] = s_k2; p.rhom[o] = rho2; }
}

// One secular root per group of G lanes (G = 64: a wave per root, G = 16: four roots per wave).  grid (ceil(m_max / (64 / G)),
// merges * B), block 64.  dynamic LDS: (2 + 64 / G) m_max doubles.  The number of roots per level does not depend on the level
// (n per matrix), the poles per root do: with a wave per root the lower levels spend their time in the five 64-lane reductions
// of an iteration on one or two poles per lane (32-37 us per level at 32 x 300); with 16 lanes per root the top level walks 19
// poles per lane in sequence (48 -> 60 us).  The launcher takes G = 16 for merges of at most 100 rows.
template <int G>
__global__ void __launch_bounds__(64) k_dc_secular(int n, int L, int lev, int nmerge, DcPtrs p, double *__restrict__ dnext_, int *__restrict__ fail_) {
    extern __shared__ __attribute__((aligned(16))) double sms[];
    constexpr int RPW = 64 / G;
    const int b = blockIdx.y / nmerge, q = blockIdx.y % nmerge, lane = threadIdx.x;
    if (!p.flag[b]) return;
    int s, c, t;
    dc_range(n, L, lev, q, s, c, t);
    const size_t o = (size_t)b * n + s;
    const int k = p.meta[o];
    const int j0 = blockIdx.x * RPW;
    if (j0 >= k) return;
    double *dl = sms, *w = dl + k;
    for (int i = lane; i < k; i += WAVE) { dl[i] = p.dl[o + i]; w[i] = p.wv[o + i]; }
    __syncthreads();
    const int g = lane / G, j = j0 + g;
    if (j < k) {
        double *delta = w + k + (size_t)g * k;
        double lam;
        int rc;
        if constexpr (G == 64) { DcWaveRed red; rc = dc_secular(red, k, j, dl, w, p.rhom[o], delta, &lam); }
        else { DcRow16Red red; rc = dc_secular(red, k, j, dl, w, p.rhom[o], delta, &lam); red.sync(); }
        double *Srow = p.S + ((size_t)b * n + s + j) * n + s;
        for (int i = lane % G; i < k; i += G) Srow[i] = delta[i];
        if (lane % G == 0) { dnext_[o + j] = lam; if (rc < 0) atomicAdd(&fail_[b], 1); }
    }
}

// Loewner weights: zhat_i = sign(w_i) sqrt( -S[i][i] prod_{j != i} S[j][i] / (dl_i - dl_j) ).   grid (ceil(m_max / 64), merges * B),
// block 1024: lane = pole i (coalesced rows of S), the sixteen waves split the j range.
__global__ void __launch_bounds__(1024) k_dc_zhat(int n, int L, int lev, int nmerge, DcPtrs p) {
    __shared__ double part[16][64];
    const int b = blockIdx.y / nmerge, q = blockIdx.y % nmerge, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (!p.flag[b]) return;
    int s, c, t;
    dc_range(n, L, lev, q, s, c, t);
    const size_t o = (size_t)b * n + s;
    const int k = p.meta[o];
    const int i = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= k) return;
    const double *S = p.S + ((size_t)b * n + s) * n + s, *dl = p.dl + o;
    double prod = 1.0;
    if (i < k) {
        const double di = dl[i];
        for (int j = wid; j < k; j += 16) {
            const double sji = S[(size_t)j * n + i];
            prod *= (j == i) ? -sji : sji / (di - dl[j]);
        }
    }
    part[wid][lane] = prod;
    __syncthreads();
    if (wid == 0 && i < k) {
        double pr = 1.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) pr *= part[u][lane];
        p.wh[o + i] = copysign(sqrt(fabs(pr)), p.wv[o + i]);
    }
}

// grid (m_max, merges * B), block 64: wave j < k writes row j of U (position-permuted columns), wave j >= k copies a deflated eigenvector
__global__ void __launch_bounds__(64) k_dc_ufill(int n, int L, int lev, int nmerge, DcPtrs p, const double *__restrict__ Xcur_, double *__restrict__ Xnext_) {
    const int b = blockIdx.y / nmerge, q = blockIdx.y % nmerge, j = blockIdx.x, lane = threadIdx.x;
    if (!p.flag[b]) return;
    int s, c, t;
    dc_range(n, L, lev, q, s, c, t);
    const int m = t - s;
    if (j >= m) return;
    const size_t o = (size_t)b * n + s;
    const int k = p.meta[o];
    if (j < k) {
        const double *Srow = p.S + ((size_t)b * n + s + j) * n + s, *wh = p.wh + o;
        const int *pos = p.pos + o;
        double *Urow = p.U + ((size_t)b * n + s + j) * n + s;
        double acc = 0.0;
        for (int i = lane; i < k; i += WAVE) { const double u = wh[i] / Srow[i]; acc = fma(u, u, acc); }
        const double inv = 1.0 / sqrt(wave_sum(acc));
        for (int i = lane; i < k; i += WAVE) Urow[pos[i]] = (wh[i] / Srow[i]) * inv;
    } else {
        const int src = p.dfl[o + (j - k)];
        const double *in = Xcur_ + ((size_t)b * n + s + src) * n + s;
        double *out = Xnext_ + ((size_t)b * n + s + j) * n + s;
        for (int x = lane; x < m; x += WAVE) out[x] = in[x];
    }
}

// Xnext[s + j][s + x] = sum_ip U[s + j][s + ip] * Xcur[s + ndp[ip]][s + x]   (j, ip < k; x < m)
// grid (ceil(m_max / 64), ceil(m_max / 64), merges * B), block 256 = 64 x 64 tile, operand layout as in k_gemm.hip.
// Columns x of the upper half receive nothing from lower-only eigenvectors and vice versa: the ip range is cut accordingly.
__global__ void __launch_bounds__(256) k_dc_gemm(int n, int L, int lev, int nmerge, DcPtrs p, const double *__restrict__ Xcur_, double *__restrict__ Xnext_) {
    __shared__ double As[64][17];
    __shared__ double Bs[16][65];
    __shared__ int rows[16];
    const int b = blockIdx.z / nmerge, q = blockIdx.z % nmerge;
    if (!p.flag[b]) return;
    int s, c, t;
    dc_range(n, L, lev, q, s, c, t);
    const int m = t - s, m1 = c - s;
    const size_t o = (size_t)b * n + s;
    const int k = p.meta[o], k1 = p.meta[o + 1], k2 = p.meta[o + 2];
    const int j0 = blockIdx.y * 64, x0 = blockIdx.x * 64;
    if (j0 >= k || x0 >= m) return;
    int ilo = 0, ihi = k;
    if (x0 + 64 <= m1) ihi = k1 + k2; else if (x0 >= m1) ilo = k1;
    const double *U = p.U + ((size_t)b * n + s) * n + s;
    const double *X = Xcur_ + (size_t)b * n * n + s;
    const int *ndp = p.ndp + o;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = (wid >> 1) * 32, wc = (wid & 1) * 32;
    dc_double4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int jj = 0; jj < 2; ++jj) acc[i][jj] = (dc_double4){0.0, 0.0, 0.0, 0.0};
    for (int i0 = ilo; i0 < ihi; i0 += 16) {
        if (tid < 16) rows[tid] = (i0 + tid < ihi) ? s + ndp[i0 + tid] : -1;
        {
            const int r = tid >> 2, cc = (tid & 3) * 4;
            const int gj = j0 + r;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int gi = i0 + cc + u; As[r][cc + u] = (gj < k && gi < ihi) ? U[(size_t)gj * n + gi] : 0.0; }
        }
        __syncthreads();
        {
            const int r = tid >> 4, cc = (tid & 15) * 4;
            const int gr = rows[r];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int gx = x0 + cc + u; Bs[r][cc + u] = (gr >= 0 && gx < m) ? X[(size_t)gr * n + gx] : 0.0; }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            double a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[wr + i * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) bb[jj] = Bs[kk + (lane >> 4)][wc + jj * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[jj], acc[i][jj], 0, 0, 0);
        }
        __syncthreads();
    }
    double *Xn = Xnext_ + (size_t)b * n * n + s;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gj = j0 + wr + i * 16 + (lane >> 4) + 4 * r;
                const int gx = x0 + wc + jj * 16 + (lane & 15);
                if (gj < k && gx < m) Xn[(size_t)(s + gj) * n + gx] = acc[i][jj][r];
            }
}

// grid B: eigenvalues back to the caller's scale
__global__ void __launch_bounds__(256) k_dc_final(int n, DcPtrs p, const double *__restrict__ dfin_, double *__restrict__ WT_) {
    const int b = blockIdx.x;
    if (!p.flag[b]) return;
    const double sc = p.scale[b];
    for (int i = threadIdx.x; i < n; i += blockDim.x) WT_[(size_t)b * n + i] = dfin_[(size_t)b * n + i] * sc;
}

}  // namespace mopt

// levels of the launch-per-stage chain for n rows (for the launch count of the timing scope)
int mopt_dc_chain_levels(int n) { return mopt::dc_levels(n); }      // (upper bound: the workgroup leaves save a level)

size_t mopt_dc_scratch_bytes(int n, int batch) {
    const size_t nv = (size_t)batch * n;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return 2 * al((size_t)batch * 8) + 10 * al(nv * 8) + 7 * al(nv * 4) + 2 * al(nv * n * 8) + 4096;
}

// Launches the whole D&C chain on `st`.  d_flag: per-matrix switch written by k_tri_bisect (device); nothing here depends on it
// on the host.  Eigenvectors end in d_Vt (vector-major), eigenvalues (unsorted) in d_WT; d_Z2 (n^2 per matrix) is the ping-pong
// partner.
int mopt_launch_dc(mopt_ctx *ctx, hipStream_t st, int n, int batch, const double *d_D, const double *d_E, int *d_flag, double *d_WT,
                   double *d_Vt, double *d_Z2, int *d_bstart, int *d_bend, int *d_fail, void *scratch) {
    using namespace mopt;
    static const bool leaf_ql = getenv("MOPT_DC_LEAFQL") != nullptr;      // developer switch: the one-wave QL leaves of round 2
    // (workgroup leaves while all of them are resident at once -- two per CU --: 64 x 600 would need 1024 workgroups of 1024 threads in four
    // rounds, 2.45 instead of 2.28 ms for its chain; its 2048 one-wave leaves run side by side)
    const bool leaf_wave = leaf_ql || ((size_t)batch << dc_levels_wg(n)) > (size_t)2 * ctx->num_cu;
    const int L = leaf_wave ? dc_levels(n) : dc_levels_wg(n);
    if (L < 1) return MOPT_OK;
    const size_t nv = (size_t)batch * n;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char *pp = (char *)scratch;
    auto take = [&](size_t bytes) { void *r = pp; pp += al(bytes); return r; };
    DcPtrs p;
    p.flag = d_flag;
    p.scale = (double *)take((size_t)batch * 8); take((size_t)batch * 8);
    p.D2 = (double *)take(nv * 8); p.E2 = (double *)take(nv * 8); p.rho = (double *)take(nv * 8);
    p.d0 = (double *)take(nv * 8); p.d1 = (double *)take(nv * 8);
    p.dl = (double *)take(nv * 8); p.wv = (double *)take(nv * 8); p.wh = (double *)take(nv * 8); p.rhom = (double *)take(nv * 8);
    take(nv * 8);
    p.bs2 = (int *)take(nv * 4); p.be2 = (int *)take(nv * 4);
    p.nd = (int *)take(nv * 4); p.ndp = (int *)take(nv * 4); p.dfl = (int *)take(nv * 4); p.pos = (int *)take(nv * 4); p.meta = (int *)take(nv * 4);
    p.S = (double *)take(nv * n * 8); p.U = (double *)take(nv * n * 8);
    double *X[2];
    X[0] = (L % 2 == 0) ? d_Vt : d_Z2;             // level l writes X[l & 1]; the last level must write d_Vt
    X[1] = (L % 2 == 0) ? d_Z2 : d_Vt;
    double *dv[2] = {p.d0, p.d1};
    hipLaunchKernelGGL(k_dc_setup, dim3(batch), dim3(256), 0, st, n, L, d_D, d_E, p, d_bstart, d_bend);
    if (leaf_wave) hipLaunchKernelGGL(k_dc_leaf, dim3(1 << L, batch), dim3(64), 0, st, n, L, p, dv[0], X[0], d_fail);
    else {
        int mcap = 0;
        for (int q = 0; q < (1 << L); ++q) { const int m = dc_bound(n, L, q + 1) - dc_bound(n, L, q); mcap = m > mcap ? m : mcap; }
        // (the layout of dc_block_body: small arrays, then X0 + max(X1 + S, the 16 leaf tables))
        const size_t small = (size_t)10 * mcap * 8 + (size_t)mcap * sizeof(DcRot) + 8 * 8 + (size_t)8 * mcap * 4 + 64;
        const size_t tabs = (size_t)3 * mcap * mcap * 8, leaf = ((size_t)mcap * mcap + (size_t)16 * 32 * DCB_LEAF) * 8;
        const size_t lds = small + (tabs > leaf ? tabs : leaf);
        if (mcap > DCB_MAX || lds > (size_t)ctx->lds_limit - 1024) { ctx->err = "k_dc_leaf_wg: leaf too large for the LDS"; return MOPT_E_ARG; }
        if (lds > 48000) MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dc_leaf_wg, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
        hipLaunchKernelGGL(k_dc_leaf_wg, dim3(1 << L, batch), dim3(1024), lds, st, n, L, p, dv[0], X[0], d_fail);
    }
    hipLaunchKernelGGL(k_dc_zero, dim3(n, batch), dim3(256), 0, st, n, (const int *)d_flag, X[1]);
    for (int lev = 1; lev <= L; ++lev) {
        const int nm = 1 << (L - lev);
        int mmax = 0;
        for (int q = 0; q < nm; ++q) {
            const int s = dc_bound(n, L, q << lev), t = dc_bound(n, L, (q + 1) << lev);
            mmax = (t - s > mmax) ? t - s : mmax;
        }
        double *Xc = X[(lev - 1) & 1], *Xn = X[lev & 1];
        const double *dc = dv[(lev - 1) & 1];
        double *dn = dv[lev & 1];
        const size_t lds_prep = (size_t)mmax * (4 * 8 + 7 * 4) + (size_t)mmax * sizeof(DcRot) + ((size_t)mmax / 64 + 2) * 8 + 64;
        if (lds_prep > 48000)
            MOPT_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_dc_prep, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 1024));
        hipLaunchKernelGGL(k_dc_prep, dim3(nm, batch), dim3(256), lds_prep, st, n, L, lev, p, dc, dn, Xc);
        if (mmax > 100) hipLaunchKernelGGL(k_dc_secular<64>, dim3(mmax, nm * batch), dim3(64), (size_t)3 * mmax * 8, st, n, L, lev, nm, p, dn, d_fail);
        else hipLaunchKernelGGL(k_dc_secular<16>, dim3((mmax + 3) / 4, nm * batch), dim3(64), (size_t)6 * mmax * 8, st, n, L, lev, nm, p, dn, d_fail);
        hipLaunchKernelGGL(k_dc_zhat, dim3((mmax + 63) / 64, nm * batch), dim3(1024), 0, st, n, L, lev, nm, p);
        hipLaunchKernelGGL(k_dc_ufill, dim3(mmax, nm * batch), dim3(64), 0, st, n, L, lev, nm, p, (const double *)Xc, Xn);
        const int T = (mmax + 63) / 64;
        hipLaunchKernelGGL(k_dc_gemm, dim3(T, T, nm * batch), dim3(256), 0, st, n, L, lev, nm, p, (const double *)Xc, Xn);
    }
    hipLaunchKernelGGL(k_dc_final, dim3(batch), dim3(256), 0, st, n, p, (const double *)dv[L & 1], d_WT);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}
