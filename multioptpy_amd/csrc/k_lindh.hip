// Lindh's 2007 model Hessian with D3 dispersion on the device: Lindh2007D3ApproxHessian.main (multioptpy/ModelHessian/lindh2007d3.py:777-822),
// the model ApproxHessian.main falls back to for unknown names (approx_hessian.py:72-74,88-91).  No connectivity: every pair contributes a
// stretch (:223-303), every triple a bend (:305-520), every quadruple i < j < k < l a torsion i-j-k-l (:522-641) and an out-of-plane term with
// centre i (:643-775) -- the reference walks them in Python loops (minutes for a hundred atoms).
//
// Assembly without atomics, one workgroup per atom-pair block (a >= b) of the Hessian:
//   * block (a, b), a > b, of the LOWER triangle: stretch(a, b) + the three bends of every third atom c (vertex a, b or c);
//   * block (b, a) of the UPPER triangle: the torsion and the out-of-plane term of every pair c < d of other atoms (the quadruple sorted);
//   * block (a, a): the stretches of all partners on and below its diagonal, and s_a s_a^T of every bend, torsion and out-of-plane term that
//     contains a BELOW its diagonal only -- the reference adds nothing to the diagonal elements there (:512-517, :636-641).
// The reference then mirrors with the rule "a lower entry below 1e-10 takes the upper one, any other lower entry wins" (:803-808): where
// stretch and bend terms are present the torsion / out-of-plane terms of the same element are DROPPED.  k_l07_mirror reproduces that
// element by element; TR/rot projection, eigen-decomposition and |lambda| * 0.1 exp(-|g|^2) follow in the launcher.
// Every term of a block is recomputed by the block that needs it (a quadruple is evaluated by its six off-diagonal and four diagonal blocks):
// ~1e8 term evaluations for 100 atoms, a few milliseconds, deterministic summation order.
#include "pipeline.h"

namespace mopt {

struct L07 { const double *X, *cov, *grp, *G, *KD; int N; };      // per geometry: coordinates, covalent radii, element group, pair tables

__device__ __forceinline__ double l07_alpha(int gi, int gj) {
    const double a[3][3] = {{1.0, 0.3949, 0.3949}, {0.3949, 0.28, 0.12}, {0.3949, 0.12, 0.06}};
    return a[gi][gj];
}
__device__ __forceinline__ double l07_d0(int gi, int gj) {
    const double d[3][3] = {{0.0, 3.6, 3.6}, {3.6, 5.3, 5.3}, {3.6, 5.3, 5.3}};
    return d[gi][gj];
}
constexpr double L07_KR = 0.45, L07_KF = 0.10, L07_KT = 0.0025, L07_KO = 0.16, L07_KD = 0.05, L07_EPS = 1.0e-12;

// pair tables: G = exp(alpha (r_cov^2 - r^2)), KD = D3 "force constant" beyond twice the covalent length (calc_d3_force_const), else 0
__global__ void __launch_bounds__(256) k_l07_pairs(int N, const double *__restrict__ X_, const double *__restrict__ par, double s6, double s8, double a1, double a2,
                                                   double *__restrict__ G_, double *__restrict__ KD_) {
    const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N * N) return;
    const int i = e / N, j = e - i * N;
    const double *X = X_ + (size_t)b * 3 * N;
    const double *cov = par, *c6 = par + N, *uff = par + 2 * N, *r4 = par + 3 * N, *grp = par + 5 * N;
    const double dx = X[3 * i] - X[3 * j], dy = X[3 * i + 1] - X[3 * j + 1], dz = X[3 * i + 2] - X[3 * j + 2];
    const double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2), rc = cov[i] + cov[j];
    const double g = exp(l07_alpha((int)grp[i], (int)grp[j]) * (rc * rc - r2));
    double kd = 0.0;
    if (i != j && r > 2.0 * rc) {
        const double C6 = sqrt(c6[i] * c6[j]), C8 = 3.0 * C6 * sqrt(r4[i] * r4[j]);
        const double bj = (uff[i] + uff[j] + a1) * a2;
        const double r6 = r2 * r2 * r2, r8 = r6 * r2, b6 = bj * bj * bj * bj * bj * bj, b8 = b6 * bj * bj;
        kd = s6 * C6 * (r6 / (r6 + b6)) / r6 + s8 * C8 * (r8 / (r8 + b8)) / r8;
    }
    G_[(size_t)b * N * N + e] = g;
    KD_[(size_t)b * N * N + e] = kd;
}

__device__ __forceinline__ void l07_sub(const double *a, const double *b, double *o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
__device__ __forceinline__ double l07_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void l07_cross(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// Bend with vertex i and partners j > k (lindh2007d3.py:305-520).  Returns the number of direction sets (0: skipped, 1: regular, 2: the two
// perpendicular directions of a linear arrangement); s[d][0..2] = derivative vectors of atoms i, j, k; g = force constant.
__device__ __forceinline__ int l07_bend(const L07 &P, int i, int j, int k, double &g, double s[2][3][3]) {
    const int N = P.N;
    const double *xi = P.X + 3 * i, *xj = P.X + 3 * j, *xk = P.X + 3 * k;
    double vij[3], vik[3], vjk[3];
    l07_sub(xi, xj, vij); l07_sub(xi, xk, vik); l07_sub(xj, xk, vjk);
    const double rij = sqrt(l07_dot(vij, vij)), rik = sqrt(l07_dot(vik, vik)), rjk = sqrt(l07_dot(vjk, vjk));
    const double dotp = l07_dot(vij, vik);
    const double cosang = dotp / (rij * rik);
    if (fabs(cosang - 1.0) < L07_EPS) return 0;
    const int gi = (int)P.grp[i], gj = (int)P.grp[j], gk = (int)P.grp[k];
    const double gij = P.G[(size_t)i * N + j] + 0.5 * L07_KD * P.KD[(size_t)i * N + j];
    const double gik = P.G[(size_t)i * N + k] + 0.5 * L07_KD * P.KD[(size_t)i * N + k];
    g = L07_KF * (gij + 0.5 * L07_KD / L07_KR * l07_d0(gi, gj)) * (gik + 0.5 * L07_KD / L07_KR * l07_d0(gi, gk));
    double cr[3];
    l07_cross(vij, vik, cr);
    const double rc2 = l07_dot(cr, cr);
    const double rcross = (rc2 > 1.0e-12) ? sqrt(rc2) : 0.0;
    if (rik <= L07_EPS || rij <= L07_EPS || rjk <= L07_EPS) return 0;
    const double sint = rcross / (rij * rik);
    if (sint > L07_EPS) {
        for (int c = 0; c < 3; ++c) {
            s[0][1][c] = (vij[c] / rij * cosang - vik[c] / rik) / (rij * sint);
            s[0][2][c] = (vik[c] / rik * cosang - vij[c] / rij) / (rik * sint);
            s[0][0][c] = -s[0][1][c] - s[0][2][c];
        }
        return 1;
    }
    double p[2][3];
    if (fabs(vij[1]) < L07_EPS && fabs(vij[2]) < L07_EPS) {
        p[0][0] = -vij[1]; p[0][1] = vij[0]; p[0][2] = 0.0;
        p[1][0] = -vij[0] * vij[2]; p[1][1] = -vij[1] * vij[2]; p[1][2] = vij[0] * vij[0] + vij[1] * vij[1];
    } else {
        p[0][0] = 1.0; p[0][1] = 0.0; p[0][2] = 0.0;
        p[1][0] = 0.0; p[1][1] = 1.0; p[1][2] = 0.0;
    }
    for (int d = 0; d < 2; ++d) {
        const double r1 = sqrt(l07_dot(p[d], p[d]));
        for (int c = 0; c < 3; ++c) {
            const double u = p[d][c] / r1;
            s[d][1][c] = -u / rij; s[d][2][c] = -u / rik;
            s[d][0][c] = -s[d][1][c] - s[d][2][c];
        }
    }
    return 2;
}

// Wilson vectors of the torsion 1-2-3-4 (calc_params.py:137-181 with its stretch2 / bend2): bt[4][3]
__device__ __forceinline__ void l07_torsion_bt(const double *x1, const double *x2, const double *x3, const double *x4, double bt[4][3]) {
    double d12[3], d23[3], d34[3];
    l07_sub(x1, x2, d12); l07_sub(x2, x3, d23); l07_sub(x3, x4, d34);
    const double r1 = sqrt(l07_dot(d12, d12)), r2 = sqrt(l07_dot(d23, d23)), r3 = sqrt(l07_dot(d34, d34));
    double e12[3], e23[3], e34[3];                          // b_r[1] of the three stretches: unit vectors (x1 - x2) / r ...
    for (int c = 0; c < 3; ++c) { e12[c] = d12[c] / r1; e23[c] = d23[c] / r2; e34[c] = d34[c] / r3; }
    // bend2(1,2,3): c_o = b_r_ij[0] . b_r_jk[1] = -(e12 . e23); bend2(2,3,4): -(e23 . e34); the reference takes sin / cos of arccos(c_o)
    const double co2 = -l07_dot(e12, e23), co3 = -l07_dot(e23, e34);
    const double f2 = acos(co2), f3 = acos(co3);
    const double s2 = sin(f2), s3 = sin(f3), c2 = cos(f2), c3 = cos(f3);
    double cA[3], cB[3];
    l07_cross(e12, e23, cA);                                // b_r_ij[1] x b_r_jk[1]
    double m34[3] = {-e34[0], -e34[1], -e34[2]}, m23[3] = {-e23[0], -e23[1], -e23[2]};
    l07_cross(m34, m23, cB);                                // b_r_kl[0] x b_r_jk[0]
    for (int c = 0; c < 3; ++c) {
        bt[0][c] = cA[c] / (r1 * s2 * s2);
        bt[3][c] = cB[c] / (r3 * s3 * s3);
        bt[1][c] = -((r2 - r1 * c2) * bt[0][c] + r3 * c3 * bt[3][c]) / r2;
        bt[2][c] = -(bt[0][c] + bt[1][c] + bt[3][c]);
    }
}

// Torsion i-j-k-l, i < j < k < l (lindh2007d3.py:522-641): false when one of the two bond angles is too flat
__device__ __forceinline__ bool l07_torsion(const L07 &P, int i, int j, int k, int l, double &t, double bt[4][3]) {
    const int N = P.N;
    const double *xi = P.X + 3 * i, *xj = P.X + 3 * j, *xk = P.X + 3 * k, *xl = P.X + 3 * l;
    double rij[3], rjk[3], rkl[3];
    l07_sub(xi, xj, rij); l07_sub(xj, xk, rjk); l07_sub(xk, xl, rkl);
    const double a2 = l07_dot(rij, rij), b2 = l07_dot(rjk, rjk), c2 = l07_dot(rkl, rkl);
    const double cosmax = 0.81915204428899180;              // cos(35 degrees)
    if (fabs(l07_dot(rij, rjk) / sqrt(a2 * b2)) > cosmax) return false;
    if (fabs(l07_dot(rkl, rjk) / sqrt(c2 * b2)) > cosmax) return false;
    const int gi = (int)P.grp[i], gj = (int)P.grp[j], gk = (int)P.grp[k], gl = (int)P.grp[l];
    const double gij = P.G[(size_t)i * N + j] + 0.5 * L07_KD * P.KD[(size_t)i * N + j];
    const double gjk = P.G[(size_t)j * N + k] + 0.5 * L07_KD * P.KD[(size_t)j * N + k];
    const double gkl = P.G[(size_t)k * N + l] + 0.5 * L07_KD * P.KD[(size_t)k * N + l];
    const double h = 0.5 * L07_KD / L07_KR;
    t = L07_KT * (gij * h * l07_d0(gi, gj)) * (gjk * h * l07_d0(gj, gk)) * (gkl * h * l07_d0(gk, gl));
    l07_torsion_bt(xi, xj, xk, xl, bt);
    return true;
}

// Out-of-plane term with centre i and partners j < k < l (lindh2007d3.py:643-775, calc_params.py:68-135): bt[0..3] for atoms (i, j, k, l)
__device__ __forceinline__ bool l07_oop(const L07 &P, int i, int j, int k, int l, double &t, double bt[4][3]) {
    const int N = P.N;
    const double *xi = P.X + 3 * i, *xj = P.X + 3 * j, *xk = P.X + 3 * k, *xl = P.X + 3 * l;
    double rij[3], rik[3], ril[3];
    l07_sub(xi, xj, rij); l07_sub(xi, xk, rik); l07_sub(xi, xl, ril);
    const double nij = sqrt(l07_dot(rij, rij)), nik = sqrt(l07_dot(rik, rik)), nil = sqrt(l07_dot(ril, ril));
    if (fabs(fabs(l07_dot(rij, rik) / (nij * nik)) - 1.0) < 0.1) return false;
    if (fabs(fabs(l07_dot(rij, ril) / (nij * nil)) - 1.0) < 0.1) return false;
    if (fabs(fabs(l07_dot(rik, ril) / (nik * nil)) - 1.0) < 0.1) return false;
    t = L07_KO * P.G[(size_t)i * N + j] * P.G[(size_t)i * N + k] * P.G[(size_t)i * N + l];      // (the D3 share is switched off there, :734)
    // outofplane2 on (x_j, x_k, x_l, x_i): a bend over (x_j, x_i, x_i + (x_k - x_i) x (x_l - x_i)); w = its derivative vector of the third point
    double r42[3], r43[3], nv[3];
    l07_sub(xk, xi, r42); l07_sub(xl, xi, r43);
    l07_cross(r42, r43, nv);
    double d1[3];
    l07_sub(xj, xi, d1);                                     // stretch2(x_j, x_i): b[0] = -(x_j - x_i) / r
    const double ra = sqrt(l07_dot(d1, d1)), rb = sqrt(l07_dot(nv, nv));
    double b10[3], b21[3];                                   // b_r_ij[0] = -d1 / ra;  second stretch (x_i, x_i + n): d = -n, b[1] = d / rb = -n / rb
    for (int c = 0; c < 3; ++c) { b10[c] = -d1[c] / ra; b21[c] = -nv[c] / rb; }
    const double co = l07_dot(b10, b21), si = sqrt(1.0 - co * co);
    double w[3];
    for (int c = 0; c < 3; ++c) w[c] = (co * b21[c] - b10[c]) / (rb * si);
    // bt[0] = (w0, w1, w2), bt[1] = (w1, w2, w0), bt[2] = (w2, w0, w1) (the reference's cyclic index walk and its two sign flips), bt[3] = -(sum);
    // c[0..3] belong to (x_j, x_k, x_l, x_i) but the reference pairs them with the atoms (i, j, k, l) in that order (:752-756)
    bt[0][0] = w[0]; bt[0][1] = w[1]; bt[0][2] = w[2];
    bt[1][0] = w[1]; bt[1][1] = w[2]; bt[1][2] = w[0];
    bt[2][0] = w[2]; bt[2][1] = w[0]; bt[2][2] = w[1];
    for (int c = 0; c < 3; ++c) bt[3][c] = -(bt[0][c] + bt[1][c] + bt[2][c]);
    return true;
}

__device__ __forceinline__ void l07_sort4(int &a, int &b, int &c, int &d) {
    int t;
#define L07_CS(x, y) if (x > y) { t = x; x = y; y = t; }
    L07_CS(a, b) L07_CS(c, d) L07_CS(a, c) L07_CS(b, d) L07_CS(b, c)
#undef L07_CS
}

// grid (N, N, batch), block 256; workgroups with blockIdx.y > blockIdx.x return.  Lo / Up: n x n per geometry (zeroed by the launcher).
__global__ void __launch_bounds__(256) k_l07_assemble(int N, const double *__restrict__ X_, const double *__restrict__ par, const double *__restrict__ G_,
                                                      const double *__restrict__ KD_, double s6, double s8, double a1, double a2, double *__restrict__ Lo_,
                                                      double *__restrict__ Up_) {
    __shared__ double red[9 * 16];
    const int a = blockIdx.x, bq = blockIdx.y, bt_ = blockIdx.z, tid = threadIdx.x, n = 3 * N;
    if (bq > a) return;
    L07 P;
    P.X = X_ + (size_t)bt_ * 3 * N; P.cov = par; P.grp = par + 5 * N; P.G = G_ + (size_t)bt_ * N * N; P.KD = KD_ + (size_t)bt_ * N * N; P.N = N;
    double *Lo = Lo_ + (size_t)bt_ * n * n, *Up = Up_ + (size_t)bt_ * n * n;
    double accL[9], accU[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) { accL[q] = 0.0; accU[q] = 0.0; }
    double g, s[2][3][3], t, bt[4][3];
    if (a != bq) {
        const int b = bq;
        // ---- bends: third atom c; vertex a (partners b, c), vertex b (partners a, c), vertex c (partners a, b); the larger partner index is "j"
        for (int c = tid; c < N; c += 256) {
            if (c == a || c == b) continue;
            for (int v = 0; v < 3; ++v) {
                const int vi = (v == 0) ? a : (v == 1) ? b : c;
                const int p0 = (v == 0) ? b : a, p1 = (v == 2) ? b : c;
                const int pj = p0 > p1 ? p0 : p1, pk = p0 > p1 ? p1 : p0;
                const int nd = l07_bend(P, vi, pj, pk, g, s);
                for (int d = 0; d < nd; ++d) {
                    const double *sa = (vi == a) ? s[d][0] : (pj == a) ? s[d][1] : s[d][2];
                    const double *sb = (vi == b) ? s[d][0] : (pj == b) ? s[d][1] : s[d][2];
#pragma unroll
                    for (int l = 0; l < 3; ++l)
#pragma unroll
                        for (int m = 0; m < 3; ++m) accL[3 * l + m] += g * sa[l] * sb[m];
                }
            }
        }
        // ---- torsions and out-of-plane terms: the other two atoms c < d
        for (int e = tid; e < N * N; e += 256) {
            const int c = e / N, d = e - c * N;
            if (c >= d || c == a || c == b || d == a || d == b) continue;
            int i = a, j = b, k = c, l = d;
            l07_sort4(i, j, k, l);
            const int ia = (a == i) ? 0 : (a == j) ? 1 : (a == k) ? 2 : 3, ib = (b == i) ? 0 : (b == j) ? 1 : (b == k) ? 2 : 3;
            if (l07_torsion(P, i, j, k, l, t, bt)) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int m = 0; m < 3; ++m) accU[3 * p + m] += t * bt[ib][p] * bt[ia][m];      // rows of the smaller atom b, columns of a
            }
            if (l07_oop(P, i, j, k, l, t, bt)) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int m = 0; m < 3; ++m) accU[3 * p + m] += t * bt[ib][p] * bt[ia][m];
            }
        }
        block_sum<9>(accL, red);
        block_sum<9>(accU, red);
        if (tid == 0) {
            // ---- stretch (a, b): -(g_mm v v^T / r^2 - D3 gradient block) into the lower block (lindh2007d3.py:223-303)
            const double *cov = par, *c6 = par + N, *uff = par + 2 * N, *r4 = par + 3 * N;
            double v[3];
            l07_sub(P.X + 3 * a, P.X + 3 * b, v);
            const double r2 = l07_dot(v, v), r = sqrt(r2);
            const double gmm = L07_KR * P.G[(size_t)a * N + b] + L07_KD * P.KD[(size_t)a * N + b];
            double dsc = 0.0;
            if (r >= 0.1) {
                const double C6 = sqrt(c6[a] * c6[b]), C8 = 3.0 * C6 * sqrt(r4[a] * r4[b]);
                const double bj = (uff[a] + uff[b] + a1) * a2;
                const double r6 = r2 * r2 * r2, r8 = r6 * r2, b6 = bj * bj * bj * bj * bj * bj, b8 = b6 * bj * bj;
                const double dm6 = r6 / (r6 + b6), dm8 = r8 / (r8 + b8);
                const double dd6 = 6.0 * r6 * b6 / ((r6 + b6) * (r6 + b6) * r), dd8 = 8.0 * r8 * b8 / ((r8 + b8) * (r8 + b8) * r);
                const double force = s6 * C6 * (6.0 * dm6 / (r6 * r) + dd6 / r6) + s8 * C8 * (8.0 * dm8 / (r8 * r) + dd8 / r8);
                dsc = force / r;
            }
            (void)cov;
            for (int l = 0; l < 3; ++l)
                for (int m = 0; m < 3; ++m) {
                    const double h = (gmm - dsc) * v[l] * v[m] / r2;
                    Lo[(size_t)(3 * a + l) * n + 3 * b + m] = accL[3 * l + m] - h;
                    Up[(size_t)(3 * b + l) * n + 3 * a + m] = accU[3 * l + m];
                }
        }
        return;
    }
    // ---- diagonal block (a, a): accL[0..2] = strictly lower elements (1,0), (2,0), (2,1) of s_a s_a^T sums; accU[0..5] = stretch sums (lower incl. diagonal)
    double accD[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) accD[q] = 0.0;
    for (int e = tid; e < N * N; e += 256) {
        const int c = e / N, d = e - c * N;
        if (c == a || d == a || c == d) continue;
        // vertex c with partners a and d (every ordered (c, d): each unordered partner pair {a, d} of vertex c once)
        {
            const int pj = a > d ? a : d, pk = a > d ? d : a;
            const int nd = l07_bend(P, c, pj, pk, g, s);
            for (int q = 0; q < nd; ++q) {
                const double *sa = (pj == a) ? s[q][1] : s[q][2];
                accD[0] += g * sa[1] * sa[0]; accD[1] += g * sa[2] * sa[0]; accD[2] += g * sa[2] * sa[1];
            }
        }
        if (c < d) {                                         // vertex a with the unordered pair {c, d}
            const int nd = l07_bend(P, a, d, c, g, s);
            for (int q = 0; q < nd; ++q) {
                const double *sa = s[q][0];
                accD[0] += g * sa[1] * sa[0]; accD[1] += g * sa[2] * sa[0]; accD[2] += g * sa[2] * sa[1];
            }
        }
    }
    // quadruples that contain a: the other three atoms c < d < e
    for (long long e3 = tid; e3 < (long long)N * N * N; e3 += 256) {
        const int c = (int)(e3 / ((long long)N * N)), rem = (int)(e3 - (long long)c * N * N), d = rem / N, f = rem - d * N;
        if (!(c < d && d < f) || c == a || d == a || f == a) continue;
        int i = a, j = c, k = d, l = f;
        l07_sort4(i, j, k, l);
        const int ia = (a == i) ? 0 : (a == j) ? 1 : (a == k) ? 2 : 3;
        if (l07_torsion(P, i, j, k, l, t, bt)) { accD[0] += t * bt[ia][1] * bt[ia][0]; accD[1] += t * bt[ia][2] * bt[ia][0]; accD[2] += t * bt[ia][2] * bt[ia][1]; }
        if (l07_oop(P, i, j, k, l, t, bt)) { accD[0] += t * bt[ia][1] * bt[ia][0]; accD[1] += t * bt[ia][2] * bt[ia][0]; accD[2] += t * bt[ia][2] * bt[ia][1]; }
    }
    // stretches of all partners: h(a, c) on and below the diagonal
    {
        const double *c6 = par + N, *uff = par + 2 * N, *r4 = par + 3 * N;
        for (int c = tid; c < N; c += 256) {
            if (c == a) continue;
            double v[3];
            l07_sub(P.X + 3 * a, P.X + 3 * c, v);
            const double r2 = l07_dot(v, v), r = sqrt(r2);
            const double gmm = L07_KR * P.G[(size_t)a * N + c] + L07_KD * P.KD[(size_t)a * N + c];
            double dsc = 0.0;
            if (r >= 0.1) {
                const double C6 = sqrt(c6[a] * c6[c]), C8 = 3.0 * C6 * sqrt(r4[a] * r4[c]);
                const double bj = (uff[a] + uff[c] + a1) * a2;
                const double r6 = r2 * r2 * r2, r8 = r6 * r2, b6 = bj * bj * bj * bj * bj * bj, b8 = b6 * bj * bj;
                const double dm6 = r6 / (r6 + b6), dm8 = r8 / (r8 + b8);
                const double dd6 = 6.0 * r6 * b6 / ((r6 + b6) * (r6 + b6) * r), dd8 = 8.0 * r8 * b8 / ((r8 + b8) * (r8 + b8) * r);
                dsc = (s6 * C6 * (6.0 * dm6 / (r6 * r) + dd6 / r6) + s8 * C8 * (8.0 * dm8 / (r8 * r) + dd8 / r8)) / r;
            }
            const double f = (gmm - dsc) / r2;
            accD[3] += f * v[0] * v[0]; accD[4] += f * v[1] * v[0]; accD[5] += f * v[1] * v[1];
            accD[6] += f * v[2] * v[0]; accD[7] += f * v[2] * v[1]; accD[8] += f * v[2] * v[2];
        }
    }
    block_sum<9>(accD, red);
    if (tid == 0) {
        const size_t o = (size_t)(3 * a) * n + 3 * a;
        Lo[o] = accD[3];
        Lo[o + n] = accD[4] + accD[0]; Lo[o + n + 1] = accD[5];
        Lo[o + 2 * n] = accD[6] + accD[1]; Lo[o + 2 * n + 1] = accD[7] + accD[2]; Lo[o + 2 * n + 2] = accD[8];
    }
}

// the reference's mirroring (lindh2007d3.py:803-808), element by element: H[p][q] = H[q][p] = |Lo[p][q]| < 1e-10 ? Up[q][p] : Lo[p][q]  (p > q)
__global__ void __launch_bounds__(256) k_l07_mirror(int n, const double *__restrict__ Lo_, const double *__restrict__ Up_, double *__restrict__ H_) {
    const int b = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * n) return;
    const int p = (int)(e / n), q = (int)(e - (size_t)p * n);
    const double *Lo = Lo_ + (size_t)b * n * n, *Up = Up_ + (size_t)b * n * n;
    double v;
    if (p == q) v = Lo[e];
    else {
        const int hi = p > q ? p : q, lo = p > q ? q : p;
        const double l = Lo[(size_t)hi * n + lo];
        v = (fabs(l) < 1.0e-10) ? Up[(size_t)lo * n + hi] : l;
    }
    H_[(size_t)b * n * n + e] = v;
}

// H = V diag(f(lambda)) V^T from the eigen-solver's outputs (Vt vector-major, W ascending through perm): mode 0: |lambda| * scale[b]
// (Lindh2007D3ApproxHessian.main, :813-820), mode 1: the dispatcher's eigenvalue smoothing `clip` (approx_hessian.py:104-120, alpha = 0.1).
// grid (ceil(n/16), ceil(n/16), batch), block 256: 16 x 16 tile, sum over the eigenvectors in LDS slices of 16.
__global__ void __launch_bounds__(256) k_l07_rebuild(int n, int mode, const double *__restrict__ Vt_, const double *__restrict__ W_, const int *__restrict__ perm_,
                                                     const int *__restrict__ einfo_, const double *__restrict__ scale_, const double *__restrict__ Hin_,
                                                     double *__restrict__ H_) {
    __shared__ double A[16][17], B[16][17], F[16];
    const int b = blockIdx.z, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int p = blockIdx.y * 16 + ty, q = blockIdx.x * 16 + tx;
    double *H = H_ + (size_t)b * n * n;
    if (einfo_[2 * b + 1] & 1) {                             // non-finite eigen-system (a NaN / Inf geometry): nothing sensible to rebuild
        if (p < n && q < n) H[(size_t)p * n + q] = Hin_ ? Hin_[(size_t)b * n * n + (size_t)p * n + q] : NAN;
        return;
    }
    const double *Vt = Vt_ + (size_t)b * n * n, *W = W_ + (size_t)b * n;
    const int *perm = perm_ + (size_t)b * n;
    double acc = 0.0;
    for (int i0 = 0; i0 < n; i0 += 16) {
        const int i = i0 + ty;                               // sorted position; eigenvector row perm[i]
        const int row = (i < n) ? perm[i] : 0;
        A[ty][tx] = (i < n && blockIdx.y * 16 + tx < n) ? Vt[(size_t)row * n + blockIdx.y * 16 + tx] : 0.0;
        B[ty][tx] = (i < n && q < n) ? Vt[(size_t)row * n + q] : 0.0;
        if (tx == 0) {
            double f = 0.0;
            if (i < n) {
                const double w = W[i];
                if (mode == 0) f = fabs(w) * scale_[b];
                else f = (fabs(w) >= 1.0) ? copysign(2.0 - 1.0 / pow(fabs(w), 0.1), w) : w;
            }
            F[ty] = f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(F[k] * A[k][ty], B[k][tx], acc);
        __syncthreads();
    }
    if (p < n && q < n) H[(size_t)p * n + q] = acc;
}

// scale[b] = 0.1 exp(-|g_b|^2)   (lindh2007d3.py:791-793); g may be null (zero gradient)
__global__ void __launch_bounds__(256) k_l07_scale(int n, const double *__restrict__ g_, double *__restrict__ scale_) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    double a = 0.0;
    if (g_) for (int i = threadIdx.x; i < n; i += 256) { const double v = g_[(size_t)b * n + i]; a += v * v; }
    a = block_sum1(a, red);
    if (threadIdx.x == 0) scale_[b] = 0.1 * exp(-a);
}

}  // namespace mopt

// Unprojected, mirrored Lindh2007D3 Hessian into d_H.  d_par as for the other models with the UFF van-der-Waals distance (bohr) in the vdw slot
// and the element group (0 H, 1 He..F, 2 else) in the model-specific slot; d_scr: 4 n^2 doubles per geometry... (here: Lo, Up, G, KD).
int mopt_launch_lindh2007d3(mopt_ctx *ctx, int N, int batch, const double *d_X, const double *d_par, const double *d3, double *d_Lo, double *d_Up, double *d_pair,
                            double *d_H) {
    using namespace mopt;
    const int n = 3 * N;
    double *G = d_pair, *KD = d_pair + (size_t)batch * N * N;
    MOPT_HIP_TRY(ctx, hipMemsetAsync(d_Lo, 0, (size_t)batch * n * n * 8, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemsetAsync(d_Up, 0, (size_t)batch * n * n * 8, ctx->stream));
    hipLaunchKernelGGL(k_l07_pairs, dim3((N * N + 255) / 256, batch), dim3(256), 0, ctx->stream, N, d_X, d_par, d3[0], d3[1], d3[2], d3[3], G, KD);
    hipLaunchKernelGGL(k_l07_assemble, dim3(N, N, batch), dim3(256), 0, ctx->stream, N, d_X, d_par, (const double *)G, (const double *)KD, d3[0], d3[1], d3[2], d3[3],
                       d_Lo, d_Up);
    hipLaunchKernelGGL(k_l07_mirror, dim3((unsigned)(((size_t)n * n + 255) / 256), batch), dim3(256), 0, ctx->stream, n, (const double *)d_Lo, (const double *)d_Up, d_H);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

int mopt_launch_l07_scale(mopt_ctx *ctx, int n, int batch, const double *d_g, double *d_scale) {
    hipLaunchKernelGGL(mopt::k_l07_scale, dim3(batch), dim3(256), 0, ctx->stream, n, d_g, d_scale);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

// d_H = V f(Lambda) V^T from wk's eigen-system (mode 0: |lambda| scale, mode 1: clip)
int mopt_launch_l07_rebuild(mopt_ctx *ctx, int n, int batch, int mode, const double *d_Vt, const double *d_W, const int *d_perm, const int *d_einfo,
                            const double *d_scale, const double *d_Hin, double *d_H) {
    const int T = (n + 15) / 16;
    hipLaunchKernelGGL(mopt::k_l07_rebuild, dim3(T, T, batch), dim3(256), 0, ctx->stream, n, mode, d_Vt, d_W, d_perm, d_einfo, d_scale, d_Hin, d_H);
    MOPT_HIP_TRY(ctx, hipGetLastError());
    return MOPT_OK;
}

// Lindh2007D3ApproxHessian.main for a batch of geometries, result (projected, eigenvalues |lambda| * 0.1 exp(-|g|^2)) into d_out [batch][n][n].
// d_g: gradients [batch][n] or null (zero).  wk: an RSIRFO workspace of this shape -- its matrices are scratch here (Gt, Vt: lower / upper
// assemblies; Z2: the mirrored matrix; A: pair tables, then the projected matrix and the eigen-solver's input).
int mopt_lindh2007d3_full(mopt_ctx *ctx, int N, int batch, const double *d_X, const double *d_par, const double *d3, const double *d_g, RsirfoWork &wk,
                          double *d_out) {
    const int n = 3 * N;
    int rc = mopt_launch_lindh2007d3(ctx, N, batch, d_X, d_par, d3, wk.Gt, wk.Vt, wk.A, wk.tri.Z2);
    if (rc != MOPT_OK) return rc;
    rc = mopt_launch_project_hess(ctx, n, batch, wk.tri.Z2, nullptr, d_X, wk.Q, wk.W6, wk.A);
    if (rc != MOPT_OK) return rc;
    rc = mopt_launch_l07_scale(ctx, n, batch, d_g, wk.be);      // (wk.sigma is the Jacobi route's shift)
    if (rc != MOPT_OK) return rc;
    const int hint = wk.defl_hint;
    wk.defl_hint = 0;                                        // (dense matrices)
    rc = mopt_eigh_auto(ctx, wk, n, batch, false, /*preserve_A=*/false, nullptr);
    wk.defl_hint = hint;
    if (rc != MOPT_OK) return rc;
    return mopt_launch_l07_rebuild(ctx, n, batch, 0, wk.Vt, wk.eigw, wk.perm, wk.einfo, wk.be, nullptr, d_out);
}

// The dispatcher's `clip` modifier (approx_hessian.py:104-120) on top of any model: d_H <- V smooth(lambda) V^T, in place.
int mopt_model_hessian_clip(mopt_ctx *ctx, int n, int batch, RsirfoWork &wk, double *d_H) {
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(wk.A, d_H, (size_t)batch * n * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    const int hint = wk.defl_hint;
    wk.defl_hint = 0;
    int rc = mopt_eigh_auto(ctx, wk, n, batch, false, /*preserve_A=*/false, nullptr);
    wk.defl_hint = hint;
    if (rc != MOPT_OK) return rc;
    return mopt_launch_l07_rebuild(ctx, n, batch, 1, wk.Vt, wk.eigw, wk.perm, wk.einfo, nullptr, nullptr, d_H);
}
