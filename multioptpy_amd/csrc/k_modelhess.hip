// Model Hessian on the device: FischerD3ApproxHessianOld.main (multioptpy/ModelHessian/fischerd3old.py:355-381), the model the
// reference's `-modelhess` switch selects by default (interface.py:325-330) and one of the two switches that turn the RFO-NEB
// path on (neb.py:794-804, recomputed every -mfc steps per image: neb.py:515-523).  The same kernels carry the two other Fischer
// models of ApproxHessian.main as a compile-free `kind` argument -- they differ in guards, in the connectivity table that excludes
// pairs and counts torsion bonds, and in the pair term:
//   MOPT_MODEL_FISCHERD3  FischerD3ApproxHessian.main (fischerd3.py:177-304): C6 of every atom scaled by its fractional coordination
//                         number (fischerd3.py:46-60, 224-229; k_mh_connect), stricter guards (fischerd3.py:88, 144-155)
//   MOPT_MODEL_FISCHER    FischerApproxHessian.main (fischer.py:216-236): no pair term, no angle guard, cross-product linearity test
//                         of the torsions (fischer.py:140-170), no sin^2 damping
//   MOPT_MODEL_MORSE      MorseApproxHessian.create_model_hessian (morse.py:50-106): every atom pair a Morse bond, k = De a^2 (4 e^-2x - 2 e^-x),
//                         x = a (r - r_i - r_j) with the GNB radii (par[5 N + i]), De = 0.10, a = 0.20; k u u^T blocks in the pair loop of
//                         k_mh_assemble, symmetrised, NOT TR/rot-projected (the reference returns it unprojected)
//   MOPT_MODEL_SCHLEGEL   SchlegelApproxHessian.main (schlegel.py:66-144): H = B^T K B over the all-pairs stretch rows of
//                         redundant_coordinate.py:15-43 with a diagonal K filled from the three tables -- every contribution lands on a
//                         bonded pair, so the model is "one spring per bond": the bond lane of k_mh_internals walks the angle and
//                         dihedral tables in the reference's order (same sum, bit for bit) and k_mh_assemble adds k u u^T blocks
//   k_mh_connect     distance tests -> two connectivity tables: 1.1 x covalent radii (BondConnectivity, Utils/bond_connectivity.py:
//                    14-41: the internal coordinates) and 1.3 x (fischerd3old.py:136-154: torsion bond counts, D3 exclusions)
//   k_mh_enumerate   bond / angle / dihedral tables in the reference's order (bond_connectivity.py:43-128; the dihedral rules depend
//                    on which angle of a pair comes first) by count -> prefix sum -> fill; one workgroup per image
//   k_mh_internals   force constant (fischerd3old.py:25-48 and the guards of :190-325) and Wilson B-vectors (calc_params.py:139-227)
//                    of every internal coordinate, one lane each
//   k_mh_assemble    H row block of one atom: sum_e fc_e b_e(m) b_e(q)^T over the internals that contain the atom, in table order,
//                    plus the D3 pair blocks (fischerd3old.py:84-131, 328-352) -- one workgroup per atom, no atomics: the result is
//                    deterministic
// followed by the symmetrisation and the TR/rot projection kernels of k_project.hip (Utils/calc_tools.py:249-316).
// Images are batched (grid.y); everything stays in HBM, the band takes the result as its per-image Hessians.
#include "pipeline.h"

namespace mopt {

struct MhWork {
    unsigned char *c11, *c13;     // [B][N][N]
    int *nb13;                    // [B][N] bonds per atom in the table the model counts torsion bonds with (1.3 x; fischerd3.py: 1.1 x)
    double *c6s;                  // [B][N] C6 scaling of the atom (1 unless MOPT_MODEL_FISCHERD3)
    int *cnt;                     // [B][4]: bonds, angles, dihedrals, overflow flag
    int *atoms;                   // [B][cap][4]
    double *fc;                   // [B][cap]
    double *bv;                   // [B][cap][12]
    int *scr;                     // [B][cap + 2 N + 8] scratch of the enumeration (counts / offsets)
    int cap, bcap, acap;
};

// X: [B][N][3]; par: cov[N], c6[N], vdw[N], r4r2[N], refcn[N]
__global__ void __launch_bounds__(128) k_mh_connect(int kind, int N, const double *__restrict__ X_, const double *__restrict__ par, MhWork w) {
    __shared__ int red[2];
    __shared__ double cnp[128];
    const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
    const double *x = X_ + (size_t)b * N * 3, *cov = par;
    unsigned char *c11 = w.c11 + ((size_t)b * N + i) * N, *c13 = w.c13 + ((size_t)b * N + i) * N;
    if (tid < 2) red[tid] = 0;
    __syncthreads();
    const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2], ci = cov[i];
    int n13 = 0, n11 = 0;
    double cn = 0.0;
    for (int j = tid; j < N; j += blockDim.x) {
        const double dx = x[3 * j] - xi, dy = x[3 * j + 1] - yi, dz = x[3 * j + 2] - zi;
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        const bool a = (j != i) && d <= (cov[j] + ci) * 1.1, c = (j != i) && d <= (ci + cov[j]) * 1.3;
        c11[j] = a ? 1 : 0; c13[j] = c ? 1 : 0;
        n13 += c ? 1 : 0; n11 += a ? 1 : 0;
        if (kind == MOPT_MODEL_FISCHERD3) {
            // "fractional coordination number" as fischerd3.py:46-60 computes it (k1 = 16, k2 = 4/3, exponent clipped to +-100).  With that
            // sign the count tends to 1 for DISTANT atoms, and the diagonal (distance set to infinity, :53) adds exp(-100) -> 1 as well:
            // reproduced as it is, the scaling of :226-227 is pinned by G10b.
            const double t = (j == i) ? -100.0 : fmin(fmax(-16.0 * ((4.0 / 3.0) * (d / (ci + cov[j])) - 1.0), -100.0), 100.0);
            cn += 1.0 / (1.0 + exp(t));
        }
    }
    atomicAdd(&red[0], n13); atomicAdd(&red[1], n11);
    cnp[tid] = cn;
    __syncthreads();
    if (tid == 0) {
        w.nb13[(size_t)b * N + i] = (kind == MOPT_MODEL_FISCHERD3) ? red[1] : red[0];
        double sc = 1.0;
        if (kind == MOPT_MODEL_FISCHERD3) {
            double s = 0.0;
            for (int t = 0; t < (int)blockDim.x; ++t) s += cnp[t];      // (fixed order: the result does not depend on scheduling)
            sc = fmin(fmax(1.0 - 0.05 * (s - par[4 * (size_t)N + i]), 0.75), 1.25);      // fischerd3.py:226-227
        }
        w.c6s[(size_t)b * N + i] = sc;
    }
}

// exclusive prefix sum of v[0 .. m) in place (one workgroup, m small: serial chunks per thread + block scan of the chunk sums)
__device__ inline int block_excl_scan(int *v, int m, int *tmp /* blockDim ints */) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (m + nt - 1) / nt, lo = tid * per, hi = min(m, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += v[i];
    tmp[tid] = s;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int t = 0; t < nt; ++t) { const int c = tmp[t]; tmp[t] = acc; acc += c; } tmp[nt] = acc; }
    __syncthreads();
    int acc = tmp[tid];
    for (int i = lo; i < hi; ++i) { const int c = v[i]; v[i] = acc; acc += c; }
    __syncthreads();
    return tmp[nt];
}

// the dihedral of the angle pair (a first, b second), bond_connectivity.py:74-126; returns false when no rule yields one
__device__ inline bool dihedral_of(const unsigned char *C, int N, int a0, int a1, int a2, int b0, int b1, int b2, int q[4]) {
    auto bonded = [&](int u, int v) { return C[(size_t)u * N + v] != 0; };
    auto tail = [&](int x) { if (bonded(a2, x)) { q[0] = a0; q[1] = a1; q[2] = a2; q[3] = x; return true; } return false; };
    auto head = [&](int x) { if (bonded(a0, x)) { q[0] = x; q[1] = a0; q[2] = a1; q[3] = a2; return true; } return false; };
    if ((a1 == b1 && a2 == b2) || (a1 == b2 && a2 == b1)) { if (tail(b0)) return true; if (head(b0)) return true; }
    if ((a1 == b1 && a0 == b0) || (a1 == b0 && a0 == b1)) { if (head(b2)) return true; if (tail(b2)) return true; }
    if ((a1 == b0 && a2 == b1) || (a1 == b1 && a2 == b0)) { if (tail(b2)) return true; if (head(b2)) return true; }
    if ((a0 == b1 && a1 == b2) || (a0 == b2 && a1 == b1)) { if (head(b0)) return true; if (tail(b0)) return true; }
    return false;
}

// grid B, block 256
__global__ void __launch_bounds__(256) k_mh_enumerate(int N, MhWork w) {
    __shared__ int tmp[257];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const unsigned char *C = w.c11 + (size_t)b * N * N;
    int *atoms = w.atoms + (size_t)b * w.cap * 4;
    int *scr = w.scr + (size_t)b * (w.cap + 2 * N + 8);
    int *cnt = w.cnt + 4 * b;
    // ---- bonds (i <= j, row-major order)
    for (int i = tid; i < N; i += nt) { int c = 0; for (int j = i + 1; j < N; ++j) c += C[(size_t)i * N + j]; scr[i] = c; }
    __syncthreads();
    const int nb = block_excl_scan(scr, N, tmp);
    const bool ovb = nb > w.bcap;
    if (!ovb)
        for (int i = tid; i < N; i += nt) { int o = scr[i]; for (int j = i + 1; j < N; ++j) if (C[(size_t)i * N + j]) { int *a = atoms + 4 * (size_t)o++; a[0] = i; a[1] = j; a[2] = -1; a[3] = -1; } }
    __syncthreads();
    // ---- angles [j, i, m]: centre i, j bonded to i, m > j bonded to i and not to j
    for (int i = tid; i < N; i += nt) {
        int c = 0;
        for (int j = 0; j < N; ++j) if (C[(size_t)i * N + j]) for (int m = j + 1; m < N; ++m) c += (C[(size_t)i * N + m] && !C[(size_t)j * N + m]) ? 1 : 0;
        scr[i] = c;
    }
    __syncthreads();
    const int na = block_excl_scan(scr, N, tmp);
    const bool ova = ovb || na > w.acap;
    const int abase = ovb ? 0 : nb;
    if (!ova)
        for (int i = tid; i < N; i += nt) {
            int o = abase + scr[i];
            for (int j = 0; j < N; ++j) if (C[(size_t)i * N + j]) for (int m = j + 1; m < N; ++m)
                if (C[(size_t)i * N + m] && !C[(size_t)j * N + m]) { int *a = atoms + 4 * (size_t)o++; a[0] = j; a[1] = i; a[2] = m; a[3] = -1; }
        }
    __syncthreads();
    // ---- dihedrals: one per angle pair (ia < ib) at most
    int nd = 0;
    if (!ova) {
        int *dc = scr + N;                                   // [na] counts -> offsets
        const int *ang = atoms + 4 * (size_t)abase;
        for (int ia = tid; ia < na; ia += nt) {
            const int a0 = ang[4 * ia], a1 = ang[4 * ia + 1], a2 = ang[4 * ia + 2];
            int c = 0, q[4];
            for (int ib = ia + 1; ib < na; ++ib) c += dihedral_of(C, N, a0, a1, a2, ang[4 * ib], ang[4 * ib + 1], ang[4 * ib + 2], q) ? 1 : 0;
            dc[ia] = c;
        }
        __syncthreads();
        nd = block_excl_scan(dc, na, tmp);
        if (abase + na + nd <= w.cap) {
            for (int ia = tid; ia < na; ia += nt) {
                const int a0 = ang[4 * ia], a1 = ang[4 * ia + 1], a2 = ang[4 * ia + 2];
                int o = abase + na + dc[ia], q[4];
                for (int ib = ia + 1; ib < na; ++ib)
                    if (dihedral_of(C, N, a0, a1, a2, ang[4 * ib], ang[4 * ib + 1], ang[4 * ib + 2], q)) { int *a = atoms + 4 * (size_t)o++; a[0] = q[0]; a[1] = q[1]; a[2] = q[2]; a[3] = q[3]; }
            }
        }
    }
    if (tid == 0) { cnt[0] = ovb ? 0 : nb; cnt[1] = ova ? 0 : na; cnt[2] = nd; cnt[3] = (ova || abase + na + nd > w.cap) ? 1 : 0; if (cnt[3]) cnt[2] = 0; }
}

// Schlegel's B parameters (bohr) by row of the periodic table as schlegel.py:27-63 assigns them (its K-Kr / Rb-Xe / Cs-Rn lists leave
// Na-Ar in the last row: reproduced through the per-atom code the host passes, par[5 N + i] = row + 8 * (element is H))
__constant__ double c_schlegel_b[7][7] = {{0.2573, 0.3401, 0.6937, 0.7126, 0.8335, 0.9491, 0.9491}, {0.3401, 0.9652, 1.2843, 1.4725, 1.6549, 1.7190, 1.7190},
                                          {0.6937, 1.2843, 1.6925, 1.8238, 2.1164, 2.3185, 2.3185}, {0.7126, 1.4725, 1.8238, 2.0203, 2.2137, 2.5206, 2.5206},
                                          {0.8335, 1.6549, 2.1164, 2.2137, 2.3718, 2.5110, 2.5110}, {0.9491, 1.7190, 2.3185, 2.5206, 2.5110, 2.5110, 2.5110},
                                          {0.9491, 1.7190, 2.3185, 2.5206, 2.5110, 2.5110, 2.5110}};

struct V3 { double x, y, z; };
__device__ inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 scale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline double norm(V3 a) { return sqrt(dot(a, a)); }
__device__ inline V3 ld3(const double *x, int a) { return {x[3 * a], x[3 * a + 1], x[3 * a + 2]}; }

// stretch2 (calc_params.py:221-227): unit vectors b0 = -(x0 - x1)/r, b1 = +(x0 - x1)/r
__device__ inline double stretch2(V3 x0, V3 x1, V3 &b0, V3 &b1) {
    const V3 d = sub(x0, x1);
    const double r = norm(d);
    b1 = scale(d, 1.0 / r); b0 = scale(d, -1.0 / r);
    return r;
}
// bend2 (calc_params.py:180-219): angle and B-vectors of x0 - x1 - x2
__device__ inline double bend2(V3 x0, V3 x1, V3 x2, V3 bf[3]) {
    V3 a0, a1, c0, c1;
    const double r1 = stretch2(x0, x1, a0, a1), r2 = stretch2(x1, x2, c0, c1);
    const double co = dot(a0, c1);
    const double crap = dot(a0, a0) + dot(c1, c1);
    double fir, si;
    if (sqrt(crap) < 1e-12) { fir = M_PI - asin(sqrt(crap)); si = sqrt(crap); }
    else { fir = acos(co); si = sqrt(1.0 - co * co); }
    if (fabs(fir - M_PI) < 1e-12) fir = M_PI;
    const double d1 = r1 * si, d2 = r2 * si;
    bf[0] = (d1 < 1e-12) ? V3{0, 0, 0} : scale(sub(scale(a0, co), c1), 1.0 / d1);
    bf[2] = (d2 < 1e-12) ? V3{0, 0, 0} : scale(sub(scale(c1, co), a0), 1.0 / d2);
    bf[1] = scale(V3{bf[0].x + bf[2].x, bf[0].y + bf[2].y, bf[0].z + bf[2].z}, -1.0);
    return fir;
}

// grid (ceil(cap / 64), B), block 64: one internal coordinate per lane
__global__ void __launch_bounds__(64) k_mh_internals(int kind, int N, const double *__restrict__ X_, const double *__restrict__ par, MhWork w) {
    const int b = blockIdx.y, e = blockIdx.x * 64 + threadIdx.x;
    const int *cnt = w.cnt + 4 * b;
    const int nb = cnt[0], na = cnt[1], nd = cnt[2];
    if (e >= nb + na + nd) return;
    const double *x = X_ + (size_t)b * N * 3, *cov = par;
    const int *at = w.atoms + ((size_t)b * w.cap + e) * 4;
    double *bv = w.bv + ((size_t)b * w.cap + e) * 12;
    double fc = 0.0;
    V3 bq[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (kind == MOPT_MODEL_SCHLEGEL) {
        if (e < nb) {
            // the spring of bond (i, j): schlegel.py:82-91 (bond), :93-106 (0.160 / 0.250 per angle that contains the bond), :108-121
            // (0.0023 - 0.07 (r_jk - cov_jk) per dihedral, on each of its three bonds), summed in table order
            const int i = at[0], j = at[1];
            const double *sch = par + 5 * (size_t)N;
            const double r = stretch2(ld3(x, i), ld3(x, j), bq[0], bq[1]);
            const double cb = c_schlegel_b[(int)sch[i] & 7][(int)sch[j] & 7], dd = r - cb;
            double k = 1.734 / (dd * dd * dd);
            const int *tab = w.atoms + (size_t)b * w.cap * 4;
            auto same = [&](int p, int q) { return (p == i && q == j) || (p == j && q == i); };
            for (int a = nb; a < nb + na; ++a) {
                const int a0 = tab[4 * a], a1 = tab[4 * a + 1], a2 = tab[4 * a + 2];
                const double v = (((int)sch[a0] & 8) || ((int)sch[a2] & 8)) ? 0.160 : 0.250;
                if (same(a0, a1)) k += v;
                if (same(a1, a2)) k += v;
            }
            for (int a = nb + na; a < nb + na + nd; ++a) {
                const int q0 = tab[4 * a], q1 = tab[4 * a + 1], q2 = tab[4 * a + 2], q3 = tab[4 * a + 3];
                const bool h0 = same(q0, q1), h1 = same(q1, q2), h2 = same(q2, q3);
                if (h0 || h1 || h2) {
                    const double v = 0.0023 - 1 * 0.07 * (norm(sub(ld3(x, q1), ld3(x, q2))) - (cov[q1] + cov[q2]));
                    if (h0) k += v;
                    if (h1) k += v;
                    if (h2) k += v;
                }
            }
            fc = k;
        }
    } else if (e < nb) {
        const int i = at[0], j = at[1];
        const double r = stretch2(ld3(x, i), ld3(x, j), bq[0], bq[1]);
        fc = (kind == MOPT_MODEL_FISCHERD3 && r < 0.1) ? 0.0 : 0.3601 * exp(-1.944 * (r - (cov[i] + cov[j])));      // (fischerd3.py:88)
    } else if (e < nb + na) {
        const int i = at[0], j = at[1], k = at[2];
        const V3 vij = sub(ld3(x, i), ld3(x, j)), vjk = sub(ld3(x, k), ld3(x, j));
        const double rij = norm(vij), rjk = norm(vjk);
        bool skip = (kind != MOPT_MODEL_FISCHER) && (rij < 0.1 || rjk < 0.1);      // (fischer.py:103-131 has no guard)
        if (!skip && kind != MOPT_MODEL_FISCHER) skip = fabs(dot(vij, vjk) / (rij * rjk)) > 0.9999;
        if (!skip) {
            const double cij = cov[i] + cov[j], cjk = cov[j] + cov[k], val = cij * cjk;
            fc = (fabs(val) < 1e-10) ? 0.0 : 0.089 + 0.11 / pow(val, -0.42) * exp(-0.44 * (rij + rjk - cij - cjk));
            bend2(ld3(x, i), ld3(x, j), ld3(x, k), bq);
        }
    } else {
        const int i = at[0], j = at[1], k = at[2], l = at[3];
        const V3 xi = ld3(x, i), xj = ld3(x, j), xk = ld3(x, k), xl = ld3(x, l);
        const V3 vji = sub(xi, xj), vjk = sub(xk, xj), vkl = sub(xl, xk);
        const double rjk = norm(vjk), cjk = cov[j] + cov[k];
        const int bsum = w.nb13[(size_t)b * N + j] + w.nb13[(size_t)b * N + k] - 2;
        const double val = rjk * cjk;
        double f = (fabs(val) < 1e-10) ? 0.0 : 0.0015 + 14.0 * pow((double)max(bsum, 0), 0.57) / pow(val, 4.0) * exp(-2.85 * (rjk - cjk));
        const double nji = norm(vji), nkl = norm(vkl);
        bool skip;
        double damp = 1.0;
        if (kind == MOPT_MODEL_FISCHER) {
            // fischer.py:140-170: sin^2 of the bond angles i-j-k and j-k-l from cross products, torsion dropped below 1e-3
            auto sin2 = [&](V3 a, V3 c, V3 e2) {
                const V3 v1 = sub(a, c), v2 = sub(e2, c);
                const V3 cr = cross(v1, v2);
                const double n1 = dot(v1, v1), n2 = dot(v2, v2);
                return (n1 * n2 < 1e-12) ? 0.0 : dot(cr, cr) / (n1 * n2);
            };
            skip = sin2(xi, xj, xk) < 1e-3 || sin2(xj, xk, xl) < 1e-3;
        } else {
            const double tiny = (kind == MOPT_MODEL_FISCHERD3) ? 0.1 : 1e-8, lim = (kind == MOPT_MODEL_FISCHERD3) ? 1e-3 : 1e-4;      // fischerd3.py:144,154 / fischerd3old.py:262,276
            skip = (nji < tiny || rjk < tiny || nkl < tiny);
            if (!skip) {
                const double c1 = dot(vji, vjk) / (nji * rjk), c2 = -dot(vjk, vkl) / (rjk * nkl);
                const double s1 = 1.0 - fmin(c1 * c1, 1.0), s2 = 1.0 - fmin(c2 * c2, 1.0);
                skip = (s1 < lim || s2 < lim);
                damp = s1 * s2;
            }
        }
        if (!skip) {
            {
                f *= damp;
                // torsion2 (calc_params.py:139-178)
                V3 bij0, bij1, bjk0, bjk1, bkl0, bkl1, t3[3];
                const double r1 = stretch2(xi, xj, bij0, bij1), r2 = stretch2(xj, xk, bjk0, bjk1), r3 = stretch2(xk, xl, bkl0, bkl1);
                const double f2 = bend2(xi, xj, xk, t3), f3 = bend2(xj, xk, xl, t3);
                const double sn2 = sin(f2), sn3 = sin(f3), cs2 = cos(f2), cs3 = cos(f3);
                bq[0] = scale(cross(bij1, bjk1), 1.0 / (r1 * sn2 * sn2));
                bq[3] = scale(cross(bkl0, bjk0), 1.0 / (r3 * sn3 * sn3));
                const double ca = (r2 - r1 * cs2), cb = r3 * cs3;
                bq[1] = scale(V3{ca * bq[0].x + cb * bq[3].x, ca * bq[0].y + cb * bq[3].y, ca * bq[0].z + cb * bq[3].z}, -1.0 / r2);
                bq[2] = scale(V3{bq[0].x + bq[1].x + bq[3].x, bq[0].y + bq[1].y + bq[3].y, bq[0].z + bq[1].z + bq[3].z}, -1.0);
                fc = f;
                if (kind == MOPT_MODEL_FISCHERD3) {              // fischerd3.py:166: a torsion with non-finite B-vectors is dropped
                    bool fin = true;
                    for (int m = 0; m < 4; ++m) fin = fin && isfinite(bq[m].x) && isfinite(bq[m].y) && isfinite(bq[m].z);
                    if (!fin) { fc = 0.0; for (int m = 0; m < 4; ++m) bq[m] = V3{0, 0, 0}; }
                }
            }
        }
    }
    w.fc[(size_t)b * w.cap + e] = fc;
    for (int m = 0; m < 4; ++m) { bv[3 * m] = bq[m].x; bv[3 * m + 1] = bq[m].y; bv[3 * m + 2] = bq[m].z; }
}

// grid (N, B), block 128.  dynamic LDS: 9 N doubles (the 3 x 3N row block of atom m).  d3 = {s6, s8, a1, a2}
__global__ void __launch_bounds__(128) k_mh_assemble(int kind, int N, const double *__restrict__ X_, const double *__restrict__ par, MhWork w, double s6, double s8,
                                                     double a1, double a2, double *__restrict__ H_) {
    extern __shared__ __attribute__((aligned(16))) double row[];      // [3][3N]
    __shared__ double dsum[128][9];
    const int b = blockIdx.y, m = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, n = 3 * N;
    const double *x = X_ + (size_t)b * N * 3;
    const double *c6 = par + N, *vdw = par + 2 * N, *r4 = par + 3 * N;
    const int *cnt = w.cnt + 4 * b;
    const int ne = (kind == MOPT_MODEL_MORSE) ? 0 : cnt[0] + cnt[1] + cnt[2];
    const int *atoms = w.atoms + (size_t)b * w.cap * 4;
    const double *fcv = w.fc + (size_t)b * w.cap, *bvv = w.bv + (size_t)b * w.cap * 12;
    for (int i = tid; i < 3 * n; i += nt) row[i] = 0.0;
    __syncthreads();
    // ---- internal coordinates that contain atom m, in table order; thread t owns the column atoms q with q % nt == t
    for (int e = 0; e < ne; ++e) {
        const int *at = atoms + 4 * (size_t)e;
        const int a0 = at[0], a1_ = at[1], a2_ = at[2], a3 = at[3];
        int pm = (a0 == m) ? 0 : (a1_ == m) ? 1 : (a2_ == m) ? 2 : (a3 == m) ? 3 : -1;
        if (pm < 0) continue;
        const double fc = fcv[e];
        if (fc == 0.0) continue;
        const double *bv = bvv + 12 * (size_t)e;
        const double bm0 = fc * bv[3 * pm], bm1 = fc * bv[3 * pm + 1], bm2 = fc * bv[3 * pm + 2];
        const int aq[4] = {a0, a1_, a2_, a3};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = aq[u];
            if (q >= 0 && (q % nt) == tid) {
                const double b0 = bv[3 * u], b1 = bv[3 * u + 1], b2 = bv[3 * u + 2];
                double *r0 = row + 3 * q, *r1 = row + n + 3 * q, *r2 = row + 2 * n + 3 * q;
                r0[0] += bm0 * b0; r0[1] += bm0 * b1; r0[2] += bm0 * b2;
                r1[0] += bm1 * b0; r1[1] += bm1 * b1; r1[2] += bm1 * b2;
                r2[0] += bm2 * b0; r2[1] += bm2 * b1; r2[2] += bm2 * b2;
            }
        }
    }
    // ---- D3 pair blocks of the non-bonded (1.3 x) partners: H(m, j) -= blk, H(m, m) += blk
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // pairs excluded by the 1.3 x table (fischerd3old.py:336) or by the 1.1 x table (fischerd3.py:196); none at all for MOPT_MODEL_FISCHER
    const unsigned char *c13 = ((kind == MOPT_MODEL_FISCHERD3) ? w.c11 : w.c13) + ((size_t)b * N + m) * N;
    const double *c6s = w.c6s + (size_t)b * N;
    const V3 xm = ld3(x, m);
    for (int j = tid; j < ((kind == MOPT_MODEL_FISCHER || kind == MOPT_MODEL_SCHLEGEL) ? 0 : N); j += nt) {
        if (kind == MOPT_MODEL_MORSE) {                      // morse.py:70-100 (all pairs; De = 0.10, a = 0.20: the dispatcher's defaults)
            if (j == m) continue;
            const V3 rv = sub(xm, ld3(x, j));
            const double r = norm(rv);
            if (!(r > 1.0e-12)) continue;
            const double *gnb = par + 5 * (size_t)N;
            const double xx = 0.20 * (r - (gnb[m] + gnb[j]));
            const double k = 0.10 * (0.20 * 0.20) * (-2.0 * exp(-xx) + 4.0 * exp(-2.0 * xx));
            const double u[3] = {rv.x / r, rv.y / r, rv.z / r};
            double *rr[3] = {row + 3 * j, row + n + 3 * j, row + 2 * n + 3 * j};
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) { const double blk = k * (u[p] * u[q]); rr[p][q] -= blk; acc[3 * p + q] += blk; }
            continue;
        }
        if (j == m || c13[j]) continue;
        const V3 rv = sub(xm, ld3(x, j));
        const double r = norm(rv);
        if ((kind == MOPT_MODEL_FISCHERD3) ? !(r > 0.1) : (r < 0.1)) continue;
        const double c6ij = sqrt((c6[m] * c6s[m]) * (c6[j] * c6s[j]));
        const double c8ij = 3.0 * c6ij * sqrt(r4[m] * r4[j]);
        const double r0 = vdw[m] + vdw[j];
        const double r2_ = r * r, r5 = r2_ * r2_ * r, r6 = r5 * r, r7 = r6 * r, r8 = r7 * r, r9 = r8 * r, r10 = r9 * r;
        const double q6 = pow(a1 * r0 + a2, 6.0), q8 = pow(a1 * r0 + a2 + 2.0, 8.0);
        const double den6 = r6 + q6, den8 = r8 + q8;
        const double f6 = r6 / den6, f8 = r8 / den8;
        const double df6 = 6.0 * r5 / den6 - 6.0 * (r6 * r6) / (den6 * den6);
        const double df8 = 8.0 * r7 / den8 - 8.0 * (r8 * r8) / (den8 * den8);
        const double g6 = -s6 * c6ij * ((-6.0 / r7) * f6 + (1.0 / r6) * df6);
        const double g8 = -s8 * c8ij * ((-8.0 / r9) * f8 + (1.0 / r8) * df8);
        const double hpar = s6 * c6ij / r8 * (42.0 * f6 - r * df6) + s8 * c8ij / r10 * (72.0 * f8 - r * df8);
        const double hperp = (g6 + g8) / r;
        const double u[3] = {rv.x / r, rv.y / r, rv.z / r};
        double *rr[3] = {row + 3 * j, row + n + 3 * j, row + 2 * n + 3 * j};
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double P = u[p] * u[q];
                const double blk = hpar * P + hperp * ((p == q ? 1.0 : 0.0) - P);
                rr[p][q] -= blk;
                acc[3 * p + q] += blk;
            }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) dsum[tid][q] = acc[q];
    __syncthreads();
    if (tid < 9) { double s = 0.0; for (int t = 0; t < nt; ++t) s += dsum[t][tid]; row[(tid / 3) * n + 3 * m + (tid % 3)] += s; }
    __syncthreads();
    double *H = H_ + (size_t)b * n * n + (size_t)3 * m * n;
    for (int i = tid; i < 3 * n; i += nt) H[i] = row[i];
}

}  // namespace mopt

// kind: MOPT_MODEL_*.  d_X: [batch][N][3] bohr; d_par: cov[N], c6[N], vdw[N], r4r2[N], refcn[N], model-specific[N] (Schlegel's element code / GNB radius); d3: s6, s8, a1, a2.  Result: d_A = the projected model Hessian
// (d_H holds the unprojected, unsymmetrised assembly).  d_Q, d_W: 6 n doubles per image.  Returns MOPT_E_ARG when an image has more
// internal coordinates than the tables hold.
int mopt_launch_model_hessian(mopt_ctx *ctx, int kind, int N, int batch, const double *d_X, const double *d_par, const double *d3, double *d_H, double *d_Q,
                              double *d_W, double *d_A, bool project) {
    using namespace mopt;
    const int n = 3 * N;
    MhWork w;
    w.bcap = 16 * N; w.acap = 96 * N; w.cap = w.bcap + w.acap + 192 * N;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t bytes = 2 * al((size_t)batch * N * N) + al((size_t)batch * N * 4) + al((size_t)batch * N * 8) + al((size_t)batch * 16) + al((size_t)batch * w.cap * 16) +
                         al((size_t)batch * w.cap * 8) + al((size_t)batch * w.cap * 96) + al((size_t)batch * (w.cap + 2 * N + 8) * 4);
    char *blk = nullptr;
    MOPT_HIP_TRY(ctx, hipMalloc((void **)&blk, bytes));
    char *p = blk;
    auto take = [&](size_t x) { void *r = p; p += al(x); return r; };
    w.c11 = (unsigned char *)take((size_t)batch * N * N); w.c13 = (unsigned char *)take((size_t)batch * N * N);
    w.nb13 = (int *)take((size_t)batch * N * 4); w.c6s = (double *)take((size_t)batch * N * 8); w.cnt = (int *)take((size_t)batch * 16);
    w.atoms = (int *)take((size_t)batch * w.cap * 16); w.fc = (double *)take((size_t)batch * w.cap * 8); w.bv = (double *)take((size_t)batch * w.cap * 96);
    w.scr = (int *)take((size_t)batch * (w.cap + 2 * N + 8) * 4);
    int rc = MOPT_OK;
    {
        mopt_scope sc(ctx, "model_hessian", 4);
        hipLaunchKernelGGL(k_mh_connect, dim3(N, batch), dim3(128), 0, ctx->stream, kind, N, d_X, d_par, w);
        hipLaunchKernelGGL(k_mh_enumerate, dim3(batch), dim3(256), 0, ctx->stream, N, w);
        hipLaunchKernelGGL(k_mh_internals, dim3((w.cap + 63) / 64, batch), dim3(64), 0, ctx->stream, kind, N, d_X, d_par, w);
        if ((size_t)9 * N * 8 > 48000)
            hipFuncSetAttribute((const void *)k_mh_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit - 8192);
        hipLaunchKernelGGL(k_mh_assemble, dim3(N, batch), dim3(128), (size_t)9 * N * 8, ctx->stream, kind, N, d_X, d_par, w, d3[0], d3[1], d3[2], d3[3], d_H);
    }
    std::vector<int> cnt((size_t)4 * batch);
    hipError_t e = hipMemcpyAsync(cnt.data(), w.cnt, cnt.size() * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    hipFree(blk);
    if (e != hipSuccess) { ctx->err = std::string("model Hessian kernels: ") + hipGetErrorString(e); return MOPT_E_DEVICE; }
    for (int b = 0; b < batch; ++b) if (cnt[4 * b + 3]) { ctx->err = "model Hessian: more internal coordinates than the tables hold (16 N bonds, 96 N angles, 192 N dihedrals)"; rc = MOPT_E_ARG; }
    if (rc != MOPT_OK) return rc;
    if (project && kind != MOPT_MODEL_MORSE) return mopt_launch_project_hess(ctx, n, batch, d_H, nullptr, d_X, d_Q, d_W, d_A);
    return mopt_launch_sym_add(ctx, n, batch, d_H, nullptr, d_A);
}
