// Angle and dihedral-angle restraints on the device: energy, gradient and Hessian of
//   StructKeepAnglePotential.calc_energy          (multioptpy/Potential/keep_angle_potential.py:58-224)
//   StructKeepDihedralAnglePotential.calc_energy  (multioptpy/Potential/keep_dihedral_angle_potential.py:60-154)
// The reference writes these energies in torch (clamps, Taylor branches near 0 / 180 degrees selected with torch.where, a smoothstep that
// switches the dihedral term off near collinear bonds) and obtains gradient and Hessian by automatic differentiation on the CPU every step
// (Potential/potential.py:127-152).  Here the energy is evaluated in HYPER-DUAL arithmetic (f + a e1 + b e2 + c e1 e2, e1^2 = e2^2 = 0): a
// thread seeds e1 on coordinate i and e2 on coordinate j of the 9 (12) coordinates of a term and reads dE/dx_i, dE/dx_j and d2E/dx_i dx_j
// off the result -- exact derivatives of exactly the expression the reference differentiates, branch by branch, with no formula for the
// second derivatives of an angle written out by hand.  45 (78) threads per term; the terms of a call are added one after the other
// (deterministic sum; they may share atoms).
#include <vector>

#include "pipeline.h"

namespace mopt {

struct HD { double v, a, b, c; };
__device__ __forceinline__ HD hd_const(double x) { return {x, 0.0, 0.0, 0.0}; }
__device__ __forceinline__ HD operator+(HD x, HD y) { return {x.v + y.v, x.a + y.a, x.b + y.b, x.c + y.c}; }
__device__ __forceinline__ HD operator-(HD x, HD y) { return {x.v - y.v, x.a - y.a, x.b - y.b, x.c - y.c}; }
__device__ __forceinline__ HD operator*(HD x, HD y) { return {x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.c * y.v + x.a * y.b + x.b * y.a + x.v * y.c}; }
__device__ __forceinline__ HD operator*(double s, HD x) { return {s * x.v, s * x.a, s * x.b, s * x.c}; }
__device__ __forceinline__ HD operator+(HD x, double s) { return {x.v + s, x.a, x.b, x.c}; }
__device__ __forceinline__ HD operator-(HD x, double s) { return {x.v - s, x.a, x.b, x.c}; }
__device__ __forceinline__ HD operator-(double s, HD x) { return {s - x.v, -x.a, -x.b, -x.c}; }
// g(x) with g' = d1, g'' = d2 at x.v
__device__ __forceinline__ HD hd_chain(HD x, double g, double d1, double d2) { return {g, d1 * x.a, d1 * x.b, d1 * x.c + d2 * x.a * x.b}; }
__device__ __forceinline__ HD operator/(HD x, HD y) { const double r = 1.0 / y.v; return x * hd_chain(y, r, -r * r, 2.0 * r * r * r); }
__device__ __forceinline__ HD hd_sqrt(HD x) { const double s = sqrt(x.v); return hd_chain(x, s, 0.5 / s, -0.25 / (s * x.v)); }
__device__ __forceinline__ HD hd_acos(HD x) { const double w = 1.0 - x.v * x.v, sw = sqrt(w); return hd_chain(x, acos(x.v), -1.0 / sw, -x.v / (w * sw)); }
__device__ __forceinline__ HD hd_atan2(HD y, HD x) {
    const double r2 = x.v * x.v + y.v * y.v, ty = x.v / r2, tx = -y.v / r2;
    const double tyy = -2.0 * x.v * y.v / (r2 * r2), txx = -tyy, txy = (y.v * y.v - x.v * x.v) / (r2 * r2);
    return {atan2(y.v, x.v), ty * y.a + tx * x.a, ty * y.b + tx * x.b,
            ty * y.c + tx * x.c + tyy * y.a * y.b + txx * x.a * x.b + txy * (y.a * x.b + x.a * y.b)};
}
// torch.clamp: the derivative passes where lo <= x <= hi, the result is the constant bound elsewhere
__device__ __forceinline__ HD hd_clamp(HD x, double lo, double hi) { return x.v < lo ? hd_const(lo) : (x.v > hi ? hd_const(hi) : x); }
__device__ __forceinline__ HD hd_clamp_min(HD x, double lo) { return x.v < lo ? hd_const(lo) : x; }

struct HD3 { HD x, y, z; };
__device__ __forceinline__ HD3 operator-(HD3 p, HD3 q) { return {p.x - q.x, p.y - q.y, p.z - q.z}; }
__device__ __forceinline__ HD dot(HD3 p, HD3 q) { return p.x * q.x + p.y * q.y + p.z * q.z; }
__device__ __forceinline__ HD3 cross(HD3 p, HD3 q) { return {p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x}; }
__device__ __forceinline__ HD3 scale(HD3 p, HD s) { return {p.x * s, p.y * s, p.z * s}; }

// the five Horner coefficients of (arccos(1 - d))^2 = d (2 + d/3 + 8 d^2/45 + 4 d^3/35 + 128 d^4/1575) (keep_angle_potential.py:50-56)
__device__ __forceinline__ HD theta_sq_taylor(HD delta) {
    HD t = hd_const(128.0 / 1575.0);
    t = delta * t + 4.0 / 35.0;
    t = delta * t + 8.0 / 45.0;
    t = delta * t + 1.0 / 3.0;
    t = delta * t + 2.0;
    return delta * t;
}

// keep_angle_potential.py:58-224; p: the three atoms, theta0 in radians
__device__ HD angle_energy(const HD3 *p, double k, double theta0) {
    constexpr double PI = 3.14159265358979323846;
    const HD3 v1 = p[0] - p[1], v2 = p[2] - p[1];
    const HD n12 = hd_clamp_min(hd_sqrt(dot(v1, v1)) * hd_sqrt(dot(v2, v2)), 1e-12);
    const HD u = hd_clamp(dot(v1, v2) / n12, -1.0, 1.0);
    const double ucp = cos(1e-3), ucn = cos(PI - 1e-3);
    const HD tsq0 = theta_sq_taylor(1.0 - u), tsqpi = theta_sq_taylor(u + 1.0);
    auto sq = [](HD x) { return x * x; };
    if (fabs(theta0) < 1e-8) {                               // branch A: equilibrium at 0 degrees
        if (u.v > ucp) return (0.5 * k) * tsq0;
        if (u.v < ucn) return (0.5 * k) * sq(PI - hd_sqrt(hd_clamp_min(tsqpi, 1e-30)));
        return (0.5 * k) * sq(hd_acos(hd_clamp(u, ucn, ucp)));
    }
    if (fabs(theta0 - PI) < 1e-8) {                          // branch B: equilibrium at 180 degrees
        if (u.v < ucn) return (0.5 * k) * tsqpi;
        if (u.v > ucp) return (0.5 * k) * sq(hd_sqrt(hd_clamp_min(tsq0, 1e-30)) - PI);
        return (0.5 * k) * sq(hd_acos(hd_clamp(u, ucn, ucp)) - PI);
    }
    if (u.v > ucp) return (0.5 * k) * sq(hd_sqrt(hd_clamp_min(tsq0, 1e-30)) - theta0);      // branch C
    if (u.v < ucn) return (0.5 * k) * sq((PI - hd_sqrt(hd_clamp_min(tsqpi, 1e-30))) - theta0);
    return (0.5 * k) * sq(hd_acos(u) - theta0);
}

// keep_dihedral_angle_potential.py:60-154; p: the four atoms, phi0 in radians
__device__ HD dihedral_energy(const HD3 *p, double k, double phi0) {
    constexpr double PI = 3.14159265358979323846, CMIN = 1e-10, CMAX = 1e-8;
    const HD3 b1 = p[1] - p[0], b2 = p[2] - p[1], b3 = p[3] - p[2];
    const HD3 n1 = cross(b1, b2), n2 = cross(b2, b3);
    const HD n1sq = dot(n1, n1), n2sq = dot(n2, n2);
    auto sw = [&](HD val) { HD t = hd_clamp((1.0 / (CMAX - CMIN)) * (val - CMIN), 0.0, 1.0); return t * t * (3.0 - 2.0 * t); };
    const HD s1 = sw(n1sq), s2 = sw(n2sq);
    const HD one = hd_const(1.0);
    const HD3 n1h = scale(n1, one / hd_clamp_min(hd_sqrt(n1sq), 1e-12)), n2h = scale(n2, one / hd_clamp_min(hd_sqrt(n2sq), 1e-12));
    const HD3 b2h = scale(b2, one / hd_clamp_min(hd_sqrt(dot(b2, b2)), 1e-12));
    const HD x = dot(n1h, n2h), y = dot(cross(n1h, n2h), b2h);
    const HD phi = hd_atan2(y, x);
    HD diff = phi - phi0;
    diff = diff - 2.0 * PI * rint(diff.v / (2.0 * PI));      // torch.round: half to even, no derivative
    return ((0.5 * k) * diff * diff) * s1 * s2;
}

// keep_outofplain_angle_potential.py:33-146; p: centre, probe, plane atom 1, plane atom 2; theta0 in radians
__device__ HD outofplane_energy(const HD3 *p, double k, double theta0) {
    const HD3 a1 = p[1] - p[0], a2 = p[2] - p[0], a3 = p[3] - p[0];
    const HD3 nv = cross(a2, a3);
    const HD nsq = dot(nv, nv);
    if (nsq.v < 1e-8) return hd_const(0.0);                  // undefined plane (collinear base vectors): energy and derivatives forced to zero
    const HD3 nh = scale(nv, hd_const(1.0) / hd_clamp_min(hd_sqrt(nsq), 1e-12));
    const HD h = dot(a1, nh);
    const HD a1n = hd_sqrt(dot(a1, a1));
    const HD rproj = hd_sqrt(hd_clamp_min(a1n * a1n - h * h, 0.0));
    const HD diff = hd_atan2(h, rproj) - theta0;
    return (0.5 * k) * diff * diff;
}

// StructKeepPotentialv2.calc_energy (keep_potential.py:79-115): 1/2 k (|c1 - c2| - r0)^2 between the centroids of two atom lists.  One workgroup;
// variables = the coordinates of the n1 + n2 list entries (an atom may occur more than once: its contributions add up, hence the atomics).
__global__ void __launch_bounds__(256) k_bias_fragment_distance(int N, const double *__restrict__ x, int n1, int n2, const int *__restrict__ atoms, double k,
                                                                double r0, double *__restrict__ E, double *__restrict__ g, double *__restrict__ H) {
    const int nv = 3 * (n1 + n2), np_ = nv * (nv + 1) / 2, n = 3 * N;
    for (int t = threadIdx.x; t < np_; t += blockDim.x) {
        int pi = 0, r = t;
        while (r >= nv - pi) { r -= nv - pi; ++pi; }
        const int pj = pi + r;
        HD3 c[2] = {{hd_const(0.0), hd_const(0.0), hd_const(0.0)}, {hd_const(0.0), hd_const(0.0), hd_const(0.0)}};
        for (int a = 0; a < n1 + n2; ++a) {
            const int at = atoms[a], f = a < n1 ? 0 : 1;
            HD q[3];
            for (int d = 0; d < 3; ++d) { const int var = 3 * a + d; q[d] = {x[3 * at + d], var == pi ? 1.0 : 0.0, var == pj ? 1.0 : 0.0, 0.0}; }
            c[f] = {c[f].x + q[0], c[f].y + q[1], c[f].z + q[2]};
        }
        const HD3 c1 = scale(c[0], hd_const(1.0 / n1)), c2 = scale(c[1], hd_const(1.0 / n2));
        const HD3 v = c1 - c2;
        const HD dist = hd_clamp_min(hd_sqrt(dot(v, v)), 1e-12);
        const HD dd = dist - r0;
        const HD e = (0.5 * k) * dd * dd;
        const int row = 3 * atoms[pi / 3] + pi % 3, col = 3 * atoms[pj / 3] + pj % 3;
        if (t == 0) E[0] = e.v;
        if (pi == pj) atomicAdd(&g[row], e.a);
        atomicAdd(&H[(size_t)row * n + col], e.c);
        if (pi != pj) atomicAdd(&H[(size_t)col * n + row], e.c);
    }
}

// The well potentials of multioptpy/Potential/switching_potential.py (round 4): ONE radial well w(r) -- linear outside [a, d], a quintic switch on
// [a, b] and [c, d], zero on (b, c) -- applied to
//   kind 0  WellPotential        (:5-67)     r = |centroid(list 1) - centroid(list 2)|
//   kind 1  WellPotentialWall    (:70-135)   r = |x_i[dir]| of every atom i of list 1, summed
//   kind 2  WellPotentialVP      (:138-196)  r = |x_i - p| of every atom i of list 1 (p: a fixed point), summed
//   kind 3  WellPotentialAround  (:199-257)  r = |x_i - centroid(list 1)| of every atom i of list 2, summed
// prm: a, b, c, d (bohr), wall energy (hartree), p (3), dir.  The branch of w is chosen on the value of r, its expression differentiated -- what
// torch does with the reference's if / elif chain.
__device__ __forceinline__ HD well_fn(HD r, const double *prm) {
    const double a = prm[0], b = prm[1], c = prm[2], d = prm[3], e0 = prm[4];
    const double ss = 0.5 / (b - a), si = 1.0 - 0.5 * b / (b - a), ls = 0.5 / (c - d), li = 1.0 - 0.5 * c / (c - d);
    const HD xs = ss * r + si, xl = ls * r + li;
    auto quintic = [&](HD x) { const HD x2 = x * x, x3 = x2 * x; return e0 * (2.0 - 20.0 * x3 + 30.0 * (x3 * x) - 12.0 * (x3 * x2)); };
    if (r.v <= a) return e0 * (2.875 - 3.75 * xs);
    if (r.v <= b) return quintic(xs);
    if (r.v < c) return hd_const(0.0);
    if (r.v < d) return quintic(xl);
    return e0 * (2.875 - 3.75 * xl);
}

// One workgroup; variables = the coordinates of the n1 + n2 list entries (as in k_bias_fragment_distance: repeats add up through the atomics).
__global__ void __launch_bounds__(256) k_bias_well(int kind, int N, const double *__restrict__ x, int n1, int n2, const int *__restrict__ atoms,
                                                   const double *__restrict__ prm, double *__restrict__ E, double *__restrict__ g, double *__restrict__ H) {
    const int nv = 3 * (n1 + n2), np_ = nv * (nv + 1) / 2, n = 3 * N;
    const int dir = (int)prm[8];
    for (int t = threadIdx.x; t < np_; t += blockDim.x) {
        int pi = 0, r = t;
        while (r >= nv - pi) { r -= nv - pi; ++pi; }
        const int pj = pi + r;
        auto coord = [&](int a) -> HD3 {
            const int at = atoms[a];
            HD q[3];
            for (int dd = 0; dd < 3; ++dd) { const int var = 3 * a + dd; q[dd] = {x[3 * at + dd], var == pi ? 1.0 : 0.0, var == pj ? 1.0 : 0.0, 0.0}; }
            return {q[0], q[1], q[2]};
        };
        auto centroid = [&](int a0, int cnt) -> HD3 {
            HD3 c = {hd_const(0.0), hd_const(0.0), hd_const(0.0)};
            for (int a = a0; a < a0 + cnt; ++a) { const HD3 q = coord(a); c = {c.x + q.x, c.y + q.y, c.z + q.z}; }
            return scale(c, hd_const(1.0 / cnt));
        };
        // (torch.linalg.norm at the zero vector: value 0 with zero derivatives -- a target atom that is its own centre)
        auto nrm = [&](HD3 v) -> HD { const HD s2 = dot(v, v); return s2.v > 0.0 ? hd_sqrt(s2) : hd_const(0.0); };
        HD e = hd_const(0.0);
        // The VALUE of the summed kinds follows the reference's accumulator (switching_potential.py:117-130: `energy = 0.0; energy += term`, the term
        // of the zero region being a float32 tensor): when the FIRST target lies in the zero region the accumulator is a float32 tensor and every
        // later in-place add is computed in double and stored in single precision.  Derivatives are not affected (the cast passes them on).
        double ev = 0.0;
        bool first = true, f32 = false;
        auto term = [&](HD r) {
            const HD w = well_fn(r, prm);
            e = e + w;
            if (first) { first = false; f32 = (r.v > prm[1] && r.v < prm[2]); ev = f32 ? (double)(float)w.v : w.v; }
            else ev = f32 ? (double)(float)(ev + w.v) : ev + w.v;
        };
        if (kind == 0) {
            const HD3 v = centroid(0, n1) - centroid(n1, n2);
            e = well_fn(nrm(v), prm);
            ev = e.v;
        } else if (kind == 1) {
            for (int a = 0; a < n1; ++a) {
                const HD3 q = coord(a);
                const HD comp = dir == 0 ? q.x : (dir == 1 ? q.y : q.z);
                term(comp.v < 0.0 ? (0.0 - comp) : comp);               // |x|: sign(x) passes, no curvature
            }
        } else if (kind == 2) {
            const HD3 pt = {hd_const(prm[5]), hd_const(prm[6]), hd_const(prm[7])};
            for (int a = 0; a < n1; ++a) term(nrm(coord(a) - pt));
        } else {
            const HD3 cen = centroid(0, n1);
            for (int a = n1; a < n1 + n2; ++a) term(nrm(coord(a) - cen));
        }
        const int row = 3 * atoms[pi / 3] + pi % 3, col = 3 * atoms[pj / 3] + pj % 3;
        if (t == 0) E[0] = ev;
        if (pi == pj) atomicAdd(&g[row], e.a);
        atomicAdd(&H[(size_t)row * n + col], e.c);
        if (pi != pj) atomicAdd(&H[(size_t)col * n + row], e.c);
    }
}

__global__ void __launch_bounds__(256) k_bias_zero(size_t total, double *__restrict__ H, int n, double *__restrict__ g, double *__restrict__ E) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) H[i] = 0.0;
    if (i < (size_t)n) g[i] = 0.0;
    if (i == 0) E[0] = 0.0;
}

// one workgroup of 128 threads; NA atoms per term (3: angle, 4: dihedral or out-of-plane); atoms: [nterms][NA] 0-based; kp: [nterms][2] = (k, angle in radians)
template <int NA, int KIND>
__global__ void __launch_bounds__(128) k_bias_angle_terms(int N, const double *__restrict__ x, int nterms, const int *__restrict__ atoms,
                                                          const double *__restrict__ kp, double *__restrict__ E, double *__restrict__ g, double *__restrict__ H) {
    constexpr int NV = 3 * NA, NP = NV * (NV + 1) / 2;
    const int t = threadIdx.x, n = 3 * N;
    // pair (i <= j) of this thread
    int pi = 0, pj = 0;
    if (t < NP) { int r = t; while (r >= NV - pi) { r -= NV - pi; ++pi; } pj = pi + r; }
    for (int term = 0; term < nterms; ++term) {
        if (t < NP) {
            HD3 p[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int at = atoms[NA * term + a];
                HD c[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int var = 3 * a + d;
                    c[d] = {x[3 * at + d], var == pi ? 1.0 : 0.0, var == pj ? 1.0 : 0.0, 0.0};
                }
                p[a] = {c[0], c[1], c[2]};
            }
            const double k = kp[2 * term], ang = kp[2 * term + 1];
            const HD e = (KIND == 0) ? angle_energy(p, k, ang) : (KIND == 1) ? dihedral_energy(p, k, ang) : outofplane_energy(p, k, ang);
            const int row = 3 * atoms[NA * term + pi / 3] + pi % 3, col = 3 * atoms[NA * term + pj / 3] + pj % 3;
            if (t == 0) E[0] += e.v;
            if (pi == pj) g[row] += e.a;
            H[(size_t)row * n + col] += e.c;
            if (pi != pj) H[(size_t)col * n + row] += e.c;
        }
        __syncthreads();
    }
}

}  // namespace mopt

// kind 0: angle restraints (3 atoms per term), kind 1: dihedral-angle restraints, kind 2: out-of-plane-angle restraints (4 atoms per term: centre,
// probe, two plane atoms).  X: N x 3 (bohr); atoms: nterms x (3 | 4)
// 0-based, all different within a term; kp: nterms x 2 = (spring constant in hartree / rad^2, equilibrium angle in RADIANS).
// Out: E (1), grad (N x 3), H (3N x 3N).  Host pointers.
extern "C" int mopt_keep_angle(mopt_ctx *ctx, int kind, int N, const double *X, int nterms, const int *atoms, const double *kp, double *E, double *grad,
                               double *H) {
    using namespace mopt;
    if (!ctx) return MOPT_E_ARG;
    const int na = kind == 0 ? 3 : 4;
    if (kind < 0 || kind > 2 || N < 1 || 3 * N > MOPT_MAX_N || nterms < 0 || !X || (nterms > 0 && (!atoms || !kp)) || !E || !grad || !H) {
        ctx->err = "mopt_keep_angle: bad argument";
        return MOPT_E_ARG;
    }
    for (int p = 0; p < nterms; ++p)
        for (int a = 0; a < na; ++a) {
            const int at = atoms[na * p + a];
            bool dup = false;
            for (int b = 0; b < a; ++b) dup = dup || atoms[na * p + b] == at;
            if (at < 0 || at >= N || dup) { ctx->err = "mopt_keep_angle: atom index out of range or repeated within a term (0-based indices)"; return MOPT_E_ARG; }
        }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t nn = (size_t)n * n;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_g = al(nn * 8), o_E = o_g + al((size_t)n * 8), o_x = o_E + 256, o_a = o_x + al((size_t)n * 8), o_k = o_a + al((size_t)(nterms + 1) * 16);
    char *d = nullptr;
    MOPT_HIP_TRY(ctx, hipMalloc((void **)&d, o_k + al((size_t)(nterms + 1) * 16)));
    struct Free { char *p; ~Free() { if (p) hipFree(p); } } guard{d};
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_x, X, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nterms > 0) {
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_a, atoms, (size_t)nterms * na * 4, hipMemcpyHostToDevice, ctx->stream));
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_k, kp, (size_t)nterms * 16, hipMemcpyHostToDevice, ctx->stream));
    }
    {
        mopt_scope sc(ctx, "bias_keep", 2);
        hipLaunchKernelGGL(k_bias_zero, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, nn, (double *)d, n, (double *)(d + o_g), (double *)(d + o_E));
        if (nterms > 0) {
            if (kind == 0)
                hipLaunchKernelGGL((k_bias_angle_terms<3, 0>), dim3(1), dim3(128), 0, ctx->stream, N, (const double *)(d + o_x), nterms, (const int *)(d + o_a),
                                   (const double *)(d + o_k), (double *)(d + o_E), (double *)(d + o_g), (double *)d);
            else if (kind == 1)
                hipLaunchKernelGGL((k_bias_angle_terms<4, 1>), dim3(1), dim3(128), 0, ctx->stream, N, (const double *)(d + o_x), nterms, (const int *)(d + o_a),
                                   (const double *)(d + o_k), (double *)(d + o_E), (double *)(d + o_g), (double *)d);
            else
                hipLaunchKernelGGL((k_bias_angle_terms<4, 2>), dim3(1), dim3(128), 0, ctx->stream, N, (const double *)(d + o_x), nterms, (const int *)(d + o_a),
                                   (const double *)(d + o_k), (double *)(d + o_E), (double *)(d + o_g), (double *)d);
        }
    }
    MOPT_HIP_TRY(ctx, hipGetLastError());
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(E, d + o_E, 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(grad, d + o_g, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(H, d, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

// StructKeepPotentialv2 (fragment-centre distance restraint): frag1 / frag2 0-based atom lists (n1 + n2 <= 64; repeats allowed), k in
// hartree / bohr^2, r0 in bohr.  Out: E (1), grad (N x 3), H (3N x 3N).  Host pointers.
extern "C" int mopt_keep_pot_v2(mopt_ctx *ctx, int N, const double *X, int n1, const int *frag1, int n2, const int *frag2, double k, double r0, double *E,
                                double *grad, double *H) {
    using namespace mopt;
    if (!ctx) return MOPT_E_ARG;
    if (N < 1 || 3 * N > MOPT_MAX_N || n1 < 1 || n2 < 1 || n1 + n2 > 64 || !X || !frag1 || !frag2 || !E || !grad || !H) { ctx->err = "mopt_keep_pot_v2: bad argument"; return MOPT_E_ARG; }
    std::vector<int> at;
    for (int a = 0; a < n1; ++a) at.push_back(frag1[a]);
    for (int a = 0; a < n2; ++a) at.push_back(frag2[a]);
    for (int v : at) if (v < 0 || v >= N) { ctx->err = "mopt_keep_pot_v2: atom index out of range (0-based)"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t nn = (size_t)n * n;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_g = al(nn * 8), o_E = o_g + al((size_t)n * 8), o_x = o_E + 256, o_a = o_x + al((size_t)n * 8);
    char *d = nullptr;
    MOPT_HIP_TRY(ctx, hipMalloc((void **)&d, o_a + 512));
    struct Free { char *p; ~Free() { if (p) hipFree(p); } } guard{d};
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_x, X, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_a, at.data(), at.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        mopt_scope sc(ctx, "bias_keep", 2);
        hipLaunchKernelGGL(k_bias_zero, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, nn, (double *)d, n, (double *)(d + o_g), (double *)(d + o_E));
        hipLaunchKernelGGL(k_bias_fragment_distance, dim3(1), dim3(256), 0, ctx->stream, N, (const double *)(d + o_x), n1, n2, (const int *)(d + o_a), k, r0,
                           (double *)(d + o_E), (double *)(d + o_g), (double *)d);
    }
    MOPT_HIP_TRY(ctx, hipGetLastError());
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(E, d + o_E, 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(grad, d + o_g, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(H, d, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

// The well potentials of Potential/switching_potential.py: kind 0 WellPotential (list1 / list2: the two fragments), 1 WellPotentialWall (list1: the target atoms;
// prm[8]: 0 / 1 / 2 = x / y / z), 2 WellPotentialVP (list1: the target atoms; prm[5..7]: the point, bohr), 3 WellPotentialAround (list1: the centre atoms, list2:
// the target atoms).  prm[0..3]: the four limit distances in bohr, prm[4]: the wall energy in hartree.  0-based atom lists, n1 + n2 <= 64 entries (repeats allowed).
// Out: E (1), grad (N x 3), H (3N x 3N).  Host pointers.
extern "C" int mopt_well_pot(mopt_ctx *ctx, int kind, int N, const double *X, int n1, const int *list1, int n2, const int *list2, const double *prm, double *E,
                             double *grad, double *H) {
    using namespace mopt;
    if (!ctx) return MOPT_E_ARG;
    const bool two = (kind == 0 || kind == 3);
    if (kind < 0 || kind > 3 || N < 1 || 3 * N > MOPT_MAX_N || n1 < 1 || (two ? n2 < 1 : n2 != 0) || n1 + n2 > 64 || !X || !list1 || (two && !list2) || !prm || !E || !grad || !H) {
        ctx->err = "mopt_well_pot: bad argument"; return MOPT_E_ARG;
    }
    if (!(prm[1] != prm[0]) || !(prm[2] != prm[3])) { ctx->err = "mopt_well_pot: limit distances a = b or c = d"; return MOPT_E_ARG; }
    std::vector<int> at;
    for (int a = 0; a < n1; ++a) at.push_back(list1[a]);
    for (int a = 0; a < n2; ++a) at.push_back(list2[a]);
    for (int v : at) if (v < 0 || v >= N) { ctx->err = "mopt_well_pot: atom index out of range (0-based)"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const int n = 3 * N;
    const size_t nn = (size_t)n * n;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_g = al(nn * 8), o_E = o_g + al((size_t)n * 8), o_x = o_E + 256, o_a = o_x + al((size_t)n * 8), o_p = o_a + 512;
    char *d = nullptr;
    MOPT_HIP_TRY(ctx, hipMalloc((void **)&d, o_p + 256));
    struct Free { char *p; ~Free() { if (p) hipFree(p); } } guard{d};
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_x, X, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_a, at.data(), at.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(d + o_p, prm, 9 * 8, hipMemcpyHostToDevice, ctx->stream));
    {
        mopt_scope sc(ctx, "bias_keep", 2);
        hipLaunchKernelGGL(k_bias_zero, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream, nn, (double *)d, n, (double *)(d + o_g), (double *)(d + o_E));
        hipLaunchKernelGGL(k_bias_well, dim3(1), dim3(256), 0, ctx->stream, kind, N, (const double *)(d + o_x), n1, n2, (const int *)(d + o_a), (const double *)(d + o_p),
                           (double *)(d + o_E), (double *)(d + o_g), (double *)d);
    }
    MOPT_HIP_TRY(ctx, hipGetLastError());
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(E, d + o_E, 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(grad, d + o_g, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipMemcpyAsync(H, d, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}
