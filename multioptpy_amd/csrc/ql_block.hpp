// Implicit QL with Wilkinson shifts and accumulated rotations (the classical tql2 / dsteqr iteration) for one symmetric tridiagonal
// block of at most 32 rows, executed by ONE wave: d and e live in registers (lane i holds d_i and e_i, scalars are fetched with
// v_readlane), the rotation chain is computed redundantly by every lane, lane r (< 32) owns row r of the eigenvector table Zs
// (LDS, column-major with stride 32: conflict-free).  The column i + 1 of a rotation is carried in a register to the next rotation
// and column i - 1 is requested one rotation ahead, so no LDS round trip sits on the dependent chain; the rotation itself comes from
// one reciprocal square root.  Shared by the leaves of the divide-and-conquer solver (k_dc.hip) and the small-block solver of the
// bisection route (k_tridiag.hip).  Zs must hold the identity on entry; returns false when an eigenvalue needed more than 60 sweeps.
#pragma once
#include "common.hpp"

namespace mopt {

constexpr double QL_EPS = 1.1102230246251565e-16;

__device__ __forceinline__ double dc_rl(double v, int lane) { return lane_bcast(v, lane); }
__device__ __forceinline__ double ql_rsqrt(double w) {      // 1/sqrt(w), w > 0 normal: hardware estimate + two Newton steps
    double y = __builtin_amdgcn_rsq(w);
    double e = fma(-(w * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(w * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}

__device__ __forceinline__ bool ql_block32(int m, double &d, double &e, double *Zs, int lane) {
    bool failed = false;
    for (int l = 0; l < m; ++l) {
        int iter = 0;
        for (;;) {
            // smallest mm >= l with a negligible e_mm (mm = m - 1 when there is none)
            const double dn = __shfl_down(d, 1, WAVE);
            const bool small = (lane >= l && lane < m - 1) ? (fabs(e) <= QL_EPS * (fabs(d) + fabs(dn))) : (lane == m - 1);
            const unsigned long long bal = __ballot(small) & ~((1ull << l) - 1ull);
            const int mm = __ffsll((long long)bal) - 1;
            if (mm == l) break;
            if (++iter > 60) { failed = true; break; }
            const double dl_ = dc_rl(d, l), el = dc_rl(e, l), dl1 = dc_rl(d, l + 1);
            double g = (dl1 - dl_) / (2.0 * el);
            double r = sqrt(fma(g, g, 1.0));
            g = dc_rl(d, mm) - dl_ + el / (g + copysign(r, g));      // (Wilkinson shift: once per QL step, IEEE division and root)
            double sn = 1.0, cs = 1.0, pp = 0.0;
            bool broke = false;
            // eigenvector columns: column i + 1 is carried in a register from one rotation to the next (the rotation of (i, i + 1)
            // leaves the new column i, which is the upper partner of the next rotation), column i - 1 is requested one rotation
            // ahead -- no LDS round trip on the chain
            const int lr = lane & 31;
            double zc = Zs[mm * 32 + lr];
            double znext = Zs[(mm - 1) * 32 + lr];
            for (int i = mm - 1; i >= l; --i) {
                const double ei = dc_rl(e, i);
                const double di1 = dc_rl(d, i + 1), di = dc_rl(d, i);
                const double z0 = znext;
                if (i > l) znext = Zs[(i - 1) * 32 + lr];
                double f = sn * ei;
                const double bq = cs * ei;
                const double x2 = fma(f, f, g * g);
                if (!(x2 >= 1e-290)) {                       // (underflow or zero: recover as tql2 does)
                    if (lane == i + 1) { e = sqrt(x2); d -= pp; }
                    if (lane == mm) e = 0.0;
                    if (lane < 32) Zs[(i + 1) * 32 + lr] = zc;
                    zc = z0;                                 // (column i is untouched; written back behind the loop)
                    broke = true;
                    if (lane < 32) Zs[i * 32 + lr] = zc;
                    break;
                }
                // the rotation from one reciprocal square root (a few ulp; the accumulated table stays orthogonal to ~m eps)
                const double rinv = ql_rsqrt(x2);
                r = x2 * rinv;
                if (lane == i + 1) e = r;
                sn = f * rinv; cs = g * rinv;
                g = di1 - pp;
                r = (di - g) * sn + 2.0 * cs * bq;
                pp = sn * r;
                if (lane == i + 1) d = g + pp;
                g = cs * r - bq;
                if (lane < 32) Zs[(i + 1) * 32 + lr] = sn * z0 + cs * zc;
                zc = cs * z0 - sn * zc;
            }
            if (!broke && lane < 32) Zs[l * 32 + lr] = zc;
            if (!broke) {
                if (lane == l) { d -= pp; e = g; }
                if (lane == mm) e = 0.0;
            }
        }
        if (failed) break;
    }
    return !failed;
}

}  // namespace mopt
