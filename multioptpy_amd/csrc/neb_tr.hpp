// TR_NEB.TR_calc decision for one image (multioptpy/Optimizer/trust_radius_neb.py:17-98), shared by the fused band step
// (band.hip) and the standalone NEB optimizers (k_nebopt.hip).  Both return the scale applied to delta (0 = frozen).
#pragma once
#include "common.hpp"

namespace mopt {

// TR_calc for one image given delta (trust_radius_neb.py:17-98).  Returns the scale applied to delta (0 = frozen).
__device__ inline double tr_scale_edge(double nrm, bool fixed) {
    if (fixed) return 0.0;
    if (nrm < 1e-15) return 0.0;
    return fmin(0.5, nrm) / nrm;
}
__device__ inline double tr_scale_interior(double dn, double r1, double r2, double c1, double c2, double fc) {
    if (fc >= 0.0) {
        if ((c1 > 0 && c2 < 0) || (c1 < 0 && c2 > 0)) {
            if (dn > r1 && c1 > 0) return r1 / dn;
            if (dn > r2 && c2 > 0) return r2 / dn;
            return 1.0;
        } else if (c1 < 0 && c2 < 0) {
            return 1.0;
        } else {
            if (dn > r1) return r1 / dn;
            if (dn > r2) return r2 / dn;
            return 1.0;
        }
    }
    return 0.0;
}


}  // namespace mopt
