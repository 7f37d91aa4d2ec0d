"""Drop-in for multioptpy/Optimizer/gradientdescent_neb.py::SteepestDescentOptimizer (mopt_neb_sd)."""
import numpy as np

from .._lib import lib, ctx, check, f64, ptr


class SteepestDescentOptimizer:
    def __init__(self, config):
        self.config = config

    def optimize(self, geometry_num_list, total_force_list, **kwargs):
        X = f64(geometry_num_list)
        F = f64(total_force_list)
        out = np.empty_like(X)
        c = ctx()
        check(lib().mopt_neb_sd(c, X.shape[0], X.shape[1], ptr(X), ptr(F), int(bool(self.config.fix_init_edge)),
                                int(bool(self.config.fix_end_edge)), float(self.config.bohr2angstroms), ptr(out)), c)
        return out
