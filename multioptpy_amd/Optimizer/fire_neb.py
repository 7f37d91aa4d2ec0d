"""Drop-in for multioptpy/Optimizer/fire_neb.py::FIREOptimizer (mopt_neb_fire): nebmain's default band optimizer."""
import numpy as np

from .._lib import lib, ctx, check, f64, ptr


class FIREOptimizer:
    def __init__(self, config):
        self.config = config
        self.dt = config.dt
        self.a = config.a
        self.n_reset = config.n_reset
        self.FIRE_N_accelerate = config.FIRE_N_accelerate
        self.FIRE_f_inc = config.FIRE_f_inc
        self.FIRE_f_accelerate = config.FIRE_f_accelerate
        self.FIRE_f_decelerate = config.FIRE_f_decelerate
        self.FIRE_a_start = config.FIRE_a_start
        self.FIRE_dt_max = config.FIRE_dt_max
        if getattr(config, "apply_convergence_criteria", False):
            raise NotImplementedError("apply_convergence_criteria: use TR_NEB.TR_calc directly (host-side check_convergence)")

    def optimize(self, geometry_num_list, total_force_list, pre_total_velocity, optimize_num, total_velocity, cos_list,
                 biased_energy_list, pre_biased_energy_list, pre_geom):
        X = f64(geometry_num_list)
        F = f64(total_force_list)
        V = f64(total_velocity)
        I, N = X.shape[0], X.shape[1]
        P = f64(pre_total_velocity) if len(pre_total_velocity) > 1 else None
        params = np.array([self.FIRE_N_accelerate, self.FIRE_f_inc, self.FIRE_f_decelerate, self.FIRE_a_start, self.FIRE_dt_max],
                          dtype=np.float64)
        state = np.array([self.dt, self.a, self.n_reset], dtype=np.float64)
        out = np.empty_like(X)
        dot = np.zeros(1)
        c = ctx()
        check(lib().mopt_neb_fire(c, I, N, ptr(X), ptr(F), ptr(V), ptr(P), int(optimize_num), int(bool(self.config.fix_init_edge)),
                                  int(bool(self.config.fix_end_edge)), ptr(params), ptr(state), float(self.config.bohr2angstroms),
                                  ptr(out), ptr(dot)), c)
        self.dt, self.a, self.n_reset = float(state[0]), float(state[1]), int(state[2])
        self.np_dot_param = float(dot[0])
        return out
