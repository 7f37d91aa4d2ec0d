"""EnhancedRSPRFO: drop-in for ``multioptpy.Optimizer.rsprfo.EnhancedRSPRFO`` (reference rsprfo.py:10-1362), the P-RFO
transition-state optimizer selected by ``-opt rsprfo_*`` (optimizer.py:435-436), running on MI355X.

Same constructor keywords, ``run(...)`` signature and return value ((3N,1) move vector to be SUBTRACTED), same public
attributes.  ``set_hessian`` / ``set_bias_hessian`` copy and symmetrise like the reference (rsprfo.py:1309-1328: with
optmain's ``-fc -1`` the quasi-Newton update is therefore discarded every step).  The step itself --
previous-step reduction ratio and trust radius, Hessian update and its eigenvalue sanity check, gradient projection,
eigen-solve, eigenvalue shifting, mode following, the max/min augmented-Hessian solves of the micro cycles, scaling and
the predicted energy change -- happens in libmopt_hip.so (``mopt_rsprfo_run``).
"""
import ctypes as C

import numpy as np

from .. import _abi
from .._lib import lib, ctx, check, f64, ptr


class EnhancedRSPRFO:
    def __init__(self, **config):
        g = config.get
        self.alpha0 = g("alpha0", 1.0)
        self.max_micro_cycles = g("max_micro_cycles", 50)
        self.saddle_order = g("saddle_order", 1)
        self.hessian_update_method = g("method", "auto")
        self.display_flag = g("display_flag", True)
        self.debug = g("debug", False)
        self.alpha_max = g("alpha_max", 1e8)
        self.alpha_min = g("alpha_min", 1e-8)
        self.alpha_step_max = g("alpha_step_max", 100.0)
        self.micro_cycle_rtol = g("micro_cycle_rtol", 1e-3)
        self.micro_cycle_atol = g("micro_cycle_atol", 1e-6)
        if self.saddle_order == 0:
            self.trust_radius_initial = g("trust_radius", 0.5)
            self.trust_radius_max = g("trust_radius_max", 0.5)
        else:
            self.trust_radius_initial = g("trust_radius", 0.1)
            self.trust_radius_max = g("trust_radius_max", 0.3)
        self.trust_radius = self.trust_radius_initial
        self.trust_radius_min = g("trust_radius_min", 0.01)
        self.eta_1 = g("eta_1", 0.1)
        self.eta_2 = g("eta_2", 0.25)
        self.eta_3 = g("eta_3", 0.75)
        self.gamma_1 = g("gamma_1", 0.25)
        self.gamma_2 = g("gamma_2", 2.0)
        self.step_rejection_enabled = g("step_rejection", True)
        self.rejection_threshold = g("rejection_threshold", -0.5)
        self.max_consecutive_rejections = g("max_consecutive_rejections", 3)
        self.consecutive_rejections = 0
        self.hessian_shift_enabled = g("hessian_shift_enabled", True)
        self.min_positive_eigval = g("min_positive_eigval", 0.001)
        self.min_negative_eigval = g("min_negative_eigval", -0.001)
        self.gradient_scaling_enabled = g("gradient_scaling_enabled", True)
        self.gradient_scaling_threshold = g("gradient_scaling_threshold", 0.001)
        self.min_step_scale = g("min_step_scale", 0.1)
        self.adaptive_trust_enabled = g("adaptive_trust_enabled", True)
        self.gradient_trust_coupling = g("gradient_trust_coupling", 0.5)
        self.adapt_trust_radius = g("adapt_trust_radius", True)
        self.config = config
        self.Initialization = True
        self.iter = 0
        self.hessian = None
        self.bias_hessian = None
        self.shifted_hessian = None
        self.mode_following_enabled = g("mode_following", True)
        self.eigvec_following = g("eigvec_following", True)
        self.overlap_threshold = g("overlap_threshold", 0.5)
        self.mixing_threshold = g("mixing_threshold", 0.3)
        self.roots = list(range(self.saddle_order))
        self.DELTA = None
        self.linesearchflag = False
        self.device = g("device", None)
        self.last_info = {}
        self._reset_state()

    def _reset_state(self):
        """rsprfo.py:888-906."""
        self.prev_eigvec_max = None
        self.prev_eigvec_min = None
        self.predicted_energy_changes = []
        self.actual_energy_changes = []
        self.reduction_ratios = []
        self.trust_radius_history = []
        self.step_quality_history = []
        self.prev_geometry = None
        self.prev_gradient = None
        self.prev_energy = None
        self.prev_move_vector = None
        self.eigvec_history = []
        self.eigval_history = []
        self.ts_mode_idx = None
        self.ts_mode_eigvec = None
        self.consecutive_rejections = 0
        self.trust_radius = self.trust_radius_initial

    # -- plugin surface (rsprfo.py:1309-1362) ---------------------------------------------
    def set_hessian(self, hessian):
        self.hessian = np.asarray(hessian).copy()
        self.hessian = 0.5 * (self.hessian + self.hessian.T)

    def set_bias_hessian(self, bias_hessian):
        self.bias_hessian = np.asarray(bias_hessian).copy()
        self.bias_hessian = 0.5 * (self.bias_hessian + self.bias_hessian.T)

    def get_hessian(self):
        return self.hessian

    def get_bias_hessian(self):
        return self.bias_hessian

    def get_shifted_hessian(self):
        return self.shifted_hessian

    def reset_trust_radius(self):
        self.trust_radius = self.trust_radius_initial

    def _params(self):
        p = _abi.RsprfoParams()
        for name in ("alpha0", "alpha_max", "alpha_min", "alpha_step_max", "micro_cycle_rtol", "micro_cycle_atol",
                     "trust_radius_min", "trust_radius_max", "eta_2", "eta_3", "gamma_1", "gamma_2", "rejection_threshold",
                     "min_positive_eigval", "min_negative_eigval", "gradient_scaling_threshold", "min_step_scale",
                     "gradient_trust_coupling", "overlap_threshold", "mixing_threshold"):
            setattr(p, name, float(getattr(self, name)))
        p.max_micro_cycles = int(self.max_micro_cycles)
        p.saddle_order = int(self.saddle_order)
        p.rule = _abi.rule_from_method(self.hessian_update_method)
        p.step_rejection_enabled = int(bool(self.step_rejection_enabled))
        p.max_consecutive_rejections = int(self.max_consecutive_rejections)
        p.hessian_shift_enabled = int(bool(self.hessian_shift_enabled))
        p.gradient_scaling_enabled = int(bool(self.gradient_scaling_enabled))
        p.adaptive_trust_enabled = int(bool(self.adaptive_trust_enabled))
        p.adapt_trust_radius = int(bool(self.adapt_trust_radius))
        p.mode_following_enabled = int(bool(self.mode_following_enabled))
        p.eigvec_following = int(bool(self.eigvec_following))
        return p

    # -- the step (rsprfo.py:713-886) -----------------------------------------------------
    def run(self, geom_num_list, B_g, pre_B_g=[], pre_geom=[], B_e=0.0, pre_B_e=0.0, pre_move_vector=[],
            initial_geom_num_list=[], g=[], pre_g=[]):
        if self.config.get("maxstep") is not None:
            raise NotImplementedError("maxstep is not supported by the MI355X EnhancedRSPRFO yet")
        process_prev = False
        if self.Initialization:
            self._reset_state()
            self.Initialization = False
        else:
            process_prev = self.prev_energy is not None and len(self.predicted_energy_changes) > 0
        if self.hessian is None:
            raise ValueError("Hessian matrix must be set before running optimization")
        x = f64(np.asarray(geom_num_list)).reshape(-1)
        n = x.size
        bg = f64(np.asarray(B_g)).reshape(-1)
        do_update = (self.prev_geometry is not None and self.prev_gradient is not None
                     and len(pre_B_g) > 0 and len(pre_geom) > 0)
        px = f64(np.asarray(pre_geom)).reshape(-1) if do_update else None
        pg = f64(np.asarray(pre_B_g)).reshape(-1) if do_update else None
        s = _abi.RsprfoState()
        s.trust_radius = float(self.trust_radius)
        s.prev_energy = float(self.prev_energy) if self.prev_energy is not None else 0.0
        s.last_predicted = float(self.predicted_energy_changes[-1]) if self.predicted_energy_changes else 0.0
        s.has_prev = int(process_prev)
        s.consecutive_rejections = int(self.consecutive_rejections)
        s.ts_mode_idx = -1 if self.ts_mode_idx is None else int(self.ts_mode_idx)
        s.iter = int(self.iter)
        prevg = f64(np.asarray(self.prev_gradient)).reshape(-1) if process_prev else None
        prevs = f64(np.asarray(self.prev_move_vector)).reshape(-1) if process_prev else None
        pmn = float(np.linalg.norm(np.asarray(pre_move_vector).flatten())) if len(pre_move_vector) > 0 else -1.0
        ts = f64(self.ts_mode_eigvec).reshape(-1) if self.ts_mode_eigvec is not None else None
        H = f64(self.hessian)
        Bh = None if self.bias_hessian is None else f64(self.bias_hessian)
        step = np.empty(n)
        Hout = np.empty((n, n))
        tsout = np.zeros(n)
        p = self._params()
        c = ctx(self.device)
        check(lib().mopt_rsprfo_run(c, n, C.byref(p), C.byref(s), ptr(H), ptr(Bh), ptr(x), ptr(bg), ptr(px), ptr(pg), float(B_e),
                                    ptr(prevg), ptr(prevs), pmn, ptr(ts), ptr(step), ptr(Hout), ptr(tsout)), c)
        if process_prev:
            self.actual_energy_changes.append(float(s.last_actual))
            self.reduction_ratios.append(float(s.last_ratio))
        self.trust_radius = float(s.trust_radius)
        self.consecutive_rejections = int(s.consecutive_rejections)
        if s.hess_updated:
            self.hessian = Hout
        if self.saddle_order > 0 and self.mode_following_enabled and s.ts_mode_idx >= 0:
            self.ts_mode_idx = int(s.ts_mode_idx)
            self.ts_mode_eigvec = tsout
        self.predicted_energy_changes.append(float(s.new_predicted))
        self.last_info = dict(shifted=int(s.shifted), micro_converged=int(s.micro_converged), n_max=int(s.n_max),
                              jacobi_sweeps=int(s.jacobi_sweeps), step_norm=float(s.step_norm), gradient_norm=float(s.gradient_norm),
                              step_accepted=int(s.step_accepted), hess_updated=int(s.hess_updated))
        move = step.reshape(-1, 1)
        self.prev_geometry = np.copy(geom_num_list)
        self.prev_gradient = np.copy(B_g)
        self.prev_energy = B_e
        self.prev_move_vector = np.copy(step)
        self.iter += 1
        return move
