"""RSIRFO: drop-in for ``multioptpy.Optimizer.rsirfo.RSIRFO`` (reference rsirfo.py:9-1754) running on MI355X.

Same constructor keywords, same ``run(...)`` signature and return value ((3N,1) step to be
SUBTRACTED, Bohr), same public attributes (``hessian``, ``bias_hessian``, ``trust_radius``,
``Initialization``, ``iteration``, ``DELTA``, ``linesearchflag``, ``NEB_mode`` ...).  The
Python object only carries state; every O(n)..O(n^3) operation of the step happens in
libmopt_hip.so (``mopt_rsirfo_run``): Hessian update, TR/rot projections, the symmetric
eigen-solve, the secular-equation RFO step and its alpha search, the predicted energy.

Object semantics kept from the reference: ``set_hessian`` stores the caller's array
by reference; a Hessian update adds the delta INTO that array in place and then rebinds
``self.hessian`` to a new symmetrised array (rsirfo.py:1368-1372).
"""
import ctypes as C

import numpy as np

from .. import _abi
from .._lib import lib, ctx, check, f64, ptr


class RSIRFO:
    def __init__(self, **config):
        g = config.get
        # rsirfo.py:26-91
        self.alpha0 = g("alpha0", 1.0)
        self.max_micro_cycles = g("max_micro_cycles", 40)
        self.saddle_order = g("saddle_order", 1)
        self.hessian_update_method = g("method", "auto")
        self.small_eigval_thresh = g("small_eigval_thresh", 1e-6)
        self.alpha_max = g("alpha_max", 1000.0)
        self.alpha_step_max = g("alpha_step_max", 10.0)
        if self.saddle_order == 0:
            self.trust_radius_initial = g("trust_radius", 0.5)
            self.trust_radius_max = g("trust_radius_max", 0.5)
        else:
            self.trust_radius_initial = g("trust_radius", 0.1)
            self.trust_radius_max = g("trust_radius_max", 0.1)
        self.trust_radius = self.trust_radius_initial
        self.trust_radius_min = g("trust_radius_min", 0.01)
        self.good_step_threshold = g("good_step_threshold", 0.75)
        self.poor_step_threshold = g("poor_step_threshold", 0.25)
        self.trust_radius_increase_factor = g("trust_radius_increase_factor", 1.2)
        self.trust_radius_decrease_factor = g("trust_radius_decrease_factor", 0.5)
        self.energy_change_threshold = g("energy_change_threshold", 1e-6)
        self.gradient_norm_threshold = g("gradient_norm_threshold", 1e-4)
        self.step_norm_tolerance = g("step_norm_tolerance", 1e-3)
        self.debug_mode = g("debug_mode", False)
        self.display_flag = g("display_flag", True)
        self.use_adaptive_trust_radius = g("use_adaptive_trust_radius", True)
        self.adaptive_trust_gradient_norm_threshold = g("adaptive_trust_gradient_norm_threshold", 1e-2)
        self.max_curvature_factor = g("max_curvature_factor", 2.5)
        self.negative_curvature_safety = g("negative_curvature_safety", 0.8)
        self.min_eigenvalue_history = []
        self.use_level_shift = g("use_level_shift", False)
        self.level_shift_value = g("level_shift_value", 1e-5)
        self.auto_level_shift = g("auto_level_shift", True)
        self.condition_number_threshold = g("condition_number_threshold", 1e8)
        self.level_shift_applied = False
        self.Initialization = True
        self.hessian = None
        self.bias_hessian = None
        self.prev_eigvec_min = None
        self.prev_eigvec_size = None
        self.predicted_energy_changes = []
        self.actual_energy_changes = []
        self.prev_geometry = None
        self.prev_gradient = None
        self.prev_energy = None
        self.converged = False
        self.iteration = 0
        self.roots = list(range(self.saddle_order))
        self.NEB_mode = False
        # set by the dispatcher (multioptpy/optimizer.py:458-460)
        self.DELTA = None
        self.linesearchflag = False
        self.device = g("device", None)
        self.last_info = {}

    # -- plugin surface (rsirfo.py:254-258, 1722-1755) ------------------------------------
    def switch_NEB_mode(self):
        self.NEB_mode = not self.NEB_mode

    def set_hessian(self, hessian):
        self.hessian = hessian

    def set_bias_hessian(self, bias_hessian):
        self.bias_hessian = bias_hessian

    def get_hessian(self):
        return self.hessian

    def get_bias_hessian(self):
        return self.bias_hessian

    def is_converged(self):
        return self.converged

    def get_predicted_energy_changes(self):
        return self.predicted_energy_changes

    def get_actual_energy_changes(self):
        return self.actual_energy_changes

    def reset_trust_radius(self):
        self.trust_radius = self.trust_radius_initial

    # -- marshalling ----------------------------------------------------------------------
    def _params(self):
        p = _abi.RsirfoParams()
        for name in ("alpha0", "alpha_max", "alpha_step_max", "small_eigval_thresh", "trust_radius_min", "trust_radius_max",
                     "good_step_threshold", "poor_step_threshold", "trust_radius_increase_factor",
                     "trust_radius_decrease_factor", "energy_change_threshold", "gradient_norm_threshold",
                     "step_norm_tolerance", "adaptive_trust_gradient_norm_threshold", "max_curvature_factor",
                     "negative_curvature_safety", "level_shift_value", "condition_number_threshold"):
            setattr(p, name, float(getattr(self, name)))
        p.max_micro_cycles = int(self.max_micro_cycles)
        p.saddle_order = int(len(self.roots))
        p.rule = _abi.rule_from_method(self.hessian_update_method)
        p.use_adaptive_trust_radius = int(bool(self.use_adaptive_trust_radius))
        p.use_level_shift = int(bool(self.use_level_shift))
        p.auto_level_shift = int(bool(self.auto_level_shift))
        p.neb_mode = int(bool(self.NEB_mode))
        return p

    def _state(self):
        s = _abi.RsirfoState()
        s.trust_radius = float(self.trust_radius)
        s.prev_energy = float(self.prev_energy) if self.prev_energy is not None else 0.0
        pe = list(self.predicted_energy_changes)[-3:]
        ae = list(self.actual_energy_changes)[-3:]
        for i, v in enumerate(pe):
            s.predicted[i] = float(v)
        for i, v in enumerate(ae):
            s.actual[i] = float(v)
        s.n_predicted, s.n_actual = len(pe), len(ae)
        s.has_prev = int(self.prev_energy is not None)
        s.iteration = int(self.iteration)
        s.converged = int(bool(self.converged))
        s.prev_eigvec_size = -1 if self.prev_eigvec_size is None else int(self.prev_eigvec_size)
        return s

    def _unpack(self, s):
        self.trust_radius = float(s.trust_radius)
        self.predicted_energy_changes = [float(s.predicted[i]) for i in range(s.n_predicted)]
        self.actual_energy_changes = [float(s.actual[i]) for i in range(s.n_actual)]
        self.iteration = int(s.iteration)
        self.converged = bool(s.converged)
        self.prev_eigvec_size = int(s.prev_eigvec_size)
        self.level_shift_applied = bool(s.level_shift_applied)
        self.last_info = dict(branch=int(s.branch), hess_updated=int(s.hess_updated), n_kept=int(s.n_kept),
                              jacobi_sweeps=int(s.jacobi_sweeps), secular_iters=int(s.secular_iters), alpha=float(s.alpha),
                              lambda_aug=float(s.lambda_aug), initial_step_norm=float(s.initial_step_norm),
                              gradient_norm=float(s.gradient_norm))

    # -- the step (rsirfo.py:285-490) -----------------------------------------------------
    def run(self, geom_num_list, B_g, pre_B_g=[], pre_geom=[], B_e=0.0, pre_B_e=0.0, pre_move_vector=[],
            initial_geom_num_list=[], g=[], pre_g=[]):
        if self.Initialization:
            self.prev_eigvec_min = None
            self.prev_eigvec_size = None
            self.predicted_energy_changes = []
            self.actual_energy_changes = []
            self.prev_geometry = None
            self.prev_gradient = None
            self.prev_energy = None
            self.converged = False
            self.iteration = 0
            self.Initialization = False
        if self.hessian is None:
            raise ValueError("Hessian matrix must be set before running optimization")
        x = f64(np.asarray(geom_num_list)).reshape(1, -1)
        n = x.shape[1]
        bg = f64(np.asarray(B_g)).reshape(1, n)
        # rsirfo.py:308 -- evaluated exactly as the reference does (len(None) raises TypeError there too)
        do_update = (self.prev_geometry is not None and self.prev_gradient is not None
                     and len(pre_g) > 0 and len(pre_geom) > 0)
        graw = f64(np.asarray(g)).reshape(1, n) if (g is not None and len(g) > 0) else bg
        px = pg = None
        if do_update:
            px = f64(np.asarray(pre_geom)).reshape(1, n)
            pg = f64(np.asarray(pre_g)).reshape(1, n)
        H = self.hessian
        if not isinstance(H, np.ndarray) or H.shape != (n, n):
            raise ValueError("Hessian must be a (3N,3N) ndarray")
        Hc = f64(H)
        Bh = None if self.bias_hessian is None else f64(self.bias_hessian)
        be = np.array([float(B_e)])
        step = np.empty((1, n))
        Hin = np.empty((n, n)) if do_update else None
        Hsym = np.empty((n, n)) if do_update else None
        p = self._params()
        s = self._state()
        c = ctx(self.device)
        check(lib().mopt_rsirfo_run(c, n, 1, C.byref(p), C.byref(s), ptr(Hc), ptr(Bh), ptr(x), ptr(bg), ptr(graw), ptr(px), ptr(pg),
                                    ptr(be), ptr(step), ptr(Hin), ptr(Hsym)), c)
        self._unpack(s)
        if do_update and s.hess_updated:
            # `self.hessian += delta_hess` mutates the array the caller handed to set_hessian ...
            if H.dtype == np.float64:
                H[...] = Hin
            else:
                H[...] = Hin.astype(H.dtype)
            # ... and `self.hessian = 0.5 * (self.hessian + self.hessian.T)` rebinds to a new array
            self.hessian = Hsym
        self.prev_geometry = geom_num_list
        self.prev_gradient = B_g
        self.prev_energy = B_e
        return step.reshape(-1, 1)
