"""MF_RSIRFO: drop-in for ``multioptpy.Optimizer.mf_rsirfo.MF_RSIRFO`` (reference mf_rsirfo.py:11-265) running on MI355X.

RS-I-RFO whose reflected modes are chosen by a mode follower instead of being the lowest ones.  The step runs in two halves of the C ABI:
``mopt_mf_rsirfo_begin`` (Hessian update, TR/rot projection, symmetric eigen-solve; eigenvalues and eigenvectors come to the host),
``ModeFollowing`` on the host (Optimizer/mode_following.py), ``mopt_mf_rsirfo_finish`` (trust radius with the tracked mode's curvature,
image function over the matched modes, RS step with the unprojected gradient, predicted change).  Same constructor keywords -- including the
extended ``method`` string "name : target_index : ema<val> : grad<val>" -- attributes and return value as the reference."""
import ctypes as C

import numpy as np

from .._lib import lib, ctx, check, f64, ptr
from .mode_following import ModeFollowing
from .rsirfo import RSIRFO


def parse_method(raw, adaptive_default=True):
    """mf_rsirfo.py:35-92 -> (update method, target index, adaptive, EMA rate, gradient weight)."""
    update_method, target, rate, grad_w = raw, 0, None, 0.0
    if ":" in raw:
        parts = raw.split(":")
        update_method = parts[0].strip()
        for part in (p.strip().lower() for p in parts[1:]):
            if not part:
                continue
            if part.isdigit():
                target = int(part)
            elif part.startswith("ema"):
                try:
                    rate = float(part[3:])
                except ValueError:
                    print(f"Warning: Invalid ema value in '{part}'. Ignoring.")
            elif part.startswith("grad"):
                try:
                    grad_w = float(part[4:])
                except ValueError:
                    print(f"Warning: Invalid grad value in '{part}'. Ignoring.")
    if rate is not None:
        adaptive = rate > 1e-12
    else:
        adaptive = adaptive_default
        rate = 1.0 if adaptive else 0.0
    return update_method, target, adaptive, rate, grad_w


class MF_RSIRFO(RSIRFO):
    def __init__(self, **config):
        update_method, target, adaptive, rate, grad_w = parse_method(config.get("method", "auto"), config.get("adaptive_mode_following", True))
        config["method"] = update_method
        super().__init__(**config)
        self.hessian_update_method = update_method
        self.use_mode_following = config.get("use_mode_following", True)
        self.mode_follower = ModeFollowing(self.saddle_order, atoms=config.get("element_list", None), initial_target_index=target,
                                           adaptive=adaptive, update_rate=rate, use_hungarian=config.get("use_hungarian", True),
                                           gradient_weight=grad_w, debug_mode=config.get("debug_mode", False))
        self.last_targets = []

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
            raise ValueError("Hessian matrix must be set")
        x = f64(np.asarray(geom_num_list)).reshape(1, -1)
        n = x.shape[1]
        bg = f64(np.asarray(B_g)).reshape(1, n)
        do_update = (self.prev_geometry is not None and self.prev_gradient is not None and len(pre_g) > 0 and len(pre_geom) > 0)   # mf_rsirfo.py:148
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
        W = np.empty(n)
        V = np.empty((n, n))
        Hin = np.empty((n, n)) if do_update else None
        Hsym = np.empty((n, n)) if do_update else None
        p, s = self._params(), self._state()
        updated = C.c_int(0)
        c = ctx(self.device)
        check(lib().mopt_mf_rsirfo_begin(c, n, C.byref(p), C.byref(s), ptr(Hc), ptr(Bh), ptr(x), ptr(bg), ptr(graw), ptr(px), ptr(pg), float(B_e),
                                         ptr(W), ptr(V), ptr(Hin), ptr(Hsym), C.byref(updated)), c)
        if do_update and updated.value:
            # F7: `self.hessian += delta` mutates the caller's array, the symmetrisation rebinds (rsirfo.py:1368-1372)
            H[...] = Hin if H.dtype == np.float64 else Hin.astype(H.dtype)
            self.hessian = Hsym
        # ---- mode following on the host (mf_rsirfo.py:173-194); eigenvectors as columns, ascending
        eigvecs = V.T
        targets = []
        if self.saddle_order > 0:
            if self.use_mode_following:
                if self.iteration == 0:
                    self.mode_follower.set_references(eigvecs, W)
                    start = self.mode_follower.target_offset
                    targets = list(range(start, start + self.saddle_order))
                else:
                    targets = self.mode_follower.get_matched_indices(eigvecs, W, current_gradient=np.asarray(B_g).ravel())
            else:
                targets = list(range(self.saddle_order))
        if len(targets) > 8:
            raise NotImplementedError("MF_RSIRFO: more than eight followed modes")
        self.last_targets = [int(t) for t in targets]
        tg = (C.c_int * 8)(*self.last_targets)
        step = np.empty((1, n))
        between = getattr(self, "_between_halves", None)          # test hook: other library calls on this context between the halves
        if between is not None:
            between()
        check(lib().mopt_mf_rsirfo_finish(c, n, C.byref(p), C.byref(s), tg, len(self.last_targets), ptr(step)), c)
        self._unpack(s)
        self.prev_geometry = geom_num_list
        self.prev_gradient = B_g
        self.prev_energy = B_e
        return step.reshape(-1, 1)
