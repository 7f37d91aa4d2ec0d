"""TEST INFRASTRUCTURE (oracle): NumPy restatement of multioptpy/Optimizer/mf_rsirfo.py::MF_RSIRFO on top of the oracle's RSIRFO.
Only tests/, smoke() and the cpu_baseline leg of bench.py may import this package.  Pinned by G15 (traces of the reference's class)."""
import numpy as np

from . import projection as pj
from .mode_following import ModeFollowing
from .rsirfo import RSIRFO


def parse_method(raw, adaptive_default=True):
    """mf_rsirfo.py:35-92"""
    name, target, rate, gw = raw, 0, None, 0.0
    if ":" in raw:
        parts = raw.split(":")
        name = parts[0].strip()
        for part in parts[1:]:
            part = part.strip().lower()
            if not part:
                continue
            if part.isdigit():
                target = int(part)
            elif part.startswith("ema"):
                try:
                    rate = float(part[3:])
                except ValueError:
                    pass
            elif part.startswith("grad"):
                try:
                    gw = float(part[4:])
                except ValueError:
                    pass
    if rate is not None:
        adaptive = rate > 1e-12
    else:
        adaptive = adaptive_default
        rate = 1.0 if adaptive else 0.0
    return name, target, adaptive, rate, gw


class MF_RSIRFO(RSIRFO):
    def __init__(self, masses=None, **config):
        """masses: per-atom masses (u) when the reference would get an element_list, else None."""
        name, target, adaptive, rate, gw = parse_method(config.get("method", "auto"), config.get("adaptive_mode_following", True))
        config = dict(config, method=name)
        super().__init__(**config)
        self.hessian_update_method = name
        self.use_mode_following = config.get("use_mode_following", True)
        self.mode_follower = ModeFollowing(self.saddle_order, masses=masses, initial_target_index=target, adaptive=adaptive, update_rate=rate,
                                           use_hungarian=config.get("use_hungarian", True), gradient_weight=gw)
        self.last_targets = []

    def run(self, geom_num_list, B_g, pre_B_g=[], pre_geom=[], B_e=0.0, pre_B_e=0.0, pre_move_vector=[], initial_geom_num_list=[], g=[], pre_g=[]):
        """mf_rsirfo.py:128-265"""
        self.trace = {}
        if self.Initialization:
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
        if self.prev_geometry is not None and self.prev_gradient is not None and len(pre_g) > 0 and len(pre_geom) > 0:
            self.update_hessian(geom_num_list, g, pre_geom, pre_g)
        gnorm = np.linalg.norm(B_g)
        if gnorm < self.gradient_norm_threshold:
            self.converged = True
        if self.actual_energy_changes and abs(self.actual_energy_changes[-1]) < self.energy_change_threshold:
            self.converged = True
        gradient = np.asarray(B_g).ravel()                         # (:163: NOT projected, unlike RSIRFO.run)
        H0 = self.hessian if self.bias_hessian is None else self.hessian + self.bias_hessian
        H = pj.project_hessian_trrot(H0, np.asarray(geom_num_list).reshape(-1, 3))
        H = 0.5 * (H + H.T)
        w, v = self.eigh_with_shift(H)
        targets = []
        if self.saddle_order > 0:
            if self.use_mode_following:
                if self.iteration == 0:
                    self.mode_follower.set_references(v, w)
                    st = self.mode_follower.target_offset
                    targets = list(range(st, st + self.saddle_order))
                else:
                    targets = self.mode_follower.get_matched_indices(v, w, current_gradient=gradient)
            else:
                targets = list(range(self.saddle_order))
        self.last_targets = [int(t) for t in targets]
        if self.iteration > 0 and self.prev_energy is not None:
            actual = B_e - self.prev_energy
            if len(self.actual_energy_changes) >= 3:
                self.actual_energy_changes.pop(0)
            self.actual_energy_changes.append(actual)
            if self.predicted_energy_changes:
                mine = w[0]
                if targets and targets[0] < len(w):
                    mine = w[targets[0]]
                self.adjust_trust_radius(actual, self.predicted_energy_changes[-1], mine, gnorm)
        P = np.eye(gradient.size)
        for idx in targets:
            if idx < len(w) and abs(w[idx]) > 1e-10:
                t = v[:, idx]
                P -= (1.0 if self.NEB_mode else 2.0) * np.outer(t, t)
        Hs = P @ H
        Hs = 0.5 * (Hs + Hs.T)
        gs = P @ gradient
        ws, vs = self.eigh_with_shift(Hs)
        keep = ~(np.abs(ws) < self.small_eigval_thresh)
        ws, vs = ws[keep], vs[:, keep]
        self.prev_eigvec_size = vs.shape[1]
        move = self.get_rs_step(ws, vs, gs)
        pred = gradient @ move + 0.5 * (move @ H @ move)
        if len(self.predicted_energy_changes) >= 3:
            self.predicted_energy_changes.pop(0)
        self.predicted_energy_changes.append(pred)
        self.prev_geometry = geom_num_list
        self.prev_gradient = B_g
        self.prev_energy = B_e
        self.iteration += 1
        return -1 * move.reshape(-1, 1)
