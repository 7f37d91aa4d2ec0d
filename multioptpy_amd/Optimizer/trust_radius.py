"""Drop-in for ``multioptpy.Optimizer.trust_radius.TrustRadius`` (the outer, composite-step trust radius of optmain;
SURVEY.md section 8 row a15, reference file multioptpy/Optimizer/trust_radius.py).

Same constructor, attributes and ``update_trust_radii`` signature.  The scalar rules are few and stay on the host;
the one O(n^2) quantity -- the quadratic model's predicted energy change g.s + 1/2 s^T H s (reference :146-151) -- is
evaluated on the device by ``mopt_quadratic_model``."""
import math

import numpy as np

from .. import ops

# (energy, gradients, step, stress) per quality label, reference :66-72
_QUALITY = {
    "VeryBasic": (1e-3, 1e-1, 1.0, 5e-2),
    "Basic": (1e-4, 1e-2, 0.1, 5e-3),
    "Normal": (1e-5, 1e-3, 0.01, 5e-4),
    "Good": (1e-6, 1e-4, 0.001, 5e-5),
    "VeryGood": (1e-7, 1e-5, 0.0001, 5e-6),
}
_R_POOR, _R_GOOD = 0.25, 0.75            # acceptance window of the actual/predicted ratio, reference :175-176


class TrustRadius:
    def __init__(self, initial_trust_radius=0.3, min_trust_radius=0.01, max_trust_radius=0.5, history_size=5,
                 adaptive_factor_scale=0.8, energy_precision_threshold=1e-8, quality="Normal"):
        self.trust_radius = initial_trust_radius
        self.min_trust_radius, self.max_trust_radius = min_trust_radius, max_trust_radius
        self.history_size = history_size
        self.adaptive_factor_scale = adaptive_factor_scale
        self.energy_precision_threshold = energy_precision_threshold
        self.energy_ratios, self.step_sizes, self.energy_changes = [], [], []
        self.iteration_count = 0
        self.last_energy = None
        self.set_convergence_quality(quality)

    # -- setters kept for interface compatibility ------------------------------------------------------------------
    def set_min_trust_radius(self, min_trust_radius):
        self.min_trust_radius = min_trust_radius

    def set_max_trust_radius(self, max_trust_radius):
        self.max_trust_radius = max_trust_radius

    def set_convergence_quality(self, quality):
        self.quality = quality if quality in _QUALITY else "Normal"
        self.convergence_criteria = dict(zip(("energy", "gradients", "step", "stress"), _QUALITY[self.quality]))

    # -- history-dependent damping ------------------------------------------------------------------------------------
    def is_approaching_convergence(self):
        """Last (up to three) energy changes all below 0.01 and below 0.005 on average (reference :105-117)."""
        tail = [abs(v) for v in self.energy_changes[-3:]]
        return len(self.energy_changes) >= 2 and max(tail) < 0.01 and sum(tail) / len(tail) < 0.005

    def calculate_adaptive_factor(self):
        """2 exp(-var(recent ratios)), damped near convergence, clamped to [1.1, 3] (reference :79-103)."""
        if not self.energy_ratios:
            return 2.0
        recent = self.energy_ratios[-self.history_size:]
        spread = float(np.var(recent)) if len(recent) > 1 else 0.0
        factor = 2.0 * math.exp(-spread)
        if self.is_approaching_convergence():
            factor *= self.adaptive_factor_scale
        return min(max(factor, 1.1), 3.0)

    # -- the update ---------------------------------------------------------------------------------------------------
    def update_trust_radii(self, B_e, pre_B_e, pre_B_g, pre_move_vector, model_hess, geom_num_list, trust_radii,
                           atom_types=None, constraints=None):
        if self.iteration_count == 0:                      # nothing to compare with on the first call
            self.iteration_count = 1
            return trust_radii
        dof = len(geom_num_list)
        step = np.asarray(pre_move_vector, dtype=np.float64).reshape(dof)
        predicted = ops.quadratic_model(np.asarray(model_hess, dtype=np.float64).reshape(dof, dof),
                                        np.asarray(pre_B_g, dtype=np.float64).reshape(dof), step)
        eps = self.energy_precision_threshold
        if abs(predicted) < eps:                           # keep the ratio finite (reference :154-158)
            predicted += math.copysign(eps, predicted) if predicted != 0.0 else 0.0
            if abs(predicted) < eps:
                predicted = eps
        gain = pre_B_e - B_e
        ratio = gain / predicted
        length = float(np.linalg.norm(step))
        self.energy_ratios.append(float(ratio))
        self.step_sizes.append(length)
        self.energy_changes.append(float(gain))
        damping = self.calculate_adaptive_factor()
        if ratio <= _R_POOR or ratio >= 2.0 - _R_POOR:
            trust_radii = trust_radii / damping            # model and surface disagree: shrink
        elif _R_GOOD <= ratio <= 2.0 - _R_GOOD and abs(length - trust_radii) < eps:
            trust_radii = trust_radii * math.sqrt(damping)  # good model and the step sat on the boundary: grow
        self.iteration_count += 1
        return np.clip(trust_radii, self.min_trust_radius, self.max_trust_radius)
