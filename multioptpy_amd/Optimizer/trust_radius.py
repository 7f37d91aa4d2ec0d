"""Drop-in for multioptpy/Optimizer/trust_radius.py::TrustRadius (the outer, composite-step trust radius of optmain).

Scalar bookkeeping stays on the host exactly as in the reference; the one O(n^2) term -- the predicted energy change
g.s + 1/2 s^T H s (trust_radius.py:146-151) -- runs on the device (mopt_quadratic_model)."""
import numpy as np

from .. import ops


class TrustRadius:
    def __init__(self, initial_trust_radius=0.3, min_trust_radius=0.01, max_trust_radius=0.5, history_size=5,
                 adaptive_factor_scale=0.8, energy_precision_threshold=1e-8, quality="Normal"):
        self.trust_radius = initial_trust_radius
        self.min_trust_radius = min_trust_radius
        self.max_trust_radius = max_trust_radius
        self.history_size = history_size
        self.adaptive_factor_scale = adaptive_factor_scale
        self.energy_precision_threshold = energy_precision_threshold
        self.energy_ratios = []
        self.step_sizes = []
        self.energy_changes = []
        self.iteration_count = 0
        self.last_energy = None
        self.set_convergence_quality(quality)

    def set_min_trust_radius(self, min_trust_radius):
        self.min_trust_radius = min_trust_radius

    def set_max_trust_radius(self, max_trust_radius):
        self.max_trust_radius = max_trust_radius

    def set_convergence_quality(self, quality):                               # trust_radius.py:57-77
        table = {"VeryBasic": (1e-3, 1e-1, 1.0, 5e-2), "Basic": (1e-4, 1e-2, 0.1, 5e-3), "Normal": (1e-5, 1e-3, 0.01, 5e-4),
                 "Good": (1e-6, 1e-4, 0.001, 5e-5), "VeryGood": (1e-7, 1e-5, 0.0001, 5e-6)}
        if quality not in table:
            quality = "Normal"
        e, g, s, st = table[quality]
        self.convergence_criteria = {"energy": e, "gradients": g, "step": s, "stress": st}
        self.quality = quality

    def calculate_adaptive_factor(self):                                      # :79-103
        if len(self.energy_ratios) == 0:
            return 2.0
        recent = self.energy_ratios[-min(self.history_size, len(self.energy_ratios)):]
        ratio_variance = np.var(recent) if len(recent) > 1 else 0.0
        base_factor = 2.0 * np.exp(-ratio_variance)
        if self.is_approaching_convergence():
            base_factor *= self.adaptive_factor_scale
        return max(1.1, min(base_factor, 3.0))

    def is_approaching_convergence(self):                                     # :105-117
        if len(self.energy_changes) < 2:
            return False
        recent = np.abs(self.energy_changes[-min(3, len(self.energy_changes)):])
        return bool(np.all(recent < 0.01) and np.mean(recent) < 0.005)

    def update_trust_radii(self, B_e, pre_B_e, pre_B_g, pre_move_vector, model_hess, geom_num_list, trust_radii,
                           atom_types=None, constraints=None):                # :120-206
        if self.iteration_count == 0:
            self.iteration_count += 1
            return trust_radii
        n = len(geom_num_list)
        Ce = ops.quadratic_model(np.asarray(model_hess, dtype="float64").reshape(n, n), np.asarray(pre_B_g).reshape(n),
                                 np.asarray(pre_move_vector).reshape(n))
        thr = self.energy_precision_threshold
        if abs(Ce) < thr:
            Ce += np.sign(Ce) * thr
            if abs(Ce) < thr:
                Ce = thr
        r = (pre_B_e - B_e) / Ce
        step_norm = float(np.linalg.norm(pre_move_vector))
        self.energy_ratios.append(float(r))
        self.step_sizes.append(step_norm)
        self.energy_changes.append(float(pre_B_e - B_e))
        adaptive_factor = self.calculate_adaptive_factor()
        r_min, r_good = 0.25, 0.75
        if r <= r_min or r >= (2.0 - r_min):
            trust_radii /= adaptive_factor
        elif r >= r_good and r <= (2.0 - r_good):
            if abs(step_norm - trust_radii) < thr:
                trust_radii *= adaptive_factor ** 0.5
        self.iteration_count += 1
        return np.clip(trust_radii, self.min_trust_radius, self.max_trust_radius)
