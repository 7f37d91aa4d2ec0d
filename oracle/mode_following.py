"""TEST INFRASTRUCTURE (oracle): NumPy restatement of multioptpy/Optimizer/mode_following.py::ModeFollowing, loop by loop as the
reference writes it (the product's own class, multioptpy_amd/Optimizer/mode_following.py, is vectorised).  Only tests/, smoke() and the
cpu_baseline leg of bench.py may import this package.  Pinned by G15 (tests/golden/g15_mf_rsirfo.npz: traces of the reference's MF_RSIRFO)."""
import numpy as np
from scipy.optimize import linear_sum_assignment


class ModeFollowing:
    def __init__(self, saddle_order, masses=None, initial_target_index=0, adaptive=True, update_rate=1.0, use_hungarian=True,
                 gradient_weight=0.0):
        """masses: per-atom masses in u or None (the reference looks them up from the element list, mode_following.py:52-60)."""
        self.saddle_order = saddle_order
        self.reference_modes = []
        self.reference_indices = []
        self.is_initialized = False
        self.target_offset = initial_target_index
        self.adaptive = adaptive
        self.update_rate = update_rate
        self.use_hungarian = use_hungarian
        self.gradient_weight = gradient_weight
        self.mass_weights = None if masses is None else np.repeat(np.asarray(masses, dtype=float), 3)
        self.mass_sqrt = None if masses is None else np.sqrt(self.mass_weights)

    def _calc_overlap(self, v1, v2):
        """mode_following.py:90-111"""
        if self.mass_weights is None:
            d, n1, n2 = np.dot(v1, v2), np.linalg.norm(v1), np.linalg.norm(v2)
        else:
            d = np.dot(v1 * self.mass_weights, v2)
            n1, n2 = np.linalg.norm(v1 * self.mass_sqrt), np.linalg.norm(v2 * self.mass_sqrt)
        if n1 < 1e-12 or n2 < 1e-12:
            return 0.0
        return d / (n1 * n2)

    def _normalize(self, v):
        """mode_following.py:113-125"""
        nrm = np.linalg.norm(v) if self.mass_weights is None else np.linalg.norm(v * self.mass_sqrt)
        return v if nrm < 1e-12 else v / nrm

    def set_references(self, eigvecs, eigvals=None):
        """mode_following.py:127-149"""
        self.reference_modes, self.reference_indices = [], []
        start, end = self.target_offset, self.target_offset + self.saddle_order
        if end > eigvecs.shape[1]:
            start, end = 0, self.saddle_order
        for i in range(start, end):
            self.reference_modes.append(eigvecs[:, i].copy())
            self.reference_indices.append(i)
        self.is_initialized = True

    def get_matched_indices(self, V, w=None, current_gradient=None):
        """mode_following.py:151-272"""
        nr, nc = len(self.reference_modes), V.shape[1]
        sim = np.zeros((nr, nc))
        sgn = np.zeros((nr, nc))
        go = np.zeros(nc)
        if self.gradient_weight > 0 and current_gradient is not None:
            gn = np.linalg.norm(current_gradient)
            if gn > 1e-10:
                ng = current_gradient / gn
                for j in range(nc):
                    go[j] = abs(self._calc_overlap(ng, V[:, j]))
        for i in range(nr):
            for j in range(nc):
                o = self._calc_overlap(self.reference_modes[i], V[:, j])
                sc = abs(o)
                if self.gradient_weight > 0:
                    sc += self.gradient_weight * go[j]
                sim[i, j] = sc
                sgn[i, j] = 1.0 if o >= 0 else -1.0
        pairs = []
        if self.use_hungarian:
            r, c = linear_sum_assignment(-1.0 * sim)
            pairs = sorted(zip(r.tolist(), c.tolist()), key=lambda t: t[0])
        else:
            used = set()
            for i in range(nr):
                best, mx = -1, -1.0
                for j in range(nc):
                    if j in used:
                        continue
                    if sim[i, j] > mx:
                        mx, best = sim[i, j], j
                if best != -1:
                    pairs.append((i, best))
                    used.add(best)
                else:
                    pairs.append((i, self.reference_indices[i] if self.reference_indices[i] < nc else 0))
        out = []
        for ri, cj in pairs:
            out.append(cj)
            if self.adaptive:
                a = self.update_rate
                new = V[:, cj] * sgn[ri, cj]
                if a >= 1.0:
                    upd = new
                elif a <= 0.0:
                    upd = self.reference_modes[ri]
                else:
                    upd = (1.0 - a) * self.reference_modes[ri] + a * new
                self.reference_modes[ri] = self._normalize(upd)
            self.reference_indices[ri] = cj
        return out
