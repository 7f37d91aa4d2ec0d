"""ModeFollowing: drop-in for ``multioptpy.Optimizer.mode_following.ModeFollowing`` (reference mode_following.py:6-272).

Host glue of MF_RSIRFO: the eigenvectors of one geometry (n x n) are on the host between the two halves of the step
(``mopt_mf_rsirfo_begin`` / ``_finish``) and the follower's work on them is O(saddle_order * n^2): overlaps of the reference vectors with
every current mode (Cartesian or mass-weighted), optional gradient bias, Hungarian (scipy ``linear_sum_assignment``, as the reference) or
greedy matching, exponential-moving-average update of the references.  Same constructor arguments, attributes and return values."""
import json
import os

import numpy as np
from scipy.optimize import linear_sum_assignment

_MASS = None


def _masses():
    global _MASS
    if _MASS is None:
        here = os.path.dirname(os.path.abspath(__file__))
        with open(os.path.join(here, "..", "ModelHessian", "element_tables.json")) as fh:
            tab = json.load(fh)
        _MASS = {e: v.get("mass") for e, v in tab["elements"].items()}
        _MASS["_order"] = tab["order"]
    return _MASS


def atomic_mass(elem):
    """Isotope mass in u for a symbol or an atomic number (Parameters/atomic_mass.py:4-30; data: ModelHessian/element_tables.json)."""
    tab = _masses()
    if isinstance(elem, (int, np.integer)):
        elem = tab["_order"][int(elem) - 1]
    m = tab.get(elem)
    if m is None:
        raise KeyError(elem)
    return m


class ModeFollowing:
    def __init__(self, saddle_order, atoms=None, initial_target_index=0, adaptive=True, update_rate=1.0, use_hungarian=True,
                 gradient_weight=0.0, debug_mode=False):
        self.saddle_order = saddle_order
        self.debug_mode = debug_mode
        self.reference_modes = []
        self.reference_indices = []
        self.is_initialized = False
        self.target_offset = initial_target_index
        self.adaptive = adaptive
        self.update_rate = update_rate
        self.use_hungarian = use_hungarian
        self.gradient_weight = gradient_weight
        self.mass_weights = None
        self.mass_sqrt = None
        self.strategies = []
        if atoms is not None:
            try:                                                      # mode_following.py:52-66
                self.mass_weights = np.repeat(np.array([atomic_mass(a) for a in atoms], dtype=float), 3)
                self.mass_sqrt = np.sqrt(self.mass_weights)
                self.strategies.append("Mass-Weighted")
            except Exception as exc:                                  # noqa: BLE001  (the reference falls back on any failure)
                print(f"[ModeFollowing] Warning: Mass init failed ({exc}). Using Cartesian.")
                self.mass_weights = self.mass_sqrt = None
                self.strategies.append("Cartesian")
        else:
            self.strategies.append("Cartesian")
        self.strategies.append(f"Adaptive(EMA={self.update_rate})" if self.adaptive else "Static")
        self.strategies.append("Hungarian" if self.use_hungarian else "Greedy")
        if self.gradient_weight > 0:
            self.strategies.append(f"GradBias({self.gradient_weight})")

    def log(self, message):
        if self.debug_mode:
            print(f"[ModeFollowing] {message}")

    # -- metric ---------------------------------------------------------------------------
    def _norms(self, cols):
        """Norm of every column of `cols` (n x m) in the current metric."""
        if self.mass_sqrt is None:
            return np.linalg.norm(cols, axis=0)
        return np.linalg.norm(cols * self.mass_sqrt[:, None], axis=0)

    def _overlaps(self, refs, cols):
        """Signed normalised overlaps S[i, j] of the rows of `refs` with the columns of `cols` (mode_following.py:90-111)."""
        left = refs if self.mass_weights is None else refs * self.mass_weights[None, :]
        dots = left @ cols
        n1 = self._norms(refs.T)
        n2 = self._norms(cols)
        small = (n1[:, None] < 1e-12) | (n2[None, :] < 1e-12)
        den = np.where(small, 1.0, n1[:, None] * n2[None, :])
        return np.where(small, 0.0, dots / den)

    def _calc_overlap(self, v1, v2):
        return float(self._overlaps(np.asarray(v1, dtype=float)[None, :], np.asarray(v2, dtype=float)[:, None])[0, 0])

    def _normalize(self, v):
        nrm = float(self._norms(np.asarray(v, dtype=float)[:, None])[0])
        return v if nrm < 1e-12 else v / nrm

    # -- references -----------------------------------------------------------------------
    def set_references(self, eigvecs, eigvals=None):
        """mode_following.py:127-149"""
        n_modes = eigvecs.shape[1]
        start, end = self.target_offset, self.target_offset + self.saddle_order
        if end > n_modes:
            start, end = 0, self.saddle_order
            print("[ModeFollowing] Error: Index out of bounds. Fallback to 0.")
        self.reference_modes = [eigvecs[:, i].copy() for i in range(start, end)]
        self.reference_indices = list(range(start, end))
        self.is_initialized = True

    def get_matched_indices(self, current_eigvecs, current_eigvals=None, current_gradient=None):
        """mode_following.py:151-272: indices of the current modes matched to the references (in reference order)."""
        if not self.is_initialized:
            raise RuntimeError("References not set.")
        n_refs, n_curr = len(self.reference_modes), current_eigvecs.shape[1]
        signed = self._overlaps(np.array(self.reference_modes, dtype=float).reshape(n_refs, -1), current_eigvecs) if n_refs else np.zeros((0, n_curr))
        sim = np.abs(signed)
        if self.gradient_weight > 0 and current_gradient is not None:
            gvec = np.asarray(current_gradient, dtype=float).ravel()
            gn = np.linalg.norm(gvec)
            if gn > 1e-10:
                sim = sim + self.gradient_weight * np.abs(self._overlaps((gvec / gn)[None, :], current_eigvecs))[0][None, :]
        sign = np.where(signed >= 0, 1.0, -1.0)
        pairs = []
        if self.use_hungarian:
            rows, cols = linear_sum_assignment(-1.0 * sim)
            pairs = sorted(zip(rows.tolist(), cols.tolist()))
        else:
            used = set()
            for i in range(n_refs):
                best, best_sim = -1, -1.0
                for j in range(n_curr):
                    if j not in used and sim[i, j] > best_sim:
                        best, best_sim = j, sim[i, j]
                if best != -1:
                    pairs.append((i, best))
                    used.add(best)
                else:
                    print(f"    [ModeFollowing] LOST TRACK of Ref {i}")
                    pairs.append((i, self.reference_indices[i] if self.reference_indices[i] < n_curr else 0))
        matched = []
        for ref_i, curr_j in pairs:
            matched.append(curr_j)
            if self.adaptive:
                alpha = self.update_rate
                new_vec = current_eigvecs[:, curr_j] * sign[ref_i, curr_j]
                if alpha >= 1.0:
                    upd = new_vec
                elif alpha <= 0.0:
                    upd = self.reference_modes[ref_i]
                else:
                    upd = (1.0 - alpha) * self.reference_modes[ref_i] + alpha * new_vec
                self.reference_modes[ref_i] = self._normalize(upd)
            self.reference_indices[ref_i] = curr_j
        return matched
