"""Oracle: Enhanced RS-P-RFO optimizer step (test infrastructure, see oracle/__init__.py).

Restates multioptpy/Optimizer/rsprfo.py (EnhancedRSPRFO, v1.21.2):
  run                          rsprfo.py:713-886
  _process_previous_step       rsprfo.py:908-962   (reduction ratio :434-472, trust radius :474-512)
  update_hessian               rsprfo.py:1190-1260 (no s.y guard; reverts when max|eig| > 1e6)
  _project_grad_tr_rot         rsprfo.py:227-285   (drops rank-deficient QR columns; < 3 atoms: no projection)
  _shift_hessian_eigenvalues   rsprfo.py:287-355
  mode following               rsprfo.py:964-1071
  micro cycles / alpha         rsprfo.py:514-711, augmented Hessian :1097-1168
Printing is omitted.
"""
import numpy as np

from . import hessian_update as hu
from . import projection as pj


def project_gradient_trrot_rsprfo(g, coords):
    """rsprfo.py:227-285."""
    g = np.asarray(g, dtype=float).ravel()
    c = np.asarray(coords, dtype=float).reshape(-1, 3)
    if c.shape[0] < 3:
        return g
    A = pj.trrot_vectors(c).T
    Q, R = np.linalg.qr(A, mode="reduced")
    valid = np.abs(np.diag(R)) > 1e-10
    Q = Q[:, valid]
    return g - Q @ (Q.T @ g)


class EnhancedRSPRFO:
    def __init__(self, **config):
        g = config.get
        self.alpha0 = g("alpha0", 1.0)
        self.max_micro_cycles = g("max_micro_cycles", 50)
        self.saddle_order = g("saddle_order", 1)
        self.hessian_update_method = g("method", "auto")
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
        self._reset_state()
        self.trust_radius = self.trust_radius_initial
        self.trace = {}

    # -- plugin surface ---------------------------------------------------------------
    def set_hessian(self, h):
        self.hessian = np.asarray(h).copy()
        self.hessian = 0.5 * (self.hessian + self.hessian.T)

    def set_bias_hessian(self, h):
        self.bias_hessian = np.asarray(h).copy()
        self.bias_hessian = 0.5 * (self.bias_hessian + self.bias_hessian.T)

    def get_hessian(self):
        return self.hessian

    def get_bias_hessian(self):
        return self.bias_hessian

    def _reset_state(self):
        self.prev_eigvec_max = None
        self.prev_eigvec_min = None
        self.predicted_energy_changes = []
        self.actual_energy_changes = []
        self.reduction_ratios = []
        self.prev_geometry = None
        self.prev_gradient = None
        self.prev_energy = None
        self.prev_move_vector = None
        self.ts_mode_idx = None
        self.ts_mode_eigvec = None
        self.consecutive_rejections = 0
        self.trust_radius = self.trust_radius_initial

    # -- pieces -----------------------------------------------------------------------
    def _adaptive_trust(self, gnorm):
        if not self.adaptive_trust_enabled:
            return self.trust_radius
        if gnorm < self.gradient_scaling_threshold:
            r = self.gradient_trust_coupling * gnorm / self.gradient_scaling_threshold * self.trust_radius_max
            r = max(r, self.trust_radius_min)
            r = min(r, self.trust_radius)
            return r
        return self.trust_radius

    def _process_previous_step(self, B_e, geom, pre_move_vector):
        if self.prev_energy is None or len(self.predicted_energy_changes) == 0:
            return True
        actual = B_e - self.prev_energy
        self.actual_energy_changes.append(actual)
        if len(pre_move_vector) > 0:
            psn = np.linalg.norm(np.asarray(pre_move_vector).flatten())
        elif self.prev_move_vector is not None:
            psn = np.linalg.norm(self.prev_move_vector.flatten())
        else:
            psn = 0.0
        H = self.hessian + self.bias_hessian if self.bias_hessian is not None else self.hessian
        H = pj.project_hessian_trrot(H, np.asarray(geom).reshape(-1, 3))
        g = self.prev_gradient.flatten()
        s = self.prev_move_vector.flatten()
        pred_red = -(g @ s + 0.5 * (s @ (H @ s)))
        if abs(pred_red) < 1e-14:
            ratio = 1.0 if abs(actual) < 1e-14 else 0.0
        else:
            ratio = actual / pred_red
            if not np.isfinite(ratio):
                ratio = 0.0
        self.reduction_ratios.append(ratio)
        at_boundary = psn >= self.trust_radius * 0.95
        if self.adapt_trust_radius:
            if ratio < self.eta_2:
                self.trust_radius = max(self.gamma_1 * psn, self.trust_radius_min)
            elif ratio > self.eta_3 and at_boundary:
                self.trust_radius = min(self.gamma_2 * self.trust_radius, self.trust_radius_max)
        if self.step_rejection_enabled and ratio < self.rejection_threshold:
            self.consecutive_rejections += 1
            if self.consecutive_rejections >= self.max_consecutive_rejections:
                self.consecutive_rejections = 0
                return True
            return False
        self.consecutive_rejections = 0
        return True

    def update_hessian(self, geom, grad, pgeom, pgrad):
        s = np.asarray(geom - pgeom).reshape(-1, 1)
        y = np.asarray(grad - pgrad).reshape(-1, 1)
        if np.linalg.norm(s) < 1e-10 or np.linalg.norm(y) < 1e-10:
            return
        delta = hu.delta_hessian(self.hessian_update_method, self.hessian, s, y)
        new = self.hessian + delta
        new = 0.5 * (new + new.T)
        if np.max(np.abs(np.linalg.eigvalsh(new))) > 1e6:
            return
        self.hessian = new

    def _shift(self, H, w, v):
        if not self.hessian_shift_enabled:
            return H, w, False
        n = len(w)
        sw = w.copy()
        applied = False
        if self.saddle_order == 0:
            mn = np.min(w)
            if mn < self.min_positive_eigval:
                sw = w + (self.min_positive_eigval - mn)
                applied = True
        else:
            order = np.argsort(w)
            for i in range(self.saddle_order):
                if w[order[i]] > self.min_negative_eigval:
                    sw[order[i]] = self.min_negative_eigval
                    applied = True
            for i in range(self.saddle_order, n):
                if w[order[i]] < 1e-6:
                    sw[order[i]] = self.min_positive_eigval
                    applied = True
        if applied:
            Hs = v @ np.diag(sw) @ v.T
            return 0.5 * (Hs + Hs.T), sw, True
        return H, w, False

    def _select_modes(self, w, v):
        n = len(w)
        if self.saddle_order == 0:
            return [], list(range(n))
        order = np.argsort(w)
        if not self.mode_following_enabled:
            mx = order[:self.saddle_order].tolist()
        elif self.ts_mode_idx is None or self.ts_mode_eigvec is None:
            self.ts_mode_idx = order[0]
            self.ts_mode_eigvec = v[:, self.ts_mode_idx].copy()
            mx = order[:self.saddle_order].tolist()
        else:
            ov = np.abs(v.T @ self.ts_mode_eigvec)
            self.trace["basis_dependent"] = self._basis_dependent(w, ov)
            best = int(np.argmax(ov))
            if ov[best] > self.overlap_threshold:
                self.ts_mode_idx = best
                self.ts_mode_eigvec = v[:, best].copy()
                mx = [best] + [i for i in order if i != best][:self.saddle_order - 1]
            elif self.eigvec_following:
                sig = np.where(ov > self.mixing_threshold)[0]
                if len(sig) == 0:
                    self.ts_mode_idx = order[0]
                    self.ts_mode_eigvec = v[:, order[0]].copy()
                    mx = order[:self.saddle_order].tolist()
                else:
                    wts = [ov[i] ** 2 * (1.0 if w[i] < 0 else 0.1) for i in sig]
                    best = int(sig[int(np.argmax(wts))])
                    self.ts_mode_idx = best
                    self.ts_mode_eigvec = v[:, best].copy()
                    mx = [best] + [i for i in order if i != best][:self.saddle_order - 1]
            else:
                self.ts_mode_idx = order[0]
                self.ts_mode_eigvec = v[:, order[0]].copy()
                mx = order[:self.saddle_order].tolist()
        mx = [int(i) for i in mx]
        return mx, [i for i in range(n) if i not in mx]

    def _basis_dependent(self, w, ov):
        """Test infrastructure, not in the reference: does the mode choice below hang on the basis LAPACK happened to return inside a
        group of EQUAL shifted eigenvalues?  `_shift` sets every eigenvalue below 1e-6 past the first `saddle_order` ones to
        min_positive_eigval (rsprfo.py:341-347): the six TR modes and every surplus negative mode become exactly degenerate, the second
        eigh (rsprfo.py:797-798) returns an arbitrary basis of that subspace, and the overlaps of its members with the tracked vector
        (rsprfo.py:1003) are decided by rounding noise.  The choice is basis-independent when no member of such a group can
        beat the best overlap outside the groups (if that one passes overlap_threshold) or pass the mixing threshold (if not)."""
        srt = np.argsort(w)
        ws, os_ = w[srt], ov[srt]
        grp = np.zeros(len(w), dtype=bool)
        worst = 0.0
        i = 0
        while i < len(ws):
            j = i
            while j + 1 < len(ws) and abs(ws[j + 1] - ws[i]) <= 1e-10 * max(1.0, abs(ws[i])):
                j += 1
            if j > i:
                grp[i:j + 1] = True
                worst = max(worst, float(np.sqrt(np.sum(os_[i:j + 1] ** 2))))
            i = j + 1
        outside = float(os_[~grp].max()) if (~grp).any() else 0.0
        # outside > overlap_threshold: that mode wins unless a group member can exceed it; otherwise a member matters as soon as it can
        # pass the mixing threshold
        return bool(worst > (outside if outside > self.overlap_threshold else self.mixing_threshold))

    @staticmethod
    def _solve_rfo(lam, gc, alpha, mode):
        """get_augmented_hessian + solve_rfo (rsprfo.py:1097-1168) -> (step, eigval)."""
        n = len(lam)
        Ha = np.zeros((n + 1, n + 1))
        np.fill_diagonal(Ha[:n, :n], lam / alpha)
        Ha[:n, n] = gc / alpha
        Ha[n, :n] = gc / alpha
        ev, vec = np.linalg.eigh(Ha)
        idx = np.argmin(ev) if mode == "min" else np.argmax(ev)
        x = vec[:, idx]
        nu = x[-1]
        if abs(nu) < 1e-12:
            nu = np.sign(nu) * 1e-12 if nu != 0 else 1e-12
        return -x[:-1] / nu, ev[idx]

    def _alpha_step(self, alpha, emax, emin, nmax, nmin, nrm, w, gt, mx, mn, R):
        eps = 1e-12
        d = 0.0
        for idxs, ev, sn in ((mx, emax, nmax), (mn, emin, nmin)):
            if len(idxs) > 0 and sn > eps:
                den = w[idxs] - ev * alpha
                den = np.where(np.abs(den) < eps, np.sign(den) * eps, den)
                sf = 1.0 + sn ** 2 * alpha
                if abs(sf) > eps:
                    d += 2.0 * ev / sf * np.sum(gt[idxs] ** 2 / den ** 3)
        if abs(d) < eps:
            return alpha * 0.5 if nrm > R else 0.0
        return (R ** 2 - nrm ** 2) / d

    def _micro_cycles(self, w, gt, mx, mn, gnorm):
        alpha = self.alpha0
        best = None
        best_diff = float("inf")
        hist = []
        R = self._adaptive_trust(gnorm)

        def scaled_best():
            bn = np.linalg.norm(best)
            b = best * (R / bn) if bn > R else best
            return b, min(bn, R)
        for mu in range(self.max_micro_cycles):
            step = np.zeros_like(gt)
            smax = np.array([])
            smin = np.array([])
            emax = emin = 0.0
            if len(mx) > 0:
                smax, emax = self._solve_rfo(w[mx], gt[mx], alpha, "max")
                step[mx] = smax
            if len(mn) > 0:
                smin, emin = self._solve_rfo(w[mn], gt[mn], alpha, "min")
                step[mn] = smin
            nrm = np.linalg.norm(step)
            hist.append(nrm)
            nmax = np.linalg.norm(smax) if len(mx) > 0 else 0.0
            nmin = np.linalg.norm(smin) if len(mn) > 0 else 0.0
            diff = abs(nrm - R)
            if diff < best_diff:
                best = step.copy()
                best_diff = diff
            if nrm <= R:
                break              # falls through to the "did not converge" exit below (rsprfo.py:594-596, 652-660)
            if nrm > 0 and diff / nrm < self.micro_cycle_rtol:
                return step * (R / nrm), R, True
            if len(hist) >= 3 and all(abs(hist[-i] - hist[-(i + 1)]) < self.micro_cycle_atol for i in range(1, 3)):
                b, bn = scaled_best()
                return b, bn, True
            astep = self._alpha_step(alpha, emax, emin, nmax, nmin, nrm, w, gt, mx, mn, R)
            if abs(astep) > self.alpha_step_max:
                astep *= self.alpha_step_max / abs(astep)
            alpha = float(np.clip(alpha + astep, self.alpha_min, self.alpha_max))
            if alpha == self.alpha_max or alpha == self.alpha_min:
                b, bn = scaled_best()
                return b, bn, True
        b, bn = scaled_best()
        return b, bn, False

    # -- the step ---------------------------------------------------------------------
    def run(self, geom_num_list, B_g, pre_B_g=[], pre_geom=[], B_e=0.0, pre_B_e=0.0, pre_move_vector=[],
            initial_geom_num_list=[], g=[], pre_g=[]):
        self.trace = {}
        if self.Initialization:
            self._reset_state()
            self.Initialization = False
        else:
            self._process_previous_step(B_e, geom_num_list, pre_move_vector)
        if self.hessian is None:
            raise ValueError("Hessian matrix must be set before running optimization")
        if (self.prev_geometry is not None and self.prev_gradient is not None
                and len(pre_B_g) > 0 and len(pre_geom) > 0):
            self.update_hessian(geom_num_list, B_g, pre_geom, pre_B_g)
        gradient = project_gradient_trrot_rsprfo(np.asarray(B_g).flatten(), geom_num_list)
        gnorm = np.linalg.norm(gradient)
        H = self.hessian + self.bias_hessian if self.bias_hessian is not None else self.hessian
        w, v = np.linalg.eigh(H)
        if not np.all(np.isfinite(w)) or not np.all(np.isfinite(v)):
            w = np.ones_like(w)
            v = np.eye(len(w))
        H, w, shifted = self._shift(H, w, v)
        if shifted:
            w, v = np.linalg.eigh(H)
        self.shifted_hessian = H
        mx, mn = self._select_modes(w, v)
        gt = v.T @ gradient
        st, sn, conv = self._micro_cycles(w, gt, mx, mn, gnorm)
        if not np.isfinite(sn) or not np.all(np.isfinite(st)):
            sd = -gt
            sdn = np.linalg.norm(sd)
            tgt = min(sdn, self.trust_radius)
            st = sd * (tgt / sdn) if sdn > 1e-12 else np.zeros_like(gt)
        move = v @ st
        sn = np.linalg.norm(move)
        scale = 1.0
        if self.gradient_scaling_enabled and not (gnorm < 1e-10 or sn < 1e-10):
            ratio = sn / gnorm
            if ratio > 50.0:
                scale = max(50.0 / ratio, self.min_step_scale)
        if scale < 1.0:
            move = move * scale
            sn = sn * scale
        eff = self._adaptive_trust(gnorm)
        if sn > eff * 1.01:
            move = move * (eff / sn)
            sn = eff
        if self.config.get("maxstep") is not None:
            ms = self.config.get("maxstep")
            if move.size % 3 == 0 and move.size > 3:
                longest = np.max(np.sqrt(np.sum(move.reshape(-1, 3) ** 2, axis=1)))
            else:
                longest = np.linalg.norm(move)
            if longest > ms:
                move = move * (ms / longest)
        pred = gradient @ move + 0.5 * (move @ (H @ move))
        self.predicted_energy_changes.append(pred)
        self.trace.update(max_indices=mx, eigvals=w, gt=gt, step_norm=np.linalg.norm(move), shifted=shifted)
        self.prev_geometry = np.copy(geom_num_list)
        self.prev_gradient = np.copy(B_g)
        self.prev_energy = B_e
        self.prev_move_vector = np.copy(move)
        self.iter += 1
        return move.reshape(-1, 1)
