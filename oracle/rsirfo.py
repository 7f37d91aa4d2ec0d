"""Oracle: RS-I-RFO optimizer step (test infrastructure, see oracle/__init__.py).

Restates multioptpy/Optimizer/rsirfo.py (RSIRFO, v1.21.2) in plain NumPy:
  run                                   rsirfo.py:285-490
  eigen-decomposition + level shift     rsirfo.py:492-657
  trust radius (adaptive/conventional)  rsirfo.py:660-887
  get_rs_step / alpha search            rsirfo.py:924-1313
  Hessian update dispatch               rsirfo.py:1316-1372
  secular equation                      rsirfo.py:1374-1575, 1688-1715
Diagnostics printing is omitted; everything that changes state or the step is kept.
"""
import numpy as np

from . import hessian_update as hu
from . import projection as pj
from .brent import brentq, BrentError


def _safe(den, floor):
    out = np.where(np.abs(den) < floor, np.sign(den) * floor, den)
    out[out == 0] = floor
    return out


def secular_f(lmd, ep, gsq):
    return lmd + np.sum(gsq / _safe(ep - lmd, 1e-30))


def secular_fp(lmd, ep, gsq):
    return 1.0 + np.sum(gsq / _safe(ep - lmd, 1e-30) ** 2)


def solve_secular_safeguarded(ep, gsq, pole, guess, stats=None):
    """rsirfo.py:1374-1503."""
    b = pole
    a = guess
    fa = secular_f(a, ep, gsq)
    gnorm = np.sqrt(np.sum(gsq))
    limit = 10
    while fa > 0 and limit > 0:
        a = a - max(gnorm, abs(a) * 0.1, 1e-8)
        fa = secular_f(a, ep, gsq)
        limit -= 1
    if fa > 0:
        return guess
    lk = guess
    if lk <= a or lk >= b:
        lk = (a + b) / 2.0
    tol = 1e-10 * abs(pole) + 1e-12
    for it in range(250):
        fl = secular_f(lk, ep, gsq)
        if abs(fl) < tol:
            if stats is not None:
                stats.append(it + 1)
            return lk
        fpl = secular_fp(lk, ep, gsq)
        dn = -fl / fpl if abs(fpl) > 1e-20 else 0.0
        ln = lk + dn
        lb = (a + b) / 2.0
        nxt = ln if (dn != 0.0 and ln > a and ln < b) else lb
        if fl > 0:
            b = lk
        else:
            a = lk
        lk = nxt
        if abs(b - a) < tol:
            return (a + b) / 2.0
    return (a + b) / 2.0


def solve_secular(eigvals, gcomp, alpha, stats=None):
    """rsirfo.py:1505-1575 (primary path)."""
    ep = eigvals / alpha
    gp = gcomp / alpha
    gsq = gp ** 2
    pole = None
    gnorm_sq = 0.0
    for i in range(len(ep)):
        gnorm_sq += gsq[i]
        if pole is None and gsq[i] > 1e-20:
            pole = ep[i]
    if pole is None:
        return ep[0]
    guess = 0.5 * (pole - np.sqrt(max(0.0, pole ** 2 + 4 * gnorm_sq)))
    return solve_secular_safeguarded(ep, gsq, pole, guess, stats)


def solve_rfo(eigvals, gcomp, alpha, stats=None):
    """rsirfo.py:1688-1715 -> (step, lambda_aug)."""
    lam = solve_secular(eigvals, gcomp, alpha, stats)
    den = _safe(eigvals / alpha - lam, 1e-20)
    return -(gcomp / alpha) / den, lam


def step_derivative(alpha, eigvals, gcomp, lam):
    """rsirfo.py:1250-1313: d||s||^2/dalpha with lambda_aug frozen."""
    d = eigvals - lam * alpha
    small = np.abs(d) < 1e-8
    if np.any(small):
        d = d.copy()
        d[small] = np.sign(d[small]) * np.maximum(1e-8, np.abs(d[small]))
    num = gcomp ** 2
    den3 = d ** 3
    valid = np.abs(den3) > 1e-10
    if not np.any(valid):
        return 1e-8
    terms = np.zeros_like(num)
    terms[valid] = num[valid] / den3[valid]
    big = np.abs(terms) > 1e20
    if np.any(big):
        terms[big] = np.sign(terms[big]) * 1e20
    res = 2.0 * lam * np.sum(terms)
    if not np.isfinite(res) or abs(res) > 1e20:
        res = np.sign(res) * 1e20 if res != 0 else 1e-8
    return res


class RSIRFO:
    def __init__(self, **config):
        g = config.get
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
        self.use_adaptive_trust_radius = g("use_adaptive_trust_radius", True)
        self.adaptive_trust_gradient_norm_threshold = g("adaptive_trust_gradient_norm_threshold", 1e-2)
        self.max_curvature_factor = g("max_curvature_factor", 2.5)
        self.negative_curvature_safety = g("negative_curvature_safety", 0.8)
        self.use_level_shift = g("use_level_shift", False)
        self.level_shift_value = g("level_shift_value", 1e-5)
        self.auto_level_shift = g("auto_level_shift", True)
        self.condition_number_threshold = g("condition_number_threshold", 1e8)
        self.level_shift_applied = False
        self.Initialization = True
        self.hessian = None
        self.bias_hessian = None
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
        # diagnostics captured for the golden comparison
        self.trace = {}

    # -- plugin surface ------------------------------------------------------
    def switch_NEB_mode(self):
        self.NEB_mode = not self.NEB_mode

    def set_hessian(self, h):
        self.hessian = h

    def set_bias_hessian(self, h):
        self.bias_hessian = h

    def get_hessian(self):
        return self.hessian

    def get_bias_hessian(self):
        return self.bias_hessian

    # -- pieces --------------------------------------------------------------
    def conditioning(self, eigvals):
        """rsirfo.py:492-551 -> (kappa or None, ill)."""
        if len(eigvals) < 2:
            return None, False
        nz = eigvals[np.abs(eigvals) > 1e-10]
        if len(nz) < 2:
            return None, True
        mx = np.max(np.abs(nz))
        mn = np.min(np.abs(nz))
        if mn < 1e-15:
            return None, True
        k = mx / mn
        return k, k > self.condition_number_threshold

    def eigh_with_shift(self, H):
        """rsirfo.py:553-657."""
        self.level_shift_applied = False
        apply = False
        w = v = None
        if self.use_level_shift:
            apply = True
        elif self.auto_level_shift:
            w, v = np.linalg.eigh(H)
            _, ill = self.conditioning(w)
            apply = bool(ill)
        if apply:
            sh = self.level_shift_value
            ws, v = np.linalg.eigh(H + sh * np.eye(H.shape[0]))
            w = ws - sh
            self.level_shift_applied = True
        elif w is None:
            w, v = np.linalg.eigh(H)
        return w, v

    def adjust_trust_radius_adaptive(self, actual, predicted, min_eig):
        """rsirfo.py:660-803."""
        if abs(predicted) < 1e-10:
            return
        ratio = actual / predicted
        ae = abs(min_eig)
        if ae > 1e-6:
            cf = min(self.max_curvature_factor, 1.0 / max(ae, 0.1))
        else:
            cf = 1.5
        if self.saddle_order > 0 and min_eig < -1e-6:
            cf *= self.negative_curvature_safety
        if ratio > 0.75:
            inc = min(1.5 * cf, self.max_curvature_factor)
            self.trust_radius = min(self.trust_radius * inc, self.trust_radius_max)
        elif ratio > 0.5:
            inc = min(1.1 * cf, 1.5)
            self.trust_radius = min(self.trust_radius * inc, self.trust_radius_max)
        elif ratio > 0.25:
            if cf > 1.2:
                self.trust_radius = min(self.trust_radius * 1.05, self.trust_radius_max)
        elif ratio > 0.1:
            self.trust_radius = max(self.trust_radius * 0.5, self.trust_radius_min)
        else:
            self.trust_radius = max(self.trust_radius * 0.25, self.trust_radius_min)
        self.trust_radius = float(np.clip(self.trust_radius, self.trust_radius_min, self.trust_radius_max))

    def adjust_trust_radius(self, actual, predicted, min_eig=None, grad_norm=None):
        """rsirfo.py:805-887."""
        if self.use_adaptive_trust_radius and min_eig is not None:
            if grad_norm is not None:
                if grad_norm < self.adaptive_trust_gradient_norm_threshold:
                    self.adjust_trust_radius_adaptive(actual, predicted, min_eig)
                    return
            else:
                self.adjust_trust_radius_adaptive(actual, predicted, min_eig)
                return
        if abs(predicted) < 1e-10:
            return
        ratio = actual / predicted
        if ratio > self.good_step_threshold:
            self.trust_radius = min(self.trust_radius * self.trust_radius_increase_factor, self.trust_radius_max)
        elif ratio < self.poor_step_threshold:
            self.trust_radius = max(self.trust_radius * self.trust_radius_decrease_factor, self.trust_radius_min)

    def update_hessian(self, geom, grad, pgeom, pgrad):
        """rsirfo.py:1316-1372."""
        s = np.asarray(geom - pgeom).reshape(-1, 1)
        y = np.asarray(grad - pgrad).reshape(-1, 1)
        if np.linalg.norm(s) < 1e-10 or np.linalg.norm(y) < 1e-10:
            return
        if (s.T @ y)[0, 0] <= 0:
            return
        delta = hu.delta_hessian(self.hessian_update_method, self.hessian, s, y)
        self.hessian += delta                                  # in place: aliases the caller's array (F7)
        self.hessian = 0.5 * (self.hessian + self.hessian.T)   # rebinding

    def alpha_search(self, eigvals, gt, alpha_init):
        """compute_rsprfo_step, rsirfo.py:986-1248 -> (step, norm, alpha)."""
        R = self.trust_radius
        tr = self.trace
        alpha = alpha_init
        try:
            s_lo, _ = solve_rfo(eigvals, gt, 1e-6)
            obj_lo = np.linalg.norm(s_lo) ** 2 - R ** 2
            s_hi, _ = solve_rfo(eigvals, gt, self.alpha_max)
            obj_hi = np.linalg.norm(s_hi) ** 2 - R ** 2
            if obj_lo * obj_hi < 0:
                bt = []
                tr["brent"] = bt

                def obj(a):
                    st, _ = solve_rfo(eigvals, gt, a)
                    return np.dot(st, st) - R ** 2
                ab = brentq(obj, 1e-6, self.alpha_max, xtol=1e-6, rtol=1e-6, maxiter=50, trace=bt)
                step, _ = solve_rfo(eigvals, gt, ab)
                nrm = np.linalg.norm(step)
                if abs(nrm - R) < self.step_norm_tolerance:
                    tr["branch"] = "brent"
                    return step, nrm, ab
                alpha = ab
        except BrentError:
            alpha = alpha_init
        tr["branch"] = "newton"
        hist = np.zeros(self.max_micro_cycles)
        hc = 0
        best = None
        best_diff = float("inf")
        a_left = a_right = None
        step = None
        nrm = None
        for mu in range(self.max_micro_cycles):
            step, lam = solve_rfo(eigvals, gt, alpha)
            nrm = np.linalg.norm(step)
            diff = abs(nrm - R)
            if diff < best_diff:
                best = step.copy()
                best_diff = diff
            objv = nrm ** 2 - R ** 2
            if objv < 0 and (a_left is None or alpha > a_left):
                a_left = alpha
            elif objv > 0 and (a_right is None or alpha < a_right):
                a_right = alpha
            if abs(objv) < 1e-8 or diff < self.step_norm_tolerance:
                best = step
                break
            if hc < self.max_micro_cycles:
                hist[hc] = nrm
                hc += 1
            d = step_derivative(alpha, eigvals, gt, lam)
            if abs(d) < 1e-10:
                if a_left is not None and a_right is not None:
                    anew = (a_left + a_right) / 2
                elif objv > 0:
                    anew = max(alpha / 2, 1e-6)
                else:
                    anew = min(alpha * 2, self.alpha_max)
            else:
                astep = float(np.clip(-objv / d, -self.alpha_step_max, self.alpha_step_max))
                anew = alpha + astep
                if a_left is not None and a_right is not None:
                    anew = max(min(anew, a_right * 0.99), a_left * 1.01)
            alpha = min(max(anew, 1e-6), self.alpha_max)
            if alpha == self.alpha_max or alpha == 1e-6:
                break
            if hc >= 3:
                i = hc - 1
                if abs(hist[i] - hist[i - 1]) < 1e-6 and abs(hist[i - 1] - hist[i - 2]) < 1e-6:
                    break
        else:
            bn = np.linalg.norm(best)
            if abs(bn - R) < self.step_norm_tolerance * 1.1:
                step, nrm = best, bn
            else:
                step = -gt
                nrm = np.linalg.norm(step)
                step = step / nrm * R if nrm > 1e-10 else np.zeros_like(gt)
                nrm = R
        return step, nrm, alpha

    def get_rs_step(self, eigvals, eigvecs, gradient):
        """rsirfo.py:924-984."""
        gt = eigvecs.T @ gradient
        self.trace["gt"] = gt
        st = []
        s0, lam0 = solve_rfo(eigvals, gt, self.alpha0, st)
        self.trace["lambda_aug0"] = lam0
        self.trace["secular_iters0"] = st[0] if st else -1
        n0 = np.linalg.norm(s0)
        self.trace["initial_norm"] = n0
        if n0 <= self.trust_radius:
            self.trace["branch"] = "within"
            self.trace["alpha"] = self.alpha0
            return eigvecs @ s0
        step, _, alpha = self.alpha_search(eigvals, gt, self.alpha0)
        self.trace["alpha"] = alpha
        return eigvecs @ step

    # -- the step ------------------------------------------------------------
    def run(self, geom_num_list, B_g, pre_B_g=[], pre_geom=[], B_e=0.0, pre_B_e=0.0,
            pre_move_vector=[], initial_geom_num_list=[], g=[], pre_g=[]):
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
            raise ValueError("Hessian matrix must be set before running optimization")
        if (self.prev_geometry is not None and self.prev_gradient is not None
                and len(pre_g) > 0 and len(pre_geom) > 0):
            self.update_hessian(geom_num_list, g, pre_geom, pre_g)
        gnorm = np.linalg.norm(B_g)
        if gnorm < self.gradient_norm_threshold:
            self.converged = True
        if self.actual_energy_changes and abs(self.actual_energy_changes[-1]) < self.energy_change_threshold:
            self.converged = True
        gradient = pj.project_gradient_trrot(np.asarray(B_g).ravel(), geom_num_list)
        coords = np.asarray(geom_num_list).reshape(-1, 3)
        H0 = self.hessian if self.bias_hessian is None else self.hessian + self.bias_hessian
        H = pj.project_hessian_trrot(H0, coords)
        H = 0.5 * (H + H.T)
        w, v = self.eigh_with_shift(H)
        if not np.all(np.isfinite(w)) or not np.all(np.isfinite(v)):
            w = np.ones_like(w)
            v = np.eye(len(w))
        if self.prev_energy is not None:
            actual = B_e - self.prev_energy
            if len(self.actual_energy_changes) >= 3:
                self.actual_energy_changes.pop(0)
            self.actual_energy_changes.append(actual)
            if self.predicted_energy_changes:
                self.adjust_trust_radius(actual, self.predicted_energy_changes[-1],
                                         w[0] if len(w) > 0 else None, gnorm)
        n = gradient.size
        P = np.eye(n)
        k = 0
        i = 0
        while k < len(self.roots):
            if abs(w[i]) > 1e-10:
                t = v[:, i]
                P -= (1.0 if self.NEB_mode else 2.0) * np.outer(t, t)
                k += 1
            i += 1
        Hs = P @ H
        Hs = 0.5 * (Hs + Hs.T)
        gs = P @ gradient
        ws, vs = self.eigh_with_shift(Hs)
        if not np.all(np.isfinite(ws)) or not np.all(np.isfinite(vs)):
            ws = np.ones_like(ws)
            vs = np.eye(len(ws))
        keep = ~(np.abs(ws) < self.small_eigval_thresh)
        ws = ws[keep]
        vs = vs[:, keep]
        self.trace["eigvals"] = w
        self.trace["eigvals_star"] = ws
        self.prev_eigvec_size = vs.shape[1]
        move = self.get_rs_step(ws, vs, gs)
        if not np.all(np.isfinite(move)):
            move = -gradient
            nrm = np.linalg.norm(move)
            if nrm > self.trust_radius:
                move *= self.trust_radius / nrm
        pred = gradient @ move + 0.5 * (move @ H @ move)
        if len(self.predicted_energy_changes) >= 3:
            self.predicted_energy_changes.pop(0)
        self.predicted_energy_changes.append(pred)
        self.prev_geometry = geom_num_list
        self.prev_gradient = B_g
        self.prev_energy = B_e
        self.iteration += 1
        return -1 * move.reshape(-1, 1)
