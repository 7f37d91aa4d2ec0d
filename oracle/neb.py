"""Oracle: NEB band force, trust clip, FIRE and the RFO-NEB blend (test infrastructure).

Restates
  multioptpy/MEP/pathopt_bneb_force.py:6-117,155-222   CaluculationBNEB (+ calculate_gamma)
  multioptpy/Coordinate/redundant_coordinate.py:377-440 B-matrix pseudo-inverse helpers
  multioptpy/Optimizer/trust_radius_neb.py:17-98        TR_NEB.TR_calc
  multioptpy/Optimizer/fire_neb.py:38-92                FIREOptimizer.optimize
  multioptpy/Optimizer/rfo_neb.py:43-83,104-208         RFOOptimizer.optimize
The per-image Hessians live in a Python list here; the reference keeps them in
tmp_hessian_{i}.npy files (a lossless float64 round trip), see SURVEY F5.
"""
import numpy as np

from .rsirfo import RSIRFO

BOHR2ANG = 0.52917721067  # neb.py:201


def local_maxima(e):
    """scipy.signal.argrelextrema(e, np.greater): strict interior maxima (order=1, clip)."""
    e = np.asarray(e)
    return [i for i in range(1, len(e) - 1) if e[i] > e[i - 1] and e[i] > e[i + 1]]


def extremum_list_index(energy_list):
    """pathopt_bneb_force.py:6-17 (two zeros appended to each list)."""
    e = np.asarray(energy_list)
    return local_maxima(e) + [0, 0], local_maxima(-e) + [0, 0]


# -- literal (reference-cost) projection ------------------------------------
def _b_matrix(c1, c2):
    """pathopt_bneb_force.py:104-117."""
    N = len(c2)
    B = np.zeros((N, 3 * N))
    for i in range(N):
        nrm = np.linalg.norm(c1[i] - c2[i]) + 1e-15
        B[i, 3 * i:3 * i + 3] = (c2[i] - c1[i]) / nrm
    return B


def _inv_b(B):
    """redundant_coordinate.py:377-400: singular values <= 1e-6 are kept as is."""
    G = B.T @ B
    U, s, VT = np.linalg.svd(G)
    sinv = np.array([1 / v if v > 1e-6 else v for v in s])
    return (VT.T @ np.diag(sinv) @ U.T @ B.T).T


def _project_literal(c1, c2, c3, g, e):
    """pathopt_bneb_force.py:67-101 -> (proj_grad, tangent_grad), both (3N,1)."""
    g = g.reshape(-1, 1)
    if e[0] < e[1] and e[1] < e[2]:
        B = _b_matrix(c2, c3)
        t = B.T @ (-(_inv_b(B) @ g))
    elif e[0] > e[1] and e[1] > e[2]:
        B = _b_matrix(c1, c2)
        t = B.T @ (-(_inv_b(B) @ g))
    else:
        Bp = _b_matrix(c2, c3)
        Bm = _b_matrix(c1, c2)
        ip = _inv_b(Bp) @ g
        im = _inv_b(Bm) @ g
        mx = max(abs(e[2] - e[1]), abs(e[1] - e[0]))
        mn = min(abs(e[2] - e[1]), abs(e[1] - e[0]))
        a = mx / (mx + mn + 1e-8)
        b = mn / (mx + mn + 1e-8)
        if e[0] < e[2]:
            t = Bp.T @ (-a * ip) + Bm.T @ (-b * im)
        else:
            t = Bp.T @ (-b * ip) + Bm.T @ (-a * im)
    return g + t, t


# -- closed form (what the B-matrix algebra reduces to per atom) --------------
def _proj_atomwise(ca, cb, g):
    """-(g_a . u_a) u_a * w with u_a = (cb-ca)/(|cb-ca|+1e-15), w = pinv weight of |u_a|^2."""
    d = cb - ca
    u = d / (np.linalg.norm(d, axis=1) + 1e-15)[:, None]
    s = np.sum(u * u, axis=1)
    w = np.where(s > 1e-6, 1.0 / np.where(s > 1e-6, s, 1.0), s)
    return -(np.sum(g * u, axis=1) * w)[:, None] * u


def _project_closed(c1, c2, c3, g, e):
    g = g.reshape(-1, 3)
    if e[0] < e[1] and e[1] < e[2]:
        t = _proj_atomwise(c2, c3, g)
    elif e[0] > e[1] and e[1] > e[2]:
        t = _proj_atomwise(c1, c2, g)
    else:
        mx = max(abs(e[2] - e[1]), abs(e[1] - e[0]))
        mn = min(abs(e[2] - e[1]), abs(e[1] - e[0]))
        a = mx / (mx + mn + 1e-8)
        b = mn / (mx + mn + 1e-8)
        tp = _proj_atomwise(c2, c3, g)
        tm = _proj_atomwise(c1, c2, g)
        t = a * tp + b * tm if e[0] < e[2] else b * tp + a * tm
    return (g + t).reshape(-1, 1), t.reshape(-1, 1)


class BNEB:
    """CaluculationBNEB (pathopt_bneb_force.py:21-222)."""

    def __init__(self, APPLY_CI_NEB=99999, literal=False):
        self.APPLY_CI_NEB = APPLY_CI_NEB
        self._project = _project_literal if literal else _project_closed
        self.tau_list = []

    def calc_force(self, X, E, G, optimize_num, element_list=None):
        I = len(E)
        self.tau_list = []
        maxima, _ = extremum_list_index(E)
        F = []
        for i in range(I):
            if i == 0 or i == I - 1:
                F.append(-1 * np.array(G[i], dtype=float))
                self.tau_list.append(np.zeros_like(G[i]).reshape(-1, 3))
                continue
            force, tg = self._project(X[i - 1], X[i], X[i + 1], np.array(G[i], dtype=float), E[i - 1:i + 2])
            if optimize_num > self.APPLY_CI_NEB and (i + 1 in maxima or i - 1 in maxima) and (i != 1 and i != I - 2):
                force = force * 0.001
            elif optimize_num > self.APPLY_CI_NEB and (i in maxima) and (i != 1 or i != I - 2):
                force = -2.0 * tg
            F.append(-1 * force.reshape(-1, 3))
            self.tau_list.append(tg.reshape(-1, 3))
        return np.array(F, dtype=float)

    def get_tau(self, i):
        return self.tau_list[i]

    @staticmethod
    def calculate_gamma(q, E, g, tangent):
        """pathopt_bneb_force.py:161-222."""
        dp = np.linalg.norm(q[1] - q[0])
        dn = np.linalg.norm(q[2] - q[1])
        if dp < 1e-6 or dn < 1e-6:
            return 0.0
        sp, sc, sn = -dp, 0.0, dn
        gp = np.dot(g[0].flatten(), ((q[1] - q[0]) / dp).flatten())
        gc = np.dot(g[1].flatten(), tangent.flatten())
        gn = np.dot(g[2].flatten(), ((q[2] - q[1]) / dn).flatten())
        A = np.array([[1, s, s ** 2, s ** 3, s ** 4, s ** 5] for s in (sp, sc, sn)]
                     + [[0, 1, 2 * s, 3 * s ** 2, 4 * s ** 3, 5 * s ** 4] for s in (sp, sc, sn)], dtype=float)
        b = np.array([E[0], E[1], E[2], gp, gc, gn], dtype=float)
        try:
            return 2.0 * np.linalg.solve(A, b)[2]
        except np.linalg.LinAlgError:
            return 0.0


def tr_calc(X, F, delta, fix_init=False, fix_end=False):
    """TR_NEB.TR_calc, trust_radius_neb.py:17-98 (CSV side effects omitted)."""
    def edge(d, fixed):
        if fixed:
            return d * 0.0
        nrm = np.linalg.norm(d)
        if nrm < 1e-15:
            return d * 0.0
        return min(0.5, nrm) * d / nrm
    out = [edge(delta[0], fix_init)]
    for i in range(1, len(delta) - 1):
        r1 = np.linalg.norm(X[i] - X[i - 1]) / 2.0
        r2 = np.linalg.norm(X[i] - X[i + 1]) / 2.0
        v1 = (X[i - 1] - X[i]) / (np.linalg.norm(X[i - 1] - X[i]) + 1e-15)
        v2 = (X[i + 1] - X[i]) / (np.linalg.norm(X[i + 1] - X[i]) + 1e-15)
        dn = np.linalg.norm(delta[i])
        nd = delta[i] / dn
        c1 = np.sum(v1 * nd)
        c2 = np.sum(v2 * nd)
        fc = np.sum(F[i] * delta[i]) / (np.linalg.norm(F[i]) * dn)
        if fc >= 0:
            if (c1 > 0 and c2 < 0) or (c1 < 0 and c2 > 0):
                if dn > r1 and c1 > 0:
                    out.append(delta[i] * r1 / dn)
                elif dn > r2 and c2 > 0:
                    out.append(delta[i] * r2 / dn)
                else:
                    out.append(delta[i])
            elif c1 < 0 and c2 < 0:
                out.append(delta[i])
            else:
                if dn > r1:
                    out.append(delta[i] * r1 / dn)
                elif dn > r2:
                    out.append(delta[i] * r2 / dn)
                else:
                    out.append(delta[i])
        else:
            out.append(delta[i] * 0.0)
    out.append(edge(delta[-1], fix_end))
    return out


class FIRE:
    """FIREOptimizer, fire_neb.py:14-92; parameters from neb.py:213-222."""

    def __init__(self, fix_init=False, fix_end=False):
        self.dt = 0.5
        self.a = 0.10
        self.n_reset = 0
        self.N_acc = 5
        self.f_inc = 1.10
        self.f_dec = 0.5
        self.a_start = 0.1
        self.dt_max = 1.0
        self.fix = (fix_init, fix_end)

    def optimize(self, X, F, pre_vel, optimize_num, vel, global_dot=None):
        """global_dot: the band-wide sum v_prev . F handed in from outside (image-sharded tests: assembled from per-rank
        partial sums); None = computed here over all images as the reference does."""
        X = np.asarray(X, dtype=float)
        F = np.asarray(F, dtype=float)
        vel = np.asarray(vel, dtype=float)
        vneb = np.empty_like(vel)
        for i in range(vel.shape[0]):
            for a in range(F.shape[1]):
                fn = np.linalg.norm(F[i][a])
                vn = np.linalg.norm(vel[i][a])
                if fn > 1e-10:
                    vneb[i][a] = (1.0 - self.a) * vel[i][a] + self.a * (vn / fn) * F[i][a]
                else:
                    vneb[i][a] = vel[i][a]
        have_prev = optimize_num != 0 and len(pre_vel) > 1
        dotp = 0.0
        if have_prev:
            if global_dot is not None:
                dotp = global_dot
            else:
                dotp = np.sum([np.dot(pre_vel[i][a], F[i][a].T) for i in range(F.shape[0]) for a in range(F.shape[1])])
        if optimize_num > 0 and dotp > 0 and len(pre_vel) > 1:
            if self.n_reset > self.N_acc:
                self.dt = min(self.dt * self.f_inc, self.dt_max)
                self.a *= self.f_inc
            self.n_reset += 1
        else:
            vneb = vneb * 0
            self.a = self.a_start
            self.dt *= self.f_dec
            self.n_reset = 0
        v = vneb + self.dt * F
        if have_prev:
            delta = self.dt * (v + np.asarray(pre_vel, dtype=float))
        else:
            delta = self.dt * v
        mv = tr_calc(X, F, delta, *self.fix)
        return (X + np.array(mv)) * BOHR2ANG


def steepest_descent(X, F, fix_init=False, fix_end=False):
    """SteepestDescentOptimizer.optimize, gradientdescent_neb.py:22-51."""
    X = np.asarray(X, dtype=float)
    F = np.asarray(F, dtype=float)
    delta = [0.5 * F[i] for i in range(len(F))]
    mv = [delta[0] * 0.0 if fix_init else delta[0]]
    for i in range(1, len(delta) - 1):
        r1 = np.linalg.norm(X[i] - X[i - 1]) / 2.0
        r2 = np.linalg.norm(X[i] - X[i + 1]) / 2.0
        nrm = np.linalg.norm(delta[i])
        if nrm > r1:
            mv.append(delta[i] * r1 / nrm)
        elif nrm > r2:
            mv.append(delta[i] * r2 / nrm)
        else:
            mv.append(delta[i])
    mv.append(delta[-1] * 0.0 if fix_end else delta[-1])
    return (X + np.array(mv)) * BOHR2ANG


def limit_step(mv, is_end):
    """rfo_neb.py:76-83."""
    nrm = np.linalg.norm(mv)
    lim = 0.2 if is_end else 0.1
    if nrm > 1e-8:
        return mv / nrm * min(lim, nrm)
    return mv


class RFONEB:
    """RFOOptimizer.optimize, rfo_neb.py:86-208, with the Hessians kept in memory."""

    def __init__(self, hessians, ratio=0.5, fix_init=False, fix_end=False, apply_ts_opt=True, literal=False):
        self.hessians = [None if h is None else np.array(h, dtype=float) for h in hessians]
        self.ratio = ratio
        self.fix = (fix_init, fix_end)
        self.apply_ts_opt = apply_ts_opt
        self.opts = []
        self.literal = literal

    def optimize(self, X, G, preX, preG, optimize_num, E, pre_vel, vel, force_calc, fire_dot=None):
        X = np.asarray(X, dtype=float)
        I, N = X.shape[0], X.shape[1]
        maxima = local_maxima(E)
        if optimize_num == 0:
            self.opts = []
            for i in range(I):
                if i == 0 or i == I - 1:
                    o = RSIRFO(method="rsirfo_block_fsb", saddle_order=0, trust_radius=0.5)
                else:
                    o = RSIRFO(method="rsirfo_block_bofill", saddle_order=0, trust_radius=0.2)
                    if not (i in maxima and self.apply_ts_opt):
                        o.switch_NEB_mode()
                self.opts.append(o)
        projF = force_calc.calc_force(X, E, G, optimize_num, None)
        deltas = []
        for i in range(I):
            H = self.hessians[i] if self.hessians[i] is not None else np.eye(3 * N)
            if not (i == 0 or i == I - 1):
                t = force_calc.get_tau(i)
                lo, hi = max(0, i - 1), min(I, i + 2)
                gamma = force_calc.calculate_gamma(X[lo:hi], E[lo:hi], G[lo:hi], t)
                H = H + gamma * np.outer(t, t)
            o = self.opts[i]
            o.set_hessian(H)
            o.set_bias_hessian(np.zeros((3 * N, 3 * N)))
            if optimize_num == 0:
                pg = px = None
            else:
                pg = preG[i].reshape(-1, 1)
                px = preX[i].reshape(-1, 1)
            bg = G[i].reshape(-1, 1)
            mv = o.run(X[i].reshape(-1, 1), bg, pg, px, 0.0, 0.0, [], [], bg, pg)
            deltas.append(limit_step(mv, i == 0 or i == I - 1).reshape(-1, 3))
            self.hessians[i] = o.get_hessian()
            o.set_hessian(None)
            o.set_bias_hessian(None)
        rfo_mv = tr_calc(X, G, deltas, *self.fix)
        fire_new = FIRE(*self.fix).optimize(X, projF, pre_vel, optimize_num, vel, global_dot=fire_dot) / BOHR2ANG
        fire_mv = fire_new - X
        out = []
        for i in range(I):
            if i == 0 or i == I - 1:
                out.append(-1.0 * rfo_mv[i])
            else:
                out.append((1.0 - self.ratio) * fire_mv[i] - self.ratio * rfo_mv[i])
        return (X + np.array(out, dtype=float)) * BOHR2ANG


# ---------------------------------------------------------------------------------------------
# spring variants
# ---------------------------------------------------------------------------------------------
class BNEB3(BNEB):
    """CaluculationBNEB3 (pathopt_bneb_force.py:305-356): BNEB projection (no CI switches) + next-nearest springs."""

    def __init__(self, APPLY_CI_NEB=99999, literal=False):
        super().__init__(APPLY_CI_NEB, literal)
        self.k = 0.05

    def calc_force(self, X, E, G, optimize_num, element_list=None):
        I = len(E)
        F = []
        nrm = np.linalg.norm
        for i in range(I):
            if i == 0 or i == I - 1:
                F.append(-1 * np.array(G[i], dtype=float))
                continue
            force, _ = self._project(X[i - 1], X[i], X[i + 1], np.array(G[i], dtype=float), E[i - 1:i + 2])
            if 1 < i < I - 2:
                c0, c1, c2, c3, c4 = X[i - 2], X[i - 1], X[i], X[i + 1], X[i + 2]
                f1 = self.k * (nrm(c1 - c2) - nrm(c0 - c1)) * (c1 - c2) / (nrm(c1 - c2) + 1e-15)
                f2 = self.k * (nrm(c2 - c3) - nrm(c1 - c2)) * (-1 * (c1 - c2) / (nrm(c1 - c2) + 1e-15) - (c2 - c3) / (nrm(c2 - c3) + 1e-15))
                f3 = self.k * (nrm(c3 - c4) - nrm(c2 - c3)) * (c3 - c4) / (nrm(c3 - c4) + 1e-15)
                spring = f1 + f2 + f3
            else:
                spring = 0.0 * force
            F.append(-1 * force.reshape(-1, 3) - 1 * spring.reshape(-1, 3))
        return np.array(F, dtype=float)


class EWBNEB(BNEB):
    """CaluculationEWBNEB (pathopt_ewbneb_force.py:21-117): BNEB projection + energy-weighted springs along the
    whole-image unit vectors to the neighbours."""

    upper, lower = 0.005, 0.0001

    def calc_force(self, X, E, G, optimize_num, element_list=None):
        I = len(E)
        maxima, _ = extremum_list_index(E)
        max_ene = max(E)
        ref = E[-1] if E[0] > E[-1] else E[0]
        ks = []
        for i in range(I):
            if i == 0 or i == I - 1:
                ks.append(0.0)
                continue
            den = max_ene - ref
            if den < 1e-8:
                den = 1e-8
            alpha = (E[i] - ref) / den
            if E[i] < ref:
                ks.append(self.lower)
            elif E[i] > ref:
                ks.append(self.upper * (1.0 - alpha) + alpha * self.lower)
            else:
                ks.append(0.0)
        F = []
        for i in range(I):
            if i == 0 or i == I - 1:
                F.append(-1 * np.array(G[i], dtype=float))
                continue
            force, tg = self._project(X[i - 1], X[i], X[i + 1], np.array(G[i], dtype=float), E[i - 1:i + 2])
            if optimize_num > self.APPLY_CI_NEB and (i + 1 in maxima or i - 1 in maxima) and (i != 1 and i != I - 2):
                force = force * 0.001
            elif optimize_num > self.APPLY_CI_NEB and (i in maxima) and (i != 1 or i != I - 2):
                force = -2.0 * tg
            fwd = X[i + 1] - X[i]
            bwd = X[i] - X[i - 1]
            nf, nb = np.linalg.norm(fwd), np.linalg.norm(bwd)
            uf = fwd / nf if nf > 1e-8 else np.zeros_like(fwd)
            ub = bwd / nb if nb > 1e-8 else np.zeros_like(bwd)
            spring = (ks[i] * uf - ks[i - 1] * ub).reshape(-1, 1)
            F.append(-1 * (force + spring).reshape(-1, 3))
        return np.array(F, dtype=float)


class DNEB:
    """CaluculationDNEB (pathopt_dneb_force.py:20-130), including its quirks."""

    def __init__(self, APPLY_CI_NEB=99999):
        self.k = 0.01
        self.APPLY_CI_NEB = APPLY_CI_NEB
        self.k_ci = 0.01

    @staticmethod
    def _unit(v):
        n = np.linalg.norm(v)
        return v / n if n != 0 else np.zeros_like(v)

    def calc_force(self, X, E, G, optimize_num, element_list=None):
        X = np.asarray(X, dtype=float)
        G = np.asarray(G, dtype=float)
        I, N = X.shape[0], X.shape[1]
        maxlist, _ = extremum_list_index(E)
        out = [-1 * G[0]]
        for i in range(1, I - 1):
            dmax = max(E[i + 1] - E[i], E[i - 1] - E[i])
            dmin = min(E[i + 1] - E[i], E[i - 1] - E[i])
            tau = np.zeros((N, 3))
            if E[i - 1] < E[i] and E[i] < E[i + 1]:
                for t in range(N):
                    tau[t] = self._unit(X[i + 1][t] - X[i][t])
            elif E[i - 1] > E[i] and E[i] > E[i + 1]:
                for t in range(N):
                    tau[t] = self._unit(X[i][t] - X[i - 1][t])
            else:
                for t in range(N):
                    tm = self._unit(X[i][t] - X[i - 1][t])
                    tp = self._unit(X[i + 1][t] - X[i][t])
                    if E[i - 1] > E[i + 1]:
                        v = tp * dmin + tm * dmax
                        # where = np.linalg.norm(v != 0): true when any component is non-zero
                        tau[t] = v / np.linalg.norm(v) if np.any(v != 0) else np.zeros(3)
                    else:
                        v = tp * dmax + tm * dmin
                        nn = np.linalg.norm(tp * dmin + tm * dmax)      # the OTHER combination normalises
                        tau[t] = v / nn if nn != 0 else np.zeros(3)
            if E[i] == E[maxlist[0]] and self.APPLY_CI_NEB < optimize_num:
                total = np.array([-self.k_ci * (G[i][f] - 2.0 * np.dot(G[i][f], tau[f]) * tau[f]) for f in range(N)])
                del maxlist[0]
            else:
                total = np.zeros((N, 3))
                for f in range(N):
                    c = self.k * np.linalg.norm(X[i + 1][f] - X[i][f]) + (-1.0) * self.k * np.linalg.norm(X[i][f] - X[i - 1][f])
                    fpar = c * tau[f]
                    fperp = G[i][f] - np.dot(G[i][f], tau[f]) * tau[f]
                    w = c - (c * tau[f]) * tau[f]
                    nw = np.linalg.norm(w)
                    ratio = np.ones(3) if nw != 0 else np.zeros(3)
                    sw = (2.0 / np.pi) * np.arctan(ratio) * (w - np.dot(w, fperp) * fperp)
                    total[f] = -fperp - fpar - sw
            mm = np.nanmean(np.nanmean(total))
            if mm > 10:
                total = total / mm
            out.append(total)
        out.append(-1 * G[-1])
        return np.array(out, dtype=float)
