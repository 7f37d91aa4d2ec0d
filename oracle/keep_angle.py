"""TEST INFRASTRUCTURE (oracle): NumPy / pure-Python restatement of the reference's angle and dihedral-angle restraints
(multioptpy/Potential/keep_angle_potential.py:58-224, keep_dihedral_angle_potential.py:60-154) with derivatives from a small hyper-dual number
class (the reference uses torch autograd).  Only tests/, smoke() and bench.py's cpu_baseline leg may import this package.  Pinned by G14b."""
import math

import numpy as np


class HD:
    """f + a e1 + b e2 + c e1 e2"""
    __slots__ = ("v", "a", "b", "c")

    def __init__(self, v, a=0.0, b=0.0, c=0.0):
        self.v, self.a, self.b, self.c = v, a, b, c

    @staticmethod
    def lift(x):
        return x if isinstance(x, HD) else HD(float(x))

    def __add__(self, o):
        o = HD.lift(o)
        return HD(self.v + o.v, self.a + o.a, self.b + o.b, self.c + o.c)
    __radd__ = __add__

    def __neg__(self):
        return HD(-self.v, -self.a, -self.b, -self.c)

    def __sub__(self, o):
        return self + (-HD.lift(o))

    def __rsub__(self, o):
        return HD.lift(o) + (-self)

    def __mul__(self, o):
        o = HD.lift(o)
        return HD(self.v * o.v, self.a * o.v + self.v * o.a, self.b * o.v + self.v * o.b, self.c * o.v + self.a * o.b + self.b * o.a + self.v * o.c)
    __rmul__ = __mul__

    def chain(self, g, d1, d2):
        return HD(g, d1 * self.a, d1 * self.b, d1 * self.c + d2 * self.a * self.b)

    def __truediv__(self, o):
        o = HD.lift(o)
        r = 1.0 / o.v
        return self * o.chain(r, -r * r, 2.0 * r ** 3)

    def __rtruediv__(self, o):
        return HD.lift(o) / self

    def sqrt(self):
        s = math.sqrt(self.v)
        return self.chain(s, 0.5 / s, -0.25 / (s * self.v))

    def acos(self):
        w = 1.0 - self.v * self.v
        return self.chain(math.acos(self.v), -1.0 / math.sqrt(w), -self.v / (w * math.sqrt(w)))

    def clamp(self, lo=None, hi=None):
        if lo is not None and self.v < lo:
            return HD(lo)
        if hi is not None and self.v > hi:
            return HD(hi)
        return self


def atan2(y, x):
    r2 = x.v * x.v + y.v * y.v
    ty, tx = x.v / r2, -y.v / r2
    tyy = -2.0 * x.v * y.v / r2 ** 2
    txy = (y.v * y.v - x.v * x.v) / r2 ** 2
    return HD(math.atan2(y.v, x.v), ty * y.a + tx * x.a, ty * y.b + tx * x.b,
              ty * y.c + tx * x.c + tyy * y.a * y.b - tyy * x.a * x.b + txy * (y.a * x.b + x.a * y.b))


def _dot(p, q):
    return p[0] * q[0] + p[1] * q[1] + p[2] * q[2]


def _cross(p, q):
    return [p[1] * q[2] - p[2] * q[1], p[2] * q[0] - p[0] * q[2], p[0] * q[1] - p[1] * q[0]]


def _sub(p, q):
    return [p[0] - q[0], p[1] - q[1], p[2] - q[2]]


def _taylor(delta):
    t = HD(128.0 / 1575.0)
    for cst in (4.0 / 35.0, 8.0 / 45.0, 1.0 / 3.0, 2.0):
        t = delta * t + cst
    return delta * t


def angle_energy(p, k, theta0):
    """keep_angle_potential.py:58-224 (theta0 in radians)"""
    v1, v2 = _sub(p[0], p[1]), _sub(p[2], p[1])
    u = (_dot(v1, v2) / (_dot(v1, v1).sqrt() * _dot(v2, v2).sqrt()).clamp(lo=1e-12)).clamp(-1.0, 1.0)
    ucp, ucn = math.cos(1e-3), math.cos(math.pi - 1e-3)
    t0, tpi = _taylor(1.0 - u), _taylor(1.0 + u)
    if abs(theta0) < 1e-8:
        if u.v > ucp:
            return 0.5 * k * t0
        if u.v < ucn:
            th = math.pi - tpi.clamp(lo=1e-30).sqrt()
            return 0.5 * k * th * th
        th = u.clamp(ucn, ucp).acos()
        return 0.5 * k * th * th
    if abs(theta0 - math.pi) < 1e-8:
        if u.v < ucn:
            return 0.5 * k * tpi
        if u.v > ucp:
            d = t0.clamp(lo=1e-30).sqrt() - math.pi
            return 0.5 * k * d * d
        d = u.clamp(ucn, ucp).acos() - math.pi
        return 0.5 * k * d * d
    if u.v > ucp:
        d = t0.clamp(lo=1e-30).sqrt() - theta0
    elif u.v < ucn:
        d = (math.pi - tpi.clamp(lo=1e-30).sqrt()) - theta0
    else:
        d = u.acos() - theta0
    return 0.5 * k * d * d


def dihedral_energy(p, k, phi0):
    """keep_dihedral_angle_potential.py:60-154 (phi0 in radians)"""
    b1, b2, b3 = _sub(p[1], p[0]), _sub(p[2], p[1]), _sub(p[3], p[2])
    n1, n2 = _cross(b1, b2), _cross(b2, b3)
    n1s, n2s = _dot(n1, n1), _dot(n2, n2)

    def sw(val):
        t = ((val - 1e-10) * (1.0 / (1e-8 - 1e-10))).clamp(0.0, 1.0)
        return t * t * (3.0 - 2.0 * t)
    s1, s2 = sw(n1s), sw(n2s)
    i1, i2 = 1.0 / n1s.sqrt().clamp(lo=1e-12), 1.0 / n2s.sqrt().clamp(lo=1e-12)
    n1h, n2h = [c * i1 for c in n1], [c * i2 for c in n2]
    ib = 1.0 / _dot(b2, b2).sqrt().clamp(lo=1e-12)
    b2h = [c * ib for c in b2]
    x, y = _dot(n1h, n2h), _dot(_cross(n1h, n2h), b2h)
    diff = atan2(y, x) - phi0
    diff = diff - 2.0 * math.pi * float(np.round(diff.v / (2.0 * math.pi)))
    return 0.5 * k * diff * diff * s1 * s2


def outofplane_energy(p, k, theta0):
    """keep_outofplain_angle_potential.py:33-146; p: centre, probe, two plane atoms"""
    a1, a2, a3 = _sub(p[1], p[0]), _sub(p[2], p[0]), _sub(p[3], p[0])
    n = _cross(a2, a3)
    nsq = _dot(n, n)
    if nsq.v < 1e-8:
        return HD(0.0)
    inv = 1.0 / nsq.sqrt().clamp(lo=1e-12)
    h = _dot(a1, [c * inv for c in n])
    a1n = _dot(a1, a1).sqrt()
    rproj = (a1n * a1n - h * h).clamp(lo=0.0).sqrt()
    diff = atan2(h, rproj) - theta0
    return 0.5 * k * diff * diff


def fragment_distance_egh(X, frag1_0, frag2_0, k, r0_bohr):
    """StructKeepPotentialv2.calc_energy (keep_potential.py:79-115) with hyper-dual derivatives: E, grad (N, 3), H (3N, 3N)."""
    X = np.asarray(X, dtype=float).reshape(-1, 3)
    N = X.shape[0]
    slots = list(frag1_0) + list(frag2_0)
    n1 = len(frag1_0)
    nv = 3 * len(slots)
    g = np.zeros((N, 3))
    H = np.zeros((3 * N, 3 * N))
    E = 0.0
    for i in range(nv):
        for j in range(i, nv):
            cs = [[HD(0.0), HD(0.0), HD(0.0)], [HD(0.0), HD(0.0), HD(0.0)]]
            for a, at in enumerate(slots):
                f = 0 if a < n1 else 1
                for d in range(3):
                    cs[f][d] = cs[f][d] + HD(X[at, d], 1.0 if 3 * a + d == i else 0.0, 1.0 if 3 * a + d == j else 0.0)
            c1 = [c * (1.0 / n1) for c in cs[0]]
            c2 = [c * (1.0 / (len(slots) - n1)) for c in cs[1]]
            v = _sub(c1, c2)
            dd = _dot(v, v).sqrt().clamp(lo=1e-12) - r0_bohr
            e = 0.5 * k * dd * dd
            E = e.v
            r, c = 3 * slots[i // 3] + i % 3, 3 * slots[j // 3] + j % 3
            if i == j:
                g[slots[i // 3], i % 3] += e.a
            H[r, c] += e.c
            if i != j:
                H[c, r] += e.c
    return E, g, H


def deg2rad_as_reference(angle_deg):
    """torch.deg2rad(torch.tensor(angle)) of the reference (keep_angle_potential.py:71-73, keep_dihedral_angle_potential.py:77-79): a float32
    tensor, so the equilibrium angle carries single precision -- 180 degrees comes out 8.7e-8 above pi and misses the `theta_0 == pi` branch."""
    return float(np.float32(angle_deg) * np.float32(math.pi / 180.0))


def energy_gradient_hessian(X, atoms0, k, angle_rad, dihedral=False, out_of_plane=False):
    """E, grad (N, 3), H (3N, 3N) of one term; atoms0: 0-based."""
    X = np.asarray(X, dtype=float).reshape(-1, 3)
    N = X.shape[0]
    nv = 3 * len(atoms0)
    f = outofplane_energy if out_of_plane else (dihedral_energy if dihedral else angle_energy)
    g = np.zeros((N, 3))
    H = np.zeros((3 * N, 3 * N))
    E = 0.0
    for i in range(nv):
        for j in range(i, nv):
            p = [[HD(X[at, d], 1.0 if 3 * a + d == i else 0.0, 1.0 if 3 * a + d == j else 0.0) for d in range(3)] for a, at in enumerate(atoms0)]
            e = f(p, k, angle_rad)
            E = e.v
            r, c = 3 * atoms0[i // 3] + i % 3, 3 * atoms0[j // 3] + j % 3
            if i == j:
                g[atoms0[i // 3], i % 3] += e.a
            H[r, c] += e.c
            if i != j:
                H[c, r] += e.c
    return E, g, H
