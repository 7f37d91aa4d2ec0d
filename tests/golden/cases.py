"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

Argon Lennard-Jones clusters as SURVEY.md section 8(d) prescribes (so that the
reference's own ``-os lj`` calculator accepts the same systems): simple-cubic
lattice, 3.9 Angstrom spacing, Gaussian jitter.  Energies/gradients/Hessians are
an analytic 12-6 LJ written here (Ar UFF parameters x = 3.817 A, D = 0.237 kcal/mol,
the numbers quoted in SURVEY.md 8(d)); they are only a smooth test surface that
feeds the reference, the oracle and the HIP path with identical numbers.
"""
import numpy as np

BOHR2ANG = 0.52917721067
_RMIN = 3.817 / BOHR2ANG            # bohr
_EPS = 0.237 / 627.509              # hartree


def cluster(N, seed_offset=0, jitter=0.15):
    """N atoms (bohr) on a simple-cubic 3.9 A lattice + N(0, jitter A)."""
    rng = np.random.default_rng(20240501 + N + seed_offset)
    m = int(np.ceil(N ** (1.0 / 3.0)))
    sites = np.array([(i, j, k) for i in range(m) for j in range(m) for k in range(m)], dtype=float)[:N]
    x = sites * 3.9 + rng.normal(0.0, jitter, size=(N, 3))
    return x / BOHR2ANG


def band(I, N):
    """I images (I,N,3) in bohr: cluster A -> A + smooth 0.6 A sine displacement along z (+ jitter)."""
    A = cluster(N)
    rng = np.random.default_rng(20240501 + N + 1)
    L = (A[:, 0].max() - A[:, 0].min()) + 1.0
    B = A.copy()
    B[:, 2] += (0.6 / BOHR2ANG) * np.sin(2 * np.pi * A[:, 0] / L)
    B += rng.normal(0.0, 0.05 / BOHR2ANG, size=B.shape)
    t = np.linspace(0.0, 1.0, I)[:, None, None]
    return (1 - t) * A[None] + t * B[None]


def lj(x, hessian=False, as_reference_calculator=False):
    """12-6 LJ energy (hartree), gradient (N,3) and optionally Hessian (3N,3N).
    as_reference_calculator=True reproduces what the reference's `-os lj` calculator hands out as its "exact" Hessian
    (Calculator/lj_calculation_tools.py:112-118 takes phi'/r with the opposite sign); only the driver goldens (G8), which
    were produced with that calculator, need it."""
    x = np.asarray(x, dtype=float).reshape(-1, 3)
    N = x.shape[0]
    d = x[:, None, :] - x[None, :, :]
    r2 = np.sum(d * d, axis=2)
    np.fill_diagonal(r2, 1.0)
    s2 = _RMIN ** 2 / r2
    s6 = s2 ** 3
    s12 = s6 ** 2
    e_pair = _EPS * (s12 - 2 * s6)
    np.fill_diagonal(e_pair, 0.0)
    E = 0.5 * np.sum(e_pair)
    # dE/dr * 1/r
    f = _EPS * (-12 * s12 + 12 * s6) / r2
    np.fill_diagonal(f, 0.0)
    G = np.sum(f[:, :, None] * d, axis=1)
    if not hessian:
        return E, G
    # second derivative pieces: phi'' and phi'/r
    h2 = _EPS * (156 * s12 - 84 * s6) / r2          # phi''
    np.fill_diagonal(h2, 0.0)
    H = np.zeros((3 * N, 3 * N))
    for a in range(N):
        for b in range(N):
            if a == b:
                continue
            dd = d[a, b]
            fr = -f[a, b] if as_reference_calculator else f[a, b]
            blk = (h2[a, b] - fr) * np.outer(dd, dd) / r2[a, b] + fr * np.eye(3)
            H[3 * a:3 * a + 3, 3 * b:3 * b + 3] -= blk
            H[3 * a:3 * a + 3, 3 * a:3 * a + 3] += blk
    return E, G, H


def spd_matrix(n, seed=0, indefinite=False):
    """H = A A^T / n + 0.1 I  (SURVEY 8(d) micro-benchmark matrix); optional sign flips."""
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n))
    H = A @ A.T / n + 0.1 * np.eye(n)
    if indefinite:
        w, v = np.linalg.eigh(H)
        w[:3] *= -1.0
        H = (v * w) @ v.T
        H = 0.5 * (H + H.T)
    return H


def sy_pair(H, seed=1, kind="normal"):
    """(s, y) pairs incl. the edge cases the update rules branch on."""
    n = H.shape[0]
    rng = np.random.default_rng(seed)
    s = 0.01 * rng.normal(size=n)
    if kind == "normal":
        y = H @ s + 0.002 * rng.normal(size=n)
    elif kind == "negative":          # s.y < 0
        y = -(H @ s) + 0.0005 * rng.normal(size=n)
    elif kind == "dd":                # 0 < s.y < 0.2 s.s -> damping active
        y = 0.05 * s + 1e-4 * rng.normal(size=n)
    elif kind == "tiny":              # denominators below 1e-10
        s = 1e-7 * rng.normal(size=n)
        y = 1e-7 * rng.normal(size=n)
    elif kind == "exact":             # y = H s: A = 0
        y = H @ s
    else:
        raise ValueError(kind)
    return s, y


class LJCalculator:
    """A Calculator plugin for argon clusters: the same interface and Hessian hand-over as the reference's
    `-os lj` calculator (Calculator/lj_calculation_tools.py:209-264), fed by the analytic `lj` above.  Upstream of the
    optimizer-step path; used by the driver tests only."""

    def __init__(self, FC_COUNT=-1, project=None, as_reference_calculator=True):
        self.as_reference_calculator = as_reference_calculator
        self.FC_COUNT = FC_COUNT
        self.hessian_flag = False
        self.Model_hess = None
        self.xtb_method = ""
        self._project = project            # callable(H, x_bohr) -> TR/rot-projected Hessian (calc_exact_hess)

    def calc_exact_hess(self, x_bohr):
        H = lj(x_bohr, hessian=True, as_reference_calculator=self.as_reference_calculator)[2]
        self.Model_hess = self._project(H, x_bohr)

    def single_point(self, file_directory, element_list, iter, electric_charge_and_multiplicity, method="", geom_num_list=None):
        x_ang = np.array(geom_num_list, dtype="float64").reshape(-1, 3)
        x_bohr = x_ang / BOHR2ANG
        e, g = lj(x_bohr)[:2]
        if self.FC_COUNT == -1 or isinstance(iter, str):
            if self.hessian_flag:
                self.calc_exact_hess(x_bohr)
        elif iter % self.FC_COUNT == 0 or self.hessian_flag:
            self.calc_exact_hess(x_bohr)
        return e, g, x_bohr, False


class LJEngine:
    """In-memory NEB engine for argon bands: energies/gradients of every image and, every FC_COUNT iterations, the
    TR/rot-projected Hessians the reference's LJEngine writes to tmp_hessian_i.npy (lj_calculation_tools.py:290-366)."""

    def __init__(self, FC_COUNT=-1, project=None, as_reference_calculator=True, MFC_COUNT=50):
        self.FC_COUNT = FC_COUNT
        self.MFC_COUNT = MFC_COUNT
        self._project = project
        self.as_reference_calculator = as_reference_calculator

    def calculate(self, X_ang, optimize_num):
        X = np.asarray(X_ang, dtype="float64") / BOHR2ANG
        I = X.shape[0]
        E = np.zeros(I)
        G = np.zeros_like(X)
        for i in range(I):
            E[i], G[i] = lj(X[i])[:2]
        hess = None
        # With its default -mfc 50 and no model Hessian the reference's engine raises (and swallows) an AttributeError on
        # iterations that are multiples of 50, iteration 0 included, before it reaches the exact-Hessian branch
        # (lj_calculation_tools.py:336-353): no Hessians are delivered there.
        skipped = self.as_reference_calculator and self.MFC_COUNT != -1 and optimize_num % self.MFC_COUNT == 0
        if self.FC_COUNT != -1 and optimize_num % self.FC_COUNT == 0 and not skipped:
            hess = [self._project(lj(X[i], hessian=True, as_reference_calculator=self.as_reference_calculator)[2], X[i]) for i in range(I)]
        return E, G, X, hess
