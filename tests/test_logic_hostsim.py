"""The scalar control logic that the kernels execute (multioptpy_amd/csrc/logic.h, brent.h), built for
the host with a serial reducer, against the golden vectors and the oracle.  No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cases
from multioptpy_amd import _abi
from oracle import hessian_update as hu
from oracle import rsirfo as orf

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def sim():
    src = os.path.join(HERE, "hostsim", "hostsim.cpp")
    so = os.path.join(HERE, "hostsim", "libmopt_hostsim.so")
    hdrs = [os.path.join(HERE, "..", "multioptpy_amd", "csrc", f) for f in ("logic.h", "brent.h")]
    if os.environ.get("MOPT_HOSTSIM_SO"):                    # (a sanitizer build handed in by tests/test_hostsim_sanitized.py)
        so = os.environ["MOPT_HOSTSIM_SO"]
    elif (not os.path.exists(so)) or any(os.path.getmtime(so) < os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", so, src])
    lib = C.CDLL(so)
    lib.sim_dd_theta.restype = C.c_double
    lib.sim_dd_theta.argtypes = [C.c_double, C.c_double, C.c_double]
    lib.sim_solve_rfo.restype = C.c_double
    lib.sim_solve_rfo.argtypes = [DP, DP, C.c_int, C.c_double, DP, C.POINTER(C.c_int)]
    lib.sim_rs_step.argtypes = [C.POINTER(_abi.RsirfoParams), C.c_double, DP, DP, C.c_int, DP, DP]
    lib.sim_brentq.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, DP, DP, C.POINTER(C.c_int)]
    lib.sim_adjust_trust_radius.argtypes = [C.POINTER(_abi.RsirfoParams), C.POINTER(_abi.RsirfoState), C.c_double,
                                            C.c_double, C.c_double, C.c_double]
    return lib


def _p(a):
    return a.ctypes.data_as(DP)


def device_formula_delta(sim, method, H, s, y):
    """What the K_update kernels compute: dots -> coefficients -> rank-k assembly (+ block symm)."""
    rule = _abi.rule_from_method(method)
    ss, sy, yy = s @ s, s @ y, y @ y
    yd = y
    if sim.sim_rule_is_dd(rule):
        th = sim.sim_dd_theta(ss, sy, 1e-12 if sim.sim_rule_is_block(rule) else 1e-10)
        if th != 1.0 or sy < 0.2 * ss:
            yd = th * y + (1.0 - th) * s
    u = H @ s
    A = yd - u
    z = y - H @ y
    dots = np.array([ss, sy, yy, s @ yd, s @ u, A @ A, A @ s, z @ z, z @ s])
    out = np.zeros(9)
    sim.sim_update_coefficients(rule, _p(dots), _p(out))
    if out[5] == 0:
        return None
    d = (out[0] * np.outer(yd, yd) + out[1] * np.outer(u, u) + out[2] * np.outer(A, A)
         + out[3] * (np.outer(A, s) + np.outer(s, A)) + out[4] * np.outer(s, s))
    if out[6]:
        d = 0.5 * ((H + d) + (H + d).T) - H
    return d


def test_update_coefficients_all_rules(sim):
    z = np.load(os.path.join(G, "g1_updates.npz"))
    ncase = sum(1 for k in z.files if k.endswith("_delta"))
    checked = 0
    for i in range(ncase):
        tag = f"c{i:03d}"
        n, indef, seed = z[tag + "_meta"]
        kind = str(z[tag + "_kind"])
        H = cases.spd_matrix(int(n), seed=int(n) + int(indef), indefinite=bool(indef))
        s, y = cases.sy_pair(H, seed=int(seed), kind=kind)
        d = device_formula_delta(sim, str(z[tag + "_method"]), H, s, y)
        ref = z[tag + "_delta"]
        if d is None:
            # guard in RSIRFO.update_hessian (|s|,|y| < 1e-10 or s.y <= 0): the rule is never called there
            assert np.linalg.norm(s) < 1e-10 or np.linalg.norm(y) < 1e-10 or s @ y <= 0
            continue
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(d - ref).max() / scale < 1e-9, (tag, str(z[tag + "_method"]), kind, np.abs(d - ref).max())
        checked += 1
    assert checked > 300


def test_brentq_iterates_bitexact(sim):
    from make_golden import BRENT_FUNCS
    z = np.load(os.path.join(G, "g4_brentq.npz"))
    for fid, (name, (f, a, b)) in enumerate(BRENT_FUNCS.items()):
        for tname, (xtol, rtol, mi) in (("loose", (1e-6, 1e-6, 50)), ("tight", (1e-12, 1e-12, 100))):
            xs = np.zeros(256)
            nx = C.c_int(0)
            root = C.c_double(0)
            rc = sim.sim_brentq(fid, a, b, xtol, rtol, mi, C.byref(root), _p(xs), C.byref(nx))
            assert rc == 0
            ref = z[f"{name}_{tname}_evals"]
            assert nx.value == ref.shape[0], (name, tname)
            # libm cos/tanh may differ from numpy's in the last ulp: compare the iterates tightly, not bitwise
            assert np.allclose(xs[:nx.value], ref[:, 0], rtol=1e-13, atol=0), (name, tname)
            assert abs(root.value - float(z[f"{name}_{tname}_root"])) <= 1e-13 * abs(root.value)


def _rand_problem(rng, m, scale):
    w = np.sort(rng.normal(size=m)) * scale
    w = w[np.abs(w) >= 1e-6]
    gt = rng.normal(size=w.size) * 10 ** rng.uniform(-4, -1)
    return np.ascontiguousarray(w), np.ascontiguousarray(gt)


def test_rs_step_matches_oracle(sim):
    rng = np.random.default_rng(5)
    for trial in range(200):
        m = int(rng.integers(3, 120))
        w, gt = _rand_problem(rng, m, 10 ** rng.uniform(-3, 0.5))
        if trial % 5 == 0:
            w = np.abs(w) + 1e-3          # positive definite
            w.sort()
        if trial % 7 == 0:
            gt[: min(3, gt.size)] = 0.0   # leading zero components: pole search
        so = int(trial % 2)
        R = [0.5, 0.1, 0.01, 0.3][trial % 4]
        o = orf.RSIRFO(method="rsirfo_bfgs", saddle_order=so)
        o.trust_radius = R
        # oracle path in the eigenbasis: get_rs_step with V = I
        s_or = o.get_rs_step(w, np.eye(w.size), gt.copy())
        p = _abi.default_params(saddle_order=so)
        step = np.zeros(w.size)
        res = np.zeros(5)
        sim.sim_rs_step(C.byref(p), R, _p(w), _p(gt), w.size, _p(step), _p(res))
        br = {0: "within", 1: "brent", 2: "newton", 3: "newton"}[int(res[0])]
        assert br == o.trace["branch"], (trial, br, o.trace["branch"])
        assert abs(res[2] - o.trace["lambda_aug0"]) <= 1e-12 * max(1.0, abs(res[2]))
        assert np.abs(step - s_or).max() <= 1e-11 * max(1.0, np.abs(s_or).max()), (trial, np.abs(step - s_or).max())


def test_rs_step_on_golden_traces(sim):
    """Eigen-basis inputs taken from the oracle replay of the G2 traces (which itself matches the reference)."""
    import test_oracle_golden as tog
    z = np.load(os.path.join(G, "g2_rsirfo.npz"))
    n_checked = 0
    for i in range(0, int(z["ncfg"]), 3):
        tag = f"t{i:02d}"
        for k, opt, step_full, _ in tog._replay(z, tag):
            w = np.ascontiguousarray(opt.trace["eigvals_star"])
            gt = np.ascontiguousarray(opt.trace["gt"])
            p = _abi.default_params(saddle_order=opt.saddle_order)
            # trust radius used for this call = value after adjust == opt.trust_radius now
            step = np.zeros(w.size)
            res = np.zeros(5)
            sim.sim_rs_step(C.byref(p), opt.trust_radius, _p(w), _p(gt), w.size, _p(step), _p(res))
            assert abs(res[3] - opt.trace["initial_norm"]) < 1e-11 * max(1.0, res[3])
            assert abs(res[1] - opt.trace["alpha"]) < 1e-9 * max(1.0, abs(res[1])), (tag, k, res[1], opt.trace["alpha"])
            n_checked += 1
    assert n_checked > 20


def test_trust_radius_logic(sim):
    rng = np.random.default_rng(9)
    for trial in range(500):
        so = int(rng.integers(0, 2))
        o = orf.RSIRFO(method="rsirfo_bfgs", saddle_order=so)
        o.trust_radius = float(rng.uniform(0.01, 0.5 if so == 0 else 0.1))
        actual = float(rng.normal() * 1e-3)
        pred = float(rng.normal() * 1e-3) if trial % 10 else 1e-12
        mine = float(rng.normal() * 10 ** rng.uniform(-8, 0))
        gn = float(10 ** rng.uniform(-4, 0))
        p = _abi.default_params(saddle_order=so)
        st = _abi.RsirfoState()
        st.trust_radius = o.trust_radius
        o.adjust_trust_radius(actual, pred, mine, gn)
        sim.sim_adjust_trust_radius(C.byref(p), C.byref(st), actual, pred, mine, gn)
        assert st.trust_radius == o.trust_radius, (trial, st.trust_radius, o.trust_radius)
