"""GPU parity: the HIP path, called through the C ABI (ctypes), against the golden vectors captured from the
reference and against the CPU oracle on identical seeded inputs.  Tolerances: step vectors / Hessians / forces
1e-8 a.u. absolute as BASELINE.json's north_star states (most checks are far tighter)."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    from multioptpy_amd import ops as o
    from multioptpy_amd._lib import ctx
    ctx()
    return o


def test_hess_update_all_rules_vs_reference(ops):
    z = np.load(os.path.join(G, "g1_updates.npz"))
    ncase = sum(1 for k in z.files if k.endswith("_delta"))
    applied = 0
    for i in range(ncase):
        tag = f"c{i:03d}"
        n, indef, seed = z[tag + "_meta"]
        kind = str(z[tag + "_kind"])
        method = str(z[tag + "_method"])
        H = cases.spd_matrix(int(n), seed=int(n) + int(indef), indefinite=bool(indef))
        s, y = cases.sy_pair(H, seed=int(seed), kind=kind)
        Hin, Hsym, info = ops.hess_update(method, H, s, y)
        ref = z[tag + "_delta"]
        if info[0] == 0:
            # RSIRFO.update_hessian's own guards (rsirfo.py:1326-1335) skipped it: H must come back untouched
            assert np.linalg.norm(s) < 1e-10 or np.linalg.norm(y) < 1e-10 or s @ y <= 0
            assert np.array_equal(Hin, H) and np.array_equal(Hsym, H)
            continue
        applied += 1
        scale = max(1.0, np.abs(ref).max())
        assert np.abs((Hin - H) - ref).max() / scale < 1e-9, (tag, method, kind)
        full = H + ref
        assert np.abs(Hsym - 0.5 * (full + full.T)).max() / scale < 1e-9, (tag, method, kind)
    assert applied > 300


def test_hess_update_batched(ops):
    rng = np.random.default_rng(0)
    B, n = 5, 33
    Hs = np.array([cases.spd_matrix(n, seed=k) for k in range(B)])
    S = 0.01 * rng.normal(size=(B, n))
    Y = np.einsum("bij,bj->bi", Hs, S) + 0.001 * rng.normal(size=(B, n))
    from oracle import hessian_update as hu
    Hin, Hsym, info = ops.hess_update("rsirfo_block_bofill", Hs, S, Y)
    for b in range(B):
        d = hu.delta_hessian("rsirfo_block_bofill", Hs[b], S[b], Y[b])
        assert np.abs(Hin[b] - (Hs[b] + d)).max() < 1e-11


def test_projections_vs_reference(ops):
    z = np.load(os.path.join(G, "g5_projections.npz"))
    for name in ("ar7", "ar22", "linear_z", "diatomic"):
        x, H, g = z[name + "_x"], z[name + "_H"], z[name + "_g"]
        Hp = ops.project_hess(H, x)
        assert np.abs(Hp - z[name + "_Hp"]).max() < 1e-11, name
        gp = ops.project_grad(g, x)
        assert np.abs(gp - z[name + "_gp"]).max() < 1e-11, name


@pytest.mark.parametrize("n", [3, 21, 45, 66, 150, 300, 337, 384, 450])      # 337/384: last register-resident MFMA back-transformation sizes, 450: VALU form, LDS rows
def test_eigh_random(ops, n):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n))
    A = 0.5 * (A + A.T)
    w, v, sw = ops.eigh(A, return_sweeps=True)
    wr = np.linalg.eigvalsh(A)
    nrm = np.abs(A).sum(axis=1).max()          # LAPACK's own bound is ~ n eps ||A||
    assert np.all(np.diff(w) >= 0)
    print("n", n, "sweeps", sw, "max eigenvalue error", np.abs(w - wr).max(), "norm", nrm)
    assert np.abs(w - wr).max() < 1e-13 * nrm, (n, np.abs(w - wr).max())
    assert np.abs(v.T @ v - np.eye(n)).max() < 1e-12
    assert np.abs(A @ v - v * w).max() < 5e-12 * nrm
    assert 0 <= sw[0] <= 16        # 0: the reduction route reports no sweeps


def test_eigh_degenerate_and_batched(ops):
    # projected identity (n-6 fold degenerate 1, six zero modes), an LJ Hessian and a diagonal matrix in one batch
    N = 22
    n = 3 * N
    x = cases.cluster(N)
    from oracle import projection as pj
    P1 = pj.project_hessian_trrot(np.eye(n), x)
    H = pj.project_hessian_trrot(cases.lj(x, hessian=True)[2], x)
    D = np.diag(np.linspace(-1, 2, n))
    A = np.array([P1, H, D])
    w, v, sw = ops.eigh(A, return_sweeps=True)
    for b in range(3):
        wr = np.linalg.eigvalsh(A[b])
        scale = max(1.0, np.abs(wr).max())
        assert np.abs(w[b] - wr).max() < 1e-13 * scale + 1e-15, (b, np.abs(w[b] - wr).max())
        assert np.abs(v[b].T @ v[b] - np.eye(n)).max() < 1e-12
        assert np.abs(A[b] @ v[b] - v[b] * w[b]).max() < 1e-12 * scale
    assert np.sum(np.abs(w[0]) < 1e-10) == 6
    assert sw[2] <= 2


def test_bneb_force_vs_reference(ops):
    z = np.load(os.path.join(G, "g6_neb_force.npz"))
    for k in range(int(z["ncase"])):
        tag = f"f{k:02d}"
        X, E, Gr = z[tag + "_X"], z[tag + "_E"], z[tag + "_G"]
        F, tau = ops.bneb_force(X, E, Gr, optimize_num=5, apply_ci_neb=(2 if int(z[tag + "_ci"]) else 99999))
        assert np.abs(F - z[tag + "_F"]).max() < 1e-12, tag
        assert np.abs(tau - z[tag + "_tau"]).max() < 1e-12, tag


def _replay_gpu(z, tag):
    from multioptpy_amd.Optimizer.rsirfo import RSIRFO
    N, so, neb, nsteps, seed = [int(v) for v in z[tag + "_meta"]]
    method = str(z[tag + "_method"])
    n = 3 * N
    h0 = str(z[tag + "_h0kind"])
    x0 = cases.cluster(N, seed_offset=seed)
    if h0 == "exact":
        H = cases.lj(x0, hessian=True)[2]
    elif h0 == "identity":
        H = np.eye(n)
    else:
        H = cases.spd_matrix(n, seed=seed)
    opt = RSIRFO(method=method, saddle_order=so, trust_radius_max=0.5 if so == 0 else 0.1, trust_radius_min=0.01)
    if neb:
        opt.switch_NEB_mode()
    px = pg = pE = None
    for k in range(nsteps):
        x = z[f"{tag}_s{k}_x"]
        g = z[f"{tag}_s{k}_g"]
        E = float(z[f"{tag}_s{k}_E"])
        opt.set_hessian(H)
        opt.set_bias_hessian(np.zeros((n, n)))
        if k == 0:
            step = opt.run(x, g, [], [], E, 0.0, [], [], g, [])
        else:
            step = opt.run(x, g, pg, px, E, pE, [], [], g, pg)
        yield k, opt, step, H
        px, pg, pE = x, g, E


def test_rsirfo_traces_vs_reference():
    """G2: consecutive RSIRFO.run calls (36 traces incl. N = 100) reproduce the reference's step, trust radius,
    predicted energy change and both Hessian objects (the caller-aliased array and self.hessian)."""
    z = np.load(os.path.join(G, "g2_rsirfo.npz"))
    worst = 0.0
    for i in range(int(z["ncfg"])):
        tag = f"t{i:02d}"
        for k, opt, step, Hcaller in _replay_gpu(z, tag):
            ref = z[f"{tag}_s{k}_step"]
            assert step.shape == ref.shape
            err = np.abs(step - ref).max()
            worst = max(worst, err)
            assert err < 1e-8, (tag, k, err)
            assert abs(opt.trust_radius - float(z[f"{tag}_s{k}_tr"])) < 1e-12, (tag, k)
            pr = float(z[f"{tag}_s{k}_pred"])
            assert abs(opt.predicted_energy_changes[-1] - pr) < 1e-8 * max(1.0, abs(pr)), (tag, k)
            if f"{tag}_s{k}_Hself" in z.files:
                assert np.abs(opt.hessian - z[f"{tag}_s{k}_Hself"]).max() < 1e-8, (tag, k)
                assert np.abs(Hcaller - z[f"{tag}_s{k}_Hcaller"]).max() < 1e-8, (tag, k)
    print("worst step deviation from the reference:", worst)


def test_rsirfo_requires_hessian():
    from multioptpy_amd.Optimizer.rsirfo import RSIRFO
    opt = RSIRFO(method="rsirfo_bfgs", saddle_order=0)
    with pytest.raises(ValueError):
        opt.run(np.zeros((9, 1)), np.zeros((9, 1)))


@pytest.mark.parametrize("variant", ["ident", "exact"])
def test_rfo_neb_band_steps_vs_reference(variant):
    """G7: four consecutive RFOOptimizer.optimize steps (I = 8, N = 7) incl. the accumulated per-image Hessians."""
    from multioptpy_amd.band import Band
    z = np.load(os.path.join(G, "g7_neb_step.npz"))
    I, N = 8, 7
    X0 = z[f"{variant}_it0_X"]
    band = Band(I, N)
    if variant == "exact":
        for i in range(I):
            band.set_hessian(i, cases.lj(X0[i], hessian=True)[2])
    for it in range(4):
        tag = f"{variant}_it{it}"
        X, E, Gr = z[tag + "_X"], z[tag + "_E"], z[tag + "_G"]
        band.upload(X, E, Gr)
        band.step(it)
        newX, F = band.download(want_force=True)
        assert np.abs(F - z[tag + "_F"]).max() < 1e-12, tag
        err = np.abs(newX - z[tag + "_newX"]).max()
        assert err < 1e-8, (tag, err)
        H = np.array([band.get_hessian(i) for i in range(I)])
        assert np.abs(H - z[tag + "_hess"]).max() < 1e-8, (tag, np.abs(H - z[tag + "_hess"]).max())
        tr = np.array([band.get_state(i).trust_radius for i in range(I)])
        assert np.allclose(tr, z[tag + "_tr"], atol=1e-12), tag
    band.close()


def test_rfo_optimizer_class_with_npy_persistence(tmp_path):
    """The drop-in class under the reference's calling convention, Hessians round-tripping through tmp_hessian_i.npy."""
    from multioptpy_amd.Optimizer.rfo_neb import RFOOptimizer
    from multioptpy_amd.MEP.pathopt_bneb_force import CaluculationBNEB
    z = np.load(os.path.join(G, "g7_neb_step.npz"))

    class Cfg:
        pass
    cfg = Cfg()
    cfg.NEB_FOLDER_DIRECTORY = str(tmp_path) + "/"
    cfg.fix_init_edge = cfg.fix_end_edge = False
    cfg.ratio_of_rfo_step = 0.5
    I = 8
    X0 = z["exact_it0_X"]
    for i in range(I):
        np.save(cfg.NEB_FOLDER_DIRECTORY + f"tmp_hessian_{i}.npy", cases.lj(X0[i], hessian=True)[2])
    opt = RFOOptimizer(cfg)
    calc = CaluculationBNEB()
    preX = preG = None
    pre_vel = []
    for it in range(3):
        tag = f"exact_it{it}"
        X, E, Gr = z[tag + "_X"], z[tag + "_E"], z[tag + "_G"]
        F = calc.calc_force(X, E, Gr, it, ["Ar"] * 7)
        assert np.abs(F - z[tag + "_F"]).max() < 1e-12
        newX = opt.optimize(X, Gr, preX, preG, it, E, E, pre_vel, F.copy(), None, preX, calc)
        assert np.abs(newX - z[tag + "_newX"]).max() < 1e-8, tag
        H = np.array([np.load(cfg.NEB_FOLDER_DIRECTORY + f"tmp_hessian_{i}.npy") for i in range(I)])
        assert np.abs(H - z[tag + "_hess"]).max() < 1e-8
        preX, preG, pre_vel = X.copy(), Gr.copy(), F.copy()


def test_band_sharded_matches_whole():
    """Image sharding: two shards of a 10-image band reproduce the single-shard result (no data-path collective:
    every shard sees the whole X/E/G and owns the RSIRFO work of its images)."""
    from multioptpy_amd.band import Band
    I, N = 10, 9
    X = cases.band(I, N)
    whole = Band(I, N)
    parts = [Band(I, N, first=0, count=5), Band(I, N, first=5, count=5)]
    for it in range(3):
        E = np.zeros(I)
        Gr = np.zeros((I, N, 3))
        for i in range(I):
            E[i], Gr[i] = cases.lj(X[i])
        whole.upload(X, E, Gr)
        whole.step(it)
        ref = whole.download()
        got = []
        for p in parts:
            p.upload(X, E, Gr)
            p.step(it)
            got.append(p.download())
        got = np.concatenate(got, axis=0)
        assert np.array_equal(got, ref)
        X = ref / cases.BOHR2ANG


@pytest.mark.parametrize("n", [5, 64, 100, 300])
def test_dgemm_mfma_f64(ops, n):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(3, n, n))
    B = rng.normal(size=(3, n, n))          # asymmetric on purpose: catches a transposed operand / C write
    C = ops.dgemm(A, B)
    ref = A @ B
    assert np.abs(C - ref).max() < 1e-12 * n


def test_band_warm_start_matches_cold():
    """Warm-starting the Jacobi eigen-solver from the previous step's eigenvectors changes only the gauge."""
    import os as _os
    from multioptpy_amd.band import Band
    I, N = 6, 12
    res = {}
    for warm in ("1", "0"):
        _os.environ["MOPT_EIGH_WARM"] = warm
        band = Band(I, N)
        X = cases.band(I, N)
        outs = []
        for it in range(5):
            E = np.zeros(I)
            Gr = np.zeros((I, N, 3))
            for i in range(I):
                E[i], Gr[i] = cases.lj(X[i])
            band.upload(X, E, Gr)
            band.step(it)
            new = band.download()
            outs.append(new)
            X = new / cases.BOHR2ANG
        res[warm] = (np.array(outs), [band.get_state(i).jacobi_sweeps for i in range(I)])
        band.close()
    _os.environ.pop("MOPT_EIGH_WARM", None)
    assert np.abs(res["1"][0] - res["0"][0]).max() < 1e-9
    print("sweeps warm", res["1"][1], "cold", res["0"][1])


def test_neb_spring_variants_vs_reference(ops):
    """G6b: DNEB, EWBNEB, BNEB3 calc_force (incl. CI-NEB switches and DNEB's quirks) against the reference."""
    z = np.load(os.path.join(G, "g6b_neb_variants.npz"))
    from multioptpy_amd.MEP.pathopt_dneb_force import CaluculationDNEB
    from multioptpy_amd.MEP.pathopt_ewbneb_force import CaluculationEWBNEB
    from multioptpy_amd.MEP.pathopt_bneb_force import CaluculationBNEB3
    for k in range(int(z["ncase"])):
        tag = f"v{k:02d}"
        X, E, Gr = z[tag + "_X"], z[tag + "_E"], z[tag + "_G"]
        apply = 2 if int(z[tag + "_ci"]) else 99999
        for cls, key in ((CaluculationDNEB, "dneb"), (CaluculationEWBNEB, "ewbneb"), (CaluculationBNEB3, "bneb3")):
            F = cls(apply).calc_force(X, E, Gr, 5, ["Ar"] * X.shape[1])
            err = np.abs(F - z[tag + "_" + key]).max()
            assert err < 1e-12, (tag, key, err)


def test_rsprfo_traces_vs_reference():
    """G3: EnhancedRSPRFO (P-RFO) consecutive run() calls against the reference: step, trust radius, predicted energy
    change, self.hessian after the (checked) update, tracked TS mode index."""
    from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO
    import test_oracle_golden as tog
    z = np.load(os.path.join(G, "g3_rsprfo.npz"))
    worst = 0.0
    for i in range(int(z["ncfg"])):
        tag = f"p{i:02d}"
        for k, opt, step in tog._replay_rsprfo(z, tag, EnhancedRSPRFO):
            ref = z[f"{tag}_s{k}_step"]
            err = np.abs(step - ref).max()
            worst = max(worst, err)
            assert err < 1e-8 * max(1.0, np.abs(ref).max()), (tag, k, err)
            assert abs(opt.trust_radius - float(z[f"{tag}_s{k}_tr"])) < 1e-12, (tag, k)
            pr = float(z[f"{tag}_s{k}_pred"])
            assert abs(opt.predicted_energy_changes[-1] - pr) < 1e-8 * max(1.0, abs(pr)), (tag, k)
            assert np.abs(opt.hessian - z[f"{tag}_s{k}_H"]).max() < 1e-8, (tag, k)
            ts = int(z[f"{tag}_s{k}_tsidx"])
            if opt.saddle_order == 1:
                assert (opt.ts_mode_idx is None and ts == -1) or int(opt.ts_mode_idx) == ts, (tag, k)
    print("RSPRFO worst step deviation from the reference:", worst)


def test_mf_rsirfo_traces_vs_reference():
    """G15: MF_RSIRFO (mode-following RS-I-RFO: mopt_mf_rsirfo_begin / ModeFollowing on the host / mopt_mf_rsirfo_finish) against twelve
    traces of the reference's class: step, trust radius, predicted change, matched mode indices, both Hessian objects (F7)."""
    from multioptpy_amd.Optimizer.mf_rsirfo import MF_RSIRFO
    import test_oracle_golden as tog
    z = np.load(os.path.join(G, "g15_mf_rsirfo.npz"))
    worst = tog._check_mf(z, lambda method, so, elems, masses, kw: MF_RSIRFO(method=method, saddle_order=so, element_list=elems, **kw), 1e-8)
    print("MF_RSIRFO worst step deviation from the reference:", worst)


def test_mf_rsirfo_through_the_dispatcher():
    """optimizer.py:438-448: the four key families reach MF_RSIRFO with the reference's keyword sets."""
    from multioptpy_amd.optimizer import CalculateMoveVector
    from multioptpy_amd.Optimizer.mf_rsirfo import MF_RSIRFO
    cmv = CalculateMoveVector(0.5, ["Ar"] * 7, saddle_order=1)
    for key, mw, adaptive in (("mwsmf_rsirfo_bofill", True, False), ("mwmf_rsirfo_fsb:1", True, True), ("smf_rsirfo_bfgs", False, False), ("mf_rsirfo_bofill:ema0.5", False, True)):
        opt = cmv.initialization([key])[0]
        assert isinstance(opt, MF_RSIRFO) and opt.DELTA == 0.5
        assert (opt.mode_follower.mass_weights is not None) == mw and opt.mode_follower.adaptive == adaptive, key


@pytest.mark.parametrize("natoms", [7, 11, 22])
def test_rsprfo_surplus_negative_modes_vs_oracle(natoms):
    """A start geometry whose exact Hessian has SEVERAL negative eigenvalues (saddle order 1): the surplus ones are shifted to
    min_positive_eigval together with the six TR modes, and the reference's second eigh returns an arbitrary basis of that
    degenerate group.  Same inputs for both classes at every step; the move vectors must agree until the oracle reports that its
    mode choice hangs on that basis (oracle/rsprfo.py _basis_dependent), and at least two shifted steps must come before that."""
    import contextlib
    import io
    from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO
    from oracle.rsprfo import EnhancedRSPRFO as Oracle
    n = 3 * natoms
    x0 = cases.cluster(natoms, jitter=0.3) * 0.95
    H0 = cases.lj(x0, hessian=True)[2]
    assert (np.linalg.eigvalsh(H0) < -1e-6).sum() >= 2
    a = EnhancedRSPRFO(method="rsprfo_bofill", saddle_order=1, element_list=["Ar"] * natoms)
    b = Oracle(method="rsprfo_bofill", saddle_order=1, element_list=["Ar"] * natoms)
    for o in (a, b):
        o.set_hessian(H0.copy())
        o.set_bias_hessian(np.zeros((n, n)))
    xs, px, pg, pe = x0.copy(), [], [], 0.0
    compared, worst = 0, 0.0
    for it in range(12):
        e, g = cases.lj(xs)[:2]
        ma = a.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, pe, [], [], g.reshape(-1, 1), pg)
        with contextlib.redirect_stdout(io.StringIO()):
            mb = b.run(xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy(), pg, px, e, pe, [], [], g.reshape(-1, 1).copy(), pg)
        if b.trace.get("basis_dependent", False):
            break
        assert a.last_info["shifted"] == int(bool(b.trace["shifted"]))
        err = np.abs(np.asarray(ma) - np.asarray(mb)).max()
        worst = max(worst, err)
        assert err < 1e-9, (it, err)
        assert abs(a.trust_radius - b.trust_radius) < 1e-12 and int(a.ts_mode_idx) == int(b.ts_mode_idx), it
        assert np.abs(np.asarray(a.hessian) - np.asarray(b.hessian)).max() < 1e-10
        compared += 1
        px, pg, pe = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy(), e
        step = np.asarray(ma).reshape(natoms, 3)
        xs = xs - step * min(1.0, 0.1 / max(np.linalg.norm(step), 1e-12))
    print(f"N = {natoms}: {compared} steps compared, worst deviation {worst:.2e}")
    assert compared >= 2


@pytest.mark.parametrize("algo", ["tridiag", "jacobi", "tridiag-stepwise", "tridiag-persistent", "tridiag-nosmall", "tridiag-persist2",
                                  "tridiag-valubacktrans", "tridiag-group8", "tridiag-small32", "tridiag-notail", "tridiag-tail192",
                                  "tridiag-tail100w16", "tridiag-jacobismall", "tridiag-dcmin64"])
def test_eigh_forced_route(ops, algo):
    """Both eigen-solvers forced for every size (MOPT_EIGH_ALGO): random, degenerate, clustered and LJ matrices.
    (By default the reduction route runs from n = 32 on and the Jacobi sweeps below.)"""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import cases
from multioptpy_amd import ops
from oracle import projection as pj
rng = np.random.default_rng(0)
mats = {}
for n in (3, 21, 66, 150, 300):
    R = rng.normal(size=(n, n)); mats[f"random{n}"] = 0.5 * (R + R.T)
N = 22; n = 3 * N; x = cases.cluster(N)
mats["projI"] = pj.project_hessian_trrot(np.eye(n), x)
mats["lj22"] = pj.project_hessian_trrot(cases.lj(x, hessian=True)[2], x)
mats["diag"] = np.diag(np.linspace(-1, 2, n))
Q, _ = np.linalg.qr(rng.normal(size=(60, 60)))
lam = np.concatenate([np.ones(20), 1 + 1e-9 * rng.normal(size=20), np.linspace(2, 3, 20)])
mats["clustered"] = (Q * lam) @ Q.T
x100 = cases.cluster(100)
mats["lj100"] = pj.project_hessian_trrot(cases.lj(x100, hessian=True)[2], x100)
# identity + rank-20 update: the reduced form has unreduced blocks of 30-50 rows around a many-fold eigenvalue 1 (late NEB steps)
U = rng.normal(size=(150, 20)); mats["ident_lowrank"] = np.eye(150) + 0.3 * (U * rng.normal(size=20)) @ U.T / 150
bad = 0
for name, A in mats.items():
    w, v = ops.eigh(A)
    wr = np.linalg.eigvalsh(A)
    nrm = max(np.abs(A).sum(axis=1).max(), 1e-300)
    e1 = np.abs(w - wr).max() / nrm; e2 = np.abs(v.T @ v - np.eye(len(w))).max(); e3 = np.abs(A @ v - v * w).max() / nrm
    print(f"{name:12s} eig {e1:.2e} orth {e2:.2e} resid {e3:.2e}")
    if not (e1 < 1e-13 and e2 < 1e-11 and e3 < 1e-12): bad += 1
B = np.array([mats["lj22"], mats["projI"], mats["diag"]])
w, v = ops.eigh(B)
for b in range(3):
    assert np.abs(w[b] - np.linalg.eigvalsh(B[b])).max() < 1e-13 * max(1.0, np.abs(B[b]).sum(axis=1).max())
# 40 matrices: more than the 32 groups of 8 workgroups that are resident at once (the persistent kernel is launched twice)
B = np.array([mats["lj22"] + 0.01 * k * np.eye(66) for k in range(40)])
w, v = ops.eigh(B)
for b in range(40):
    assert np.abs(w[b] - np.linalg.eigvalsh(B[b])).max() < 1e-13 * max(1.0, np.abs(B[b]).sum(axis=1).max())
    assert np.abs(v[b].T @ v[b] - np.eye(66)).max() < 1e-11
sys.exit(bad)
'''
    extra = {"tridiag-stepwise": {"MOPT_TRI_NOPERSIST": "1"}, "tridiag-persistent": {"MOPT_TRI_FORCEPERSIST": "1"},
             "tridiag-nosmall": {"MOPT_TRI_SMALL": "0"}, "tridiag-persist2": {"MOPT_TRI_P3": "0"},
             "tridiag-valubacktrans": {"MOPT_BT_MFMA": "0"}, "tridiag-group8": {"MOPT_TRI_GMAX": "8"},
             "tridiag-small32": {"MOPT_TRI_SMALL": "32"}, "tridiag-notail": {"MOPT_TRI_TAIL": "0"},
             # single-compute-unit tridiagonalisation (k_tritail.hip): whole matrices up to n = 192, and as the 192- / 100-row tail of
             # the persistent kernel for n = 300 (8- and 16-wave forms of the three-chunk kernel)
             "tridiag-tail192": {"MOPT_TRI_TAILONLY": "192", "MOPT_TRI_TAILMAX": "192"},
             "tridiag-tail100w16": {"MOPT_TRI_TAILONLY": "192", "MOPT_TRI_TAILMAX": "100", "MOPT_TAIL_W8": "0"},
             "tridiag-jacobismall": {"MOPT_TRI_QL": "0"},
             # blocks of 33..64 rows by the two-wave Jacobi (the default until the end of round 2; now they go to divide and conquer)
             "tridiag-dcmin64": {"MOPT_TRI_DCMIN": "64"}}.get(algo, {})
    env = dict(os.environ, MOPT_EIGH_ALGO=algo.split("-")[0], **extra)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    print(r.stdout)
    print(r.stderr[-2000:])
    assert r.returncode == 0


@pytest.mark.parametrize("algo", ["tridiag", "jacobi"])
def test_traces_forced_route(algo):
    """The RSIRFO traces (G2), the band steps (G7) and the RS-P-RFO traces (G3) with one eigen-solver forced for every size."""
    import subprocess
    import sys
    env = dict(os.environ, MOPT_EIGH_ALGO=algo)
    root = os.path.dirname(os.path.dirname(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-k",
                        "(rsirfo or band or rsprfo or shard) and not forced"], env=env, capture_output=True, text=True, cwd=root)
    print(r.stdout[-3000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0


def test_eigensolver_failure_is_loud():
    """A failure flag from the reduction route (injected) must surface as MOPT_E_NUMERIC / MoptError, never as a silent result."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from multioptpy_amd import ops
from multioptpy_amd._lib import MoptError
R = np.random.default_rng(0).normal(size=(2, 40, 40)); A = R + R.transpose(0, 2, 1)
try:
    ops.eigh(A)
except MoptError as e:
    assert "-5" in str(e), str(e)
    sys.exit(0)
sys.exit(1)
'''
    env = dict(os.environ, MOPT_EIGH_ALGO="tridiag", MOPT_TRI_INJECT_FAIL="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0


def test_kabsch_vs_reference(ops):
    """a20: Kabsch on the device against the reference's output (G5), incl. the in-place centring of the drop-in class."""
    from multioptpy_amd.Utils.calc_tools import Calculationtools
    z = np.load(os.path.join(G, "g5_projections.npz"))
    keys = [k for k in z.files if "kabsch" in k]
    assert keys, z.files
    P, Q = z["kabsch_P"].copy(), z["kabsch_Q"].copy()
    Pa, Qc = ops.kabsch(P, Q)
    assert np.abs(Pa - z["kabsch_Pout"]).max() < 1e-12 and np.abs(Qc - z["kabsch_Qout"]).max() < 1e-12
    P2, Q2 = z["kabsch_P"].copy(), z["kabsch_Q"].copy()
    Pr, Qr = Calculationtools().kabsch_algorithm(P2, Q2)
    assert Qr is Q2 and np.abs(Q2 - z["kabsch_Qout"]).max() < 1e-12 and np.abs(P2.mean(axis=0)).max() < 1e-12
    # planar and random cases against the oracle
    from oracle import projection as pj
    rng = np.random.default_rng(5)
    for N, planar in ((3, True), (4, True), (7, False), (100, False)):
        A = rng.normal(size=(N, 3))
        if planar:
            A[:, 2] = 0.0
        th = 0.7
        Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
        B = A @ Rz.T + 0.01 * rng.normal(size=(N, 3)) * (0 if planar else 1) + 0.3
        ref, _ = pj.kabsch(A.copy(), B.copy())
        got, _ = ops.kabsch(A, B)
        assert np.abs(got - ref).max() < 1e-11, (N, planar)


def test_outer_trust_radius_vs_reference():
    """a15: TrustRadius.update_trust_radii sequences (G9), predicted energy change on the device."""
    from multioptpy_amd.Optimizer.trust_radius import TrustRadius
    z = np.load(os.path.join(G, "g9_trust_radius.npz"))
    for c in range(int(z["ncases"])):
        n, nk, tmax, tmin = z[f"c{c}_meta"]
        TR = TrustRadius()
        if tmax > 0:
            TR.set_max_trust_radius(float(tmax))
        if tmin > 0:
            TR.set_min_trust_radius(float(tmin))
        tr = 0.5 if tmax < 0 else float(tmax)
        H = z[f"c{c}_H"]
        for k in range(int(nk)):
            e1, e0 = z[f"c{c}_k{k}_e"]
            tr = float(TR.update_trust_radii(e1, e0, z[f"c{c}_k{k}_g"].reshape(-1, 1), z[f"c{c}_k{k}_mv"].reshape(-1, 1), H,
                                             np.zeros((int(n), 1)), tr))
            assert abs(tr - z[f"c{c}_tr"][k]) < 1e-12, (c, k)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_optmain_driver_vs_reference(tag):
    """a15/a20/a23: the in-memory optmain iteration on the device path against the reference driver's own trajectory (G8):
    geometry, move vector, new geometry, Hessian, outer trust radius, displacement and convergence thresholds per iteration."""
    import cases
    from oracle import projection as pj
    from multioptpy_amd import drivers
    from test_oracle_golden import _check_optmain
    z = np.load(os.path.join(G, "g8_optmain.npz"))

    def run(x0, elems, method, order, nstep, fc, rec, pc=None):
        calc = cases.LJCalculator(FC_COUNT=fc, project=pj.project_hessian_trrot)      # the Calculator is upstream of the path
        drivers.optmain_steps(calc, x0, elems, method, order, nstep, fc, record=rec, projection_constraint=pc)      # (d: `-pc`, CRSIRFO via the dispatcher)

    worst = _check_optmain(run, z, tag, 1e-8)
    print("optmain", tag, "worst move-vector deviation from the reference", worst)


@pytest.mark.parametrize("kind", ["keep", "afir", "both", "angles", "wells"])
def test_optmain_driver_with_bias_vs_oracle(kind):
    """a23 + f3: the in-memory optmain iteration with bias potentials -- harmonic distance restraints (mopt_keep_pot) and / or the AFIR
    term (mopt_afir): bias energy and gradient added to the calculator's, bias Hessian handed to the optimizer -- against the oracle's
    driver with the oracle's analytic bias (both pinned by the reference's autograd outputs, G11 / G14): six iterations on Ar7."""
    import cases
    from oracle import driver, projection as pj, afir as oafir, keep_pot as okp, keep_angle as oka, model_hessian as mh
    from multioptpy_amd import drivers
    N = 7
    x0 = cases.cluster(N, jitter=0.2) * cases.BOHR2ANG
    el = ["Ar"] * N
    cov = mh.element_params(el)["cov"]
    kp = [dict(pair=[1, 4], k=0.05, r0=3.2), dict(pair=[4, 6], k=0.1, r0=4.5)] if kind in ("keep", "both") else None
    af = dict(gamma=40.0, fragm_1=[1, 2, 3], fragm_2=[5, 6, 7]) if kind in ("afir", "both") else None
    ka = [dict(atoms=[1, 2, 3], k=0.08, angle=75.0), dict(atoms=[4, 5, 6], k=0.05, angle=110.0)] if kind == "angles" else None
    kd = [dict(atoms=[1, 2, 5, 7], k=0.06, angle=40.0)] if kind == "angles" else None
    # round 4: the well potentials through the drivers' generic hook (objects with calc_energy_gradient_hessian)
    wells = []
    if kind == "wells":
        from multioptpy_amd.Potential.switching_potential import WellPotential, WellPotentialAround
        from oracle import well_pot as owp
        wells = [("well", [1, 2], [6, 7], 150.0, [1.0, 2.0, 3.0, 4.5]), ("around", [1, 2, 3], [4, 5, 6, 7], 80.0, [0.5, 1.0, 2.5, 3.5])]
        wobj = [WellPotential(well_pot_wall_energy=wells[0][3], well_pot_fragm_1=wells[0][1], well_pot_fragm_2=wells[0][2], well_pot_limit_dist=wells[0][4]),
                WellPotentialAround(around_well_pot_wall_energy=wells[1][3], around_well_pot_center=wells[1][1], around_well_pot_limit_dist=wells[1][4], around_well_pot_target=wells[1][2])]

    def bias(x):
        E, g, H = 0.0, np.zeros((N, 3)), np.zeros((3 * N, 3 * N))
        if af:
            e1, g1, h1 = oafir.afir(x, cov, af["fragm_1"], af["fragm_2"], af["gamma"])
            E, g, H = E + e1, g + g1, H + h1
        for p in kp or []:
            e1, g1, h1 = okp.keep_pot(x, p["pair"], p["k"], p["r0"])
            E, g, H = E + e1, g + g1, H + h1
        for terms, dih in ((ka, False), (kd, True)):
            for p in terms or []:
                e1, g1, h1 = oka.energy_gradient_hessian(x, [a - 1 for a in p["atoms"]], p["k"], oka.deg2rad_as_reference(p["angle"]), dihedral=dih)
                E, g, H = E + e1, g + g1, H + h1
        for wk_, l1, l2, e0, lim in wells:
            e1, g1, h1 = owp.well_potential(wk_, x, l1, l2, e0, lim)
            E, g, H = E + e1, g + g1, H + h1
        return E, g, H

    ra, rb = [], []
    drivers.optmain_steps(cases.LJCalculator(FC_COUNT=-1, project=pj.project_hessian_trrot), x0, el, "rsirfo_bfgs", 0, 6, -1, record=ra, afir=af, keep_pot=kp, keep_angle=ka, keep_dihedral=kd,
                          bias_potentials=wobj if kind == "wells" else None)
    driver.run_optmain(cases.LJCalculator(FC_COUNT=-1, project=pj.project_hessian_trrot), x0, el, "rsirfo_bfgs", 0, 6, -1, record=rb, bias=bias)
    assert len(ra) == len(rb) >= 3
    worst = 0.0
    for a, b in zip(ra, rb):
        for key in ("geom", "B_g", "mv", "aligned", "H"):
            worst = max(worst, float(np.abs(np.asarray(a[key]) - np.asarray(b[key])).max()))
        assert abs(a["B_e"] - b["B_e"]) < 1e-10 and abs(a["tr"] - b["tr"]) < 1e-10
    assert worst < 1e-8, worst
    assert np.abs(ra[0]["B_g"] - cases.lj(ra[0]["geom"])[1]).max() > 1e-4      # the bias is really there
    print("biased optmain", kind, "worst deviation from the oracle's driver", worst)


@pytest.mark.parametrize("tag", ["n", "e", "f", "s", "a", "c", "m"])
def test_nebmain_driver_vs_reference(tag):
    """a23: the in-memory nebmain iteration (RFO branch) on the device path against the reference driver's own band
    trajectory (G8): geometry read by the engine, optimizer output, geometry after edge fixing + Kabsch chain, Hessians."""
    import cases
    from oracle import projection as pj
    from multioptpy_amd import drivers
    from test_oracle_golden import _check_nebmain

    def run(ends, fc, nstep, p, fe, rec, sd, **extra):          # extra: align_distances (`-ad`), climbing_image (`-ci`)
        eng = cases.LJEngine(FC_COUNT=fc, project=pj.project_hessian_trrot)           # the engine is upstream of the path
        drivers.nebmain_steps(eng, ends, ["Ar"] * ends.shape[1], nstep, p, fc, fe, record=rec, steepest_descent=sd, **extra)

    worst = _check_nebmain(run, np.load(os.path.join(G, "g8_nebmain.npz")), tag, 1e-8)
    print("nebmain", tag, "worst optimizer-output deviation from the reference", worst)


def test_neb_tr_fire_standalone_vs_reference():
    """a19: TR_NEB.TR_calc and FIREOptimizer.optimize as standalone drop-ins against the reference's outputs (G7)."""
    from multioptpy_amd.Optimizer.trust_radius_neb import TR_NEB
    from multioptpy_amd.Optimizer.fire_neb import FIREOptimizer
    import cases
    z = np.load(os.path.join(G, "g7_neb_step.npz"))
    mv = TR_NEB().TR_calc(z["tr_X"], z["tr_F"], z["tr_delta"], None, None, None)
    assert np.abs(np.array(mv) - z["tr_out"]).max() < 1e-12

    class Cfg:
        pass
    cfg = Cfg()
    cfg.NEB_FOLDER_DIRECTORY = None
    cfg.fix_init_edge = cfg.fix_end_edge = cfg.apply_convergence_criteria = False
    cfg.bohr2angstroms = cases.BOHR2ANG
    cfg.dt, cfg.a, cfg.n_reset = 0.5, 0.10, 0
    cfg.FIRE_N_accelerate, cfg.FIRE_f_inc, cfg.FIRE_f_accelerate = 5, 1.10, 0.99
    cfg.FIRE_f_decelerate, cfg.FIRE_a_start, cfg.FIRE_dt_max = 0.5, 0.1, 1.0
    fo = FIREOptimizer(cfg)
    nx = fo.optimize(z["tr_X"], z["tr_F"], z["fire_prevel"], 3, z["fire_vel"], None, None, None, None)
    assert np.abs(nx - z["fire_out"]).max() < 1e-12
    assert np.allclose([fo.dt, fo.a, fo.n_reset], z["fire_state"], rtol=0, atol=1e-15)


def _oracle_band_steps(X0, nsteps, literal=False, hessians=None):
    """The oracle's RFO-NEB loop on the LJ surface: yields (X, E, G, newX_angstrom) per step."""
    from oracle import neb as oneb
    I, N = X0.shape[0], X0.shape[1]
    calc = oneb.BNEB(literal=literal)
    opt = oneb.RFONEB([None] * I if hessians is None else hessians, literal=literal)
    X = X0.copy()
    preX = preG = None
    pre_vel = []
    for it in range(nsteps):
        E = np.zeros(I)
        Gr = np.zeros((I, N, 3))
        for i in range(I):
            E[i], Gr[i] = cases.lj(X[i])[:2]
        F = calc.calc_force(X, E, Gr, it, None)
        newX = np.array(opt.optimize(X, Gr, preX, preG, it, E, pre_vel, np.array(F), calc))
        yield X, E, Gr, newX
        preX, preG, pre_vel = X.copy(), Gr.copy(), np.array(F)
        X = newX / cases.BOHR2ANG


def test_band_headline_size_vs_oracle():
    """The BASELINE headline size, 32 images x 100 atoms (n = 300): three consecutive band steps against the oracle on the
    same inputs (the oracle itself is pinned by the reference's goldens at I = 8, N = 7)."""
    from multioptpy_amd.band import Band
    I, N = 32, 100
    band = Band(I, N)
    worst = 0.0
    for it, (X, E, Gr, ref) in enumerate(_oracle_band_steps(cases.band(I, N), 3)):
        band.upload(X, E, Gr)
        band.step(it)
        newX, _ = band.download(want_force=True)
        err = np.abs(newX - ref).max()
        worst = max(worst, err)
        assert err < 1e-8, (it, err)
    band.close()
    print("32 x 100 band: worst deviation from the oracle", worst, "angstrom")


@pytest.mark.parametrize("shape", [(3, 3), (4, 5), (3, 5), (3, 4), (9, 3)])
def test_band_tiny_shapes_vs_oracle(shape):
    """Small band shapes: one interior image (the smallest band the C ABI accepts is I = 3), triatomics (n = 9: three internal
    degrees of freedom).  (Diatomics are left out on purpose: there the reference's gradient projection takes a full 6 x 6
    orthogonal Q, the projected gradient is rounding noise (1e-23) and RSIRFO rescales that noise to the trust radius --
    the reference's own result is not reproducible to any tolerance.)"""
    from multioptpy_amd.band import Band
    I, N = shape
    rng = np.random.default_rng(100 * I + N)
    base = rng.normal(size=(N, 3)) * 4.0 + np.arange(N)[:, None] * 6.0
    X0 = np.array([base + 0.15 * i * rng.normal(size=(N, 3)) for i in range(I)])
    band = Band(I, N)
    for it, (X, E, Gr, ref) in enumerate(_oracle_band_steps(X0, 3, literal=False)):
        band.upload(X, E, Gr)
        band.step(it)
        newX, _ = band.download(want_force=True)
        assert np.abs(newX - ref).max() < 1e-8, (shape, it, np.abs(newX - ref).max())
    band.close()


def test_band_long_run_vs_oracle():
    """Twelve consecutive band steps (I = 8, N = 14, exact LJ Hessians to start with: dense spectra with clusters, then
    quasi-Newton updates) against the oracle: no drift, no solver failure, trust radii identical."""
    from multioptpy_amd.band import Band
    I, N = 8, 14
    X0 = cases.band(I, N)
    hess = [cases.lj(X0[i], hessian=True)[2] for i in range(I)]
    band = Band(I, N)
    for i in range(I):
        band.set_hessian(i, hess[i])
    worst = 0.0
    for it, (X, E, Gr, ref) in enumerate(_oracle_band_steps(X0, 12, hessians=[h.copy() for h in hess])):
        band.upload(X, E, Gr)
        band.step(it)
        newX, _ = band.download(want_force=True)
        err = np.abs(newX - ref).max()
        worst = max(worst, err)
        assert err < 1e-8, (it, err)
    band.close()
    print("12 steps: worst deviation", worst)
