"""The CPU oracle against the golden vectors captured from the reference (G1, G2/G4, G5, G6, G7)."""
import os

import numpy as np
import pytest

import cases
from oracle import hessian_update as hu
from oracle import projection as pj
from oracle import neb as oneb
from oracle.rsirfo import RSIRFO
from oracle.brent import brentq

G = os.path.join(os.path.dirname(__file__), "golden")


def test_g1_update_rules():
    z = np.load(os.path.join(G, "g1_updates.npz"))
    ncase = sum(1 for k in z.files if k.endswith("_delta"))
    assert ncase == 520
    worst = 0.0
    for i in range(ncase):
        tag = f"c{i:03d}"
        n, indef, seed = z[tag + "_meta"]
        H = cases.spd_matrix(int(n), seed=int(n) + int(indef), indefinite=bool(indef))
        s, y = cases.sy_pair(H, seed=int(seed), kind=str(z[tag + "_kind"]))
        d = hu.delta_hessian(str(z[tag + "_method"]), H, s, y)
        ref = z[tag + "_delta"]
        scale = max(1.0, np.abs(ref).max())
        err = np.abs(d - ref).max() / scale
        worst = max(worst, err)
        assert err < 1e-10, (tag, str(z[tag + "_method"]), str(z[tag + "_kind"]), err)
    assert worst < 1e-10


def _replay(z, tag):
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


def test_g2_rsirfo_traces():
    z = np.load(os.path.join(G, "g2_rsirfo.npz"))
    ncfg = int(z["ncfg"])
    assert ncfg == 36
    for i in range(ncfg):
        tag = f"t{i:02d}"
        for k, opt, step, Hcaller in _replay(z, tag):
            ref = z[f"{tag}_s{k}_step"]
            assert step.shape == ref.shape
            assert np.abs(step - ref).max() < 1e-9, (tag, k, np.abs(step - ref).max())
            assert abs(opt.trust_radius - float(z[f"{tag}_s{k}_tr"])) < 1e-12, (tag, k)
            pr = float(z[f"{tag}_s{k}_pred"])
            assert abs(opt.predicted_energy_changes[-1] - pr) < 1e-10 * max(1.0, abs(pr)), (tag, k)
            if f"{tag}_s{k}_Hself" in z.files:
                assert np.abs(opt.hessian - z[f"{tag}_s{k}_Hself"]).max() < 1e-9, (tag, k)
                assert np.abs(Hcaller - z[f"{tag}_s{k}_Hcaller"]).max() < 1e-9, (tag, k)
            # alpha / lambda_aug of the final solve
            sl = z[f"{tag}_s{k}_solve"]
            if f"{tag}_s{k}_brent_root" in z.files:
                assert opt.trace["branch"] in ("brent", "newton")
                assert abs(opt.trace["alpha"] - float(z[f"{tag}_s{k}_brent_root"])) < 1e-9 or opt.trace["branch"] == "newton"
            assert abs(opt.trace["lambda_aug0"] - sl[0, 1]) < 1e-9 * max(1.0, abs(sl[0, 1]))


def test_g4_brentq_iterates():
    """Every (x, f(x)) the real scipy.optimize.brentq evaluated is reproduced by oracle.brent
    (call-site tolerances of rsirfo.py:1042 and tight ones).  Inside the reference's RSIRFO the
    bracket test never succeeds (the step norm does not depend on alpha), so brentq is pinned here."""
    from make_golden import BRENT_FUNCS
    z = np.load(os.path.join(G, "g4_brentq.npz"))
    for name, (f, a, b) in BRENT_FUNCS.items():
        for tname, kw in (("loose", dict(xtol=1e-6, rtol=1e-6, maxiter=50)), ("tight", dict(xtol=1e-12, rtol=1e-12, maxiter=100))):
            tr = []
            r = brentq(f, a, b, trace=tr, **kw)
            ref = z[f"{name}_{tname}_evals"]
            mine = np.array(tr)
            assert mine.shape == ref.shape, (name, tname, mine.shape, ref.shape)
            assert np.array_equal(mine[:, 0], ref[:, 0]), (name, tname)
            assert r == float(z[f"{name}_{tname}_root"])


def test_brentq_textbook():
    r = brentq(lambda t: t ** 3 - 2 * t - 5, 2.0, 3.0, xtol=1e-12, rtol=1e-12)
    assert abs(r - 2.0945514815423265) < 1e-11


def test_g5_projections():
    z = np.load(os.path.join(G, "g5_projections.npz"))
    for name in ("ar7", "ar22", "linear_z", "diatomic"):
        x, H, g = z[name + "_x"], z[name + "_H"], z[name + "_g"]
        Hp = pj.project_hessian_trrot(H, x)
        assert np.abs(Hp - z[name + "_Hp"]).max() < 1e-12, name
        gp = pj.project_gradient_trrot(g, x)
        assert np.abs(gp - z[name + "_gp"]).max() < 1e-12, name
    P, Q = z["kabsch_P"].copy(), z["kabsch_Q"].copy()
    Pk, Qk = pj.kabsch(P, Q)
    assert np.abs(Pk - z["kabsch_Pout"]).max() < 1e-12
    assert np.abs(Qk - z["kabsch_Qout"]).max() < 1e-12


@pytest.mark.parametrize("literal", [False, True])
def test_g6_bneb_force(literal):
    z = np.load(os.path.join(G, "g6_neb_force.npz"))
    for k in range(int(z["ncase"])):
        tag = f"f{k:02d}"
        X, E, Gr = z[tag + "_X"], z[tag + "_E"], z[tag + "_G"]
        calc = oneb.BNEB(APPLY_CI_NEB=(2 if int(z[tag + "_ci"]) else 99999), literal=literal)
        F = calc.calc_force(X, E, Gr, 5)
        assert np.abs(F - z[tag + "_F"]).max() < 1e-12, (tag, np.abs(F - z[tag + "_F"]).max())
        assert np.abs(np.array(calc.tau_list) - z[tag + "_tau"]).max() < 1e-12
        for i in range(1, 7):
            gam = calc.calculate_gamma(X[i - 1:i + 2], E[i - 1:i + 2], Gr[i - 1:i + 2], calc.get_tau(i))
            ref = z[tag + "_gamma"][i]
            assert abs(gam - ref) < 1e-9 * max(1.0, abs(ref)), (tag, i, gam, ref)


def test_g7_tr_and_fire():
    z = np.load(os.path.join(G, "g7_neb_step.npz"))
    mv = oneb.tr_calc(z["tr_X"], z["tr_F"], list(z["tr_delta"]))
    assert np.abs(np.array(mv) - z["tr_out"]).max() < 1e-13
    f = oneb.FIRE()
    nx = f.optimize(z["tr_X"], z["tr_F"], z["fire_prevel"], 3, z["fire_vel"])
    assert np.abs(nx - z["fire_out"]).max() < 1e-12
    assert np.allclose([f.dt, f.a, f.n_reset], z["fire_state"])


@pytest.mark.parametrize("variant", ["ident", "exact"])
def test_g7_rfo_neb_steps(variant):
    z = np.load(os.path.join(G, "g7_neb_step.npz"))
    I, N = 8, 7
    X0 = z[f"{variant}_it0_X"]
    hess = [None] * I if variant == "ident" else [cases.lj(X0[i], hessian=True)[2] for i in range(I)]
    opt = oneb.RFONEB(hess)
    calc = oneb.BNEB()
    preX = preG = None
    pre_vel = []
    for it in range(4):
        tag = f"{variant}_it{it}"
        X, E, Gr = z[tag + "_X"], z[tag + "_E"], z[tag + "_G"]
        F = calc.calc_force(X, E, Gr, it)
        assert np.abs(F - z[tag + "_F"]).max() < 1e-12
        newX = opt.optimize(X, Gr, preX, preG, it, E, pre_vel, F.copy(), calc)
        assert np.abs(newX - z[tag + "_newX"]).max() < 1e-9, (tag, np.abs(newX - z[tag + "_newX"]).max())
        assert np.abs(np.array(opt.hessians) - z[tag + "_hess"]).max() < 1e-9
        assert np.allclose([o.trust_radius for o in opt.opts], z[tag + "_tr"], atol=1e-12)
        preX, preG, pre_vel = X.copy(), Gr.copy(), F.copy()


@pytest.mark.parametrize("literal", [False, True])
def test_g6b_neb_variants(literal):
    z = np.load(os.path.join(G, "g6b_neb_variants.npz"))
    for k in range(int(z["ncase"])):
        tag = f"v{k:02d}"
        X, E, Gr = z[tag + "_X"], z[tag + "_E"], z[tag + "_G"]
        apply = 2 if int(z[tag + "_ci"]) else 99999
        F = oneb.BNEB3(apply, literal=literal).calc_force(X, E, Gr, 5)
        assert np.abs(F - z[tag + "_bneb3"]).max() < 1e-12, (tag, "bneb3")
        F = oneb.EWBNEB(apply, literal=literal).calc_force(X, E, Gr, 5)
        assert np.abs(F - z[tag + "_ewbneb"]).max() < 1e-12, (tag, "ewbneb")
        if not literal:
            F = oneb.DNEB(apply).calc_force(X, E, Gr, 5)
            assert np.abs(F - z[tag + "_dneb"]).max() < 1e-12, (tag, "dneb", np.abs(F - z[tag + "_dneb"]).max())


def _replay_rsprfo(z, tag, cls):
    N, so, nsteps, seed = [int(v) for v in z[tag + "_meta"]]
    method, mode = str(z[tag + "_method"]), str(z[tag + "_mode"])
    n = 3 * N
    H0 = cases.lj(cases.cluster(N, seed_offset=seed), hessian=True)[2]
    opt = cls(method=method, saddle_order=so, trust_radius_max=0.3 if so > 0 else 0.5, trust_radius_min=0.01)
    px = pg = pE = pmove = None
    for k in range(nsteps):
        x, g, E = z[f"{tag}_s{k}_x"], z[f"{tag}_s{k}_g"], float(z[f"{tag}_s{k}_E"])
        if mode == "every" or k == 0:
            opt.set_hessian(H0)
            opt.set_bias_hessian(np.zeros((n, n)))
        if k == 0:
            step = opt.run(x, g, [], [], E, 0.0, [], [], g, [])
        else:
            step = opt.run(x, g, pg, px, E, pE, pmove, [], g, pg)
        yield k, opt, step
        px, pg, pE, pmove = x, g, E, z[f"{tag}_s{k}_step"]


def test_g3_rsprfo_traces():
    from oracle.rsprfo import EnhancedRSPRFO
    z = np.load(os.path.join(G, "g3_rsprfo.npz"))
    for i in range(int(z["ncfg"])):
        tag = f"p{i:02d}"
        for k, opt, step in _replay_rsprfo(z, tag, EnhancedRSPRFO):
            ref = z[f"{tag}_s{k}_step"]
            assert np.abs(step - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), (tag, k, np.abs(step - ref).max())
            assert abs(opt.trust_radius - float(z[f"{tag}_s{k}_tr"])) < 1e-12, (tag, k)
            pr = float(z[f"{tag}_s{k}_pred"])
            assert abs(opt.predicted_energy_changes[-1] - pr) < 1e-9 * max(1.0, abs(pr)), (tag, k)
            assert np.abs(opt.hessian - z[f"{tag}_s{k}_H"]).max() < 1e-9, (tag, k)
            ts = int(z[f"{tag}_s{k}_tsidx"])
            if opt.saddle_order == 1:      # order 2: both shifted modes sit at -0.001, the tracked index is gauge dependent
                assert (opt.ts_mode_idx is None and ts == -1) or int(opt.ts_mode_idx) == ts, (tag, k)


def test_g9_outer_trust_radius():
    """a15: TrustRadius.update_trust_radii sequences."""
    from oracle.driver import TrustRadius
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
            tr = TR.update_trust_radii(e1, e0, z[f"c{c}_k{k}_g"], z[f"c{c}_k{k}_mv"], H, np.zeros(int(n)), tr)
            assert abs(tr - z[f"c{c}_tr"][k]) < 1e-14, (c, k)


def _check_optmain(run, z, tag, tol):
    import cases
    argv = str(z[tag + "_argv"]).split()
    method = argv[argv.index("-opt") + 1]
    order = int(argv[argv.index("-order") + 1]) if "-order" in argv else 0
    fc = int(argv[argv.index("-fc") + 1]) if "-fc" in argv else -1
    nstep = int(argv[argv.index("-ns") + 1])
    x0 = z[tag + "_x0_ang"]
    # the start file holds 10 decimals
    x0 = np.array([[float(f"{v:.10f}") for v in r] for r in x0])
    rec = []
    if "-pc" in argv:                                          # `-pc name atoms name atoms ...` (interface.py:466-475)
        i = argv.index("-pc") + 1
        conds, atoms = [], []
        while i + 1 < len(argv) and not argv[i].startswith("-"):
            conds.append(argv[i]); atoms.append([int(a) for a in argv[i + 1].split(",")]); i += 2
        run(x0, ["Ar"] * len(x0), method, order, nstep, fc, rec, dict(conditions=conds, atoms=atoms))
    else:
        run(x0, ["Ar"] * len(x0), method, order, nstep, fc, rec)
    assert len(rec) == int(z[tag + "_nit"])
    worst = 0.0
    for k, r in enumerate(rec):
        sc = z[f"{tag}_i{k}_sc"]
        assert np.abs(r["geom"] - z[f"{tag}_i{k}_geom"]).max() < tol, (tag, k, "geom")
        assert abs(r["B_e"] - sc[0]) < 1e-12 and abs(r["tr"] - sc[2]) < 1e-12, (tag, k, "scalars")
        d = np.abs(r["mv"] - z[f"{tag}_i{k}_mv"]).max()
        worst = max(worst, d)
        assert d < tol, (tag, k, "move_vector", d)
        assert np.abs(r["new"] - z[f"{tag}_i{k}_new"]).max() < tol, (tag, k, "new_geometry")
        assert np.abs(r["H"] - z[f"{tag}_i{k}_H"]).max() < 1e-8, (tag, k, "hessian")
        cv = z[f"{tag}_i{k}_conv"]
        assert bool(cv[0]) == r["conv"][0] and abs(cv[1] - r["conv"][1]) < 1e-12 and abs(cv[2] - r["conv"][2]) < 1e-12
        assert np.abs(r["disp"] - z[f"{tag}_i{k}_disp"]).max() < tol, (tag, k, "displacement")
    return worst


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_g8_optmain_driver(tag):
    """a15/a20/a23: the oracle's in-memory optmain iteration against the reference driver's own trajectory (d: `-opt crsirfo_bfgs -pc bond
    1,2 angle 3,4,5` -- the constrained optimizer through the dispatcher)."""
    import cases
    from oracle import driver, projection as pj
    z = np.load(os.path.join(G, "g8_optmain.npz"))

    def run(x0, elems, method, order, nstep, fc, rec, pc=None):
        calc = cases.LJCalculator(FC_COUNT=fc, project=pj.project_hessian_trrot)
        driver.run_optmain(calc, x0, elems, method, order, nstep, fc, record=rec, projection_constraint=pc)

    worst = _check_optmain(run, z, tag, 1e-9)
    print("optmain", tag, "worst move-vector deviation", worst)


def _check_nebmain(run, z, tag, tol):
    argv = str(z[tag + "_argv"]).split()
    fc = int(argv[argv.index("-fc") + 1]) if "-fc" in argv else -1
    nstep = int(argv[argv.index("-ns") + 1])
    p = int(argv[argv.index("-p") + 1])
    fe = int(argv[argv.index("-fe") + 1]) if "-fe" in argv else 0
    sd = int(argv[argv.index("-sd") + 1]) if "-sd" in argv else 99999
    ends = z[tag + "_ends_bohr"] * cases.BOHR2ANG
    ends = np.array([[[float(f"{v:.10f}") for v in r] for r in x] for x in ends])       # the start files hold 10 decimals
    rec = []
    extra = {}
    if "-ad" in argv:
        extra["align_distances"] = int(argv[argv.index("-ad") + 1])
    if "-ci" in argv:
        extra["climbing_image"] = (int(argv[argv.index("-ci") + 1]), int(argv[argv.index("-ci") + 2]))
    if "-ma" in argv:                                          # `-ma gamma fragm_1 fragm_2` (AFIR, kJ/mol; 1-based atoms)
        i = argv.index("-ma")
        extra["afir"] = dict(gamma=float(argv[i + 1]), fragm_1=[int(a) for a in argv[i + 2].split(",")], fragm_2=[int(a) for a in argv[i + 3].split(",")])
    run(ends, fc, nstep, p, fe, rec, sd, **extra)
    assert len(rec) == int(z[tag + "_nit"])
    worst = 0.0
    for k, r in enumerate(rec):
        assert np.abs(r["X"] - z[f"{tag}_i{k}_X"]).max() < tol, (tag, k, "geometry read by the engine")
        if f"{tag}_i{k}_fire" in z.files and r.get("fire") is not None and z[f"{tag}_i{k}_fire"][0] > 0:
            assert np.allclose(r["fire"], z[f"{tag}_i{k}_fire"], rtol=0, atol=1e-14), (tag, k, "FIRE dt / a / n_reset")
        assert np.abs(r["E"] - z[f"{tag}_i{k}_E"]).max() < 1e-10, (tag, k, "energies")
        d = np.abs(r["opt"] - z[f"{tag}_i{k}_opt"]).max()
        worst = max(worst, d)
        assert d < tol, (tag, k, "optimizer output", d)
        assert np.abs(r["post"] - z[f"{tag}_i{k}_post"]).max() < tol, (tag, k, "after constraints + Kabsch chain")
        for i in range(len(r["hess"])):
            key = f"{tag}_i{k}_hout{i}"
            if key in z.files:
                assert np.abs(r["hess"][i] - z[key]).max() < 1e-8, (tag, k, i, "Hessian after the step")
    return worst


@pytest.mark.parametrize("tag", ["n", "e", "f", "s", "a", "c", "m"])
def test_g8_nebmain_driver(tag):
    """a23: the oracle's in-memory nebmain iteration (RFO branch; FIRE branch; FIRE then steepest descent; RFO branch with image
    re-distribution `-ad 2` and with the climbing-image move `-ci 1 2`) against the reference driver's own band trajectory."""
    from oracle import driver

    def run(ends, fc, nstep, p, fe, rec, sd, afir=None, **extra):
        eng = cases.LJEngine(FC_COUNT=fc, project=pj.project_hessian_trrot)
        bias = None
        if afir is not None:                                   # (m: `-ma`, the per-image bias of neb.py:806-832 with the oracle's analytic AFIR, pinned by G11)
            from oracle import afir as oafir, model_hessian as mh
            cov = mh.element_params(["Ar"] * ends.shape[1])["cov"]
            bias = lambda x: oafir.afir(x, cov, afir["fragm_1"], afir["fragm_2"], afir["gamma"])
        driver.run_nebmain(eng, ends, ["Ar"] * ends.shape[1], nstep, p, fc, fe, record=rec, literal=True, sd_from=sd, bias=bias, **extra)

    worst = _check_nebmain(run, np.load(os.path.join(G, "g8_nebmain.npz")), tag, 1e-9)
    print("nebmain", tag, "worst optimizer-output deviation", worst)


def test_g10_model_hessian_oracle_vs_reference():
    """G10: FischerD3ApproxHessianOld.main / ApproxHessian.main of the reference (organic test geometries, argon clusters, a
    distorted geometry, linear CO2) against the oracle's restatement."""
    from oracle import model_hessian as mh
    z = np.load(os.path.join(G, "g10_model_hessian.npz"))
    for t in z["tags"]:
        el = [str(e) for e in z[t + "_elems"]]
        H = mh.approx_hessian(z[t + "_x"], el, None, "fischerd3old")
        assert np.abs(H - z[t + "_H"]).max() < 1e-13, t
        assert np.array_equal(z[t + "_H"], z[t + "_Hdisp"]), t
    with pytest.raises(NotImplementedError):
        mh.approx_hessian(z["co2_x"], ["O", "C", "O"], None, "swartd3")          # (not restated)


def test_g10b_model_hessian_variants_oracle_vs_reference():
    """G10b: FischerD3ApproxHessian.main (coordination-number scaled D3) and FischerApproxHessian.main (no dispersion) of the
    reference, through the classes and through ApproxHessian.main's substring dispatch, against the oracle's restatements; the
    three Fischer models are told apart by the data (they differ by far more than the tolerance)."""
    from oracle import model_hessian as mh
    z = np.load(os.path.join(G, "g10b_model_hessian_variants.npz"))
    zo = np.load(os.path.join(G, "g10_model_hessian.npz"))
    for t in z["tags"]:
        el = [str(e) for e in z[t + "_elems"]]
        H3 = mh.approx_hessian(z[t + "_x"], el, None, "fischerd3")
        Hp = mh.approx_hessian(z[t + "_x"], el, None, "fischer")
        assert np.abs(H3 - z[t + "_Hd3"]).max() < 1e-13, t
        assert np.abs(Hp - z[t + "_Hplain"]).max() < 1e-13, t
        Hs = mh.approx_hessian(z[t + "_x"], el, None, "schlegel")
        assert np.abs(Hs - z[t + "_Hsch"]).max() < 1e-13, t
        assert np.array_equal(z[t + "_Hsch"], z[t + "_Hschdisp"]), t
        assert np.abs(mh.approx_hessian(z[t + "_x"], el, None, "morse") - z[t + "_Hmorse"]).max() < 1e-13, t
        assert np.array_equal(z[t + "_Hd3"], z[t + "_Hd3disp"]) and np.array_equal(z[t + "_Hplain"], z[t + "_Hplaindisp"]), t
    for t in ("aldol_rxn", "claisen_rearrengment", "ar13"):
        assert np.abs(z[t + "_Hd3"] - zo[t + "_H"]).max() > 1e-5, t
        assert np.abs(z[t + "_Hplain"] - zo[t + "_H"]).max() > 1e-5, t


def test_g14_keep_pot_oracle_vs_reference():
    """G14: StructKeepPotential energy, gradient and Hessian of the reference (torch autograd) against the oracle's analytic forms."""
    from oracle import keep_pot as kp
    z = np.load(os.path.join(G, "g14_keep_pot.npz"))
    for t in z["tags"]:
        E, g, H = kp.keep_pot(z[t + "_x"], [int(v) for v in z[t + "_pair"]], float(z[t + "_k"]), float(z[t + "_r0"]))
        assert abs(E - float(z[t + "_E"])) < 1e-13 and np.abs(g - z[t + "_g"]).max() < 1e-13 and np.abs(H - z[t + "_H"]).max() < 1e-12, t
    for t in z["atags"]:
        E, g, H = kp.anharmonic_keep_pot(z[t + "_x"], [int(v) for v in z[t + "_pair"]], float(z[t + "_k"]), float(z[t + "_D"]), float(z[t + "_r0"]))
        assert abs(E - float(z[t + "_E"])) < 1e-13 and np.abs(g - z[t + "_g"]).max() < 1e-13 and np.abs(H - z[t + "_H"]).max() < 1e-12, t


def test_g11_afir_oracle_vs_reference():
    """G11: AFIR energy, gradient and Hessian (the reference's torch autograd outputs) against the oracle's analytic form."""
    from oracle import afir, model_hessian as mh
    z = np.load(os.path.join(G, "g11_afir.npz"))
    for t in z["tags"]:
        el = [str(e) for e in z[t + "_elems"]]
        cov = mh.element_params(el)["cov"]
        E, g, H = afir.afir(z[t + "_x"], cov, list(z[t + "_f1"]), list(z[t + "_f2"]), float(z[t + "_gamma"]))
        assert abs(E - float(z[t + "_E"])) < 1e-14 and np.abs(g - z[t + "_g"]).max() < 1e-14 and np.abs(H - z[t + "_H"]).max() < 1e-14, t


def _g12():
    z = np.load(os.path.join(G, "g12_constraints.npz"))
    names = [str(s) for s in z["names"]]
    atoms, k = [], 0
    for ln in z["atoms_len"]:
        atoms.append([int(a) for a in z["atoms_flat"][k:k + int(ln)]])
        k += int(ln)
    return z, names, atoms


def test_g12_constraints_oracle_vs_reference():
    """G12: ProjectOutConstrain (initial values, SHAKE-like adjustment, constraint rows up to sign, gradient projection, three Hessian
    projections with history) and three CRSIRFO.run calls of the reference against the oracle."""
    import contextlib
    import io
    from oracle import constraint as oc
    z, names, atoms = _g12()
    pc = oc.ProjectOutConstrain(names, atoms)
    assert np.abs(np.array(pc.initialize(z["x0"])) - z["init"]).max() < 1e-14
    for k in range(3):
        x, g, H = z[f"p{k}_x"], z[f"p{k}_g"], z[f"p{k}_Hin"]
        B = pc.constraint_vectors(x)
        for i in range(len(B)):
            assert min(np.abs(B[i] - z[f"p{k}_B"][i]).max(), np.abs(B[i] + z[f"p{k}_B"][i]).max()) < 1e-12
        assert np.abs(pc.calc_project_out_grad(x, g) - z[f"p{k}_gproj"]).max() < 1e-13
        assert np.abs(pc.calc_project_out_hess(x, g, H) - z[f"p{k}_Hproj"]).max() < 1e-11
    assert np.abs(pc.adjust_init_coord(z["shake_in"].copy()) - z["shake_out"]).max() < 1e-13
    pc2 = oc.ProjectOutConstrain(names[:2], atoms[:2])
    pc2.initialize(z["x0"])
    opt = oc.CRSIRFO(constraints=pc2, method="rsirfo_fsb", saddle_order=0)
    K = z["K"]
    px = pg = None
    pE = 0.0
    for it in range(3):
        x, g, E = z[f"c{it}_x"], z[f"c{it}_g"], float(z[f"c{it}_E"])
        opt.set_bias_hessian(np.zeros_like(K))
        with contextlib.redirect_stdout(io.StringIO()):
            if it == 0:
                opt.set_hessian(K.copy())
                s = opt.run(x.reshape(-1, 1).copy(), g.copy(), [], [], E, 0.0, [], [], g.copy(), [])
            else:
                s = opt.run(x.reshape(-1, 1).copy(), g.copy(), pg, px, E, pE, [], [], g.copy(), pg)
        assert np.abs(s - z[f"c{it}_step"]).max() < 1e-9, it
        assert abs(opt.trust_radius - float(z[f"c{it}_tr"])) < 1e-12 and np.abs(opt.hessian - z[f"c{it}_H"]).max() < 1e-12
        px, pg, pE = x.reshape(-1, 1).copy(), g.copy(), E


def test_g12_constraint_host_parts_of_the_plugin_class():
    """The host side of multioptpy_amd's ProjectOutConstrain (no GPU: rows, basis, gradient projection, SHAKE-like adjustment)."""
    from multioptpy_amd.Constraint.constraint_condition import ProjectOutConstrain
    z, names, atoms = _g12()
    pc = ProjectOutConstrain(names, atoms)
    assert np.abs(np.array(pc.initialize(z["x0"])) - z["init"]).max() < 1e-14
    for k in range(3):
        B = pc._get_all_constraint_vectors(z[f"p{k}_x"])
        for i in range(len(B)):
            assert min(np.abs(B[i] - z[f"p{k}_B"][i]).max(), np.abs(B[i] + z[f"p{k}_B"][i]).max()) < 1e-12
        assert np.abs(pc.calc_project_out_grad(z[f"p{k}_x"], z[f"p{k}_g"]) - z[f"p{k}_gproj"]).max() < 1e-13
    assert np.abs(pc.adjust_init_coord(z["shake_in"].copy()) - z["shake_out"]).max() < 1e-13
    with pytest.raises(NotImplementedError):
        ProjectOutConstrain(["no_such_kind"], [[1, 2, 3]])


def _g12b_sets(z):
    for si in range(int(z["nsets"])):
        tag = f"k{si}"
        names = [str(s) for s in z[tag + "_names"]]
        atoms, k = [], 0
        for ln in z[tag + "_atoms_len"]:
            atoms.append([int(a) for a in z[tag + "_atoms_flat"][k:k + int(ln)]])
            k += int(ln)
        yield tag, names, atoms


def _check_g12b(z, make, rows_of, tol_h):
    """ProjectOutConstrain objects built by make(names, atoms) against G12b: initial values, rows (up to sign), projections, adjustment."""
    special = ("rot", "eigvec", "atoms_pair")
    for tag, names, atoms in _g12b_sets(z):
        pc = make(names, atoms)
        xinit = z["x0"].copy()
        init = pc.initialize(xinit, hessian=z["Hm"])
        sc = np.array([float(v) for v, nm in zip(init, names) if nm not in special])
        assert sc.size == z[tag + "_init_scalars"].size and (sc.size == 0 or np.abs(sc - z[tag + "_init_scalars"]).max() < 1e-13), tag
        for k in range(2):
            x, g, H = z[f"{tag}_p{k}_x"], z[f"{tag}_p{k}_g"], z[f"{tag}_p{k}_Hin"]
            B = rows_of(pc, x)
            assert B.shape == z[f"{tag}_p{k}_B"].shape, tag
            for i in range(len(B)):
                assert min(np.abs(B[i] - z[f"{tag}_p{k}_B"][i]).max(), np.abs(B[i] + z[f"{tag}_p{k}_B"][i]).max()) < 1e-10, (tag, k, i)
            assert np.abs(pc.calc_project_out_grad(x.copy(), g.copy()) - z[f"{tag}_p{k}_gproj"]).max() < 1e-11, (tag, k)
            if tol_h is not None:
                assert np.abs(pc.calc_project_out_hess(x.copy(), g.copy(), H.copy()) - z[f"{tag}_p{k}_Hproj"]).max() < tol_h, (tag, k)
        out = pc.adjust_init_coord(z[tag + "_shake_in"].copy(), hessian=z["Hm"])
        assert np.abs(out - z[tag + "_shake_out"]).max() < 1e-10, tag
        assert pc.last_shake_iter == int(z[tag + "_shake_iters"]), tag
        assert np.abs(xinit - z[tag + "_xinit_after"]).max() < 1e-12, tag           # the stored initial geometry was centred in place


def test_g12b_constraint_kinds_oracle_vs_reference():
    """G12b: fbond / rot / eigvec / atoms_pair (five constraint sets on the aldol geometry) -- the oracle against the reference's outputs."""
    from oracle import constraint as oc
    z = np.load(os.path.join(G, "g12b_constraint_kinds.npz"))
    _check_g12b(z, lambda names, atoms: oc.ProjectOutConstrain(names, atoms), lambda pc, x: pc.constraint_vectors(x), 1e-9)


def test_g13_path_tools_oracle_and_host_glue_vs_reference():
    """G13: path length, equal-arc-length re-distribution, natural-spline extrema, IDPP objective / gradient / three-image relaxation
    and the climbing-image move -- the oracle's loops AND the package's host glue (numpy on (I, N, 3) arrays, as in the reference;
    no device code involved) against the reference's outputs."""
    from oracle import path_tools as pt
    from multioptpy_amd.Utils.calc_tools import calc_path_length_list, apply_climbing_image
    from multioptpy_amd.Interpolation.linear_interpolation import distribute_geometry
    from multioptpy_amd.Interpolation.interpolation import spline_interpolation
    from multioptpy_amd.Potential.idpp import IDPP
    z = np.load(os.path.join(G, "g13_path_tools.npz"))
    X, Xu, E = z["X"], z["Xu"], z["E"]
    for f in (pt.path_lengths, calc_path_length_list):
        assert np.abs(np.array(f(list(X))) - z["path_length"]).max() < 1e-14
    for f in (pt.redistribute, distribute_geometry):
        assert np.abs(np.array(f(list(X))) - z["distributed"]).max() < 1e-13
        assert np.abs(np.array(f(list(Xu))) - z["distributed_u"]).max() < 1e-13
    for f in (pt.spline_extrema, spline_interpolation):
        mx, mn = f(z["path_length"], E)
        assert np.abs(np.array(mx).reshape(-1, 2) - z["spline_max"]).max() < 1e-12
        assert np.abs(np.array(mn).reshape(-1, 2) - z["spline_min"]).max() < 1e-12
    trio = [t.copy() for t in z["trio"]]
    for f in (lambda im: pt.idpp_value_gradient(im, 1), lambda im: IDPP().get_func_and_deriv(im, 3, 1)):
        val, grad = f(trio)
        assert abs(val - float(z["idpp_obj"])) < 1e-12 * max(1.0, abs(float(z["idpp_obj"])))
        assert np.abs(grad - z["idpp_grad"]).max() < 1e-12 * max(1.0, np.abs(z["idpp_grad"]).max())
    rel_o = pt.idpp_relax([t.copy() for t in z["trio"]])
    rel_p = IDPP().opt_path([t.copy() for t in z["trio"]], ["Ar"] * 7)
    print("IDPP relaxation: oracle", np.abs(np.array(rel_o) - z["idpp_relaxed"]).max(), "package", np.abs(np.array(rel_p) - z["idpp_relaxed"]).max())
    assert np.abs(np.array(rel_o) - z["idpp_relaxed"]).max() < 1e-9
    assert np.abs(np.array(rel_p) - z["idpp_relaxed"]).max() < 1e-9
    ci_o = pt.climbing_image_move(list(X), E)
    ci_p = apply_climbing_image([x.copy() for x in X], list(E), ["Ar"] * 7)
    moved = [i for i in range(len(X)) if np.abs(z["ci_out"][i] - X[i]).max() > 0]
    assert moved == [3, 6]
    print("climbing-image move: oracle", np.abs(np.array(ci_o) - z["ci_out"]).max(), "package", np.abs(np.array(ci_p) - z["ci_out"]).max())
    assert np.abs(np.array(ci_o) - z["ci_out"]).max() < 1e-9
    assert np.abs(np.array(ci_p) - z["ci_out"]).max() < 1e-9


def test_oracle_lindh2007d3_vs_reference():
    """G10c: the oracle's Lindh2007D3 restatement (all pairs / triples / quadruples, the reference's triangle-wise accumulation and its
    mirroring rule, projection, |lambda| scaling by the gradient norm) against Lindh2007D3ApproxHessian.main, the dispatcher by name,
    the dispatcher's fall-back for an unknown name, and the `clip` modifier."""
    import os
    import numpy as np
    from oracle import model_hessian as mh
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10c_lindh2007d3.npz"))
    for tag in z["tags"]:
        x, el, g = z[tag + "_x"], [str(e) for e in z[tag + "_elems"]], z[tag + "_g"]
        assert np.array_equal(z[tag + "_H"], z[tag + "_Hdisp"]) and np.array_equal(z[tag + "_H"], z[tag + "_Hfallback"])
        if len(el) > 15:
            continue                                        # (the Python quadruple loops: kept to seconds)
        H = mh.approx_hessian(x, el, g, "lindh2007d3")
        assert np.abs(H - z[tag + "_H"]).max() < 1e-12, tag
        assert np.abs(mh.approx_hessian(x, el, g, "some_unknown_name") - z[tag + "_Hfallback"]).max() < 1e-12, tag
        assert np.abs(mh.approx_hessian(x, el, g, "lindh2007d3_clip") - z[tag + "_Hclip"]).max() < 1e-12, tag


def test_oracle_sr_modifier_and_modifier_order_vs_reference():
    """G10e: the `sr` modifier (ShortRangeCorrectionHessian.main) and combinations of ts / sr / clip in the reference's order, through the
    oracle's dispatcher by name, against ApproxHessian.main of the reference."""
    import os
    import numpy as np
    from oracle import model_hessian as mh
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10e_sr_modifier.npz"))
    worst = 0.0
    for tag in z["tags"]:
        x, el, g = z[tag + "_x"], [str(e) for e in z[tag + "_elems"]], z[tag + "_g"]
        for name in (str(v) for v in z["names"]):
            if "lindh" in name and len(el) > 15:
                continue                                    # (the Python quadruple loops: kept to seconds)
            ref = z[f"{tag}_H_{name}"]
            if ref.size == 0:
                continue                                    # (the reference's own IndexError, see G10d)
            H = mh.approx_hessian(x, el, g, name)
            d = np.abs(H - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, d)
            assert d < (1e-8 if "ts" in name else 1e-10), (tag, name, d)      # (ts: the reflected eigenvector of a near-degenerate pair amplifies 1e-16)
    print("oracle sr / modifier order vs the reference: worst relative deviation", worst)


def _check_g14c(fn, tol):
    """fn(kind, x, list1, list2, wall energy kJ/mol, limits angstrom, point, direction) -> (E, g, H) against G14c (the reference's torch autograd outputs)."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14c_well_potentials.npz"))
    worst = 0.0
    for i in range(int(z["ncase"])):
        tag = f"w{i:02d}"
        kind = str(z[tag + "_kind"])
        point = z[tag + "_point"] if (tag + "_point") in z.files else np.zeros(3)
        direction = {"x": 0, "y": 1, "z": 2}[str(z[tag + "_dir"])] if (tag + "_dir") in z.files else 0
        E, g, H = fn(kind, z[tag + "_x"], [int(a) for a in z[tag + "_l1"]], [int(a) for a in z[tag + "_l2"]], float(z[tag + "_e0"]), list(z[tag + "_lim"]), point, direction)
        refH = z[tag + "_H"]
        sc = max(1.0, np.abs(refH).max())
        assert abs(E - float(z[tag + "_E"])) < tol * max(1.0, abs(float(z[tag + "_E"]))), (tag, kind, E, float(z[tag + "_E"]))
        assert np.abs(g - z[tag + "_g"]).max() < tol * sc, (tag, kind, np.abs(g - z[tag + "_g"]).max())
        assert np.abs(H - refH).max() < tol * sc, (tag, kind, np.abs(H - refH).max())
        worst = max(worst, np.abs(H - refH).max() / sc, np.abs(g - z[tag + "_g"]).max() / sc)
    return worst


def test_g14c_well_potentials_oracle_vs_reference():
    """G14c: WellPotential / WellPotentialWall / WellPotentialVP / WellPotentialAround -- the oracle's hand-written radial derivatives against the
    reference's torch.func.jacrev / hessian outputs, every region of the well."""
    from oracle import well_pot
    worst = _check_g14c(well_pot.well_potential, 1e-10)
    print("oracle well potentials vs the reference: worst deviation", worst)


def _replay_mf(z, tag, make):
    """Feeds the recorded inputs of a G15 trace to an MF_RSIRFO object built by make(method, saddle_order, elems, masses, kw); yields (k, opt, step)."""
    N, so, nst = (int(v) for v in z[tag + "_meta"])
    n = 3 * N
    method = str(z[tag + "_method"])
    kw = eval(str(z[tag + "_kw"]), {"__builtins__": {}}, {"True": True, "False": False})      # a literal dict written by make_golden.py
    elems = [str(e) for e in z[tag + "_elems"]] or None
    masses = z[tag + "_masses"] if z[tag + "_masses"].size else None
    opt = make(method, so, elems, masses, kw)
    H = z[tag + "_H0"].copy()
    px = pg = pE = None
    for k in range(nst):
        x, g, E = z[f"{tag}_s{k}_x"], z[f"{tag}_s{k}_g"], float(z[f"{tag}_s{k}_E"])
        opt.set_hessian(H)
        opt.set_bias_hessian(np.zeros((n, n)))
        if k == 0:
            step = opt.run(x.copy(), g.copy(), [], [], E, 0.0, [], [], g.copy(), [])
        else:
            step = opt.run(x.copy(), g.copy(), pg, px, E, pE, [], [], g.copy(), pg)
        yield k, opt, np.asarray(step), H
        px, pg, pE = x, g, E


def _check_mf(z, make, tol):
    worst = 0.0
    for i in range(int(z["ncfg"])):
        tag = f"m{i:02d}"
        for k, opt, step, H in _replay_mf(z, tag, make):
            ref = z[f"{tag}_s{k}_step"]
            err = np.abs(step - ref).max()
            worst = max(worst, err)
            assert err < tol * max(1.0, np.abs(ref).max()), (tag, k, err)
            assert abs(opt.trust_radius - float(z[f"{tag}_s{k}_tr"])) < 1e-12, (tag, k)
            pr = float(z[f"{tag}_s{k}_pred"])
            assert abs(opt.predicted_energy_changes[-1] - pr) < tol * max(1.0, abs(pr)), (tag, k)
            assert list(opt.mode_follower.reference_indices) == [int(t) for t in z[f"{tag}_s{k}_targets"]], (tag, k)
            assert np.abs(np.asarray(opt.hessian) - z[f"{tag}_s{k}_Hself"]).max() < tol, (tag, k)
            assert np.abs(H - z[f"{tag}_s{k}_Hcaller"]).max() < tol, (tag, k)        # F7: the caller's array after the in-place update
    return worst


def test_g15_mf_rsirfo_traces():
    """G15: the oracle's MF_RSIRFO + ModeFollowing against twelve traces of the reference's class (target index, EMA, gradient bias,
    mass-weighted overlaps, Hungarian / greedy matching, static references, saddle orders 0 .. 2)."""
    from oracle.mf_rsirfo import MF_RSIRFO
    z = np.load(os.path.join(G, "g15_mf_rsirfo.npz"))
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        worst = _check_mf(z, lambda method, so, elems, masses, kw: MF_RSIRFO(masses=masses, method=method, saddle_order=so, **kw), 1e-9)
    print("oracle MF_RSIRFO vs reference: worst step deviation", worst)


def _check_g14b(fn, tol):
    """fn(kind, x, atoms 1-based, k, angle in degrees) -> (E, g, H) against G14b (the reference's torch autograd outputs)."""
    z = np.load(os.path.join(G, "g14b_keep_angle.npz"))
    worst = 0.0
    for i in range(int(z["ncase"])):
        tag = f"a{i:02d}"
        kind = str(z[tag + "_kind"])
        atoms = [int(a) for a in z[tag + "_atoms"]]
        if kind == "frag":                               # (two atom lists; the "angle" slot carries the distance in angstrom)
            n1 = int(z[tag + "_n1"])
            atoms = [atoms[:n1], atoms[n1:]]
        E, g, H = fn(kind, z[tag + "_x"], atoms, float(z[tag + "_k"]), float(z[tag + "_angle_deg"]))
        refH = z[tag + "_H"]
        if not np.isfinite(refH).all():          # (torch.where multiplies the unselected branch's infinite derivative by zero: NaN in the reference itself)
            continue
        sc = max(1.0, np.abs(refH).max())
        assert abs(E - float(z[tag + "_E"])) < tol * max(1.0, abs(float(z[tag + "_E"]))), (tag, kind)
        assert np.abs(g - z[tag + "_g"]).max() < tol * sc, (tag, kind, np.abs(g - z[tag + "_g"]).max())
        assert np.abs(H - refH).max() < tol * sc, (tag, kind, np.abs(H - refH).max())
        worst = max(worst, np.abs(H - refH).max() / sc)
    return worst


def test_g14b_keep_angle_oracle_vs_reference():
    """G14b: angle, dihedral-angle, out-of-plane-angle and fragment-centre distance restraints (Taylor branches near 0 / 180 degrees, periodic wrap,
    collinearity switch and guard) -- the oracle's
    hyper-dual restatement against the reference's torch.func.jacrev / hessian outputs."""
    from oracle import keep_angle as ka

    def fn(kind, x, atoms, k, ang):
        if kind == "frag":
            return ka.fragment_distance_egh(x, [a - 1 for a in atoms[0]], [a - 1 for a in atoms[1]], k, ang / 0.52917721067)
        return ka.energy_gradient_hessian(x, [a - 1 for a in atoms], k, ka.deg2rad_as_reference(ang), dihedral=(kind == "dihedral"), out_of_plane=(kind == "oop"))
    print("oracle angle restraints vs reference: worst relative Hessian deviation", _check_g14b(fn, 1e-9))
