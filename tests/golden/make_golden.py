"""Generate the golden vectors by IMPORTING the reference (build container only).

Usage:  python tests/golden/make_golden.py        (needs /root/reference, read-only)

The reference is loaded through a package shim that bypasses
multioptpy/__init__.py (which imports ASE unguarded).  Nothing
from the reference is copied: only inputs and the outputs it computes are stored,
as small .npz files next to this script.  The reference never travels to the GPU
box; tests read only the .npz files.
"""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

REF = "/root/reference/multioptpy"


def load_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("multioptpy")
    pkg.__path__ = [REF]
    sys.modules["multioptpy"] = pkg


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


# every dispatch key of rsirfo.py:208-251 plus an unknown name (falls back to flowchart)
UPDATE_METHODS = [
    "rsirfo_flowchart", "rsirfo_block_cfd_fsb_dd", "rsirfo_block_cfd_fsb_weighted", "rsirfo_block_cfd_fsb",
    "rsirfo_block_cfd_bofill_weighted", "rsirfo_block_cfd_bofill", "rsirfo_block_bfgs_dd", "rsirfo_block_bfgs",
    "rsirfo_block_fsb_dd", "rsirfo_block_fsb_weighted", "rsirfo_block_fsb", "rsirfo_block_bofill_weighted",
    "rsirfo_block_bofill", "rsirfo_bfgs_dd", "rsirfo_bfgs", "rsirfo_sr1", "rsirfo_pcfd_bofill", "rsirfo_cfd_fsb_dd",
    "rsirfo_cfd_fsb", "rsirfo_cfd_bofill", "rsirfo_fsb_dd", "rsirfo_fsb", "rsirfo_bofill", "rsirfo_psb", "rsirfo_msp",
    "rsirfo_auto",
]


def g1_updates():
    """G1: delta_H of every rule through RSIRFO's own dispatch list."""
    from multioptpy.Optimizer import rsirfo
    out = {}
    idx = 0
    for n in (21, 42):
        for indef in (False, True):
            H = cases.spd_matrix(n, seed=n + int(indef), indefinite=indef)
            for kind in ("normal", "negative", "dd", "tiny", "exact"):
                sd = 7 * n + idx
                s, y = cases.sy_pair(H, seed=sd, kind=kind)
                for m in UPDATE_METHODS:
                    opt = rsirfo.RSIRFO(method=m, saddle_order=0)
                    low = m.lower()
                    fn = opt.default_update_method[1]
                    for key, _, f in opt.updater_dispatch_list:
                        if key in low:
                            fn = f
                            break
                    with quiet():
                        d = fn(H.copy(), s.reshape(-1, 1).copy(), y.reshape(-1, 1).copy())
                    tag = f"c{idx:03d}"
                    out[tag + "_meta"] = np.array([n, int(indef), sd])
                    out[tag + "_kind"] = np.array(kind)
                    out[tag + "_method"] = np.array(m)
                    out[tag + "_delta"] = np.asarray(d, dtype=float)
                    idx += 1
    np.savez_compressed(os.path.join(HERE, "g1_updates.npz"), **out)
    print("G1:", idx, "cases")


def _trace_rsirfo(cls_module, method, saddle_order, N, nsteps, neb_mode, h0_kind, seed, out, tag):
    """Drive consecutive RSIRFO.run calls along an LJ surface, record inputs + outputs."""
    import scipy.optimize
    rsirfo = cls_module
    x = cases.cluster(N, seed_offset=seed)
    n = 3 * N
    E, G, Hx = cases.lj(x, hessian=True)
    if h0_kind == "exact":
        H = Hx.copy()
    elif h0_kind == "identity":
        H = np.eye(n)
    else:
        H = cases.spd_matrix(n, seed=seed)
    opt = rsirfo.RSIRFO(method=method, saddle_order=saddle_order, trust_radius_max=0.5 if saddle_order == 0 else 0.1,
                        trust_radius_min=0.01)
    if neb_mode:
        opt.switch_NEB_mode()
    brent_calls = []
    real_brentq = scipy.optimize.brentq

    def wrapped_brentq(f, a, b, **kw):
        evals = []

        def g(z):
            v = f(z)
            evals.append((float(z), float(v)))
            return v
        r = real_brentq(g, a, b, **kw)
        brent_calls.append((np.array(evals), float(r)))
        return r
    rsirfo.brentq = wrapped_brentq
    solve_log = []
    real_solve = rsirfo.RSIRFO.solve_rfo

    def wrapped_solve(self, eigvals, gc, alpha, mode="min"):
        r = real_solve(self, eigvals, gc, alpha, mode)
        solve_log.append((float(alpha), float(r[1])))
        return r
    rsirfo.RSIRFO.solve_rfo = wrapped_solve
    try:
        px = pg = pE = None
        out[tag + "_meta"] = np.array([N, saddle_order, int(neb_mode), nsteps, seed])
        out[tag + "_method"] = np.array(method)
        out[tag + "_h0kind"] = np.array(h0_kind)
        out[tag + "_H0"] = H.copy() if n <= 66 else np.zeros(0)
        for k in range(nsteps):
            E, G = cases.lj(x)
            g = G.reshape(-1, 1)
            xx = x.reshape(-1, 1).copy()
            opt.set_hessian(H)               # optmain re-sets the same array every step (F7)
            opt.set_bias_hessian(np.zeros((n, n)))
            del brent_calls[:]
            del solve_log[:]
            with quiet():
                if k == 0:
                    step = opt.run(xx, g, [], [], E, 0.0, [], [], g, [])
                else:
                    step = opt.run(xx, g, pg, px, E, pE, [], [], g, pg)
            out[f"{tag}_s{k}_x"] = xx.copy()
            out[f"{tag}_s{k}_g"] = g.copy()
            out[f"{tag}_s{k}_E"] = np.array(E)
            out[f"{tag}_s{k}_step"] = np.asarray(step).copy()
            out[f"{tag}_s{k}_tr"] = np.array(opt.trust_radius)
            out[f"{tag}_s{k}_pred"] = np.array(opt.predicted_energy_changes[-1])
            out[f"{tag}_s{k}_solve"] = np.array(solve_log).reshape(-1, 2)
            if brent_calls:
                out[f"{tag}_s{k}_brent_evals"] = brent_calls[0][0]
                out[f"{tag}_s{k}_brent_root"] = np.array(brent_calls[0][1])
            if n <= 66 or k == nsteps - 1:
                out[f"{tag}_s{k}_Hself"] = np.array(opt.hessian).copy()
                out[f"{tag}_s{k}_Hcaller"] = H.copy()
            px, pg, pE = xx, g, E
            x = (xx - step).reshape(-1, 3)
    finally:
        rsirfo.brentq = real_brentq
        rsirfo.RSIRFO.solve_rfo = real_solve


def g2_rsirfo():
    """G2 (+G4 brentq traces): consecutive RSIRFO.run calls."""
    from multioptpy.Optimizer import rsirfo
    out = {}
    cfgs = []
    for N, nst in ((7, 4), (11, 4), (22, 4)):
        for so in (0, 1):
            for method in ("rsirfo_bfgs", "rsirfo_fsb", "rsirfo_bofill", "rsirfo_block_fsb", "rsirfo_block_bofill"):
                cfgs.append((method, so, N, nst, False, "exact", 0))
    cfgs.append(("rsirfo_block_bofill", 0, 7, 4, True, "identity", 1))
    cfgs.append(("rsirfo_block_bofill", 1, 7, 4, True, "exact", 2))
    cfgs.append(("rsirfo_block_fsb", 0, 14, 5, False, "identity", 3))
    cfgs.append(("rsirfo_bfgs", 0, 11, 4, False, "spd", 4))
    cfgs.append(("rsirfo_bfgs", 0, 100, 3, False, "exact", 5))
    cfgs.append(("rsirfo_block_bofill", 0, 100, 3, True, "identity", 6))
    for i, c in enumerate(cfgs):
        _trace_rsirfo(rsirfo, c[0], c[1], c[2], c[3], c[4], c[5], c[6], out, f"t{i:02d}")
    out["ncfg"] = np.array(len(cfgs))
    np.savez_compressed(os.path.join(HERE, "g2_rsirfo.npz"), **out)
    print("G2:", len(cfgs), "traces")


def g5_projections():
    from multioptpy.Optimizer import rsirfo
    from multioptpy.Utils.calc_tools import Calculationtools
    out = {}
    geoms = {
        "ar7": cases.cluster(7),
        "ar22": cases.cluster(22),
        "linear_z": np.array([[0, 0, -2.0], [0, 0, 0.0], [0, 0, 2.5]]),
        "diatomic": np.array([[0.1, 0.2, 0.3], [1.5, -0.4, 0.9]]),
    }
    opt = rsirfo.RSIRFO(method="rsirfo_bfgs", saddle_order=0)
    for name, x in geoms.items():
        n = x.size
        H = cases.spd_matrix(n, seed=n)
        rng = np.random.default_rng(n)
        g = rng.normal(size=n)
        with quiet():
            Hp = Calculationtools().project_out_hess_tr_and_rot_for_coord(H, x.copy(), x.copy(), False)
            gp = opt._project_grad_tr_rot(g.copy(), x.reshape(-1, 1).copy())
        out[name + "_x"] = x
        out[name + "_H"] = H
        out[name + "_g"] = g
        out[name + "_Hp"] = Hp
        out[name + "_gp"] = gp
    P = cases.cluster(9)
    rng = np.random.default_rng(3)
    Q = P + 0.05 * rng.normal(size=P.shape)
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    Q = Q @ R.T + 0.7
    Pk, Qk = Calculationtools().kabsch_algorithm(P.copy(), Q.copy())
    out["kabsch_P"], out["kabsch_Q"], out["kabsch_Pout"], out["kabsch_Qout"] = P, Q, Pk, Qk
    np.savez_compressed(os.path.join(HERE, "g5_projections.npz"), **out)
    print("G5 done")


def _band_inputs(I, N, bump):
    X = cases.band(I, N)
    E = np.zeros(I)
    G = np.zeros((I, N, 3))
    for i in range(I):
        E[i], G[i] = cases.lj(X[i])
    # optional synthetic energy profile to visit all three tangent cases
    if bump is not None:
        E = E + np.asarray(bump)
    return X, E, G


def g6_neb_forces():
    from multioptpy.MEP import pathopt_bneb_force as bf
    out = {}
    k = 0
    for N in (7, 14):
        for bump in (None, [0, 1e-3, 2e-3, 1.5e-3, 1e-3, 1.2e-3, 2.5e-3, 0.5e-3], [0, -1e-3, -2e-3, -3e-3, -2e-3, -1e-3, 0, 1e-3]):
            for ci in (False, True):
                X, E, G = _band_inputs(8, N, bump)
                if k % 3 == 1:
                    X[3, 2] = X[2, 2]          # zero displacement of one atom between neighbours
                calc = bf.CaluculationBNEB(APPLY_CI_NEB=(2 if ci else 99999))
                with quiet():
                    F = calc.calc_force(X.copy(), E.copy(), G.copy(), 5, ["Ar"] * N)
                tau = np.array(calc.tau_list)
                gam = np.zeros(8)
                for i in range(1, 7):
                    gam[i] = calc.calculate_gamma(X[i - 1:i + 2], E[i - 1:i + 2], G[i - 1:i + 2], calc.get_tau(i))
                tag = f"f{k:02d}"
                out[tag + "_X"], out[tag + "_E"], out[tag + "_G"] = X, E, G
                out[tag + "_ci"] = np.array(int(ci))
                out[tag + "_F"], out[tag + "_tau"], out[tag + "_gamma"] = F, tau, gam
                k += 1
    out["ncase"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "g6_neb_force.npz"), **out)
    print("G6:", k, "cases")


def g3_rsprfo():
    """G3: consecutive EnhancedRSPRFO.run calls (P-RFO TS search and saddle_order 0), with the Hessian re-set every
    step from the caller's array (optmain with -fc -1: the update is discarded) or set once."""
    from multioptpy.Optimizer import rsprfo
    out = {}
    cfgs = [("rsprfo_bofill", 1, 7, 5, "once", 0), ("rsprfo_bofill", 1, 11, 5, "every", 1), ("rsprfo_bfgs", 0, 7, 4, "once", 2),
            ("rsprfo_fsb", 1, 22, 4, "once", 3), ("rsprfo_block_bofill", 2, 11, 4, "once", 4), ("rsprfo_bofill", 1, 14, 6, "once", 5),
            ("rsprfo_bofill", 1, 22, 5, "once", 6)]      # p06: the Diels-Alder-sized P-RFO/Bofill TS search (BASELINE config 3)
    for ci, (method, so, N, nsteps, mode, seed) in enumerate(cfgs):
        tag = f"p{ci:02d}"
        x = cases.cluster(N, seed_offset=seed)
        n = 3 * N
        H0 = cases.lj(x, hessian=True)[2]
        opt = rsprfo.EnhancedRSPRFO(method=method, saddle_order=so, trust_radius_max=0.3 if so > 0 else 0.5, trust_radius_min=0.01)
        out[tag + "_meta"] = np.array([N, so, nsteps, seed])
        out[tag + "_method"] = np.array(method)
        out[tag + "_mode"] = np.array(mode)
        px = pg = pE = pmove = None
        for k in range(nsteps):
            E, G = cases.lj(x)
            g = G.reshape(-1, 1)
            xx = x.reshape(-1, 1).copy()
            if mode == "every" or k == 0:
                opt.set_hessian(H0)
                opt.set_bias_hessian(np.zeros((n, n)))
            with quiet():
                if k == 0:
                    step = opt.run(xx, g, [], [], E, 0.0, [], [], g, [])
                else:
                    step = opt.run(xx, g, pg, px, E, pE, pmove, [], g, pg)
            out[f"{tag}_s{k}_x"], out[f"{tag}_s{k}_g"], out[f"{tag}_s{k}_E"] = xx.copy(), g.copy(), np.array(E)
            out[f"{tag}_s{k}_step"] = np.asarray(step).copy()
            out[f"{tag}_s{k}_tr"] = np.array(opt.trust_radius)
            out[f"{tag}_s{k}_pred"] = np.array(opt.predicted_energy_changes[-1])
            out[f"{tag}_s{k}_H"] = np.array(opt.hessian).copy()
            out[f"{tag}_s{k}_tsidx"] = np.array(-1 if opt.ts_mode_idx is None else int(opt.ts_mode_idx))
            px, pg, pE, pmove = xx, g, E, np.asarray(step).copy()
            x = (xx - step).reshape(-1, 3)
    out["ncfg"] = np.array(len(cfgs))
    np.savez_compressed(os.path.join(HERE, "g3_rsprfo.npz"), **out)
    print("G3:", len(cfgs), "traces")


def _read_xyz(path):
    """Element symbols and coordinates (angstrom) of a reference test geometry (first line: charge/multiplicity or atom count)."""
    elems, xyz = [], []
    for line in open(path):
        t = line.split()
        if len(t) == 4:
            try:
                xyz.append([float(t[1]), float(t[2]), float(t[3])])
                elems.append(t[0])
            except ValueError:
                pass
    return elems, np.array(xyz)


def g10_model_hessian():
    """G10: FischerD3ApproxHessianOld.main (the `-modelhess` default) and ApproxHessian.main for organic test geometries of the
    reference (test/*.xyz: bonds, angles, dihedrals, D3 pairs) and argon clusters (D3 pairs only)."""
    from multioptpy.ModelHessian.fischerd3old import FischerD3ApproxHessianOld
    try:
        from multioptpy.ModelHessian.approx_hessian import ApproxHessian
    except Exception as e:          # optional heavy imports of other models
        print("ApproxHessian dispatcher not importable here:", e)
        ApproxHessian = None
    out = {}
    tags = []
    for name in ("SN2", "aldol_rxn", "claisen_rearrengment", "diels_alder_rxn", "epoxidation"):
        elems, xyz = _read_xyz(f"/root/reference/test/{name}.xyz")
        tags.append((name, elems, xyz / cases.BOHR2ANG))
    for N in (7, 13):
        tags.append((f"ar{N}", ["Ar"] * N, cases.cluster(N)))
    # a distorted copy (different connectivity near the thresholds) and a linear triatomic
    elems, xyz = _read_xyz("/root/reference/test/aldol_rxn.xyz")
    rng = np.random.default_rng(10)
    tags.append(("aldol_jitter", elems, (xyz + rng.normal(0.0, 0.08, xyz.shape)) / cases.BOHR2ANG))
    tags.append(("co2", ["O", "C", "O"], np.array([[-1.16, 0.0, 0.0], [0.0, 0.0, 0.0], [1.16, 0.0, 0.0]]) / cases.BOHR2ANG))
    for tag, elems, x in tags:
        g = np.zeros_like(x)
        with quiet():
            H = FischerD3ApproxHessianOld().main(x.copy(), list(elems), g)
            Hd = ApproxHessian().main(x.copy(), list(elems), g, approx_hess_type="fischerd3old") if ApproxHessian else H
        out[tag + "_x"], out[tag + "_H"], out[tag + "_Hdisp"] = x, np.array(H), np.array(Hd)
        out[tag + "_elems"] = np.array(elems)
    out["tags"] = np.array([t[0] for t in tags])
    np.savez_compressed(os.path.join(HERE, "g10_model_hessian.npz"), **out)
    print("G10:", len(tags), "geometries")

def g10b_model_hessian_variants():
    """G10b: the two other Fischer models behind ApproxHessian.main -- FischerD3ApproxHessian (coordination-number scaled D3 pairs,
    fischerd3.py) and FischerApproxHessian (no dispersion term, fischer.py) -- on G10's geometries, each through the class and through
    the dispatcher's substring test."""
    from multioptpy.ModelHessian.fischerd3 import FischerD3ApproxHessian
    from multioptpy.ModelHessian.fischer import FischerApproxHessian
    from multioptpy.ModelHessian.approx_hessian import ApproxHessian
    from multioptpy.ModelHessian.schlegel import SchlegelApproxHessian
    z = np.load(os.path.join(HERE, "g10_model_hessian.npz"))
    out = {"tags": z["tags"]}
    for tag in z["tags"]:
        x, elems = z[tag + "_x"], [str(e) for e in z[tag + "_elems"]]
        g = np.zeros_like(x)
        with quiet():
            H3 = FischerD3ApproxHessian().main(x.copy(), list(elems), g)
            H3d = ApproxHessian().main(x.copy(), list(elems), g, approx_hess_type="fischerd3")
            Hp = FischerApproxHessian().main(x.copy(), list(elems), g)
            Hpd = ApproxHessian().main(x.copy(), list(elems), g, approx_hess_type="fischer")
            Hs = SchlegelApproxHessian().main(x.copy(), list(elems), g)
            Hsd = ApproxHessian().main(x.copy(), list(elems), g, approx_hess_type="schlegel")
            Hm = ApproxHessian().main(x.copy(), list(elems), g, approx_hess_type="morse")
        out[tag + "_Hmorse"] = np.array(Hm)
        out[tag + "_Hsch"], out[tag + "_Hschdisp"] = np.array(Hs), np.array(Hsd)
        out[tag + "_x"], out[tag + "_elems"] = x, z[tag + "_elems"]
        out[tag + "_Hd3"], out[tag + "_Hd3disp"], out[tag + "_Hplain"], out[tag + "_Hplaindisp"] = np.array(H3), np.array(H3d), np.array(Hp), np.array(Hpd)
    np.savez_compressed(os.path.join(HERE, "g10b_model_hessian_variants.npz"), **out)
    print("G10b:", len(out["tags"]), "geometries x 2 models")



def g14_keep_pot():
    """G14: StructKeepPotential.calc_energy with its gradient and Hessian as the reference obtains them (torch.func.jacrev / hessian,
    Potential/potential.py:127-152): bonded and distant pairs, stretched and compressed, parameters from the constructor and from
    bias_pot_params."""
    import torch
    from multioptpy.Potential.keep_potential import StructKeepPotential
    out = {}
    elems, xyz = _read_xyz("/root/reference/test/aldol_rxn.xyz")
    x = xyz / cases.BOHR2ANG
    cases_ = [("a", x, [1, 2], 0.1, 1.0, False), ("b", x, [3, 11], 0.5, 4.5, True), ("c", x, [7, 2], 2.0, 0.3, False),
              ("d", cases.cluster(13), [1, 13], 0.05, 8.0, True)]
    for tag, xb, pair, k, r0, via_params in cases_:
        pot = StructKeepPotential(keep_pot_spring_const=k, keep_pot_distance=r0, keep_pot_atom_pairs=pair)
        g = torch.tensor(xb, dtype=torch.float64)
        par = torch.tensor([k, r0], dtype=torch.float64) if via_params else []
        e = pot.calc_energy(g, par)
        grad = torch.func.jacrev(pot.calc_energy, argnums=0)(g, par)
        hess = torch.func.hessian(pot.calc_energy, argnums=0)(g, par).reshape(3 * len(xb), 3 * len(xb))
        out[tag + "_x"], out[tag + "_pair"], out[tag + "_k"], out[tag + "_r0"] = xb, np.array(pair), np.array(k), np.array(r0)
        out[tag + "_E"], out[tag + "_g"], out[tag + "_H"] = np.array(float(e)), grad.detach().numpy(), hess.detach().numpy()
    from multioptpy.Potential.anharmonic_keep_potential import StructAnharmonicKeepPotential
    atags = []
    for tag, xb, pair, k, D, r0 in (("an_a", x, [1, 2], 0.3, 0.15, 1.2), ("an_b", x, [3, 11], 0.1, 0.02, 3.0), ("an_c", cases.cluster(13), [2, 9], 0.2, 0.05, 5.0),
                                    ("an_zero", x, [1, 2], 0.3, 0.0, 1.2)):
        pot = StructAnharmonicKeepPotential(anharmonic_keep_pot_spring_const=k, anharmonic_keep_pot_potential_well_depth=D,
                                            anharmonic_keep_pot_atom_pairs=pair, anharmonic_keep_pot_distance=r0)
        g = torch.tensor(xb, dtype=torch.float64)
        e = pot.calc_energy(g, [])
        if D != 0.0:
            grad = torch.func.jacrev(pot.calc_energy, argnums=0)(g, []).detach().numpy()
            hess = torch.func.hessian(pot.calc_energy, argnums=0)(g, []).reshape(3 * len(xb), 3 * len(xb)).detach().numpy()
        else:                                       # (a constant tensor: the reference's autograd returns zeros)
            grad, hess = np.zeros_like(xb), np.zeros((3 * len(xb), 3 * len(xb)))
        out[tag + "_x"], out[tag + "_pair"], out[tag + "_k"], out[tag + "_D"], out[tag + "_r0"] = xb, np.array(pair), np.array(k), np.array(D), np.array(r0)
        out[tag + "_E"], out[tag + "_g"], out[tag + "_H"] = np.array(float(e)), grad, hess
        atags.append(tag)
    out["atags"] = np.array(atags)
    out["tags"] = np.array([c[0] for c in cases_])
    np.savez_compressed(os.path.join(HERE, "g14_keep_pot.npz"), **out)
    print("G14:", len(cases_), "cases")


def g11_afir():
    """G11: AFIRPotential.calc_energy with its gradient and Hessian as the reference obtains them (torch.func.jacrev / hessian,
    Potential/potential.py:127-152) for several geometries, fragment pairs and gamma values (kJ/mol, both signs and zero)."""
    import torch
    from multioptpy.Potential.AFIR_potential import AFIRPotential
    out = {}
    cases_ = []
    elems, xyz = _read_xyz("/root/reference/test/aldol_rxn.xyz")
    x = xyz / cases.BOHR2ANG
    cases_.append(("aldol_a", elems, x, [1, 2, 3], [5, 6, 7, 8], 100.0))
    cases_.append(("aldol_b", elems, x, [1], [11], -50.0))
    cases_.append(("aldol_c", elems, x, [4, 5], [9, 10, 11], 0.0))
    elems2, xyz2 = _read_xyz("/root/reference/test/diels_alder_rxn.xyz")
    cases_.append(("da", elems2, xyz2 / cases.BOHR2ANG, list(range(1, 11)), list(range(11, 23)), 250.0))
    cases_.append(("ar13", ["Ar"] * 13, cases.cluster(13), [1, 2, 3, 4, 5, 6], [7, 8, 9, 10, 11, 12, 13], 30.0))
    for tag, el, xb, f1, f2, gam in cases_:
        pot = AFIRPotential(AFIR_gamma=gam, AFIR_Fragm_1=f1, AFIR_Fragm_2=f2, element_list=list(el))
        g = torch.tensor(xb, dtype=torch.float64)
        par = torch.tensor([gam], dtype=torch.float64)
        e = pot.calc_energy(g, par)
        grad = torch.func.jacrev(pot.calc_energy, argnums=0)(g, par)
        hess = torch.func.hessian(pot.calc_energy, argnums=0)(g, par).reshape(3 * len(el), 3 * len(el))
        out[tag + "_x"], out[tag + "_elems"] = xb, np.array(el)
        out[tag + "_f1"], out[tag + "_f2"], out[tag + "_gamma"] = np.array(f1), np.array(f2), np.array(gam)
        out[tag + "_E"], out[tag + "_g"], out[tag + "_H"] = np.array(float(e)), grad.detach().numpy(), hess.detach().numpy()
    out["tags"] = np.array([c[0] for c in cases_])
    np.savez_compressed(os.path.join(HERE, "g11_afir.npz"), **out)
    print("G11:", len(cases_), "cases")


def g12_constraints():
    """G12: ProjectOutConstrain (Constraint/constraint_condition.py:340-754) -- initial values, SHAKE-like adjustment, gradient
    projection, three consecutive Hessian projections (history of constraint bases) -- and three consecutive CRSIRFO.run calls
    (Optimizer/crsirfo.py:15-182) on a quadratic model surface, with bond / angle / dihedral / x constraints on the aldol geometry."""
    from multioptpy.Constraint.constraint_condition import ProjectOutConstrain
    from multioptpy.Optimizer.crsirfo import CRSIRFO
    from multioptpy.ModelHessian.fischerd3old import FischerD3ApproxHessianOld
    elems, xyz = _read_xyz("/root/reference/test/aldol_rxn.xyz")
    x0 = xyz / cases.BOHR2ANG
    N = len(elems)
    with quiet():
        K = FischerD3ApproxHessianOld().main(x0.copy(), list(elems), np.zeros_like(x0))
    rng = np.random.default_rng(12)
    out = {"x0": x0, "elems": np.array(elems), "K": K}
    names = ["bond", "angle", "dihedral", "x"]
    atoms = [[1, 3], [2, 1, 3], [4, 1, 3, 5], [7]]
    out["names"] = np.array(names)
    out["atoms_flat"] = np.array([a for lst in atoms for a in lst])
    out["atoms_len"] = np.array([len(lst) for lst in atoms])
    # ---- projections
    with quiet():
        pc = ProjectOutConstrain(list(names), [list(a) for a in atoms])
        init = pc.initialize(x0.copy())
        out["init"] = np.array(init, dtype=float)
        xs = [x0 + rng.normal(0, 0.03, x0.shape) for _ in range(3)]
        for k, x in enumerate(xs):
            g = rng.normal(size=(N, 3))
            H = K + 0.05 * np.eye(3 * N) * k
            out[f"p{k}_x"], out[f"p{k}_g"], out[f"p{k}_Hin"] = x, g, H
            out[f"p{k}_B"] = np.array(pc._get_all_constraint_vectors(x))
            out[f"p{k}_gproj"] = np.array(pc.calc_project_out_grad(x.copy(), g.copy()))
            out[f"p{k}_Hproj"] = np.array(pc.calc_project_out_hess(x.copy(), g.copy(), H.copy()))
        xa = x0 + rng.normal(0, 0.05, x0.shape)
        out["shake_in"] = xa
        out["shake_out"] = np.array(pc.adjust_init_coord(xa.copy()))
        out["shake_iters"] = np.array(pc.last_shake_iter)
    # ---- CRSIRFO on E = 1/2 (x - x0) K (x - x0) + d.(x - x0)
    d = 0.01 * rng.normal(size=x0.size)
    # a physical linear term: no net force or torque (the model surface's curvature K has exact TR/rot zero modes at x0, and CRSIRFO
    # does not project them out -- a gradient with components along them would make the step ill-conditioned for any solver)
    Tr = np.zeros((6, x0.size))
    xc = x0 - x0.mean(axis=0)
    for a in range(len(x0)):
        for c in range(3):
            Tr[c, 3 * a + c] = 1.0
        Tr[3, 3 * a:3 * a + 3] = [0.0, -xc[a, 2], xc[a, 1]]
        Tr[4, 3 * a:3 * a + 3] = [xc[a, 2], 0.0, -xc[a, 0]]
        Tr[5, 3 * a:3 * a + 3] = [-xc[a, 1], xc[a, 0], 0.0]
    Qtr, _ = np.linalg.qr(Tr.T)
    d = d - Qtr @ (Qtr.T @ d)
    out["d"] = d
    with quiet():
        pc2 = ProjectOutConstrain(list(names[:2]), [list(a) for a in atoms[:2]])
        pc2.initialize(x0.copy())
        opt = CRSIRFO(constraints=pc2, method="rsirfo_fsb", saddle_order=0)
        x = x0 + rng.normal(0, 0.02, x0.shape)
        px = pg = None
        pE = 0.0
        for it in range(3):
            dx = (x - x0).reshape(-1)
            E = 0.5 * dx @ K @ dx + d @ dx
            g = (K @ dx + d).reshape(-1, 1)
            opt.set_bias_hessian(np.zeros_like(K))
            if it == 0:
                opt.set_hessian(K.copy())
                s = opt.run(x.reshape(-1, 1).copy(), g.copy(), [], [], E, 0.0, [], [], g.copy(), [])
            else:
                s = opt.run(x.reshape(-1, 1).copy(), g.copy(), pg, px, E, pE, [], [], g.copy(), pg)
            out[f"c{it}_x"], out[f"c{it}_g"], out[f"c{it}_E"] = x.copy(), g.copy(), np.array(E)
            out[f"c{it}_step"] = np.asarray(s).copy()
            out[f"c{it}_tr"] = np.array(opt.trust_radius)
            out[f"c{it}_H"] = np.array(opt.hessian).copy()
            px, pg, pE = x.reshape(-1, 1).copy(), g.copy(), E
            x = (x.reshape(-1, 1) - s).reshape(-1, 3)
    np.savez_compressed(os.path.join(HERE, "g12_constraints.npz"), **out)
    print("G12 written")


BRENT_FUNCS = {
    "cubic": (lambda t: t ** 3 - 2 * t - 5, 2.0, 3.0),
    "cosx": (lambda t: np.cos(t) - t, 0.0, 1.0),
    "recip": (lambda t: 1.0 / t - 5.0, 1e-6, 1000.0),
    "steep": (lambda t: np.tanh(40 * (t - 0.3)) + 0.2, 1e-6, 1000.0),
    "norm2": (lambda t: 0.81 / (1 + t) ** 2 - 0.25, 1e-6, 1000.0),
}


def g4_brentq():
    """G4: every (x, f(x)) that the real scipy.optimize.brentq evaluates, with the call-site tolerances
    of rsirfo.py:1042 (xtol = rtol = 1e-6, maxiter = 50) and with tight ones."""
    import scipy.optimize
    out = {}
    for name, (f, a, b) in BRENT_FUNCS.items():
        for tname, kw in (("loose", dict(xtol=1e-6, rtol=1e-6, maxiter=50)), ("tight", dict(xtol=1e-12, rtol=1e-12, maxiter=100))):
            ev = []

            def g(t, f=f, ev=ev):
                v = f(t)
                ev.append((float(t), float(v)))
                return v
            r = scipy.optimize.brentq(g, a, b, **kw)
            out[f"{name}_{tname}_evals"] = np.array(ev)
            out[f"{name}_{tname}_root"] = np.array(r)
    out["scipy_version"] = np.array(scipy.__version__)
    np.savez_compressed(os.path.join(HERE, "g4_brentq.npz"), **out)
    print("G4 done, scipy", scipy.__version__)


def g6b_neb_variants():
    """G6b: the spring/tangent variants named in the north star: DNEB, EWBNEB, BNEB3 calc_force."""
    from multioptpy.MEP import pathopt_bneb_force as bf
    from multioptpy.MEP import pathopt_dneb_force as df
    from multioptpy.MEP import pathopt_ewbneb_force as ef
    out = {}
    k = 0
    bumps = (None, [0, 1e-3, 2e-3, 1.5e-3, 1e-3, 1.2e-3, 2.5e-3, 0.5e-3], [0, -1e-3, -2e-3, -3e-3, -2e-3, -1e-3, 0, 1e-3],
             [2e-3, 1e-3, 3e-3, 3e-3, 1e-3, 0.5e-3, 0.7e-3, 0.0])
    for N in (7, 14):
        for bump in bumps:
            for ci in (False, True):
                X, E, G = _band_inputs(8, N, bump)
                if k % 3 == 1:
                    X[3, 2] = X[2, 2]
                tag = f"v{k:02d}"
                out[tag + "_X"], out[tag + "_E"], out[tag + "_G"] = X, E, G
                out[tag + "_ci"] = np.array(int(ci))
                apply = 2 if ci else 99999
                with quiet():
                    out[tag + "_dneb"] = df.CaluculationDNEB(apply).calc_force(X.copy(), E.copy(), G.copy(), 5, ["Ar"] * N)
                    out[tag + "_ewbneb"] = ef.CaluculationEWBNEB(apply).calc_force(X.copy(), E.copy(), G.copy(), 5, ["Ar"] * N)
                    out[tag + "_bneb3"] = bf.CaluculationBNEB3(apply).calc_force(X.copy(), E.copy(), G.copy(), 5, ["Ar"] * N)
                k += 1
    out["ncase"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "g6b_neb_variants.npz"), **out)
    print("G6b:", k, "cases")


class _Cfg:
    pass


def g7_neb_step():
    """G7: TR_calc, FIRE and RFOOptimizer.optimize over consecutive band steps (I=8, N=7)."""
    from multioptpy.MEP import pathopt_bneb_force as bf
    from multioptpy.Optimizer import rfo_neb, fire_neb, trust_radius_neb
    out = {}
    tmp = tempfile.mkdtemp(prefix="mopt_g7_") + "/"
    cfg = _Cfg()
    cfg.NEB_FOLDER_DIRECTORY = tmp
    cfg.fix_init_edge = False
    cfg.fix_end_edge = False
    cfg.apply_convergence_criteria = False
    cfg.ratio_of_rfo_step = 0.5
    cfg.bohr2angstroms = cases.BOHR2ANG
    cfg.dt, cfg.a, cfg.n_reset = 0.5, 0.10, 0
    cfg.FIRE_N_accelerate, cfg.FIRE_f_inc, cfg.FIRE_f_accelerate = 5, 1.10, 0.99
    cfg.FIRE_f_decelerate, cfg.FIRE_a_start, cfg.FIRE_dt_max = 0.5, 0.1, 1.0
    I, N = 8, 7
    cfg.element_list = ["Ar"] * N
    for variant, exact in (("ident", False), ("exact", True)):
        for f in os.listdir(tmp):
            os.remove(tmp + f)
        X = cases.band(I, N)
        if exact:
            for i in range(I):
                np.save(tmp + f"tmp_hessian_{i}.npy", cases.lj(X[i], hessian=True)[2])
        calc = bf.CaluculationBNEB()
        optm = rfo_neb.RFOOptimizer(cfg)
        preX = preG = None
        pre_vel = []
        for it in range(4):
            E = np.zeros(I)
            G = np.zeros((I, N, 3))
            for i in range(I):
                E[i], G[i] = cases.lj(X[i])
            with quiet():
                F = calc.calc_force(X, E, G, it, cfg.element_list)
                vel = np.array(F, dtype=float)
                newX = optm.optimize(X, G, preX, preG, it, E, E, pre_vel, vel, None, preX, calc)
            tag = f"{variant}_it{it}"
            out[tag + "_X"], out[tag + "_E"], out[tag + "_G"], out[tag + "_F"] = X.copy(), E, G, F
            out[tag + "_newX"] = np.array(newX)
            out[tag + "_hess"] = np.array([np.load(tmp + f"tmp_hessian_{i}.npy") for i in range(I)])
            out[tag + "_tr"] = np.array([o.trust_radius for o in optm.optimizer_instance_list])
            preX, preG, pre_vel = X.copy(), G.copy(), vel.copy()
            X = np.array(newX) / cases.BOHR2ANG
    # standalone TR_calc and FIRE on fixed inputs
    X = cases.band(I, N)
    rng = np.random.default_rng(11)
    Fr = 0.01 * rng.normal(size=X.shape)
    delta = [0.3 * rng.normal(size=(N, 3)) * (0.02 + 0.2 * (i % 3)) for i in range(I)]
    trn = trust_radius_neb.TR_NEB(NEB_FOLDER_DIRECTORY=tmp)
    with quiet():
        mv = trn.TR_calc(X, Fr, [d.copy() for d in delta], None, None, None)
    out["tr_X"], out["tr_F"], out["tr_delta"], out["tr_out"] = X, Fr, np.array(delta), np.array(mv)
    fo = fire_neb.FIREOptimizer(cfg)
    vel = 0.01 * rng.normal(size=X.shape)
    pre_vel = 0.01 * rng.normal(size=X.shape)
    with quiet():
        nx = fo.optimize(X, Fr, pre_vel, 3, vel, None, None, None, None)
    out["fire_vel"], out["fire_prevel"], out["fire_out"] = vel, pre_vel, np.array(nx)
    out["fire_state"] = np.array([fo.dt, fo.a, fo.n_reset])
    np.savez_compressed(os.path.join(HERE, "g7_neb_step.npz"), **out)
    print("G7 done")


def g9_trust_radius():
    """G9: TrustRadius.update_trust_radii sequences (trust_radius.py:120-206) on synthetic (E, g, s, H) histories."""
    from multioptpy.Optimizer.trust_radius import TrustRadius
    out = {}
    for c, (n, seed, tmax, tmin) in enumerate([(21, 3, None, None), (21, 4, 0.3, 0.05), (39, 5, None, None), (9, 6, 0.1, None)]):
        rng = np.random.default_rng(seed)
        H = cases.spd_matrix(n, seed=seed)
        TR = TrustRadius()
        if tmax is not None:
            TR.set_max_trust_radius(tmax)
        if tmin is not None:
            TR.set_min_trust_radius(tmin)
        tr = 0.5 if tmax is None else tmax
        x = np.zeros(n)
        res = []
        for k in range(14):
            g = 0.02 * rng.normal(size=n)
            mv = 0.05 * rng.normal(size=n)
            if k % 4 == 1:
                mv *= tr / np.linalg.norm(mv)              # a step that was limited by the trust radius
            if k % 5 == 3:
                mv *= 1e-5                                 # |Ce| below the precision threshold
            pred = float(g @ mv + 0.5 * mv @ H @ mv)
            e0 = float(rng.normal())
            e1 = e0 - pred * (1.0 + (0.1, 0.6, 2.5, -0.5, 0.02)[k % 5] * rng.normal())
            with quiet():
                tr = float(TR.update_trust_radii(e1, e0, g.reshape(-1, 1), mv.reshape(-1, 1), H, x.reshape(-1, 1), tr))
            out[f"c{c}_k{k}_g"] = g; out[f"c{c}_k{k}_mv"] = mv; out[f"c{c}_k{k}_e"] = np.array([e1, e0])
            res.append(tr)
        out[f"c{c}_H"] = H
        out[f"c{c}_meta"] = np.array([n, 14, -1.0 if tmax is None else tmax, -1.0 if tmin is None else tmin])
        out[f"c{c}_tr"] = np.array(res)
    out["ncases"] = np.array(4)
    np.savez_compressed(os.path.join(HERE, "g9_trust_radius.npz"), **out)
    print("G9: 4 sequences")


def _run_optmain(tag, N, argv, out, scale=0.92, jitter=0.3):
    """Run the unmodified optmain driver (multioptpy/optimization.py) with the LJ calculator in a scratch directory and
    record, per iteration, what goes into and comes out of CalculateMoveVector.calc_move_vector, the convergence test,
    and the trust radius; plus the final geometry and the energy_profile.csv rows."""
    import multioptpy.interface
    import multioptpy.optimization
    import multioptpy.optimizer
    stub = types.ModuleType("multioptpy.Calculator.ase_calculation_tools")
    stub.ASEEngine = stub.Calculation = type("ASEUnavailable", (), {})
    sys.modules.setdefault(stub.__name__, stub)
    x0 = cases.cluster(N, jitter=jitter) * scale * cases.BOHR2ANG          # angstrom, as the XYZ file holds it
    cwd = os.getcwd()
    rec = []
    conv = []
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            with open("ar.xyz", "w") as f:
                f.write(f"{N}\n0 1\n")
                for r in x0:
                    f.write(f"Ar {r[0]:.10f} {r[1]:.10f} {r[2]:.10f}\n")
            parser = multioptpy.interface.init_parser()
            args = multioptpy.interface.optimizeparser(parser, ["ar.xyz", "-os", "lj"] + argv)
            CMV = multioptpy.optimizer.CalculateMoveVector
            CC = multioptpy.optimization.ConvergenceChecker
            orig, orig_cc = CMV.calc_move_vector, CC.check_convergence

            def wrap(self, it, geom, B_g, pre_B_g, pre_geom, B_e, pre_B_e, pre_mv, init, g, pre_g, insts, pc=False, print_flag=True):
                r = dict(geom=np.array(geom, dtype=float), B_g=np.array(B_g, dtype=float), B_e=float(B_e), pre_B_e=float(pre_B_e))
                o = orig(self, it, geom, B_g, pre_B_g, pre_geom, B_e, pre_B_e, pre_mv, init, g, pre_g, insts, pc, print_flag)
                r.update(tr=float(self.trust_radii), new=np.array(o[0], dtype=float), mv=np.array(o[1], dtype=float),
                         H=np.array(insts[0].hessian, dtype=float), opt_tr=float(getattr(insts[0], "trust_radius", np.nan)))
                rec.append(r)
                return o

            def wrap_cc(self, state, disp, insts):
                o = orig_cc(self, state, disp, insts)
                conv.append((bool(o[0]), float(o[1]), float(o[2]), np.array(disp, dtype=float)))
                return o

            CMV.calc_move_vector, CC.check_convergence = wrap, wrap_cc
            try:
                with quiet():
                    bpa = multioptpy.optimization.Optimize(args)
                    bpa.run()
            finally:
                CMV.calc_move_vector, CC.check_convergence = orig, orig_cc
            import glob
            prof = glob.glob(os.path.join(td, "*", "*", "energy_profile.csv"))
            rows = [ln.strip().split(",") for ln in open(prof[0]).read().splitlines()[1:]] if prof else []
            opt = glob.glob(os.path.join(td, "*", "*", "*_optimized.xyz"))
            final = None
            if opt:
                lines = open(opt[0]).read().splitlines()[2:2 + N]
                final = np.array([[float(v) for v in ln.split()[1:4]] for ln in lines])
        finally:
            os.chdir(cwd)
    out[tag + "_x0_ang"] = x0
    out[tag + "_argv"] = np.array(" ".join(argv))
    out[tag + "_nit"] = np.array(len(rec))
    for k, r in enumerate(rec):
        for key in ("geom", "B_g", "new", "mv", "H"):
            out[f"{tag}_i{k}_{key}"] = r[key]
        out[f"{tag}_i{k}_sc"] = np.array([r["B_e"], r["pre_B_e"], r["tr"], r["opt_tr"]])
        out[f"{tag}_i{k}_conv"] = np.array([float(conv[k][0]), conv[k][1], conv[k][2]])
        out[f"{tag}_i{k}_disp"] = conv[k][3]
    out[tag + "_profile"] = np.array([[float(v) for v in row] for row in rows]) if rows else np.zeros((0, 0))
    if final is not None:
        out[tag + "_final_ang"] = final
    print("G8", tag, "iterations:", len(rec), "converged:", bool(conv and conv[-1][0]))


def g8_optmain():
    """G8: per-iteration trajectories of the reference's own optmain driver (Ar clusters, -os lj)."""
    out = {}
    _run_optmain("a", 7, ["-opt", "rsirfo_bfgs", "-ns", "8"], out)
    _run_optmain("b", 13, ["-opt", "rsirfo_block_fsb", "-ns", "8", "-fc", "5"], out)
    _run_optmain("c", 7, ["-opt", "rsprfo_bofill", "-order", "1", "-ns", "6", "-fc", "5"], out)
    # `-pc`: a bond and an angle kept by the constrained optimizer the dispatcher builds for crsirfo_* keys (optimizer.py:450-451)
    _run_optmain("d", 7, ["-opt", "crsirfo_bfgs", "-ns", "6", "-pc", "bond", "1,2", "angle", "3,4,5"], out)
    # mode-following RS-I-RFO through the dispatcher (optimizer.py:438-448): Cartesian overlaps, and mass-weighted ones following mode 1
    _run_optmain("e", 7, ["-opt", "mf_rsirfo_bofill", "-order", "1", "-ns", "6", "-fc", "5"], out)
    _run_optmain("f", 7, ["-opt", "mwmf_rsirfo_fsb:1", "-order", "1", "-ns", "6", "-fc", "5"], out)
    np.savez_compressed(os.path.join(HERE, "g8_optmain.npz"), **out)


def _run_nebmain(tag, N, argv, out):
    """Run the unmodified nebmain driver (multioptpy/neb.py) with the LJ engine in a scratch directory.  Per iteration:
    what the engine returned (E, G, X read back from the XYZ files), the tmp_hessian_i.npy files the optimizer saw, what
    RFOOptimizer.optimize returned, and the geometry after constraints/Kabsch chain that goes to the next input files."""
    stub = types.ModuleType("multioptpy.Calculator.ase_calculation_tools")      # the one unguarded ASE import
    stub.ASEEngine = stub.Calculation = type("ASEUnavailable", (), {})
    sys.modules.setdefault(stub.__name__, stub)
    import multioptpy.interface
    import multioptpy.neb
    from multioptpy.Optimizer import rfo_neb
    from multioptpy.Calculator import lj_calculation_tools as ljt
    X = cases.band(8, N)
    cwd = os.getcwd()
    calc_rec, opt_rec, post_rec, fire_state = [], [], [], []
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            os.makedirs("ends")
            for k, x in enumerate((X[0], X[-1])):
                with open(f"ends/ar_{k}.xyz", "w") as f:
                    f.write(f"{N}\n0 1\n")
                    for r in x * cases.BOHR2ANG:
                        f.write(f"Ar {r[0]:.10f} {r[1]:.10f} {r[2]:.10f}\n")
            parser = multioptpy.interface.init_parser()
            args = multioptpy.interface.nebparser(parser, ["ends", "-os", "lj"] + argv)
            from multioptpy.Optimizer import fire_neb, gradientdescent_neb
            o_calc, o_opt, o_print = ljt.LJEngine.calculate, rfo_neb.RFOOptimizer.optimize, multioptpy.neb.NEB.print_geometry_list
            o_fire, o_sd = fire_neb.FIREOptimizer.optimize, gradientdescent_neb.SteepestDescentOptimizer.optimize

            nested = [0]          # RFOOptimizer.optimize runs a fresh FIREOptimizer inside (rfo_neb.py:187): not a driver-level call

            def w_fire(self, geom, force, *rest):
                o = o_fire(self, geom, force, *rest)
                if not nested[0]:
                    opt_rec.append((np.array(o, dtype=float), {}, {}))
                    fire_state.append((self.dt, self.a, self.n_reset))
                return o

            def w_sd(self, geom, force, **kw):
                o = o_sd(self, geom, force, **kw)
                opt_rec.append((np.array(o, dtype=float), {}, {}))
                fire_state.append((-1.0, -1.0, -1))
                return o

            def w_calc(self, file_directory, optimize_num, pre_total_velocity, config):
                r = o_calc(self, file_directory, optimize_num, pre_total_velocity, config)
                calc_rec.append((np.array(r[0]), np.array(r[1]), np.array(r[2])))
                return r

            def w_opt(self, geom, grads, pre_geom, pre_grads, optimize_num, E, *rest):
                import glob
                hs = {}
                for f in glob.glob(os.path.join(self.config.NEB_FOLDER_DIRECTORY, "tmp_hessian_*.npy")):
                    hs[int(os.path.basename(f)[len("tmp_hessian_"):-4])] = np.load(f)
                nested[0] += 1
                try:
                    o = o_opt(self, geom, grads, pre_geom, pre_grads, optimize_num, E, *rest)
                finally:
                    nested[0] -= 1
                hs_after = {}
                for f in glob.glob(os.path.join(self.config.NEB_FOLDER_DIRECTORY, "tmp_hessian_*.npy")):
                    hs_after[int(os.path.basename(f)[len("tmp_hessian_"):-4])] = np.load(f)
                opt_rec.append((np.array(o, dtype=float), hs, hs_after))
                return o

            def w_print(self, new_geometry, element_list, cm):
                post_rec.append(np.array(new_geometry, dtype=float))
                return o_print(self, new_geometry, element_list, cm)

            ljt.LJEngine.calculate, rfo_neb.RFOOptimizer.optimize, multioptpy.neb.NEB.print_geometry_list = w_calc, w_opt, w_print
            fire_neb.FIREOptimizer.optimize, gradientdescent_neb.SteepestDescentOptimizer.optimize = w_fire, w_sd
            try:
                with quiet():
                    neb = multioptpy.neb.NEB(args)
                    neb.run()
            finally:
                ljt.LJEngine.calculate, rfo_neb.RFOOptimizer.optimize, multioptpy.neb.NEB.print_geometry_list = o_calc, o_opt, o_print
                fire_neb.FIREOptimizer.optimize, gradientdescent_neb.SteepestDescentOptimizer.optimize = o_fire, o_sd
        finally:
            os.chdir(cwd)
    nit = len(opt_rec)
    post = post_rec[-nit:]                    # (print_geometry_list is also called once while the input band is set up)
    out[tag + "_argv"] = np.array(" ".join(argv))
    out[tag + "_ends_bohr"] = np.array([X[0], X[-1]])
    out[tag + "_nit"] = np.array(nit)
    for k in range(nit):
        out[f"{tag}_i{k}_E"], out[f"{tag}_i{k}_G"], out[f"{tag}_i{k}_X"] = calc_rec[k]
        out[f"{tag}_i{k}_opt"] = opt_rec[k][0]
        out[f"{tag}_i{k}_post"] = post[k]
        if fire_state:
            out[f"{tag}_i{k}_fire"] = np.array(fire_state[k], dtype=float)
        for i, h in opt_rec[k][1].items():
            out[f"{tag}_i{k}_hin{i}"] = h
        for i, h in opt_rec[k][2].items():
            out[f"{tag}_i{k}_hout{i}"] = h
    print("G8 neb", tag, "iterations:", nit, "images:", calc_rec[0][0].shape[0])


def g13_path_tools():
    """G13: the path tools nebmain calls between the optimizer and the next input files (neb.py:561-572): calc_path_length_list,
    distribute_geometry (`-ad`), the natural-spline extrema and the IDPP objective / gradient / three-image relaxation behind
    apply_climbing_image (`-ci`), and apply_climbing_image itself on a 10-image Ar7 band with a synthetic two-barrier profile."""
    from multioptpy.Utils import calc_tools as ct
    from multioptpy.Interpolation import linear_interpolation as li
    from multioptpy.Interpolation import interpolation as ip
    from multioptpy.Potential import idpp as idpp_mod
    out = {}
    rng = np.random.default_rng(13)
    X = cases.band(10, 7) * cases.BOHR2ANG
    X = X + 0.05 * rng.normal(size=X.shape)
    out["X"] = X
    out["path_length"] = np.array(ct.calc_path_length_list(list(X)))
    out["distributed"] = np.array(li.distribute_geometry(list(X)))
    # uneven band: two images close together, one far apart
    Xu = X.copy()
    Xu[3] = 0.9 * Xu[2] + 0.1 * Xu[4]
    Xu[6] = 0.2 * Xu[5] + 0.8 * Xu[7]
    out["Xu"] = Xu
    out["distributed_u"] = np.array(li.distribute_geometry(list(Xu)))
    E = np.array([0.0, 0.004, 0.011, 0.019, 0.013, 0.009, 0.015, 0.012, 0.005, 0.001])
    out["E"] = E
    with quiet():
        mx, mn = ip.spline_interpolation(out["path_length"], E)
    out["spline_max"] = np.array(mx, dtype=float).reshape(-1, 2)
    out["spline_min"] = np.array(mn, dtype=float).reshape(-1, 2)
    inst = idpp_mod.IDPP()
    trio = [X[3].copy(), 0.4 * X[3] + 0.6 * X[4] + 0.03 * rng.normal(size=X[3].shape), X[4].copy()]
    out["trio"] = np.array(trio)
    f, g = inst.get_func_and_deriv(trio, 3, 1)
    out["idpp_obj"], out["idpp_grad"] = np.array(f), np.array(g)
    with quiet():
        rel = idpp_mod.IDPP().opt_path([t.copy() for t in trio], ["Ar"] * 7)
    out["idpp_relaxed"] = np.array(rel)
    with quiet():
        ci = ct.apply_climbing_image([x.copy() for x in X], list(E), ["Ar"] * 7)
    out["ci_out"] = np.array(ci)
    np.savez_compressed(os.path.join(HERE, "g13_path_tools.npz"), **out)
    moved = [i for i in range(len(X)) if np.abs(out["ci_out"][i] - X[i]).max() > 0]
    print("G13 written; spline maxima:", out["spline_max"][:, 0], "images moved by the climbing-image step:", moved)


def g8_nebmain():
    """G8 (NEB half): the reference's nebmain loop on an Ar7 band, RFO branch, exact Hessians every 3 iterations."""
    out = {}
    _run_nebmain("a", 7, ["-fc", "3", "-ns", "5", "-p", "5", "-ad", "2"], out)         # images re-distributed on iterations 2 and 4
    _run_nebmain("c", 7, ["-fc", "3", "-ns", "5", "-p", "5", "-ci", "1", "2"], out)    # climbing-image move on iteration 3
    _run_nebmain("n", 7, ["-fc", "3", "-ns", "5", "-p", "5"], out)
    _run_nebmain("m", 7, ["-fc", "2", "-ns", "4", "-p", "5", "-ma", "150", "1,2", "5,6"], out)      # AFIR bias per image (neb.py:806-832)
    _run_nebmain("e", 7, ["-fc", "2", "-ns", "4", "-p", "5", "-fe", "3"], out)
    _run_nebmain("f", 7, ["-ns", "9", "-p", "5"], out)                       # FIRE branch (the default optimizer of nebmain)
    _run_nebmain("s", 7, ["-ns", "6", "-p", "5", "-sd", "3", "-fe", "1"], out)     # FIRE, then steepest descent from iteration 3
    np.savez_compressed(os.path.join(HERE, "g8_nebmain.npz"), **out)
def g10c_lindh2007d3():
    """G10c: Lindh2007D3ApproxHessian.main (lindh2007d3.py:777-822) -- the model ApproxHessian.main falls back to for unknown names
    (approx_hessian.py:88-91) -- on G10's geometries with a small non-zero gradient (it scales the eigenvalues), through the class, through
    the dispatcher by name and through the fall-back, and the dispatcher's `clip` modifier (approx_hessian.py:104-110) on top of it."""
    from multioptpy.ModelHessian.lindh2007d3 import Lindh2007D3ApproxHessian
    from multioptpy.ModelHessian.approx_hessian import ApproxHessian
    z = np.load(os.path.join(HERE, "g10_model_hessian.npz"))
    out = {"tags": z["tags"]}
    rng = np.random.default_rng(107)
    for tag in z["tags"]:
        x, elems = z[tag + "_x"], [str(e) for e in z[tag + "_elems"]]
        g = rng.normal(0.0, 0.02, x.shape)
        with quiet():
            H = Lindh2007D3ApproxHessian().main(x.copy(), list(elems), g.copy())
            Hd = ApproxHessian().main(x.copy(), list(elems), g.copy(), approx_hess_type="lindh2007d3")
            Hf = ApproxHessian().main(x.copy(), list(elems), g.copy(), approx_hess_type="no_such_model")
            Hc = ApproxHessian().main(x.copy(), list(elems), g.copy(), approx_hess_type="lindh2007d3_clip")
        out[tag + "_x"], out[tag + "_elems"], out[tag + "_g"] = x, z[tag + "_elems"], g
        out[tag + "_H"], out[tag + "_Hdisp"], out[tag + "_Hfallback"], out[tag + "_Hclip"] = np.array(H), np.array(Hd), np.array(Hf), np.array(Hc)
    np.savez_compressed(os.path.join(HERE, "g10c_lindh2007d3.npz"), **out)
    print("G10c:", len(out["tags"]), "geometries")


def g15_mf_rsirfo():
    """G15: consecutive MF_RSIRFO.run calls of the reference (mode following: target index, EMA, gradient bias, mass-weighted overlaps,
    Hungarian / greedy matching, static references) on the LJ surface, geometry advanced by the returned step."""
    from multioptpy.Optimizer.mf_rsirfo import MF_RSIRFO
    from multioptpy.Parameters.atomic_mass import atomic_mass
    mixed = ["Ar", "Ne", "Kr", "He", "Xe", "Ar", "Ne", "Kr", "He", "Xe", "Ar", "Ne", "Kr", "He"]
    cfgs = [
        # (method string, saddle_order, N, steps, element_list or None, extra keywords, jitter)
        ("mf_rsirfo_bofill", 1, 7, 6, None, {}, 0.3),
        ("mf_rsirfo_fsb:1", 1, 7, 6, None, {}, 0.3),
        ("mf_rsirfo_bofill:0:ema0.5", 1, 11, 6, None, {}, 0.3),
        ("mf_rsirfo_bofill:ema0.3:grad0.3", 1, 11, 6, None, {}, 0.3),
        ("mwmf_rsirfo_bofill", 1, 7, 6, mixed[:7], {}, 0.3),
        ("mwsmf_rsirfo_fsb:1", 1, 11, 6, mixed[:11], {"adaptive_mode_following": False}, 0.3),
        ("smf_rsirfo_bfgs", 1, 7, 5, None, {"adaptive_mode_following": False}, 0.3),
        ("mf_rsirfo_bofill:1", 2, 11, 6, None, {}, 0.3),
        ("mf_rsirfo_bofill", 2, 7, 5, None, {"use_hungarian": False}, 0.3),
        ("mf_rsirfo_block_fsb", 1, 14, 5, mixed[:14], {}, 0.2),
        ("mf_rsirfo_bfgs", 0, 7, 4, None, {}, 0.2),
        ("mf_rsirfo_bofill:2", 1, 22, 4, None, {}, 0.3),
    ]
    out = {"ncfg": np.array(len(cfgs))}
    for ci, (method, so, N, nst, elems, kw, jit) in enumerate(cfgs):
        tag = f"m{ci:02d}"
        n = 3 * N
        x = cases.cluster(N, jitter=jit) * 0.95
        H = cases.lj(x, hessian=True)[2].copy()
        with quiet():
            opt = MF_RSIRFO(method=method, saddle_order=so, element_list=elems, **kw)
        out[tag + "_meta"] = np.array([N, so, nst])
        out[tag + "_method"] = np.array(method)
        out[tag + "_kw"] = np.array(repr(kw))
        out[tag + "_masses"] = np.array([atomic_mass(e) for e in elems]) if elems else np.zeros(0)
        out[tag + "_elems"] = np.array(elems if elems else [])
        out[tag + "_H0"] = H.copy()
        px = pg = pE = None
        for k in range(nst):
            E, G = cases.lj(x)
            g = G.reshape(-1, 1)
            xx = x.reshape(-1, 1).copy()
            opt.set_hessian(H)               # optmain re-sets the same array every step (F7)
            opt.set_bias_hessian(np.zeros((n, n)))
            with quiet():
                if k == 0:
                    step = opt.run(xx, g, [], [], E, 0.0, [], [], g, [])
                else:
                    step = opt.run(xx, g, pg, px, E, pE, [], [], g, pg)
            out[f"{tag}_s{k}_x"] = xx.copy()
            out[f"{tag}_s{k}_g"] = g.copy()
            out[f"{tag}_s{k}_E"] = np.array(E)
            out[f"{tag}_s{k}_step"] = np.asarray(step).copy()
            out[f"{tag}_s{k}_tr"] = np.array(opt.trust_radius)
            out[f"{tag}_s{k}_pred"] = np.array(opt.predicted_energy_changes[-1])
            out[f"{tag}_s{k}_targets"] = np.array(opt.mode_follower.reference_indices, dtype=np.int64)
            out[f"{tag}_s{k}_Hself"] = np.array(opt.hessian).copy()
            out[f"{tag}_s{k}_Hcaller"] = H.copy()
            px, pg, pE = xx, g, E
            x = (xx - step).reshape(-1, 3)
    np.savez_compressed(os.path.join(HERE, "g15_mf_rsirfo.npz"), **out)
    print("G15:", len(cfgs), "traces")


def g12b_constraint_kinds():
    """G12b: the constraint kinds fbond / rot / eigvec / atoms_pair of ProjectOutConstrain (constraint_condition.py:385-437, 481-497,
    556-612, 764-797) next to a plain bond: initial values, constraint rows, gradient projection, two Hessian projections, and the
    alignment + SHAKE-like adjustment (which centres the stored initial geometry in place)."""
    from multioptpy.Constraint.constraint_condition import ProjectOutConstrain
    from multioptpy.ModelHessian.fischerd3old import FischerD3ApproxHessianOld
    elems, xyz = _read_xyz("/root/reference/test/aldol_rxn.xyz")
    x0 = xyz / cases.BOHR2ANG
    N = len(elems)
    n = 3 * N
    with quiet():
        K = FischerD3ApproxHessianOld().main(x0.copy(), list(elems), np.zeros_like(x0))
    Hm = K + np.diag(0.05 + 0.01 * np.arange(n) / n)          # no zero modes, no exact degeneracies: the eigvec kind picks a well-defined mode
    rng = np.random.default_rng(1212)
    sets = [
        (["fbond", "bond"], [[1, 2, 3, 4, 2], [5, 6]]),
        (["rot", "bond"], [[1, 2, 3], [4, 5]]),
        (["eigvec", "bond"], [[1], [2, 3]]),
        (["atoms_pair", "atoms_pair", "angle"], [[1, 2], [3, 4], [5, 1, 6]]),
        (["fbond", "rot", "eigvec", "atoms_pair", "x"], [[4, 5, 6, 7, 8, 3], [1, 2, 3], [2], [9, 10], [11]]),
    ]
    out = {"x0": x0, "elems": np.array(elems), "Hm": Hm, "nsets": np.array(len(sets))}
    for si, (names, atoms) in enumerate(sets):
        tag = f"k{si}"
        out[tag + "_names"] = np.array(names)
        out[tag + "_atoms_flat"] = np.array([a for lst in atoms for a in lst])
        out[tag + "_atoms_len"] = np.array([len(lst) for lst in atoms])
        with quiet():
            pc = ProjectOutConstrain(list(names), [list(a) for a in atoms])
            xinit = x0.copy()
            init = pc.initialize(xinit, hessian=Hm)
            out[tag + "_init_scalars"] = np.array([float(v) for v, nm in zip(init, names) if nm not in ("rot", "eigvec", "atoms_pair")])
            for k in range(2):
                x = x0 + rng.normal(0, 0.03, x0.shape)
                g = rng.normal(size=(N, 3))
                H = Hm + 0.05 * np.eye(n) * k
                out[f"{tag}_p{k}_x"], out[f"{tag}_p{k}_g"], out[f"{tag}_p{k}_Hin"] = x, g, H
                out[f"{tag}_p{k}_B"] = np.array(pc._get_all_constraint_vectors(x))
                out[f"{tag}_p{k}_gproj"] = np.array(pc.calc_project_out_grad(x.copy(), g.copy()))
                out[f"{tag}_p{k}_Hproj"] = np.array(pc.calc_project_out_hess(x.copy(), g.copy(), H.copy()))
            xa = x0 + rng.normal(0, 0.05, x0.shape)
            out[tag + "_shake_in"] = xa.copy()
            out[tag + "_shake_out"] = np.array(pc.adjust_init_coord(xa.copy(), hessian=Hm))
            out[tag + "_shake_iters"] = np.array(pc.last_shake_iter)
            out[tag + "_xinit_after"] = xinit.copy()         # the initial geometry object after the Kabsch call centred it
    np.savez_compressed(os.path.join(HERE, "g12b_constraint_kinds.npz"), **out)
    print("G12b written:", len(sets), "constraint sets")


def g10d_ts_modifier():
    """G10d: the `ts` modifier of ApproxHessian.main (approx_hessian.py:95-98 -> ModelHessian/tshess.py:14-40: the lowest non-zero mode of
    the model Hessian is reflected) on G10's geometries, on top of fischerd3old and of schlegel."""
    from multioptpy.ModelHessian.approx_hessian import ApproxHessian
    z = np.load(os.path.join(HERE, "g10_model_hessian.npz"))
    out = {"tags": z["tags"]}
    for tag in z["tags"]:
        x, elems = z[tag + "_x"], [str(e) for e in z[tag + "_elems"]]
        g = np.zeros_like(x)
        out[tag + "_x"], out[tag + "_elems"] = x, z[tag + "_elems"]
        for model in ("fischerd3old", "schlegel"):
            try:
                with quiet():
                    Hm = ApproxHessian().main(x.copy(), list(elems), g.copy(), approx_hess_type=model + "_ts")
                out[f"{tag}_Hts_{model}"] = np.array(Hm)
            except IndexError:                                # every eigenvalue below 1e-8 in magnitude (tshess.py:31 indexes past the end)
                out[f"{tag}_Hts_{model}"] = np.zeros(0)
    np.savez_compressed(os.path.join(HERE, "g10d_ts_modifier.npz"), **out)
    print("G10d:", len(out["tags"]), "geometries")


def g10e_sr_modifier():
    """G10e: the `sr` modifier of ApproxHessian.main (approx_hessian.py:100-102 -> ModelHessian/shortrange.py) and modifier combinations in the
    reference's order ts -> sr -> clip, through the dispatcher by name, on G10's geometries."""
    from multioptpy.ModelHessian.approx_hessian import ApproxHessian
    z = np.load(os.path.join(HERE, "g10_model_hessian.npz"))
    names = ["fischerd3old_sr", "schlegel_sr_clip", "fischer_ts_sr", "morse_ts_clip", "lindh2007d3_sr", "fischerd3_ts_sr_clip"]
    out = {"tags": z["tags"], "names": np.array(names)}
    rng = np.random.default_rng(211)
    for tag in z["tags"]:
        x, elems = z[tag + "_x"], [str(e) for e in z[tag + "_elems"]]
        g = rng.normal(0.0, 0.02, x.shape)
        out[tag + "_x"], out[tag + "_elems"], out[tag + "_g"] = x, z[tag + "_elems"], g
        for name in names:
            try:
                with quiet():
                    Hm = ApproxHessian().main(x.copy(), list(elems), g.copy(), approx_hess_type=name)
                out[f"{tag}_H_{name}"] = np.array(Hm)
            except IndexError:                                # (`ts` on a model Hessian without a non-zero mode, see G10d)
                out[f"{tag}_H_{name}"] = np.zeros(0)
    np.savez_compressed(os.path.join(HERE, "g10e_sr_modifier.npz"), **out)
    print("G10e:", len(out["tags"]), "geometries x", len(names), "names")


def g14c_well_potentials():
    """G14c: the four well potentials of Potential/switching_potential.py (WellPotential, WellPotentialWall, WellPotentialVP, WellPotentialAround): energy,
    torch.func.jacrev gradient and torch.func.hessian Hessian as Potential/potential.py:127-152 forms them, with the distance in each of the five
    regions of the well (linear wall below a, quintic on [a, b], zero, quintic on [c, d), linear wall beyond d)."""
    import torch
    from multioptpy.Potential.switching_potential import WellPotential, WellPotentialWall, WellPotentialVP, WellPotentialAround
    rng = np.random.default_rng(1415)
    N = 7
    base = rng.normal(0.0, 2.0, (N, 3))
    lim = [1.0, 2.0, 4.0, 5.5]                                 # angstrom
    cases_ = []
    # kind 0: fragment centres at chosen distances (bohr): inside every region
    for dist in (1.2, 2.9, 5.6, 8.8, 11.5):
        x = base.copy()
        f1, f2 = [1, 2, 3], [5, 6]
        c1 = x[[0, 1, 2]].mean(axis=0); c2 = x[[4, 5]].mean(axis=0)
        v = c2 - c1
        x[[4, 5]] += (c1 + v / np.linalg.norm(v) * dist) - c2
        cases_.append(("well", x, (f1, f2), 120.0, lim, None))
    cases_.append(("well", base.copy(), ([7], [1, 2, 3, 4]), 60.0, [0.5, 1.0, 1.5, 2.5], None))
    # kind 1: walls along x / y / z (targets on both sides of the plane)
    for d_, tg in (("x", [1, 2, 3, 4, 5, 6, 7]), ("y", [2, 4, 6]), ("z", [1, 7])):
        cases_.append(("wall", base * 1.8, tg, 80.0, lim, d_))
    # kind 2: a void point; kind 3: around a centre
    cases_.append(("vp", base * 1.5, [1, 2, 3, 4, 5, 6, 7], 100.0, lim, [0.31, -0.2, 0.45]))
    cases_.append(("vp", base * 2.5, [2, 5], 40.0, [0.8, 1.1, 3.0, 3.2], [1.0, 1.0, -1.0]))
    cases_.append(("around", base * 1.7, ([1, 2], [3, 4, 5, 6, 7]), 90.0, lim, None))
    cases_.append(("around", base * 2.2, ([4], [1, 2, 4, 6]), 30.0, [0.5, 1.5, 4.5, 6.0], None))      # (a target that is also the centre: r = 0)
    out = {"ncase": np.array(len(cases_))}
    for ci, (kind, x, atoms, e0, lims, extra) in enumerate(cases_):
        tag = f"w{ci:02d}"
        if kind == "well":
            pot = WellPotential(well_pot_wall_energy=e0, well_pot_fragm_1=atoms[0], well_pot_fragm_2=atoms[1], well_pot_limit_dist=lims)
            out[tag + "_l1"], out[tag + "_l2"] = np.array(atoms[0]), np.array(atoms[1])
        elif kind == "wall":
            pot = WellPotentialWall(wall_well_pot_wall_energy=e0, wall_well_pot_direction=extra, wall_well_pot_limit_dist=lims, wall_well_pot_target=atoms)
            out[tag + "_l1"], out[tag + "_l2"], out[tag + "_dir"] = np.array(atoms), np.zeros(0, dtype=int), np.array(extra)
        elif kind == "vp":
            pot = WellPotentialVP(void_point_well_pot_wall_energy=e0, void_point_well_pot_coordinate=list(extra), void_point_well_pot_limit_dist=lims, void_point_well_pot_target=atoms)
            out[tag + "_l1"], out[tag + "_l2"], out[tag + "_point"] = np.array(atoms), np.zeros(0, dtype=int), np.array(extra, dtype=float)
        else:
            pot = WellPotentialAround(around_well_pot_wall_energy=e0, around_well_pot_center=atoms[0], around_well_pot_limit_dist=lims, around_well_pot_target=atoms[1])
            out[tag + "_l1"], out[tag + "_l2"] = np.array(atoms[0]), np.array(atoms[1])
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        with quiet():
            E = pot.calc_energy(xt, [])
            g = torch.func.jacrev(pot.calc_energy, argnums=0)(xt, [])
            H = torch.func.hessian(pot.calc_energy, argnums=0)(xt, [])
        out[tag + "_kind"], out[tag + "_x"], out[tag + "_e0"], out[tag + "_lim"] = np.array(kind), x, np.array(e0), np.array(lims, dtype=float)
        out[tag + "_E"] = np.array(float(E))
        out[tag + "_g"] = np.asarray(g.detach().numpy(), dtype=float).reshape(N, 3)
        out[tag + "_H"] = np.asarray(H.detach().numpy(), dtype=float).reshape(3 * N, 3 * N)
    np.savez_compressed(os.path.join(HERE, "g14c_well_potentials.npz"), **out)
    print("G14c:", len(cases_), "cases")


def g14b_keep_angle():
    """G14b: StructKeepAnglePotential / StructKeepDihedralAnglePotential (Potential/keep_angle_potential.py, keep_dihedral_angle_potential.py):
    energy, torch.func.jacrev gradient and torch.func.hessian Hessian as Potential/potential.py:127-152 forms them -- ordinary geometries, angles
    within 1e-3 rad of 0 and 180 degrees (Taylor branches), equilibrium angles 0 / 180 / general, dihedrals across the +-180 wrap and inside the
    collinearity switch."""
    import torch
    from multioptpy.Potential.keep_angle_potential import StructKeepAnglePotential
    from multioptpy.Potential.keep_dihedral_angle_potential import StructKeepDihedralAnglePotential
    rng = np.random.default_rng(1414)
    N = 6
    cases_ = []
    base = rng.normal(0.0, 1.5, (N, 3))

    def with_angle(theta):          # atoms 1-2-3 with the given angle at atom 2
        x = base.copy()
        x[1] = [0.0, 0.0, 0.0]
        x[0] = [1.7, 0.0, 0.0]
        x[2] = [1.3 * math.cos(theta), 1.3 * math.sin(theta), 0.0]
        return x
    import math
    for theta, th0 in ((1.9, 109.5), (0.7, 60.0), (math.pi - 4e-4, 180.0), (math.pi - 4e-4, 120.0), (3e-4, 0.0), (3e-4, 90.0), (2.5, 180.0), (0.4, 0.0), (math.pi - 0.2, 170.0)):
        cases_.append(("angle", with_angle(theta) + rng.normal(0, 1e-5, (N, 3)) * (0 if theta < 1e-3 or theta > math.pi - 1e-3 else 1), [1, 2, 3], 0.35, th0))
    cases_.append(("angle", base.copy(), [4, 6, 5], 0.2, 100.0))

    def with_dihedral(phi, bend=1.9):      # atoms 1-2-3-4
        x = base.copy()
        x[1] = [0.0, 0.0, 0.0]
        x[2] = [1.5, 0.0, 0.0]
        x[0] = [1.2 * math.cos(bend), 1.2 * math.sin(bend), 0.0]
        x[3] = [1.5 - 1.1 * math.cos(bend), 1.1 * math.sin(bend) * math.cos(phi), 1.1 * math.sin(bend) * math.sin(phi)]
        return x
    for phi, ph0, bend in ((1.0, 60.0, 1.9), (-2.9, 175.0, 1.9), (3.0, -170.0, 2.0), (0.3, 0.0, 1.2), (2.0, 120.0, math.pi - 6e-5), (1.2, 30.0, math.pi - 2e-5), (-1.0, -90.0, 1.7)):
        cases_.append(("dihedral", with_dihedral(phi, bend), [1, 2, 3, 4], 0.25, ph0))
    cases_.append(("dihedral", base.copy(), [2, 5, 3, 6], 0.15, 45.0))
    # out-of-plane angles (centre, probe, two plane atoms) and fragment-centre distances
    from multioptpy.Potential.keep_outofplain_angle_potential import StructKeepOutofPlainAnglePotential
    from multioptpy.Potential.keep_potential import StructKeepPotentialv2
    cases_.append(("oop", base.copy(), [1, 2, 3, 4], 0.3, 20.0))
    cases_.append(("oop", base.copy(), [5, 2, 6, 3], 0.2, -35.0))
    xo = base.copy()
    xo[0] = [0.0, 0.0, 0.0]; xo[2] = [1.4, 0.0, 0.0]; xo[3] = [2.8, 3e-5, 0.0]; xo[1] = [0.3, 0.4, 1.1]        # plane atoms collinear with the centre: guard
    cases_.append(("oop", xo, [1, 2, 3, 4], 0.3, 10.0))
    cases_.append(("frag", base.copy(), [[1, 2, 3], [4, 5]], 0.12, 2.1))
    cases_.append(("frag", base.copy(), [[6], [1, 2, 3, 4]], 0.2, 1.4))
    out = {"ncase": np.array(len(cases_))}
    for ci, (kind, x, atoms, k, ang) in enumerate(cases_):
        tag = f"a{ci:02d}"
        if kind == "angle":
            pot = StructKeepAnglePotential(keep_angle_spring_const=k, keep_angle_angle=ang, keep_angle_atom_pairs=atoms)
        elif kind == "oop":
            pot = StructKeepOutofPlainAnglePotential(keep_out_of_plain_angle_spring_const=k, keep_out_of_plain_angle_angle=ang, keep_out_of_plain_angle_atom_pairs=atoms)
        elif kind == "frag":
            pot = StructKeepPotentialv2(keep_pot_v2_spring_const=k, keep_pot_v2_distance=ang, keep_pot_v2_fragm1=atoms[0], keep_pot_v2_fragm2=atoms[1])
        else:
            pot = StructKeepDihedralAnglePotential(keep_dihedral_angle_spring_const=k, keep_dihedral_angle_angle=ang, keep_dihedral_angle_atom_pairs=atoms)
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        E = pot.calc_energy(xt, [])
        g = torch.func.jacrev(pot.calc_energy, argnums=0)(xt, [])
        H = torch.func.hessian(pot.calc_energy, argnums=0)(xt, [])
        out[tag + "_kind"] = np.array(kind)
        out[tag + "_x"] = x
        out[tag + "_atoms"] = np.array(atoms if kind != "frag" else atoms[0] + atoms[1])
        out[tag + "_n1"] = np.array(len(atoms[0]) if kind == "frag" else 0)
        out[tag + "_k"] = np.array(k)
        out[tag + "_angle_deg"] = np.array(ang)
        out[tag + "_E"] = np.array(float(E))
        out[tag + "_g"] = g.detach().numpy().copy()
        out[tag + "_H"] = H.detach().numpy().reshape(3 * N, 3 * N).copy()
    np.savez_compressed(os.path.join(HERE, "g14b_keep_angle.npz"), **out)
    print("G14b:", len(cases_), "cases")


if __name__ == "__main__":
    load_reference()
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g6b", "g7", "g8", "g8n", "g9", "g10", "g10b", "g10c", "g10d", "g10e", "g11", "g12", "g12b", "g13", "g14", "g14b", "g14c", "g15"]
    if "g13" in which:
        g13_path_tools()
    if "g1" in which:
        g1_updates()
    if "g2" in which:
        g2_rsirfo()
    if "g3" in which:
        g3_rsprfo()
    if "g10" in which:
        g10_model_hessian()
    if "g14" in which:
        g14_keep_pot()
    if "g10b" in which:
        g10b_model_hessian_variants()
    if "g10c" in which:
        g10c_lindh2007d3()
    if "g15" in which:
        g15_mf_rsirfo()
    if "g14b" in which:
        g14b_keep_angle()
    if "g10d" in which:
        g10d_ts_modifier()
    if "g10e" in which:
        g10e_sr_modifier()
    if "g14c" in which:
        g14c_well_potentials()
    if "g12b" in which:
        g12b_constraint_kinds()
    if "g11" in which:
        g11_afir()
    if "g12" in which:
        g12_constraints()
    if "g4" in which:
        g4_brentq()
    if "g5" in which:
        g5_projections()
    if "g6" in which:
        g6_neb_forces()
    if "g6b" in which:
        g6b_neb_variants()
    if "g7" in which:
        g7_neb_step()
    if "g8" in which:
        g8_optmain()
    if "g8n" in which:
        g8_nebmain()
    if "g9" in which:
        g9_trust_radius()
