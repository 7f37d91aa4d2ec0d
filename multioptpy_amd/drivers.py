"""In-memory counterparts of the reference's iteration loops: the harness that owns the
Calculator call and feeds the device-backed optimizer step.  No files: the reference's per-iteration XYZ round trip
(`%17.12f` angstrom, fileio.py:442-446) is kept as a quantisation of the geometry so that trajectories stay comparable
with the reference's (switch it off with quantise=False).

`optmain_steps` follows multioptpy/optimization.py:1780-2092 for the unbiased, unconstrained, all-active case:
  compute E, g -> Hessian hand-over with the reference's aliasing (set_bias_hessian / set_hessian) -> calc_move_vector
  -> Kabsch onto the current geometry (which centres it in place) -> displacement -> 4-criterion convergence test with
  force-adaptive displacement thresholds -> shift "previous" state.
The Calculator is any object with the reference's plugin interface:
  single_point(file_directory, element_number_list, iter, [charge, mult], method, geom_num_list=<angstrom (N,3)>)
      -> (energy [hartree], gradient (N,3) [hartree/bohr], geometry (N,3) [bohr], finish_flag)
  and the attribute Model_hess (the Hessian hand-over of lj_calculation_tools.py:195-207, tblite_calculation_tools.py)."""
import copy

import numpy as np

from .optimizer import CalculateMoveVector, BOHR2ANGSTROMS
from .Utils.calc_tools import Calculationtools, apply_climbing_image
from .Interpolation.linear_interpolation import distribute_geometry


def quantise_xyz(x_ang):
    return np.array([[float(f"{v:>17.12f}") for v in row] for row in np.asarray(x_ang)], dtype="float64")


def calculate_rms_safely(vector, threshold=1e-10):
    """optimization.py:1246-1252."""
    filtered = vector[np.abs(vector) > threshold]
    return float(np.sqrt((filtered ** 2).mean())) if filtered.size > 0 else 0.0


class ConvergenceChecker:
    """optimization.py:1254-1289; default thresholds optimization.py:129-132."""

    def __init__(self, max_force=3e-4, rms_force=2e-4, max_displacement=1.5e-3, rms_displacement=1e-3):
        self.MAX_FORCE_THRESHOLD = max_force
        self.RMS_FORCE_THRESHOLD = rms_force
        self.MAX_DISPLACEMENT_THRESHOLD = max_displacement
        self.RMS_DISPLACEMENT_THRESHOLD = rms_displacement

    def check_convergence(self, effective_gradient, displacement_vector, optimizer_instances):
        max_force = np.abs(effective_gradient).max()
        rms_force = calculate_rms_safely(effective_gradient)
        delta_max = max(0.0, self.MAX_FORCE_THRESHOLD - 1 * max_force)
        delta_rms = max(0.0, self.RMS_FORCE_THRESHOLD - 1 * rms_force)
        max_displacement = np.abs(displacement_vector).max()
        rms_displacement = calculate_rms_safely(displacement_vector)
        max_th = max(self.MAX_DISPLACEMENT_THRESHOLD, self.MAX_DISPLACEMENT_THRESHOLD + delta_max)
        rms_th = max(self.RMS_DISPLACEMENT_THRESHOLD, self.RMS_DISPLACEMENT_THRESHOLD + delta_rms)
        if (max_force < self.MAX_FORCE_THRESHOLD and rms_force < self.RMS_FORCE_THRESHOLD
                and max_displacement < max_th and rms_displacement < rms_th):
            return True, max_th, rms_th
        for opt in optimizer_instances:
            if getattr(opt, "proj_grad_converged", False):
                return True, max_th, rms_th
        return False, max_th, rms_th


def optmain_steps(calculator, start_geometry_ang, element_list, opt_method, saddle_order=0, NSTEP=300, FC_COUNT=-1,
                  max_trust_radius=None, min_trust_radius=None, quantise=True, record=None, print_flag=False, model_hessian=None,
                  MFC_COUNT=50, afir=None, keep_pot=None, projection_constraint=None, keep_angle=None, keep_dihedral=None, bias_potentials=None):
    """Run up to NSTEP optimizer iterations; returns dict(converged, iterations, geometry_bohr, energy).
    projection_constraint: `-pc` as dict(conditions=[...], atoms=[[...], ...], constants=[...]) (interface.py:453-475): a ProjectOutConstrain
    object goes to CalculateMoveVector, whose crsirfo_* keys then build the constrained optimizer (optimizer.py:450-451, optimization.py:1810-1829).
    The driver's OWN projection steps (gradient / Hessian projection, SHAKE-like adjustment of the new geometry; optimization.py:1182-1198,
    1345-1368) run only when an -opt entry is literally "crsirfo" (optimization.py:1163) -- a name that matches no optimizer key, so the
    reference then steps with FIRE; that combination is outside this package and raises."""
    natom = len(element_list)
    geometry = np.array(start_geometry_ang, dtype="float64") / BOHR2ANGSTROMS
    initial_geometry = copy.deepcopy(geometry)
    pre_geometry = np.zeros((natom, 3))
    pre_bias_gradient = np.zeros((natom, 3))
    pre_raw_gradient = np.zeros((natom, 3))
    pre_move_vector = np.zeros((natom, 3))
    pre_bias_energy = 0.0
    Model_hess = np.eye(natom * 3)
    methods = opt_method if isinstance(opt_method, (list, tuple)) else [opt_method]
    PC = None
    if projection_constraint:
        from .Constraint.constraint_condition import ProjectOutConstrain
        PC = ProjectOutConstrain(projection_constraint["conditions"], projection_constraint["atoms"], projection_constraint.get("constants", []))
        if any(str(m).lower() == "crsirfo" for m in methods):
            raise NotImplementedError("-opt crsirfo (the literal name) selects the reference's FIRE optimizer with driver-side projections: not part of this package")
    CMV = CalculateMoveVector("x", element_list, saddle_order, FC_COUNT, 0.0, None, max_trust_radius=max_trust_radius,
                              min_trust_radius=min_trust_radius, projection_constraint=PC)
    optimizer_instances = CMV.initialization(list(methods))
    for opt in optimizer_instances:
        opt.set_hessian(Model_hess)
    # (`-pc` quadruples the displacement thresholds: optimization.py:133-135)
    convergence = ConvergenceChecker(max_displacement=6e-3, rms_displacement=4e-3) if projection_constraint else ConvergenceChecker()
    tools = Calculationtools()
    converged = False
    energy = None
    it = -1
    for it in range(NSTEP):
        # StandardHandler.compute (optimization.py:318-343)
        x_ang = geometry * BOHR2ANGSTROMS
        if quantise:
            x_ang = quantise_xyz(x_ang)
        calculator.Model_hess = copy.deepcopy(Model_hess)
        e, g, geom_bohr, exit_flag = calculator.single_point(None, None, it, [0, 1], getattr(calculator, "xtb_method", ""),
                                                             geom_num_list=x_ang)
        if exit_flag:
            break
        geometry = np.array(geom_bohr, dtype="float64")
        Model_hess = copy.deepcopy(calculator.Model_hess)
        energy = e
        # periodic model Hessian (optimization.py:1884-1907): replaces Model_hess every MFC_COUNT iterations when no exact Hessian is used
        if model_hessian is not None and it % MFC_COUNT == 0 and FC_COUNT < 1:
            from .ModelHessian.approx_hessian import ApproxHessian
            Model_hess = ApproxHessian().main(geometry, element_list, np.asarray(g, dtype="float64"), model_hessian)
            calculator.Model_hess = Model_hess
        bias_hessian = np.zeros((3 * natom, 3 * natom))
        bias_energy, bias_gradient = 0.0, np.zeros((natom, 3))
        if afir is not None:
            # BiasPotentialCalculation.main (Potential/potential.py:127-152) for the AFIR term (`-ma gamma fragm_1 fragm_2`): energy,
            # gradient and Hessian of the bias from the device kernels instead of torch autograd
            from .Potential.AFIR_potential import AFIRPotential
            pot = AFIRPotential(AFIR_gamma=afir["gamma"], AFIR_Fragm_1=afir["fragm_1"], AFIR_Fragm_2=afir["fragm_2"], element_list=element_list)
            bias_energy, bias_gradient, bias_hessian = pot.calc_energy_gradient_hessian(geometry, [afir["gamma"]])
        if keep_pot:
            # harmonic distance restraints (`-kp k r0 i,j`, Potential/keep_potential.py): list of dict(pair=[i, j] 1-based, k=..., r0=... angstrom)
            from .Potential.keep_potential import keep_pot_sum
            ke, kg, kh = keep_pot_sum(geometry, [p["pair"] for p in keep_pot], [p["k"] for p in keep_pot], [p["r0"] for p in keep_pot])
            bias_energy, bias_gradient, bias_hessian = bias_energy + ke, bias_gradient + kg, bias_hessian + kh
        for terms, dih in ((keep_angle, False), (keep_dihedral, True)):
            if terms:
                # angle / dihedral-angle restraints (`-ka k theta i,j,k`, `-kda k phi i,j,k,l`; Potential/keep_angle_potential.py,
                # keep_dihedral_angle_potential.py): list of dict(atoms=[...] 1-based, k=..., angle=... degrees)
                from .Potential.keep_angle_potential import keep_angle_sum
                ke, kg, kh = keep_angle_sum(geometry, [p["atoms"] for p in terms], [p["k"] for p in terms], [p["angle"] for p in terms], dihedral=dih)
                bias_energy, bias_gradient, bias_hessian = bias_energy + ke, bias_gradient + kg, bias_hessian + kh
        for pot in (bias_potentials or ()):
            # any further bias term of Potential/potential.py that has a device counterpart (e.g. the well potentials of Potential/switching_potential.py):
            # an object with calc_energy_gradient_hessian(geometry in bohr) -> (E, grad (N, 3), H (3N, 3N)), summed as potential.py:127-152 sums them
            ke, kg, kh = pot.calc_energy_gradient_hessian(geometry)
            bias_energy, bias_gradient, bias_hessian = bias_energy + ke, bias_gradient + kg, bias_hessian + kh
        if it == 0:                                                     # _save_init_geometry (optimization.py:2206-2217)
            centre = tools.calc_center(geometry, element_list)
            pre_geometry = (geometry - centre) - centre
        # HessianManager.calc_eff_hess_for_fix_atoms_and_set_hess (optimization.py:1311-1365), all atoms active
        for inst in optimizer_instances:
            inst.set_bias_hessian(bias_hessian)
            if it % FC_COUNT == 0 or (model_hessian is not None and it % MFC_COUNT == 0):      # optimization.py:1354-1367
                inst.set_hessian(Model_hess)
        g_raw = copy.deepcopy(np.asarray(g, dtype="float64"))
        B_g = copy.deepcopy(np.asarray(g, dtype="float64")) + bias_gradient
        e = e + bias_energy                                             # B_e: the biased energy drives the trust radius (optimizer.py:740-841)
        effective_gradient = g_raw + (B_g - g_raw)
        geometry_in = geometry.copy() if record is not None else None
        new_geometry, move_vector, optimizer_instances = CMV.calc_move_vector(
            it, geometry, B_g, pre_bias_gradient, pre_geometry, e, pre_bias_energy, pre_move_vector, initial_geometry, g_raw,
            pre_raw_gradient, optimizer_instances, False, print_flag)
        new_raw = new_geometry.copy() if record is not None else None
        # ConstraintManager.project_out_translation_rotation (optimization.py:1207-1213)
        aligned, _ = tools.kabsch_algorithm(new_geometry / BOHR2ANGSTROMS, geometry)
        new_geometry = aligned * BOHR2ANGSTROMS
        displacement_vector = move_vector if it == 0 else new_geometry / BOHR2ANGSTROMS - geometry
        converge_flag, max_th, rms_th = convergence.check_convergence(effective_gradient, displacement_vector, optimizer_instances)
        if record is not None:
            record.append(dict(geom=geometry_in, B_g=B_g.copy(), B_e=float(e), pre_B_e=float(pre_bias_energy), tr=float(CMV.trust_radii),
                               new=new_raw, aligned=new_geometry.copy(), mv=move_vector.copy(),
                               H=np.array(optimizer_instances[0].hessian, dtype="float64"),
                               conv=(converge_flag, max_th, rms_th), disp=np.array(displacement_vector, dtype="float64")))
        if converge_flag:
            converged = True
            break
        pre_bias_energy = e
        pre_bias_gradient = B_g
        pre_raw_gradient = g_raw
        pre_geometry = geometry
        pre_move_vector = move_vector
        geometry = new_geometry / BOHR2ANGSTROMS
    return dict(converged=converged, iterations=it + 1, geometry_bohr=geometry, energy=energy)


# ---------------------------------------------------------------------------------------------------------------------
# nebmain (multioptpy/neb.py:489-601), RFO branch
# ---------------------------------------------------------------------------------------------------------------------
def make_band(frames_ang, partition):
    """NEB.make_geometry_list (neb.py:1013-1096) for plain linear interpolation: `partition` extra nodes per segment; the
    first frame appears twice, as in the reference (images = 1 + (partition + 1)(frames - 1) + 1)."""
    frames = [np.array(f, dtype="float64") for f in frames_ang]
    nodes = []
    for k in range(len(frames) - 1):
        delta = (frames[k + 1] - frames[k]) / (partition + 1)
        for i in range(partition + 1):
            nodes.append(frames[k] + delta * i)
    nodes.append(frames[-1])
    return np.array([frames[0]] + nodes, dtype="float64")


def nebmain_steps(engine, frames_ang, element_list, NSTEP=10, partition=5, FC_COUNT=1, fixedges=0, APPLY_CI_NEB=99999,
                  ratio_of_rfo_step=0.5, quantise=True, record=None, device=None, steepest_descent=99999, devices=None,
                  transport="rccl", model_hessian=None, MFC_COUNT=50, align_distances=0, climbing_image=(999999, 999999), afir=None, keep_pot=None, keep_angle=None, keep_dihedral=None, bias_potentials=None):
    """Band relaxation as in nebmain: with FC_COUNT != -1 the Hessian-based branch (`-fc`: RFOOptimizer on a BNEB force),
    otherwise the default FIRE optimizer, replaced by steepest descent from iteration `steepest_descent` on (`-sd`,
    neb.py:794-804, 939-948).
    `engine.calculate(X_ang (I,N,3), optimize_num)` -> (E (I,), G (I,N,3), X_bohr (I,N,3), hessians | None) is the in-memory
    form of the reference's Engine.calculate and its tmp_hessian_i.npy side channel.  Per iteration, in the reference's order
    (neb.py:497-597): energies/gradients -> BNEB force -> RFOOptimizer.optimize (which recomputes the force, rfo_neb.py:131)
    -> climbing-image move when due (`-ci start interval`, neb.py:561-564) -> edge fixing -> sequential Kabsch k -> k+1 (centres image
    k+1 in place) -> equal-arc-length re-distribution every `align_distances` iterations (`-ad`, neb.py:649-766; the other
    `-ad*` strategies are not built) -> `%17.12f` quantisation.
    afir / keep_pot (as for optmain_steps): the bias of NEB._apply_bias_potential (neb.py:806-832) per image -- energy and gradient added to the
    engine's, the bias Hessian added to the image's resident Hessian every step (or taken AS the Hessian while the image has none)."""
    from .MEP.pathopt_bneb_force import CaluculationBNEB
    from .Optimizer.rfo_neb import RFOOptimizer
    from .Optimizer.fire_neb import FIREOptimizer
    from .Optimizer.gradientdescent_neb import SteepestDescentOptimizer

    class _Config:
        pass

    config = _Config()
    config.NEB_FOLDER_DIRECTORY = None
    config.fix_init_edge = fixedges in (1,) or fixedges >= 3
    config.fix_end_edge = fixedges in (2,) or fixedges >= 3
    config.ratio_of_rfo_step = ratio_of_rfo_step
    config.bohr2angstroms = BOHR2ANGSTROMS
    config.apply_convergence_criteria = False
    config.dt, config.a, config.n_reset = 0.5, 0.10, 0                      # NEBConfig defaults (neb.py:213-222)
    config.FIRE_N_accelerate, config.FIRE_f_inc, config.FIRE_f_accelerate = 5, 1.10, 0.99
    config.FIRE_f_decelerate, config.FIRE_a_start, config.FIRE_dt_max = 0.5, 0.1, 1.0
    # neb.py:797: the Hessian-based branch runs with exact Hessians (-fc) or with model Hessians (-modelhess, every -mfc steps)
    rfo_branch = FC_COUNT != -1 or (MFC_COUNT != -1 and bool(model_hessian))
    geometry_ang = make_band(frames_ang, partition)
    if quantise:
        geometry_ang = np.array([quantise_xyz(x) for x in geometry_ang])
    STRING_FORCE_CALC = CaluculationBNEB(APPLY_CI_NEB)
    optimizer = (RFOOptimizer(config, persist_hessians=False, device=device, devices=devices, transport=transport) if rfo_branch
                 else FIREOptimizer(config))
    tools = Calculationtools()
    pre_geom = pre_gradients = None
    pre_total_velocity = []
    init_geometry = None
    for optimize_num in range(NSTEP):
        E, G, X, hessians = engine.calculate(geometry_ang, optimize_num)
        E = np.asarray(E, dtype="float64")
        G = np.asarray(G, dtype="float64")
        X = np.asarray(X, dtype="float64")
        if hessians is not None and rfo_branch:
            optimizer.set_image_hessians(hessians)
        E_raw = E.copy()
        if afir is not None or keep_pot or keep_angle or keep_dihedral or bias_potentials:
            Bh = []
            for i in range(len(X)):
                be, bg, bh = 0.0, np.zeros_like(X[i]), np.zeros((3 * X.shape[1], 3 * X.shape[1]))
                if afir is not None:
                    from .Potential.AFIR_potential import AFIRPotential
                    pot = AFIRPotential(AFIR_gamma=afir["gamma"], AFIR_Fragm_1=afir["fragm_1"], AFIR_Fragm_2=afir["fragm_2"], element_list=element_list)
                    e1, g1, h1 = pot.calc_energy_gradient_hessian(X[i], [afir["gamma"]])
                    be, bg, bh = be + e1, bg + g1, bh + h1
                if keep_pot:
                    from .Potential.keep_potential import keep_pot_sum
                    e1, g1, h1 = keep_pot_sum(X[i], [p["pair"] for p in keep_pot], [p["k"] for p in keep_pot], [p["r0"] for p in keep_pot])
                    be, bg, bh = be + e1, bg + g1, bh + h1
                for terms, dih in ((keep_angle, False), (keep_dihedral, True)):
                    if terms:
                        from .Potential.keep_angle_potential import keep_angle_sum
                        e1, g1, h1 = keep_angle_sum(X[i], [p["atoms"] for p in terms], [p["k"] for p in terms], [p["angle"] for p in terms], dihedral=dih)
                        be, bg, bh = be + e1, bg + g1, bh + h1
                for pot in (bias_potentials or ()):
                    e1, g1, h1 = pot.calc_energy_gradient_hessian(X[i])
                    be, bg, bh = be + e1, bg + g1, bh + h1
                E[i] += be
                G[i] += bg
                Bh.append(bh)
            if rfo_branch and (FC_COUNT > 0 or (MFC_COUNT > 0 and bool(model_hessian))):      # neb.py:819
                optimizer.set_bias_hessians(Bh)
        if FC_COUNT == -1 and model_hessian and optimize_num % MFC_COUNT == 0:          # neb.py:515-523
            optimizer.request_model_hessian(element_list, model_hessian)
        if optimize_num == 0:
            init_geometry = X
        total_force = STRING_FORCE_CALC.calc_force(X, E, G, optimize_num, element_list)
        total_velocity = np.array(total_force, dtype="float64")            # NEB.force2velocity (neb.py:1159-1162)
        if rfo_branch:
            new_geometry = optimizer.optimize(X, G, pre_geom, pre_gradients, optimize_num, E, E, pre_total_velocity, total_velocity,
                                              None, pre_geom, STRING_FORCE_CALC)
        elif optimize_num < steepest_descent:
            new_geometry = optimizer.optimize(X, total_force, pre_total_velocity, optimize_num, total_velocity, None, E, E, pre_geom)
        else:
            new_geometry = SteepestDescentOptimizer(config).optimize(X, total_force)
        new_geometry = np.array(new_geometry, dtype="float64")
        opt_out = new_geometry.copy() if record is not None else None
        ci_start, ci_interval = climbing_image
        if optimize_num > ci_start and (optimize_num - ci_start) % ci_interval == 0:
            new_geometry = np.array(apply_climbing_image(list(new_geometry), E, element_list), dtype="float64")
        # NEB._apply_constraints (neb.py:966-998), no fixed atoms
        if config.fix_init_edge:
            new_geometry[0] = init_geometry[0] * BOHR2ANGSTROMS
        if config.fix_end_edge:
            new_geometry[-1] = init_geometry[-1] * BOHR2ANGSTROMS
        for k in range(len(new_geometry) - 1):
            aligned, _ = tools.kabsch_algorithm(new_geometry[k], new_geometry[k + 1])
            new_geometry[k] = copy.copy(aligned)
        if optimize_num > 0 and align_distances >= 1 and optimize_num % align_distances == 0:      # NEB._align_geometries
            new_geometry = np.array(distribute_geometry(np.array(new_geometry)), dtype="float64")
        if record is not None:
            record.append(dict(E=E_raw, G=G.copy(), X=X.copy(), opt=opt_out, post=new_geometry.copy(),
                               hess=[optimizer.get_image_hessian(i) for i in range(len(X))] if rfo_branch else [],
                               fire=None if rfo_branch else (optimizer.dt, optimizer.a, optimizer.n_reset)))
        pre_geom, pre_gradients, pre_total_velocity = copy.copy(X), copy.copy(G), copy.copy(total_velocity)
        geometry_ang = np.array([quantise_xyz(x) for x in new_geometry]) if quantise else new_geometry
    return geometry_ang
