"""ctypes mirror of include/mopt_hip.h (structs, enums, prototypes)."""
import ctypes as C

RULE_KEYS = [
    # dispatch keys of multioptpy/Optimizer/rsirfo.py:208-251 in priority order == MOPT_RULE_* values
    "flowchart",
    "block_cfd_fsb_dd", "block_cfd_fsb_weighted", "block_cfd_fsb",
    "block_cfd_bofill_weighted", "block_cfd_bofill",
    "block_bfgs_dd", "block_bfgs",
    "block_fsb_dd", "block_fsb_weighted", "block_fsb",
    "block_bofill_weighted", "block_bofill",
    "bfgs_dd", "bfgs", "sr1", "pcfd_bofill",
    "cfd_fsb_dd", "cfd_fsb", "cfd_bofill",
    "fsb_dd", "fsb", "bofill", "psb", "msp",
]


def rule_from_method(method):
    """First substring match wins, unknown names fall back to the flowchart rule (rsirfo.py:1341-1356)."""
    low = str(method).lower()
    for i, key in enumerate(RULE_KEYS):
        if key in low:
            return i
    return 0


class RsirfoParams(C.Structure):
    _fields_ = [
        ("alpha0", C.c_double), ("alpha_max", C.c_double), ("alpha_step_max", C.c_double),
        ("small_eigval_thresh", C.c_double),
        ("trust_radius_min", C.c_double), ("trust_radius_max", C.c_double),
        ("good_step_threshold", C.c_double), ("poor_step_threshold", C.c_double),
        ("trust_radius_increase_factor", C.c_double), ("trust_radius_decrease_factor", C.c_double),
        ("energy_change_threshold", C.c_double), ("gradient_norm_threshold", C.c_double),
        ("step_norm_tolerance", C.c_double),
        ("adaptive_trust_gradient_norm_threshold", C.c_double),
        ("max_curvature_factor", C.c_double), ("negative_curvature_safety", C.c_double),
        ("level_shift_value", C.c_double), ("condition_number_threshold", C.c_double),
        ("max_micro_cycles", C.c_int32), ("saddle_order", C.c_int32), ("rule", C.c_int32),
        ("use_adaptive_trust_radius", C.c_int32),
        ("use_level_shift", C.c_int32), ("auto_level_shift", C.c_int32),
        ("neb_mode", C.c_int32), ("reserved", C.c_int32),
    ]


class RsirfoState(C.Structure):
    _fields_ = [
        ("trust_radius", C.c_double), ("prev_energy", C.c_double),
        ("predicted", C.c_double * 3), ("actual", C.c_double * 3),
        ("n_predicted", C.c_int32), ("n_actual", C.c_int32),
        ("has_prev", C.c_int32), ("iteration", C.c_int32), ("converged", C.c_int32),
        ("prev_eigvec_size", C.c_int32), ("level_shift_applied", C.c_int32),
        ("branch", C.c_int32), ("hess_updated", C.c_int32), ("n_kept", C.c_int32),
        ("jacobi_sweeps", C.c_int32), ("secular_iters", C.c_int32),
        ("alpha", C.c_double), ("lambda_aug", C.c_double),
        ("initial_step_norm", C.c_double), ("gradient_norm", C.c_double),
    ]


def default_params(**kw):
    """The defaults of RSIRFO.__init__ (rsirfo.py:26-91) for saddle_order given in kw."""
    so = kw.get("saddle_order", 1)
    p = RsirfoParams()
    p.alpha0 = 1.0
    p.alpha_max = 1000.0
    p.alpha_step_max = 10.0
    p.small_eigval_thresh = 1e-6
    p.trust_radius_min = 0.01
    p.trust_radius_max = 0.5 if so == 0 else 0.1
    p.good_step_threshold = 0.75
    p.poor_step_threshold = 0.25
    p.trust_radius_increase_factor = 1.2
    p.trust_radius_decrease_factor = 0.5
    p.energy_change_threshold = 1e-6
    p.gradient_norm_threshold = 1e-4
    p.step_norm_tolerance = 1e-3
    p.adaptive_trust_gradient_norm_threshold = 1e-2
    p.max_curvature_factor = 2.5
    p.negative_curvature_safety = 0.8
    p.level_shift_value = 1e-5
    p.condition_number_threshold = 1e8
    p.max_micro_cycles = 40
    p.saddle_order = so
    p.rule = 0
    p.use_adaptive_trust_radius = 1
    p.use_level_shift = 0
    p.auto_level_shift = 1
    p.neb_mode = 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class RsprfoParams(C.Structure):
    _fields_ = [
        ("alpha0", C.c_double), ("alpha_max", C.c_double), ("alpha_min", C.c_double), ("alpha_step_max", C.c_double),
        ("micro_cycle_rtol", C.c_double), ("micro_cycle_atol", C.c_double),
        ("trust_radius_min", C.c_double), ("trust_radius_max", C.c_double),
        ("eta_2", C.c_double), ("eta_3", C.c_double), ("gamma_1", C.c_double), ("gamma_2", C.c_double),
        ("rejection_threshold", C.c_double),
        ("min_positive_eigval", C.c_double), ("min_negative_eigval", C.c_double),
        ("gradient_scaling_threshold", C.c_double), ("min_step_scale", C.c_double), ("gradient_trust_coupling", C.c_double),
        ("overlap_threshold", C.c_double), ("mixing_threshold", C.c_double),
        ("max_micro_cycles", C.c_int32), ("saddle_order", C.c_int32), ("rule", C.c_int32),
        ("step_rejection_enabled", C.c_int32), ("max_consecutive_rejections", C.c_int32),
        ("hessian_shift_enabled", C.c_int32), ("gradient_scaling_enabled", C.c_int32), ("adaptive_trust_enabled", C.c_int32),
        ("adapt_trust_radius", C.c_int32),
        ("mode_following_enabled", C.c_int32), ("eigvec_following", C.c_int32),
        ("reserved", C.c_int32),
    ]


class RsprfoState(C.Structure):
    _fields_ = [
        ("trust_radius", C.c_double), ("prev_energy", C.c_double), ("last_predicted", C.c_double),
        ("last_ratio", C.c_double), ("last_actual", C.c_double), ("new_predicted", C.c_double),
        ("has_prev", C.c_int32), ("consecutive_rejections", C.c_int32), ("ts_mode_idx", C.c_int32), ("iter", C.c_int32),
        ("hess_updated", C.c_int32), ("shifted", C.c_int32), ("step_accepted", C.c_int32), ("micro_converged", C.c_int32),
        ("n_max", C.c_int32), ("jacobi_sweeps", C.c_int32),
        ("step_norm", C.c_double), ("gradient_norm", C.c_double),
    ]
