"""CRSIRFO: drop-in for ``multioptpy.Optimizer.crsirfo.CRSIRFO`` (reference crsirfo.py:5-182), the constrained RS-I-RFO optimizer.

Host: the SHAKE-like coordinate correction (through the constraints object, in place -- the reference's gradient transport never
fires for contiguous ndarray input because its displacement is taken after the in-place correction, crsirfo.py:62-83; kept as is),
the rank and an orthonormal basis of the constraints' ROW space (c x c Gram matrix: O(c^2 n)) and the convergence test on the
projected gradient.  Device (``mopt_crsirfo_run_rowspace``): the null-space basis U (Householder QR of the row space; it never crosses
PCIe), Hessian update, U^T (H + bias) U on the FP64 matrix cores, the symmetric eigen-solve and the RS-I-RFO step in the subspace,
step = U step_sub."""
import ctypes as C

import numpy as np

from .rsirfo import RSIRFO
from .._lib import lib, ctx, check, f64, ptr


class CRSIRFO(RSIRFO):
    def __init__(self, constraints=None, **config):
        super().__init__(**config)
        self.constraints_obj = constraints
        self.null_space_basis = None
        self.svd_threshold = config.get("svd_threshold", 1e-5)
        self.proj_grad_converged = False

    def _row_space(self, geom):
        """Orthonormal rows spanning the row space of the normalised constraint vectors, (r, n) -- r = 0: no active constraint.  The
        reference takes the null space from the full SVD of the 3N x c matrix (crsirfo.py:26-46: singular values above
        max(svd_threshold, 1e-6 S_max) count); the same rank follows from the c x c Gram matrix (its eigenvalues are the squared singular
        values), and the null-space basis itself is formed on the device from these rows (mopt_crsirfo_run_rowspace)."""
        n = len(geom) * 3
        if self.constraints_obj is None:
            return np.zeros((0, n))
        B = self.constraints_obj._get_all_constraint_vectors(np.asarray(geom).reshape(-1, 3))
        if B is None or len(B) == 0:
            return np.zeros((0, n))
        lengths = np.linalg.norm(B, axis=1)
        Bn = B / np.where(lengths < 1e-12, 1.0, lengths)[:, None]
        lam, W = np.linalg.eigh(Bn @ Bn.T)
        sing = np.sqrt(np.clip(lam, 0.0, None))
        keep = sing > max(self.svd_threshold, (sing.max() if sing.size else 1.0) * 1e-6)
        if not np.any(keep):
            return np.zeros((0, n))
        return np.ascontiguousarray((W[:, keep].T @ Bn) / sing[keep][:, None])

    def run(self, geom_num_list, B_g, pre_B_g=[], pre_geom=[], B_e=0.0, pre_B_e=0.0, pre_move_vector=[], initial_geom_num_list=[], g=[], pre_g=[]):
        if self.Initialization:
            self.prev_eigvec_min = None
            self.prev_eigvec_size = None
            self.predicted_energy_changes = []
            self.actual_energy_changes = []
            self.prev_geometry = None
            self.prev_gradient = None
            self.prev_energy = None
            self.proj_grad_converged = False
            self.iteration = 0
            self.Initialization = False
        if self.hessian is None:
            raise ValueError("Hessian matrix must be set before running optimization")
        gradient_full = np.asarray(B_g, dtype=np.float64).ravel()
        original_shape = geom_num_list.shape
        geom_flat = geom_num_list.ravel()                               # (view, as in the reference)
        if self.constraints_obj is not None:
            corrected = self.constraints_obj.adjust_init_coord(geom_num_list.reshape(-1, 3)).ravel()
            disp = corrected - geom_flat
            if np.linalg.norm(disp) > 1e-6:
                H_eff = self.hessian
                if self.bias_hessian is not None:
                    H_eff += self.bias_hessian
                gradient_full = gradient_full + np.dot(H_eff, disp)
            geom_num_list = corrected.reshape(original_shape)
        x = f64(np.asarray(geom_num_list)).reshape(1, -1)
        n = x.shape[1]
        Qc = self._row_space(np.asarray(geom_num_list).reshape(-1, 3))
        if Qc.shape[0] >= n:
            return np.zeros_like(gradient_full).reshape(-1, 1)
        do_update = (self.prev_geometry is not None and self.prev_gradient is not None and len(pre_g) > 0 and len(pre_geom) > 0)
        # |U^T g| = |g - Qc^T Qc g| for any orthonormal basis U of the complement
        gsub_norm = float(np.linalg.norm(gradient_full - Qc.T @ (Qc @ gradient_full)))
        if gsub_norm < self.gradient_norm_threshold and not do_update:
            # crsirfo.py:116-125 (with a pending Hessian update the device call below performs it first, as the reference does, and the
            # test is repeated afterwards)
            self.proj_grad_converged = True
            self.prev_geometry, self.prev_gradient, self.prev_energy = geom_num_list, B_g, B_e
            return np.zeros_like(gradient_full).reshape(-1, 1)
        H = self.hessian
        if not isinstance(H, np.ndarray) or H.shape != (n, n):
            raise ValueError("Hessian must be a (3N,3N) ndarray")
        bg = f64(gradient_full).reshape(1, n)
        graw = f64(np.asarray(g)).reshape(1, n) if (g is not None and len(g) > 0) else bg
        px = pg = None
        if do_update:
            px = f64(np.asarray(pre_geom)).reshape(1, n)
            pg = f64(np.asarray(pre_g)).reshape(1, n)
        Bh = None if self.bias_hessian is None else f64(self.bias_hessian)
        be = np.array([float(B_e)])
        step = np.empty((1, n))
        Hin = np.empty((n, n)) if do_update else None
        Hsym = np.empty((n, n)) if do_update else None
        p = self._params()
        s = self._state()
        Qd = f64(Qc) if Qc.shape[0] > 0 else None
        c = ctx(self.device)
        check(lib().mopt_crsirfo_run_rowspace(c, n, Qc.shape[0], C.byref(p), C.byref(s), ptr(f64(H)), ptr(Bh), ptr(x), ptr(bg), ptr(graw), ptr(px), ptr(pg),
                                              ptr(be), ptr(Qd), ptr(step), ptr(Hin), ptr(Hsym)), c)
        converged_now = gsub_norm < self.gradient_norm_threshold
        if do_update and s.hess_updated:
            H[...] = Hin if H.dtype == np.float64 else Hin.astype(H.dtype)
            self.hessian = Hsym
        if converged_now:
            self.proj_grad_converged = True
            self.prev_geometry, self.prev_gradient, self.prev_energy = geom_num_list, B_g, B_e
            return np.zeros_like(gradient_full).reshape(-1, 1)
        self._unpack(s)
        self.prev_geometry = geom_num_list
        self.prev_gradient = B_g
        self.prev_energy = B_e
        return step.reshape(-1, 1)
