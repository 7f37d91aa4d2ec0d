"""Drop-in for multioptpy/Optimizer/trust_radius_neb.py::TR_NEB (mopt_neb_tr_calc).

The reference also appends the neighbour distances of every call to procrustes_distance_{1,2}.csv in the NEB folder
(trust_radius_neb.py:79-83); that logging is kept only when NEB_FOLDER_DIRECTORY is given."""
import numpy as np

from .._lib import lib, ctx, check, f64, ptr


class TR_NEB:
    def __init__(self, **config):
        self.NEB_FOLDER_DIRECTORY = config.get("NEB_FOLDER_DIRECTORY", None)
        self.fix_init_edge = config.get("fix_init_edge", False)
        self.fix_end_edge = config.get("fix_end_edge", False)
        self.apply_convergence_criteria = config.get("apply_convergence_criteria", False)
        self.threshold_max_force = 0.00045
        self.threshold_rms_force = 0.00030
        self.threshold_max_displacement = 0.0018
        self.threshold_rms_displacement = 0.0012

    def TR_calc(self, geometry_num_list, total_force_list, total_delta, biased_energy_list, pre_biased_energy_list, pre_geom):
        X = f64(geometry_num_list)
        F = f64(total_force_list)
        D = f64(np.asarray(total_delta, dtype=np.float64))
        I, N = X.shape[0], X.shape[1]
        mv = np.empty_like(X)
        c = ctx()
        check(lib().mopt_neb_tr_calc(c, I, N, ptr(X), ptr(F), ptr(D), int(bool(self.fix_init_edge)), int(bool(self.fix_end_edge)),
                                     ptr(mv)), c)
        if self.NEB_FOLDER_DIRECTORY is not None:
            d1 = [str(np.linalg.norm(X[i] - X[i - 1])) for i in range(1, I - 1)]
            d2 = [str(np.linalg.norm(X[i] - X[i + 1])) for i in range(1, I - 1)]
            with open(self.NEB_FOLDER_DIRECTORY + "procrustes_distance_1.csv", "a") as f:
                f.write(",".join(d1) + "\n")
            with open(self.NEB_FOLDER_DIRECTORY + "procrustes_distance_2.csv", "a") as f:
                f.write(",".join(d2) + "\n")
        move_vector = [mv[i] for i in range(I)]
        if self.apply_convergence_criteria:
            move_vector = self.check_convergence(F, move_vector)
        return move_vector

    def check_convergence(self, total_force_list, move_vec_list):
        """trust_radius_neb.py:100-124 (signed maxima, as in the reference)."""
        for i in range(1, len(total_force_list) - 1):
            max_grad = np.max(total_force_list[i])
            rms_grad = np.sqrt(np.sum(total_force_list[i] ** 2) / (len(total_force_list[i]) * 3))
            max_move = np.max(move_vec_list[i])
            rms_move = np.sqrt(np.sum(move_vec_list[i] ** 2) / (len(move_vec_list[i]) * 3))
            if (max_grad < self.threshold_max_force and rms_grad < self.threshold_rms_force
                    and max_move < self.threshold_max_displacement and rms_move < self.threshold_rms_displacement):
                move_vec_list[i] = move_vec_list[i] * 0.0
        return move_vec_list
