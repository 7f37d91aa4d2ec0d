"""Drop-in for multioptpy/Potential/idpp.py::IDPP (image dependent pair potential, arXiv:1406.1512) as the climbing-image move of
nebmain uses it (calc_tools.py:880-882: three images, the middle one relaxes).  Host glue on (N, 3) arrays."""
import numpy as np


def _dist(pos):
    return np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(axis=-1))


class IDPP:
    def __init__(self):
        self.iteration = 2000
        self.lr = 0.01
        self.threshold = 1e-4

    def calc_dist_matrix(self, pos):
        return _dist(np.asarray(pos, dtype="float64"))

    def calc_idpp_dist_matrix(self, pos_list, n_node, number_of_node):
        """idpp.py:51-60: pair distances interpolated linearly between the first and the last image."""
        d0, d1 = _dist(np.asarray(pos_list[0], dtype="float64")), _dist(np.asarray(pos_list[-1], dtype="float64"))
        return d0 + number_of_node * (d1 - d0) / (n_node - 1)

    def get_func_and_deriv(self, pos_list, n_node, number_of_node):
        """idpp.py:18-49, 67-73: S = sum_{i<j} (r_ij + 1e-15)^-4 (d_ij - r_ij)^2 and dS/dx (the gradient sums the upper triangle:
        row i gets +t_ij, row j gets -t_ij with t_ij = (w' (d - r)^2 - 2 w (d - r)) (x_i - x_j) / (r + 1e-15))."""
        pos = np.asarray(pos_list[number_of_node], dtype="float64")
        target = self.calc_idpp_dist_matrix(pos_list, n_node, number_of_node)
        dist = _dist(pos)
        iu = np.triu_indices(len(pos), k=1)
        obj = np.sum((dist[iu] + 1e-15) ** (-4) * (target[iu] - dist[iu]) ** 2.0)
        diff = pos[:, None, :] - pos[None, :, :]
        r = np.linalg.norm(diff, axis=-1)
        unit = np.zeros_like(diff)
        nz = r > 0
        unit[nz] = diff[nz] / (r[nz][:, None] + 1e-15)
        dm = target - dist
        t = (-4 * (r + 1e-15) ** (-5) * dm ** 2 - 2.0 * (r + 1e-15) ** (-4) * dm)[:, :, None] * unit
        # accumulated pair by pair in upper-triangle order, all +t_ij first and all -t_ij afterwards: the summation order of the reference
        grad = np.zeros_like(pos)
        np.add.at(grad, iu[0], t[iu])
        np.subtract.at(grad, iu[1], t[iu])
        return obj, grad

    def opt_path(self, geometry_list, element_list, memory_size=30):
        """idpp.py:76-183: per interior image an L-BFGS direction (two-loop recursion over the pairs of that image, initial scaling
        s.y / y.y of the newest pair, 0 when y.y <= 1e-10), applied with the sign conventions of the reference -- it negates the
        gradient, takes -H(-g) as direction and then SUBTRACTS min(lr, |dir|) dir/|dir| -- a pair is stored when s.y > 1e-10 with
        y = g_new - (-g_old); stops when the largest objective is below the threshold."""
        geometry_list = [np.array(g, dtype="float64") for g in geometry_list] if not isinstance(geometry_list, list) else geometry_list
        n = len(geometry_list)
        mem = [dict(s=[], y=[], rho=[]) for _ in range(n)]

        def direction(g, m):
            if not m["s"]:
                return -g
            q = g.copy()
            alphas = [0.0] * len(m["s"])
            for k in range(len(m["s"]) - 1, -1, -1):
                alphas[k] = m["rho"][k] * np.sum(m["s"][k] * q)
                q = q - alphas[k] * m["y"][k]
            yy = np.sum(m["y"][-1] * m["y"][-1])
            gamma = np.sum(m["s"][-1] * m["y"][-1]) / yy if np.abs(yy) > 1e-10 else 0
            r = gamma * q
            for k in range(len(m["s"])):
                beta = m["rho"][k] * np.sum(m["y"][k] * r)
                r = r + m["s"][k] * (alphas[k] - beta)
            return -r

        for _ in range(self.iteration):
            objs = []
            for j in range(1, n - 1):
                before = geometry_list[j].copy()
                obj, g = self.get_func_and_deriv(geometry_list, n, j)
                objs.append(obj)
                g = -g
                dvec = direction(g, mem[j])
                dn = np.linalg.norm(dvec)
                if dn > 1e-10:
                    geometry_list[j] -= min(self.lr, dn) * (dvec / dn)
                _, g_new = self.get_func_and_deriv(geometry_list, n, j)
                s = geometry_list[j] - before
                y = g_new - g
                sy = np.sum(s * y)
                if sy > 1e-10:
                    if len(mem[j]["s"]) >= memory_size:
                        for key in ("s", "y", "rho"):
                            mem[j][key].pop(0)
                    mem[j]["s"].append(s)
                    mem[j]["y"].append(y)
                    mem[j]["rho"].append(1.0 / sy)
            if max(objs) < self.threshold:
                break
        return geometry_list
