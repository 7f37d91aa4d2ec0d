"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain loops) of the path tools nebmain calls between the optimizer and the next input
files.  Never imported by the product (multioptpy_amd/); pinned by tests/golden/g13_path_tools.npz and the `a` / `c` runs of
g8_nebmain.npz, all recorded from the reference.

  path_lengths          multioptpy/Utils/calc_tools.py:853-862     calc_path_length_list
  redistribute          multioptpy/Interpolation/linear_interpolation.py:308-336   distribute_geometry  (`-ad`, neb.py:649-766)
  spline_extrema        multioptpy/Interpolation/interpolation.py:4-45, 92-155     natural cubic spline, Newton per segment
  idpp_value_gradient   multioptpy/Potential/idpp.py:18-73
  idpp_relax            multioptpy/Potential/idpp.py:76-183       per-image L-BFGS with a capped step length
  climbing_image_move   multioptpy/Utils/calc_tools.py:865-883     apply_climbing_image  (`-ci`, neb.py:561-564)
"""
import numpy as np


def path_lengths(geoms):
    L = [0.0]
    for k in range(len(geoms) - 1):
        a = geoms[k] - np.mean(geoms[k], axis=0)
        b = geoms[k + 1] - np.mean(geoms[k + 1], axis=0)
        L.append(L[-1] + np.linalg.norm(b - a))
    return L


def redistribute(geoms):
    n = len(geoms)
    L = path_lengths(geoms)
    if L[-1] < 1e-8:
        return list(geoms)
    spacing = L[-1] / (n - 1)
    out = [geoms[0]]
    for i in range(1, n - 1):
        d = i * spacing
        placed = None
        for j in range(n - 1):
            if L[j] <= d <= L[j + 1]:                       # first bracketing segment wins
                placed = geoms[j] + (geoms[j + 1] - geoms[j]) * ((d - L[j]) / (L[j + 1] - L[j]))
                break
        out.append(geoms[-1] if placed is None else placed)
    out.append(geoms[-1])
    return out


def spline_extrema(x, y):
    n = len(x)
    a = np.array(y, dtype=float)
    h = np.diff(x)
    M = np.zeros((n, n))
    r = np.zeros(n)
    M[0, 0] = 1.0
    M[n - 1, n - 1] = 1.0
    for i in range(1, n - 1):
        M[i, i - 1], M[i, i], M[i, i + 1] = h[i - 1], 2 * (h[i - 1] + h[i]), h[i]
        r[i] = 3 * ((a[i + 1] - a[i]) / h[i] - (a[i] - a[i - 1]) / h[i - 1])
    c = np.linalg.solve(M, r)
    maxima, minima = [], []
    for i in range(n - 1):
        b_i = (a[i + 1] - a[i]) / h[i] - h[i] * (2 * c[i] + c[i + 1]) / 3
        d_i = (c[i + 1] - c[i]) / (3 * h[i])
        t = (x[i] + x[i + 1]) / 2
        for _ in range(500):
            f = b_i + 2 * c[i] * (t - x[i]) + 3 * d_i * (t - x[i]) ** 2
            df = 2 * c[i] + 6 * d_i * (t - x[i])
            if abs(df) < 1e-12:
                break
            tn = t - f / df
            done = abs(tn - t) < 1e-10
            t = tn
            if done:
                break
        if t >= x[i] - 1e-10 and t <= x[i + 1] + 1e-10:
            s = a[i] + b_i * (t - x[i]) + c[i] * (t - x[i]) ** 2 + d_i * (t - x[i]) ** 3
            s2 = 2 * c[i] + 6 * d_i * (t - x[i])
            if s2 < 0:
                maxima.append((t, s))
            elif s2 > 0:
                minima.append((t, s))
    return maxima, minima


def idpp_value_gradient(images, k):
    """Objective and gradient of image k of `images` (targets: pair distances interpolated between the first and the last image)."""
    n_img = len(images)
    pos = images[k]
    N = len(pos)

    def dmat(p):
        return np.sqrt(np.sum((p[:, None, :] - p[None, :, :]) ** 2, axis=-1))
    d0, d1 = dmat(images[0]), dmat(images[-1])
    target = d0 + k * (d1 - d0) / (n_img - 1)
    dist = dmat(pos)
    val = 0.0
    grad = np.zeros_like(pos)
    terms = {}
    for i in range(N):
        for j in range(i + 1, N):
            val += 0.0                                      # (the value is summed by numpy over the triangle, see below)
            vec = pos[i] - pos[j]
            rr = np.linalg.norm(vec)
            u = vec / (rr + 1e-15) if rr > 0 else np.zeros(3)
            dm = target[i, j] - dist[i, j]
            terms[(i, j)] = (-4 * (rr + 1e-15) ** (-5) * dm ** 2 - 2.0 * (rr + 1e-15) ** (-4) * dm) * u
    iu = np.triu_indices(N, k=1)
    val = np.sum((dist[iu] + 1e-15) ** (-4) * (target[iu] - dist[iu]) ** 2.0)
    for (i, j), t in terms.items():
        grad[i] += t
    for (i, j), t in terms.items():
        grad[j] -= t
    return val, grad


def idpp_relax(images, iterations=2000, lr=0.01, threshold=1e-4, memory=30):
    images = [np.array(g, dtype=float) for g in images]
    n = len(images)
    S = [[] for _ in range(n)]
    Y = [[] for _ in range(n)]
    RHO = [[] for _ in range(n)]
    for _ in range(iterations):
        worst = None
        for j in range(1, n - 1):
            x_old = images[j].copy()
            val, g = idpp_value_gradient(images, j)
            worst = val if worst is None else max(worst, val)
            g = -g
            if not S[j]:
                direction = -g
            else:
                q = g.copy()
                al = []
                for i in range(len(S[j]) - 1, -1, -1):
                    a_i = RHO[j][i] * np.sum(S[j][i] * q)
                    al.insert(0, a_i)
                    q = q - a_i * Y[j][i]
                yy = np.sum(Y[j][-1] * Y[j][-1])
                gamma = np.sum(S[j][-1] * Y[j][-1]) / yy if abs(yy) > 1e-10 else 0
                r = gamma * q
                for i in range(len(S[j])):
                    r = r + S[j][i] * (al[i] - RHO[j][i] * np.sum(Y[j][i] * r))
                direction = -r
            nd = np.linalg.norm(direction)
            if nd > 1e-10:
                images[j] = images[j] - min(lr, nd) * (direction / nd)
            _, g2 = idpp_value_gradient(images, j)
            s, y = images[j] - x_old, g2 - g
            sy = np.sum(s * y)
            if sy > 1e-10:
                if len(S[j]) >= memory:
                    S[j].pop(0), Y[j].pop(0), RHO[j].pop(0)
                S[j].append(s), Y[j].append(y), RHO[j].append(1.0 / sy)
        if worst < threshold:
            break
    return images


def climbing_image_move(geoms, energies):
    geoms = [np.array(g, dtype=float) for g in geoms]
    L = path_lengths(geoms)
    maxima, _ = spline_extrema(L, energies)
    for dist, _e in maxima:
        for i in range(2, len(L) - 2):
            if L[i] >= dist or dist >= L[i + 1]:
                continue
            t = (dist - L[i]) / (L[i + 1] - L[i])
            mid = geoms[i] + (geoms[i + 1] - geoms[i]) * t
            geoms[i] = idpp_relax([geoms[i], mid, geoms[i + 1]])[1]
    return geoms
