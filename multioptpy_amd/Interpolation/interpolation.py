"""Drop-in for multioptpy/Interpolation/interpolation.py::spline_interpolation as used by apply_climbing_image (calc_tools.py:869):
extrema of the natural cubic spline through (path length, energy)."""
import numpy as np


def compute_natural_spline_coefficients(x, y):
    """interpolation.py:4-45: S_i(t) = a_i + b_i (t - x_i) + c_i (t - x_i)^2 + d_i (t - x_i)^3 with S'' = 0 at both ends.  The
    reference solves the dense (n x n) system with numpy.linalg.solve; the same call is made here so that c agrees to the last bit."""
    x = np.asarray(x, dtype="float64")
    a = np.array(y, dtype="float64")
    n = len(x)
    h = np.diff(x)
    M = np.zeros((n, n))
    rhs = np.zeros(n)
    M[0, 0] = M[-1, -1] = 1.0
    k = np.arange(1, n - 1)
    M[k, k - 1] = h[k - 1]
    M[k, k] = 2 * (h[k - 1] + h[k])
    M[k, k + 1] = h[k]
    rhs[k] = 3 * ((a[k + 1] - a[k]) / h[k] - (a[k] - a[k - 1]) / h[k - 1])
    c = np.linalg.solve(M, rhs)
    b = (a[1:] - a[:-1]) / h - h * (2 * c[:-1] + c[1:]) / 3
    d = (c[1:] - c[:-1]) / (3 * h)
    return a, b, c, d, h


def find_extrema(x, a, b, c, d):
    """interpolation.py:106-146: per segment, Newton on S_i' from the midpoint (tolerance 1e-10 on the update, 500 iterations, stop when
    |S_i''| < 1e-12); a root inside [x_i - 1e-10, x_i+1 + 1e-10] is a maximum when S_i'' < 0 and a minimum when S_i'' > 0."""
    maxima, minima = [], []
    for i in range(len(x) - 1):
        t = (x[i] + x[i + 1]) / 2
        for _ in range(500):
            f = b[i] + 2 * c[i] * (t - x[i]) + 3 * d[i] * (t - x[i]) ** 2
            df = 2 * c[i] + 6 * d[i] * (t - x[i])
            if abs(df) < 1e-12:
                break
            t_new = t - f / df
            if abs(t_new - t) < 1e-10:
                t = t_new
                break
            t = t_new
        if x[i] - 1e-10 <= t <= x[i + 1] + 1e-10:
            val = a[i] + b[i] * (t - x[i]) + c[i] * (t - x[i]) ** 2 + d[i] * (t - x[i]) ** 3
            curv = 2 * c[i] + 6 * d[i] * (t - x[i])
            if curv < 0:
                maxima.append((t, val))
            elif curv > 0:
                minima.append((t, val))
    return maxima, minima


def spline_interpolation(x_data, y_data):
    """interpolation.py:148-155 without the 100 000-point evaluation whose result the reference discards."""
    a, b, c, d, _ = compute_natural_spline_coefficients(x_data, y_data)
    return find_extrema(np.asarray(x_data, dtype="float64"), a, b, c, d)
