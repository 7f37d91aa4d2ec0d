"""Oracle: Brent's root finder as SciPy implements it (test infrastructure).

The reference calls scipy.optimize.brentq (scipy ~= 1.13, pyproject.toml:26) at
multioptpy/Optimizer/rsirfo.py:1042 with xtol = rtol = 1e-6, maxiter = 50.
SciPy is a third-party dependency that is not in /root/reference; this restates
its published algorithm (scipy/optimize/Zeros/brentq.c) and is pinned by the
G4 traces in tests/golden (every (x, f(x)) the real brentq evaluated).
"""
import math


class BrentError(Exception):
    pass


def brentq(f, xa, xb, xtol=2e-12, rtol=8.881784197001252e-16, maxiter=100, trace=None):
    xpre, xcur = xa, xb
    xblk = 0.0
    fblk = 0.0
    spre = 0.0
    scur = 0.0
    fpre = f(xpre)
    fcur = f(xcur)
    if trace is not None:
        trace.append((xpre, fpre))
        trace.append((xcur, fcur))
    if fpre == 0:
        return xpre
    if fcur == 0:
        return xcur
    if math.copysign(1.0, fpre) == math.copysign(1.0, fcur):
        raise BrentError("f(a) and f(b) must have different signs")
    for _ in range(maxiter):
        if fpre != 0 and fcur != 0 and (math.copysign(1.0, fpre) != math.copysign(1.0, fcur)):
            xblk = xpre
            fblk = fpre
            spre = scur = xcur - xpre
        if abs(fblk) < abs(fcur):
            xpre = xcur
            xcur = xblk
            xblk = xpre
            fpre = fcur
            fcur = fblk
            fblk = fpre
        delta = (xtol + rtol * abs(xcur)) / 2
        sbis = (xblk - xcur) / 2
        if fcur == 0 or abs(sbis) < delta:
            return xcur
        if abs(spre) > delta and abs(fcur) < abs(fpre):
            if xpre == xblk:
                stry = -fcur * (xcur - xpre) / (fcur - fpre)
            else:
                dpre = (fpre - fcur) / (xpre - xcur)
                dblk = (fblk - fcur) / (xblk - xcur)
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre))
            if 2 * abs(stry) < min(abs(spre), 3 * abs(sbis) - delta):
                spre = scur
                scur = stry
            else:
                spre = sbis
                scur = sbis
        else:
            spre = sbis
            scur = sbis
        xpre = xcur
        fpre = fcur
        if abs(scur) > delta:
            xcur += scur
        else:
            xcur += delta if sbis > 0 else -delta
        fcur = f(xcur)
        if trace is not None:
            trace.append((xcur, fcur))
    raise BrentError("failed to converge after %d iterations" % maxiter)
