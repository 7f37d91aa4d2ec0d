"""ApproxHessian: drop-in for ``multioptpy.ModelHessian.approx_hessian.ApproxHessian`` (reference approx_hessian.py:26-112).

``main(coord, element_list, cart_gradient, approx_hess_type)`` keeps the reference's signature (coord in bohr) and its dispatch by
substring.  On the device (``mopt_model_hessian`` of include/mopt_hip.h): the Fischer + D3 model the reference's ``-modelhess``
switch selects by default (``fischerd3old``, interface.py:325-330; ModelHessian/fischerd3old.py:355-381), the Fischer model with
coordination-number scaled D3 pairs (``fischerd3``, ModelHessian/fischerd3.py:177-304), the plain Fischer model (``fischer``,
ModelHessian/fischer.py:216-236), Schlegel's model (``schlegel``, ModelHessian/schlegel.py:66-144), the Morse model (``morse``,
ModelHessian/morse.py:50-106) and Lindh's 2007 model with D3 dispersion (``lindh2007d3``, ModelHessian/lindh2007d3.py:777-822) -- which is
also what a name the dispatcher does not know falls back to, as in the reference (approx_hessian.py:88-91) -- and the ``clip`` modifier
(approx_hessian.py:104-120).  The Swart, Lindh-1995 / -D2 / -D4, Schlegel-D3/D4, Fischer-D4 and GFN models raise NotImplementedError; there
is no CPU fallback.  Modifiers, in the reference's order (approx_hessian.py:95-110): ``ts`` (tshess.py: eigen-solve on the device, rank-two
correction on the host array), ``sr`` (shortrange.py: ``mopt_model_hessian_sr``, round 4), ``clip`` (inside the model call when it is the only
modifier, ``mopt_hessian_clip`` behind the others).

Per-element parameters (covalent radii, C6, van-der-Waals radii, r4/r2, D3 scaling) are the numbers the reference's
``multioptpy/Parameters`` functions return, kept as data in ``element_tables.json`` (tools/extract_element_tables.py)."""
import ctypes as C
import json
import os

import numpy as np

from .. import _lib
from .._lib import lib, check, f64, ptr

MODEL_FISCHERD3OLD, MODEL_FISCHERD3, MODEL_FISCHER, MODEL_SCHLEGEL, MODEL_MORSE, MODEL_LINDH2007D3 = 0, 1, 2, 3, 4, 5
FLAG_CLIP = 1
FLAG_TS = 2          # (host-side flags of this module: the `ts` and `sr` modifiers; never passed to the C ABI)
FLAG_SR = 4
# Pauling electronegativities as ShortRangeCorrectionHessian.estimate_atomic_charges tabulates them (shortrange.py:160-172; anything else: 2.0)
_PAULING = {"H": 2.20, "He": 0.00, "Li": 0.98, "Be": 1.57, "B": 2.04, "C": 2.55, "N": 3.04, "O": 3.44, "F": 3.98, "Ne": 0.00, "Na": 0.93, "Mg": 1.31,
            "Al": 1.61, "Si": 1.90, "P": 2.19, "S": 2.58, "Cl": 3.16, "Ar": 0.00, "K": 0.82, "Ca": 1.00, "Sc": 1.36, "Ti": 1.54, "V": 1.63, "Cr": 1.66,
            "Mn": 1.55, "Fe": 1.83, "Co": 1.88, "Ni": 1.91, "Cu": 1.90, "Zn": 1.65, "Ga": 1.81, "Ge": 2.01, "As": 2.18, "Se": 2.55, "Br": 2.96, "Kr": 0.00}
SR_PARAMS = (0.2, 0.78, 0.5, 15.0)      # omega (1/bohr), cx_sr, scaling factor, cut-off (bohr): shortrange.py:19-34
_TAB = None


def _tables():
    global _TAB
    if _TAB is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "element_tables.json")) as fh:
            _TAB = json.load(fh)
    return _TAB


def _schlegel_code(e, order):
    """Row of Schlegel's B table for an element as the reference assigns it (schlegel.py:27-63: H-He, Li-Ne, K-Kr, Rb-Xe, Cs-Rn,
    everything else -- Na-Ar included -- in the last row), + 8 for hydrogen (its angle constant, schlegel.py:101-106)."""
    z = order.index(e) + 1 if e in order else 999
    row = 0 if z <= 2 else 1 if z <= 10 else 2 if 19 <= z <= 36 else 3 if 37 <= z <= 54 else 4 if 55 <= z <= 86 else 5
    return row + (8 if e == "H" else 0)


def _lindh_group(e, order):
    """select_idx of the Lindh models (lindh2007d3.py:60-82): H -> 0, atomic numbers 2 .. 9 -> 1, everything else -> 2"""
    z = order.index(e) + 1
    return 0 if z < 2 else (1 if z < 10 else 2)


def element_parameters(element_list, kind=MODEL_FISCHERD3OLD):
    """(par, d3): par = [cov | c6 | vdw | r4r2 | refcn | model-specific] as one (6 N,) array, d3 = (s6, s8, a1, a2) -- the arguments
    of the C ABI; the last array is Schlegel's element code, the GNB radius for the Morse model or the element group of the Lindh model
    (whose van-der-Waals slot carries the UFF distance)."""
    t = _tables()
    el = t["elements"]

    def last(e):
        if kind == MODEL_MORSE:
            return el[e]["gnb"]
        if kind == MODEL_LINDH2007D3:
            return _lindh_group(e, t["order"])
        return _schlegel_code(e, t["order"])
    keys = ("cov", "c6", "uffvdw" if kind == MODEL_LINDH2007D3 else "vdw", "r4r2", "refcn")
    try:
        cols = [[el[str(e)][k] for e in element_list] for k in keys] + [[last(str(e)) for e in element_list]]
        if any(v is None for c in cols for v in c):
            raise KeyError(next(str(e) for e in element_list if any(el[str(e)][k] is None for k in keys)))
        par = np.concatenate([np.array(c, dtype=np.float64) for c in cols])
    except KeyError as e:
        raise KeyError(f"no model-Hessian parameters for element {e}") from None
    d3 = np.array([t["d3"][k] for k in ("s6", "s8", "a1", "a2")], dtype=np.float64)
    return np.ascontiguousarray(par), d3


def model_kind(approx_hess_type, with_flags=False):
    """The reference's substring dispatch (approx_hessian.py:35-110), restricted to what runs on the device.  with_flags: (kind, flags)."""
    name = str(approx_hess_type).lower()
    if "gfnff" in name or "gfn0xtb" in name:
        raise NotImplementedError(f"model Hessian '{approx_hess_type}' is not available on the device")
    # the reference tests the substrings in this order (approx_hessian.py:41-91): fischerd3old, fischerd3, fischerd4, schlegel*, swart*,
    # lindh2007d3, lindh2007d4, lindh2007, lindh, fischer, morse; anything else: Lindh (2007) D3
    kind, base = None, ""
    if "fischerd3old" in name:
        kind, base = MODEL_FISCHERD3OLD, "fischerd3old"
    elif "fischerd3" in name:
        kind, base = MODEL_FISCHERD3, "fischerd3"
    elif "fischerd4" in name or "schlegeld3" in name or "schlegeld4" in name:
        pass
    elif "schlegel" in name:
        kind, base = MODEL_SCHLEGEL, "schlegel"
    elif "swart" in name:
        pass
    elif "lindh2007d3" in name:
        kind, base = MODEL_LINDH2007D3, "lindh2007d3"
    elif "lindh" in name:
        pass
    elif "fischer" in name:
        kind, base = MODEL_FISCHER, "fischer"
    elif "morse" in name:
        kind, base = MODEL_MORSE, "morse"
    else:
        kind = MODEL_LINDH2007D3                              # "Approximate Hessian type not recognized. Using default Lindh (2007) D3 model..."
    if kind is None:
        raise NotImplementedError(f"model Hessian '{approx_hess_type}' is not available on the device (available: fischerd3old, fischerd3, "
                                  "fischer, schlegel, morse, lindh2007d3)")
    flags = 0
    # approx_hessian.py:93-110: modifiers applied after the model, in the order ts, sr, clip (the tests see the whole name)
    if "ts" in name:
        flags |= FLAG_TS
    if "sr" in name:
        flags |= FLAG_SR
    if "clip" in name:
        flags |= FLAG_CLIP
    return (kind, flags) if with_flags else kind


def sr_charges(element_list):
    """estimate_atomic_charges (shortrange.py:147-186): 0.2 (mean electronegativity - electronegativity), the mean as sum / len of the list"""
    en = [_PAULING.get(str(e), 2.0) for e in element_list]
    avg = sum(en) / len(en)
    q = np.zeros(len(en))
    for i, v in enumerate(en):
        q[i] = 0.2 * (avg - v)
    return q


def sr_correction(H, coords, element_list, device=None):
    """ShortRangeCorrectionHessian.main on the device for one (3N, 3N) or a batch (I, 3N, 3N) of Hessians; returns new arrays."""
    Hc = np.array(H, dtype=np.float64, order="C")
    X = f64(coords).reshape(-1, len(element_list), 3)
    B, N = X.shape[0], X.shape[1]
    cov = np.array([_tables()["elements"][str(e)]["cov"] for e in element_list], dtype=np.float64)
    chg = sr_charges(element_list)
    prm = np.array(SR_PARAMS, dtype=np.float64)
    c = _lib.ctx(device)
    check(lib().mopt_model_hessian_sr(c, N, B, ptr(X), ptr(cov), ptr(chg), ptr(prm), ptr(Hc)), c)
    return Hc


def clip_modification(H, device=None):
    """The `clip` modifier behind another modifier (approx_hessian.py:104-110): V smooth(lambda) V^T on the device."""
    Hc = np.array(H, dtype=np.float64, order="C")
    n = Hc.shape[-1]
    c = _lib.ctx(device)
    check(lib().mopt_hessian_clip(c, n, int(Hc.size // (n * n)), ptr(Hc)), c)
    return Hc


def ts_modification(H):
    """TransitionStateHessian.create_ts_hessian (ModelHessian/tshess.py:14-40): unless the model Hessian already has an eigenvalue below -1e-8,
    its lowest mode behind the leading |lambda| < 1e-8 ones is reflected: sym((1 - 2 v v^T) H) = H - v (H v)^T - (H v) v^T.  The eigen-solve
    runs on the device (ops.eigh); the rank-two correction is O(n^2) host work on the array that is being handed to the caller anyway."""
    from .. import ops
    w, v = ops.eigh(np.ascontiguousarray(H, dtype=np.float64))
    if np.any(w < -1e-8):
        return H
    count = 0
    for lam in w:
        if abs(lam) < 1e-8:
            count += 1
        else:
            break
    t = np.array(v[:, count])
    Ht = H @ t
    return H - np.outer(t, H.T @ t) - np.outer(Ht, t)


class ApproxHessian:
    def __init__(self, device=None):
        self.device = device

    def main(self, coord, element_list, cart_gradient=None, approx_hess_type="fischerd3old"):
        kind, flags = model_kind(approx_hess_type, with_flags=True)
        X = f64(coord).reshape(-1, 3)
        N = X.shape[0]
        par, d3 = element_parameters(element_list, kind)
        G = None if cart_gradient is None else f64(cart_gradient).reshape(-1, 3)
        H = np.empty((3 * N, 3 * N))
        c = _lib.ctx(self.device)
        later = flags & (FLAG_TS | FLAG_SR)                   # another modifier stands between the model and the clipping
        check(lib().mopt_model_hessian_ex(c, kind, 0 if later else (flags & FLAG_CLIP), N, 1, ptr(X), ptr(par), ptr(d3), ptr(G), 1, ptr(H)), c)
        if flags & FLAG_TS:
            H = ts_modification(H)
        if flags & FLAG_SR:
            H = sr_correction(H, X, element_list, self.device)
        if later and (flags & FLAG_CLIP):
            H = clip_modification(H, self.device)
        return H

    def batch(self, coords, element_list, approx_hess_type="fischerd3old", project=True, cart_gradients=None):
        """All images of a band in one call: coords (I, N, 3) bohr -> (I, 3N, 3N)."""
        kind, flags = model_kind(approx_hess_type, with_flags=True)
        X = f64(coords)
        I, N = X.shape[0], X.shape[1]
        par, d3 = element_parameters(element_list, kind)
        G = None if cart_gradients is None else f64(cart_gradients).reshape(I, N, 3)
        H = np.empty((I, 3 * N, 3 * N))
        c = _lib.ctx(self.device)
        later = flags & (FLAG_TS | FLAG_SR)
        check(lib().mopt_model_hessian_ex(c, kind, 0 if later else (flags & FLAG_CLIP), N, I, ptr(X), ptr(par), ptr(d3), ptr(G), 1 if project else 0, ptr(H)), c)
        if flags & FLAG_TS:
            for i in range(I):
                H[i] = ts_modification(H[i])
        if flags & FLAG_SR:
            H = sr_correction(H, X, element_list, self.device)
        if later and (flags & FLAG_CLIP):
            H = clip_modification(H, self.device)
        return H
