"""ApproxHessian: drop-in for ``multioptpy.ModelHessian.approx_hessian.ApproxHessian`` (reference approx_hessian.py:26-112).

``main(coord, element_list, cart_gradient, approx_hess_type)`` keeps the reference's signature (coord in bohr) and its dispatch by
substring.  On the device (``mopt_model_hessian`` of include/mopt_hip.h): the Fischer + D3 model the reference's ``-modelhess``
switch selects by default (``fischerd3old``, interface.py:325-330; ModelHessian/fischerd3old.py:355-381), the Fischer model with
coordination-number scaled D3 pairs (``fischerd3``, ModelHessian/fischerd3.py:177-304), the plain Fischer model (``fischer``,
ModelHessian/fischer.py:216-236) and Schlegel's model (``schlegel``, ModelHessian/schlegel.py:66-144).  Other model names raise NotImplementedError; there is no CPU fallback.

Per-element parameters (covalent radii, C6, van-der-Waals radii, r4/r2, D3 scaling) are the numbers the reference's
``multioptpy/Parameters`` functions return, kept as data in ``element_tables.json`` (tools/extract_element_tables.py)."""
import ctypes as C
import json
import os

import numpy as np

from .. import _lib
from .._lib import lib, check, f64, ptr

MODEL_FISCHERD3OLD, MODEL_FISCHERD3, MODEL_FISCHER, MODEL_SCHLEGEL, MODEL_MORSE = 0, 1, 2, 3, 4
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


def element_parameters(element_list, kind=MODEL_FISCHERD3OLD):
    """(par, d3): par = [cov | c6 | vdw | r4r2 | refcn | model-specific] as one (6 N,) array, d3 = (s6, s8, a1, a2) -- the arguments
    of the C ABI; the last array is Schlegel's element code, or the GNB radius for the Morse model."""
    t = _tables()
    el = t["elements"]
    try:
        par = np.concatenate([np.array([el[str(e)][k] for e in element_list], dtype=np.float64) for k in ("cov", "c6", "vdw", "r4r2", "refcn")]
                             + [np.array([(el[str(e)]["gnb"] if kind == MODEL_MORSE else _schlegel_code(str(e), t["order"])) for e in element_list],
                                         dtype=np.float64)])
    except KeyError as e:
        raise KeyError(f"no model-Hessian parameters for element {e}") from None
    d3 = np.array([t["d3"][k] for k in ("s6", "s8", "a1", "a2")], dtype=np.float64)
    return np.ascontiguousarray(par), d3


def model_kind(approx_hess_type):
    """The reference's substring dispatch (approx_hessian.py:35-97), restricted to what runs on the device."""
    name = str(approx_hess_type).lower()
    if "gfnff" in name or "gfn0xtb" in name:
        raise NotImplementedError(f"model Hessian '{approx_hess_type}' is not available on the device")
    # the reference tests the substrings in this order (approx_hessian.py:41-91): fischerd3old, fischerd3, fischerd4, schlegel*, swart*,
    # lindh*, fischer, morse
    kind, base = None, None
    if "fischerd3old" in name:
        kind, base = MODEL_FISCHERD3OLD, "fischerd3old"
    elif "fischerd3" in name:
        kind, base = MODEL_FISCHERD3, "fischerd3"
    elif "fischerd4" in name or "schlegeld3" in name or "schlegeld4" in name:
        pass
    elif "schlegel" in name:
        kind, base = MODEL_SCHLEGEL, "schlegel"
    elif any(k in name for k in ("swart", "lindh")):
        pass
    elif "fischer" in name:
        kind, base = MODEL_FISCHER, "fischer"
    elif "morse" in name:
        kind, base = MODEL_MORSE, "morse"
    if kind is None:
        raise NotImplementedError(f"model Hessian '{approx_hess_type}' is not available on the device (available: fischerd3old, fischerd3, fischer, schlegel, morse)")
    for suffix in ("ts", "sr", "clip"):                  # approx_hessian.py:93-110: modifiers applied after the model
        if suffix in name.replace(base, ""):
            raise NotImplementedError(f"model Hessian modifier '{suffix}' is not available on the device")
    return kind


class ApproxHessian:
    def __init__(self, device=None):
        self.device = device

    def main(self, coord, element_list, cart_gradient=None, approx_hess_type="fischerd3old"):
        kind = model_kind(approx_hess_type)
        X = f64(coord).reshape(-1, 3)
        N = X.shape[0]
        par, d3 = element_parameters(element_list, kind)
        H = np.empty((3 * N, 3 * N))
        c = _lib.ctx(self.device)
        check(lib().mopt_model_hessian(c, kind, N, 1, ptr(X), ptr(par), ptr(d3), 1, ptr(H)), c)
        return H

    def batch(self, coords, element_list, approx_hess_type="fischerd3old", project=True):
        """All images of a band in one call: coords (I, N, 3) bohr -> (I, 3N, 3N)."""
        kind = model_kind(approx_hess_type)
        X = f64(coords)
        I, N = X.shape[0], X.shape[1]
        par, d3 = element_parameters(element_list, kind)
        H = np.empty((I, 3 * N, 3 * N))
        c = _lib.ctx(self.device)
        check(lib().mopt_model_hessian(c, kind, N, I, ptr(X), ptr(par), ptr(d3), 1 if project else 0, ptr(H)), c)
        return H
