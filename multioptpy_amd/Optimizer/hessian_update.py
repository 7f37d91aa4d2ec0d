"""ModelHessianUpdate: drop-in for ``multioptpy.Optimizer.hessian_update.ModelHessianUpdate`` (reference hessian_update.py:7-433), the
class the IRC integrators instantiate to reuse the quasi-Newton update rules (IRC/lqa.py:59,286; IRC/hpc.py:163,421) and that
RSIRFO / EnhancedRSPRFO dispatch to.  Every public ``*_hessian_update(hess, displacement, delta_grad)`` returns the update matrix
delta-H (to be ADDED by the caller), computed by the update kernels of libmopt_hip.so (``mopt_hess_delta``: no guards other than the
rule's own denominators)."""
import numpy as np

from .. import _abi
from .._lib import lib, ctx, check, f64, ptr


class ModelHessianUpdate:
    def __init__(self, device=None):
        self.device = device
        self.Initialization = True

    def _delta(self, key, hess, displacement, delta_grad):
        H = f64(hess)
        n = H.shape[0]
        s = f64(np.asarray(displacement)).reshape(1, n)
        y = f64(np.asarray(delta_grad)).reshape(1, n)
        dH = np.empty((n, n))
        c = ctx(self.device)
        check(lib().mopt_hess_delta(c, _abi.rule_from_method(key), n, 1, ptr(H), ptr(s), ptr(y), ptr(dH)), c)
        return dH

    def flowchart_hessian_update(self, hess, displacement, delta_grad, method="auto"):
        return self._delta("flowchart", hess, displacement, delta_grad)


def _make(key):
    def f(self, hess, displacement, delta_grad):
        return self._delta(key, hess, displacement, delta_grad)
    return f


# method name of the reference -> dispatch key of rsirfo.py:208-251 (same kernels underneath)
for _name, _key in (("BFGS_hessian_update", "bfgs"), ("SR1_hessian_update", "sr1"), ("PSB_hessian_update", "psb"), ("FSB_hessian_update", "fsb"),
                    ("CFD_FSB_hessian_update", "cfd_fsb"), ("Bofill_hessian_update", "bofill"), ("CFD_Bofill_hessian_update", "cfd_bofill"),
                    ("pCFD_Bofill_hessian_update", "pcfd_bofill"), ("MSP_hessian_update", "msp"), ("BFGS_hessian_update_dd", "bfgs_dd"),
                    ("FSB_hessian_update_dd", "fsb_dd"), ("CFD_FSB_hessian_update_dd", "cfd_fsb_dd")):
    setattr(ModelHessianUpdate, _name, _make(_key))
