"""multioptpy_amd -- MI355X-native optimizer-step path behind MultiOptPy's Optimizer plugin surface.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/mopt_hip.h)
and the host-side mirrors of the reference classes for this path:

    multioptpy_amd.Optimizer.rsirfo.RSIRFO          <- multioptpy.Optimizer.rsirfo.RSIRFO
    multioptpy_amd.MEP.pathopt_bneb_force           <- multioptpy.MEP.pathopt_bneb_force.CaluculationBNEB
    multioptpy_amd.Optimizer.rfo_neb.RFOOptimizer   <- multioptpy.Optimizer.rfo_neb.RFOOptimizer

Importing the package does not need a GPU; calling into it does, and there is no CPU fallback.
"""
__version__ = "0.1.0"
