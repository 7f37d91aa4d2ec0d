"""Loader for libmopt_hip.so (the C ABI of include/mopt_hip.h).  There is no CPU fallback:
if the library is missing or no MI355X is visible, every entry point raises."""
import ctypes as C
import os
import threading

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOPT_HIP_LIB") or os.path.join(_HERE, "libmopt_hip.so")      # override: A/B builds of the same ABI

DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)


class MoptError(RuntimeError):
    pass


_lib = None
_ctx = {}
_lock = threading.Lock()


def _proto(lib):
    vp = C.c_void_p
    lib.mopt_abi_version.restype = C.c_int
    lib.mopt_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.mopt_ctx_destroy.argtypes = [vp]
    lib.mopt_last_error.argtypes = [vp]
    lib.mopt_last_error.restype = C.c_char_p
    lib.mopt_ctx_sync.argtypes = [vp]
    if hasattr(lib, "mopt_debug_occupy"):
        lib.mopt_debug_occupy.argtypes = [vp, C.c_double, C.c_int, C.c_int]
    lib.mopt_debug_poison.argtypes = [C.c_int]
    if hasattr(lib, "mopt_model_hessian_sr"):
        lib.mopt_model_hessian_sr.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP, DP, DP]
        lib.mopt_hessian_clip.argtypes = [vp, C.c_int, C.c_int, DP]
    if hasattr(lib, "mopt_debug_last_deflation"):
        lib.mopt_debug_last_deflation.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
    lib.mopt_ctx_timing.argtypes = [vp, C.c_int]
    lib.mopt_ctx_timing_read.argtypes = [vp, C.c_char_p, DP, C.POINTER(C.c_int64)]
    lib.mopt_hess_update.argtypes = [vp, C.c_int, C.c_int, C.c_int, DP, DP, DP, DP, DP, DP]
    if hasattr(lib, "mopt_hess_delta"):
        lib.mopt_hess_delta.argtypes = [vp, C.c_int, C.c_int, C.c_int, DP, DP, DP, DP]
    lib.mopt_project_trrot_hess.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP]
    lib.mopt_project_trrot_grad.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP]
    lib.mopt_eigh.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP, IP]
    lib.mopt_dgemm.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP]
    lib.mopt_quadratic_model.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP, DP]
    lib.mopt_clip_and_move.argtypes = [vp, C.c_int, DP, DP, C.c_double, C.c_double, DP, DP, DP]
    lib.mopt_kabsch.argtypes = [vp, C.c_int, DP, DP, DP, DP]
    lib.mopt_neb_tr_calc.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP, C.c_int, C.c_int, DP]
    lib.mopt_neb_fire.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP, DP, C.c_int, C.c_int, C.c_int, DP, DP, C.c_double, DP, DP]
    lib.mopt_neb_sd.argtypes = [vp, C.c_int, C.c_int, DP, DP, C.c_int, C.c_int, C.c_double, DP]
    lib.mopt_bneb_force.argtypes = [vp, C.c_int, C.c_int, DP, DP, DP, C.c_int, C.c_int, DP, DP]
    lib.mopt_neb_force.argtypes = [vp, C.c_int, C.c_int, C.c_int, DP, DP, DP, C.c_int, C.c_int, DP, DP]
    lib.mopt_rsirfo_run.argtypes = [vp, C.c_int, C.c_int, C.POINTER(_abi.RsirfoParams), C.POINTER(_abi.RsirfoState),
                                    DP, DP, DP, DP, DP, DP, DP, DP, DP, DP, DP]
    if hasattr(lib, "mopt_mf_rsirfo_begin"):
        lib.mopt_mf_rsirfo_begin.argtypes = [vp, C.c_int, C.POINTER(_abi.RsirfoParams), C.POINTER(_abi.RsirfoState), DP, DP, DP, DP, DP, DP, DP, C.c_double,
                                             DP, DP, DP, DP, C.POINTER(C.c_int)]
        lib.mopt_mf_rsirfo_finish.argtypes = [vp, C.c_int, C.POINTER(_abi.RsirfoParams), C.POINTER(_abi.RsirfoState), C.POINTER(C.c_int), C.c_int, DP]
    lib.mopt_rsprfo_run.argtypes = [vp, C.c_int, C.POINTER(_abi.RsprfoParams), C.POINTER(_abi.RsprfoState), DP, DP, DP, DP, DP, DP,
                                    C.c_double, DP, DP, C.c_double, DP, DP, DP, DP]
    if hasattr(lib, "mopt_band_create"):
        lib.mopt_band_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
        lib.mopt_band_destroy.argtypes = [vp]
        lib.mopt_band_set_shard.argtypes = [vp, C.c_int, C.c_int]
        lib.mopt_band_set_hessian.argtypes = [vp, C.c_int, DP]
        lib.mopt_band_get_hessian.argtypes = [vp, C.c_int, DP]
        lib.mopt_band_add_hessian.argtypes = [vp, C.c_int, DP]
        lib.mopt_band_get_state.argtypes = [vp, C.c_int, C.POINTER(_abi.RsirfoState)]
        lib.mopt_band_configure.argtypes = [vp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.mopt_band_upload.argtypes = [vp, DP, DP, DP]
        lib.mopt_band_step.argtypes = [vp, C.c_int]
        lib.mopt_band_download.argtypes = [vp, DP, DP, DP]
        lib.mopt_band_upload_dev.argtypes = [vp, vp, vp, vp, DP]
        lib.mopt_band_download_dev.argtypes = [vp, vp]
        lib.mopt_band_failures.argtypes = [vp, IP]
    if hasattr(lib, "mopt_crsirfo_run"):
        lib.mopt_constraint_project_hess.argtypes = [vp, C.c_int, DP, C.c_int, DP, C.c_int, DP, C.c_double, C.c_double, DP]
        lib.mopt_crsirfo_run.argtypes = [vp, C.c_int, C.c_int, C.POINTER(_abi.RsirfoParams), C.POINTER(_abi.RsirfoState),
                                         DP, DP, DP, DP, DP, DP, DP, DP, DP, DP, DP, DP]
        lib.mopt_crsirfo_run_rowspace.argtypes = [vp, C.c_int, C.c_int, C.POINTER(_abi.RsirfoParams), C.POINTER(_abi.RsirfoState),
                                         DP, DP, DP, DP, DP, DP, DP, DP, DP, DP, DP, DP]
    if hasattr(lib, "mopt_keep_angle"):
        lib.mopt_keep_angle.argtypes = [vp, C.c_int, C.c_int, DP, C.c_int, IP, DP, DP, DP, DP]
        lib.mopt_keep_pot_v2.argtypes = [vp, C.c_int, DP, C.c_int, IP, C.c_int, IP, C.c_double, C.c_double, DP, DP, DP]
    if hasattr(lib, "mopt_well_pot"):
        lib.mopt_well_pot.argtypes = [vp, C.c_int, C.c_int, DP, C.c_int, IP, C.c_int, IP, DP, DP, DP, DP]
    if hasattr(lib, "mopt_keep_pot"):
        lib.mopt_keep_pot.argtypes = [vp, C.c_int, DP, C.c_int, IP, DP, DP, DP, DP]
    if hasattr(lib, "mopt_afir"):
        lib.mopt_afir.argtypes = [vp, C.c_int, DP, DP, C.c_int, IP, C.c_int, IP, C.c_double, DP, DP, DP]
    if hasattr(lib, "mopt_model_hessian"):
        lib.mopt_model_hessian.argtypes = [vp, C.c_int, C.c_int, C.c_int, DP, DP, DP, C.c_int, DP]
        lib.mopt_band_model_hessian.argtypes = [vp, C.c_int, DP, DP]
        lib.mopt_model_hessian_ex.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, DP, DP, DP, DP, C.c_int, DP]
        lib.mopt_band_model_hessian_ex.argtypes = [vp, C.c_int, C.c_int, DP, DP]
    if hasattr(lib, "mopt_halo_exchange"):
        lib.mopt_comm_unique_id.argtypes = [vp, C.c_int]
        lib.mopt_comm_create_rccl_rank.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
        lib.mopt_comm_create_rccl_all.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp)]
        lib.mopt_comm_create_peer.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp)]
        lib.mopt_comm_destroy.argtypes = [vp]
        lib.mopt_band_attach_comm.argtypes = [vp, vp]
        lib.mopt_band_upload_shard.argtypes = [vp, DP, DP, DP]
        lib.mopt_band_upload_shard_dev.argtypes = [vp, vp, vp, vp]
        lib.mopt_halo_exchange.argtypes = [C.c_int, C.POINTER(vp)]
        lib.mopt_band_step_multi.argtypes = [C.c_int, C.POINTER(vp), C.c_int]


def lib():
    """The loaded shared library; raises MoptError when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise MoptError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  multioptpy_amd has no CPU fallback.")
                l = C.CDLL(LIB_PATH)
                _proto(l)
                if l.mopt_abi_version() != 1:
                    raise MoptError("libmopt_hip.so ABI version mismatch")
                _lib = l
    return _lib


def ctx(device=None):
    """One context per device per process (LOCAL_RANK selects the device under torch.distributed.run)."""
    if device is None:
        device = int(os.environ.get("MOPT_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _lock:
        c = _ctx.get(device)
    if c is None:
        l = lib()
        h = C.c_void_p()
        rc = l.mopt_ctx_create(device, C.byref(h))
        if rc != 0 or not h:
            raise MoptError(f"mopt_ctx_create(device={device}) failed with {rc}: no usable HIP device "
                            "(multioptpy_amd needs an MI355X; there is no CPU fallback)")
        with _lock:
            _ctx[device] = h
        c = h
    return c


def _drain():
    """Interpreter exit: let every context's streams run dry before the HIP runtime is torn down (asynchronous bookkeeping copies of the
    library may still be in flight; objects with __del__ -- bands, communicators -- are destroyed afterwards, against live contexts)."""
    if _lib is None:
        return
    for c in list(_ctx.values()):
        try:
            _lib.mopt_ctx_sync(c)
        except Exception:
            pass


import atexit  # noqa: E402
atexit.register(_drain)


def check(rc, c=None):
    if rc != 0:
        msg = lib().mopt_last_error(c).decode() if c else ""
        if rc == -4:
            raise ValueError(msg or "call sequence error")
        raise MoptError(f"libmopt_hip error {rc}: {msg}")


def f64(a):
    """C-contiguous float64 view/copy."""
    return np.ascontiguousarray(a, dtype=np.float64)


_first_double = C.c_double.from_buffer


def ptr(a):
    """Pointer argument for an ndarray (or None).  A c_double that shares the array's buffer is what ctypes passes by reference for a
    POINTER(c_double) parameter; making one costs 0.3 us against 2.1 us for ndarray.ctypes.data_as -- a dozen of them are a tenth of a
    small-geometry RSIRFO.run.  Read-only or empty arrays take the slow way."""
    if a is None:
        return None
    try:
        return _first_double(a)
    except (TypeError, ValueError, BufferError):
        return a.ctypes.data_as(DP)
