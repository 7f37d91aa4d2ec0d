"""Device-resident NEB band (``mopt_band_*`` of include/mopt_hip.h)."""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib, ctx, check, f64, ptr


class Band:
    """I images of N atoms; this process owns images [first, first+count) (image sharding, one process per GPU)."""

    def __init__(self, I, N, first=0, count=None, device=None, ratio_of_rfo_step=0.5, fix_init_edge=False,
                 fix_end_edge=False, apply_ts_opt=True, apply_ci_neb=99999):
        self.I, self.N, self.n = int(I), int(N), 3 * int(N)
        self.first = int(first)
        self.count = self.I - self.first if count is None else int(count)
        self._ctx = ctx(device)
        self._h = C.c_void_p()
        check(lib().mopt_band_create(self._ctx, self.I, self.N, C.byref(self._h)), self._ctx)
        if self.first != 0 or self.count != self.I:
            check(lib().mopt_band_set_shard(self._h, self.first, self.count), self._ctx)
        check(lib().mopt_band_configure(self._h, float(ratio_of_rfo_step), int(bool(fix_init_edge)), int(bool(fix_end_edge)),
                                        int(bool(apply_ts_opt)), int(apply_ci_neb)), self._ctx)

    def close(self):
        if self._h:
            lib().mopt_band_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_hessian(self, image, H):
        Hc = None if H is None else f64(H)
        check(lib().mopt_band_set_hessian(self._h, int(image), ptr(Hc)), self._ctx)

    def add_hessian(self, image, B):
        """H_image += sym(B) on the device (a bias Hessian, neb.py:806-832)."""
        Bc = f64(B)
        check(lib().mopt_band_add_hessian(self._h, int(image), ptr(Bc)), self._ctx)

    def get_hessian(self, image):
        H = np.empty((self.n, self.n))
        check(lib().mopt_band_get_hessian(self._h, int(image), ptr(H)), self._ctx)
        return H

    def get_state(self, image):
        s = _abi.RsirfoState()
        check(lib().mopt_band_get_state(self._h, int(image), C.byref(s)), self._ctx)
        return s

    def upload(self, X, E, G):
        X = f64(X).reshape(self.I, self.n)
        G = f64(G).reshape(self.I, self.n)
        E = f64(E).reshape(self.I)
        check(lib().mopt_band_upload(self._h, ptr(X), ptr(E), ptr(G)), self._ctx)

    def model_hessian(self, element_list, approx_hess_type="fischerd3old", coords=None, gradients=None):
        """Replace the Hessians of the owned images by the model Hessian of the geometries last uploaded, on the device
        (neb.py:515-523 with the tmp_hessian_i.npy hand-over removed).  Names with the `ts` or `sr` modifier (host-array steps between the
        model and the clipping) go through ApproxHessian().batch and set_hessian instead and need the images' coordinates (I or count, N, 3;
        bohr) and gradients."""
        from .ModelHessian.approx_hessian import element_parameters, model_kind, FLAG_CLIP, ApproxHessian
        kind, flags = model_kind(approx_hess_type, with_flags=True)
        if flags & ~FLAG_CLIP:
            if coords is None:
                raise NotImplementedError("band-resident model Hessians take the `clip` modifier only; for `ts` / `sr` pass coords (and gradients)")
            Xc = f64(coords).reshape(-1, self.N, 3)
            Gc = None if gradients is None else f64(gradients).reshape(-1, self.N, 3)
            if Xc.shape[0] == self.I and self.count != self.I:
                Xc = Xc[self.first:self.first + self.count]
                Gc = None if Gc is None else Gc[self.first:self.first + self.count]
            Hs = ApproxHessian().batch(Xc, element_list, approx_hess_type, cart_gradients=Gc)
            for k in range(self.count):
                self.set_hessian(self.first + k, Hs[k])
            return
        par, d3 = element_parameters(element_list, kind)
        check(lib().mopt_band_model_hessian_ex(self._h, kind, flags, ptr(par), ptr(d3)), self._ctx)

    def attach(self, comm):
        """Join an exchange group (multioptpy_amd.parallel): this band must own the rank's image block."""
        check(lib().mopt_band_attach_comm(self._h, comm._h), self._ctx)
        self._comm = comm

    def upload_shard(self, X_own, E_own, G_own):
        """Inputs of the owned images only (count, N, 3) / (count,); neighbours arrive through parallel.halo_exchange."""
        X = f64(X_own).reshape(self.count, self.n)
        G = f64(G_own).reshape(self.count, self.n)
        E = f64(E_own).reshape(self.count)
        check(lib().mopt_band_upload_shard(self._h, ptr(X), ptr(E), ptr(G)), self._ctx)

    def upload_shard_dev(self, dX_own, dE_own, dG_own):
        check(lib().mopt_band_upload_shard_dev(self._h, C.c_void_p(int(dX_own)), C.c_void_p(int(dE_own)), C.c_void_p(int(dG_own))), self._ctx)

    def upload_dev(self, dX, dE, dG, E_host=None):
        """Device pointers (ints, e.g. torch.Tensor.data_ptr()) of X (I,N,3), E (I), G (I,N,3) float64."""
        Eh = None if E_host is None else f64(E_host).reshape(self.I)
        check(lib().mopt_band_upload_dev(self._h, C.c_void_p(int(dX)), C.c_void_p(int(dE)), C.c_void_p(int(dG)), ptr(Eh)), self._ctx)

    def download_dev(self, d_newX):
        check(lib().mopt_band_download_dev(self._h, C.c_void_p(int(d_newX))), self._ctx)

    def timing(self, enable):
        check(lib().mopt_ctx_timing(self._ctx, int(bool(enable))), self._ctx)

    def timing_read(self, family):
        ms = C.c_double(0.0)
        nl = C.c_int64(0)
        check(lib().mopt_ctx_timing_read(self._ctx, family.encode(), C.byref(ms), C.byref(nl)), self._ctx)
        return ms.value, nl.value

    def step(self, optimize_num):
        check(lib().mopt_band_step(self._h, int(optimize_num)), self._ctx)

    def sync(self):
        check(lib().mopt_ctx_sync(self._ctx), self._ctx)

    def failures(self):
        """Sticky failure bits of the steps since the last look (4: eigen-solver failure, 8: hand-off time-out); a time-out of the last
        step is repaired, anything else raises MoptError."""
        bits = C.c_int(0)
        check(lib().mopt_band_failures(self._h, C.byref(bits)), self._ctx)
        return bits.value

    def download(self, want_force=False, want_rfo=False):
        newX = np.empty((self.count, self.N, 3))
        F = np.empty((self.I, self.N, 3)) if want_force else None
        R = np.empty((self.count, self.N, 3)) if want_rfo else None
        check(lib().mopt_band_download(self._h, ptr(newX), ptr(F), ptr(R)), self._ctx)
        out = [newX]
        if want_force:
            out.append(F)
        if want_rfo:
            out.append(R)
        return out[0] if len(out) == 1 else tuple(out)
