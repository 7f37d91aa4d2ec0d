"""RFOOptimizer: drop-in for ``multioptpy.Optimizer.rfo_neb.RFOOptimizer`` (reference rfo_neb.py:86-208).

``optimize(...)`` keeps the reference's positional argument list (neb.py:926-931) and returns the new geometry
list in Angstrom.  The whole step -- BNEB force, Ayala curvature correction, one RSIRFO.run per image, step limit,
TR_calc, FIRE and the blend -- runs in libmopt_hip.so on a device-resident band (``mopt_band_step``).

Hessian persistence.  The reference loads ``tmp_hessian_{i}.npy`` before and saves it after every image step
(rfo_neb.py:18-25,175) because other parts of nebmain write those files (exact Hessians from the calculator, model
Hessians, bias Hessians).  With ``persist_hessians=True`` (default when ``config.NEB_FOLDER_DIRECTORY`` is set) this
class does the same round trip so it can sit under the unmodified driver; with ``False`` the Hessians never leave HBM.
"""
import os

import numpy as np

from ..band import Band


class RFOOptimizer:
    def __init__(self, config, persist_hessians=None, device=None, devices=None, transport="rccl"):
        """devices=[d0, d1, ...]: image-sharded band, one contiguous image block per listed GPU (the number of images must divide
        evenly); neighbour coordinates / gradients / energies travel inside the library (RCCL over xGMI, or peer copies with
        transport="peer").  Default: the whole band on one device."""
        self.config = config
        self.devices = None if devices is None else list(devices)
        self.transport = transport
        self.apply_ts_opt = True
        self.ratio_of_rfo_step = getattr(config, "ratio_of_rfo_step", 0.5)
        self.folder = getattr(config, "NEB_FOLDER_DIRECTORY", None)
        self.persist = (self.folder is not None) if persist_hessians is None else bool(persist_hessians)
        self.device = device
        self.band = None
        self.optimizer_instance_list = []     # kept for API compatibility (state lives on the device)
        self._pending_hessians = None         # in-memory counterpart of the tmp_hessian_i.npy side channel
        self._pending_model = None
        self._pending_bias = None             # bias Hessians of the next step (neb.py:806-832)
        self._have_files = False              # the band holds "tmp_hessian_i.npy" contents (stepped before, or Hessians were handed over)

    def close(self):
        """Release the device-resident band (Hessians, optimizer state, exchange streams and communicators)."""
        if self.band is not None:
            self.band.close()
            self.band = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_apply_ts_opt(self, apply_ts_opt):
        self.apply_ts_opt = apply_ts_opt

    def set_image_hessians(self, hessians):
        """Hand over one Hessian per image (or None to keep an image's current one) for the next `optimize` call: what the
        calculators do in the reference by writing tmp_hessian_{i}.npy (e.g. lj_calculation_tools.py:350-353)."""
        self._pending_hessians = list(hessians)

    def set_bias_hessians(self, bias_hessians):
        """One bias Hessian per image for the next `optimize` call: nebmain's _apply_bias_potential (neb.py:806-832) adds it to the
        image's tmp_hessian_{i}.npy when that file exists (every step: it accumulates, as in the reference) and, when no file exists yet,
        writes a non-zero bias Hessian AS the file -- the optimizer then starts that image from the bias Hessian alone."""
        self._pending_bias = [None if b is None else np.asarray(b, dtype=np.float64) for b in bias_hessians]

    def request_model_hessian(self, element_list, approx_hess_type="fischerd3old"):
        """The next `optimize` call first replaces every image's Hessian by the model Hessian of its geometry, computed on the
        device into the band's resident Hessians (the reference: ApproxHessian().main per image + tmp_hessian_{i}.npy, neb.py:515-523)."""
        self._pending_model = (list(element_list), approx_hess_type)

    def get_image_hessian(self, i):
        """The Hessian of image i as the reference would find it in tmp_hessian_{i}.npy after the step."""
        return self.band.get_hessian(i)

    def _file(self, i):
        return os.path.join(self.folder, f"tmp_hessian_{i}.npy")

    def optimize(self, geometry_num_list, total_force_list, prev_geometry_num_list, prev_total_force_list, optimize_num,
                 biased_energy_list, pre_biased_energy_list, pre_total_velocity, total_velocity, cos_list, pre_geom,
                 STRING_FORCE_CALC):
        X = np.asarray(geometry_num_list, dtype=np.float64)
        G = np.asarray(total_force_list, dtype=np.float64)
        E = np.asarray(biased_energy_list, dtype=np.float64)
        I, N = X.shape[0], X.shape[1]
        if self.band is None or optimize_num == 0 or (self.band.I, self.band.N) != (I, N):
            if self.band is not None:
                self.band.close()
            ci = getattr(STRING_FORCE_CALC, "APPLY_CI_NEB", 99999)
            opts = dict(ratio_of_rfo_step=self.ratio_of_rfo_step, fix_init_edge=getattr(self.config, "fix_init_edge", False),
                        fix_end_edge=getattr(self.config, "fix_end_edge", False), apply_ts_opt=self.apply_ts_opt, apply_ci_neb=ci)
            if self.devices is not None and len(self.devices) > 1:
                from ..parallel import ShardedBand
                self.band = ShardedBand(I, N, self.devices, transport=self.transport, **opts)
            else:
                self.band = Band(I, N, device=self.device if self.devices is None else self.devices[0], **opts)
        if self.persist:
            for i in range(I):
                f = self._file(i)
                self.band.set_hessian(i, np.load(f) if os.path.exists(f) else None)
        fresh = optimize_num == 0                              # (the band was re-created above: identity Hessians, no "files")
        if fresh:
            self._have_files = False
        have = [self._have_files or self.persist] * I
        if self._pending_hessians is not None:
            for i, h in enumerate(self._pending_hessians):
                if h is not None:
                    self.band.set_hessian(i, np.asarray(h, dtype=np.float64))
                    have[i] = True
            self._pending_hessians = None
        if self._pending_bias is not None:
            for i, b in enumerate(self._pending_bias):
                if b is None:
                    continue
                if have[i]:
                    self.band.add_hessian(i, b)
                elif not np.allclose(b, 0):
                    self.band.set_hessian(i, b)
            self._pending_bias = None
        self._have_files = True                                  # (every image's Hessian is "saved" behind the step, rfo_neb.py:175)
        self.band.upload(X, E, G)
        if self._pending_model is not None:
            self.band.model_hessian(*self._pending_model, coords=X, gradients=G)
            self._pending_model = None
        self.band.step(optimize_num)
        newX, F = self.band.download(want_force=True)
        if hasattr(STRING_FORCE_CALC, "tau_list"):
            pass        # tau stays on the device; calc_force of the mirror class refreshes tau_list when called
        if self.persist:
            for i in range(I):
                np.save(self._file(i), self.band.get_hessian(i))
        return newX
