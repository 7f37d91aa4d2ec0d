"""Developer diagnostic: the 8 x 40 identity-start band of test_band_step_overlap_switches_agree, several times in one process,
printing the per-step deviation from the oracle (environment switches are taken from the caller)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
import test_gpu_configs as t  # noqa: E402
from multioptpy_amd.band import Band  # noqa: E402

steps = list(t._oracle_band_steps(cases.band(8, 40), 6, hessians=None))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    band = Band(8, 40)
    errs = []
    for it, (X, E, Gr, ref) in enumerate(steps):
        band.upload(X, E, Gr)
        band.step(it)
        newX, _ = band.download(want_force=True)
        errs.append(float(np.abs(newX - ref).max()))
    band.close()
    print("rep", rep, " ".join(f"{e:.1e}" for e in errs), flush=True)
