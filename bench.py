#!/usr/bin/env python
"""bench.py -- optimizer-step wall time of the RFO-NEB band step on MI355X.

Metric (BASELINE.json): optimizer-step ms (Hessian update + RFO solve + NEB force, QM call excluded) on a
32-image x 100-atom NEB band, at 1/2/4/8 GPUs (images sharded over the ranks, one process per GPU).

One "step" = RFOOptimizer.optimize of the reference (multioptpy/Optimizer/rfo_neb.py:104-208) for the whole band:
BNEB force, Ayala curvature correction, one RSIRFO.run per image (block-Bofill/FSB update, TR/rot projections,
symmetric eigen-solve, RS-I-RFO step), step limit, TR_calc, FIRE, blend.  Inputs (geometries, energies, gradients of
an argon LJ surface, synthetic) for every step are staged in HBM before the timed region.  For N > 1 every rank is handed
only its own images' energies and gradients per step (what its calculator would produce); the timed region contains the halo
exchange a sharded run needs -- neighbour coordinates and gradients, all energies, FIRE's partial sums -- inside the library over
RCCL (csrc/comm.hip), with no host wait between steps.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector/matrix peak (vendor spec; not in the local guide)


def lj_hessians(X):
    """Exact LJ Hessians of every image (I, n, n), vectorised (same formula as tests/golden/cases.py:lj, which loops in Python)."""
    import cases
    I, N = X.shape[0], X.shape[1]
    out = np.empty((I, 3 * N, 3 * N))
    eye3 = np.eye(3)
    for i in range(I):
        x = X[i]
        d = x[:, None, :] - x[None, :, :]
        r2 = np.sum(d * d, axis=2)
        np.fill_diagonal(r2, 1.0)
        s2 = cases._RMIN ** 2 / r2
        s6 = s2 ** 3
        s12 = s6 ** 2
        f = cases._EPS * (-12 * s12 + 12 * s6) / r2
        h2 = cases._EPS * (156 * s12 - 84 * s6) / r2
        np.fill_diagonal(f, 0.0)
        np.fill_diagonal(h2, 0.0)
        blk = ((h2 - f) / r2)[:, :, None, None] * (d[:, :, :, None] * d[:, :, None, :]) + f[:, :, None, None] * eye3
        H4 = -blk
        diag = blk.sum(axis=1)
        for a in range(N):
            H4[a, a] = diag[a]
        out[i] = H4.transpose(0, 2, 1, 3).reshape(3 * N, 3 * N)
    return out


def make_trajectory(I, N, nsteps, device, hess0=None):
    """Untimed: run the band optimizer itself for nsteps to obtain realistic per-step inputs (X, E, G)."""
    import cases
    from multioptpy_amd.band import Band
    band = Band(I, N, device=device)
    X = cases.band(I, N)
    if hess0 is not None:
        for i in range(I):
            band.set_hessian(i, hess0[i])
    traj = []
    for it in range(nsteps):
        E = np.zeros(I)
        G = np.zeros((I, N, 3))
        for i in range(I):
            E[i], G[i] = cases.lj(X[i])
        traj.append((X.copy(), E, G))
        band.upload(X, E, G)
        band.step(it)
        X = band.download() / cases.BOHR2ANG
    band.close()
    return traj


def _oracle_steps(traj, I, N, budget_s, max_samples, hess0=None):
    """Run the oracle's RFO-NEB step over the first steps of the trajectory; returns (per-step seconds, new geometries)."""
    import contextlib
    import io
    from oracle import neb as oneb
    opt = oneb.RFONEB([None] * I if hess0 is None else [h.copy() for h in hess0])
    calc = oneb.BNEB(literal=True)
    preX = preG = None
    pre_vel = []
    times, outs = [], []
    t_begin = time.perf_counter()
    for it, (X, E, G) in enumerate(traj):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            F = calc.calc_force(X, E, G, it)            # neb.py:542 (first of the two calls per step)
            newX = opt.optimize(X, G, preX, preG, it, E, pre_vel, F.copy(), calc)
        dt = time.perf_counter() - t0
        times.append(dt)
        outs.append(newX)
        preX, preG, pre_vel = X.copy(), G.copy(), F.copy()
        if len(times) - 1 >= max_samples or (it >= 1 and (time.perf_counter() - t_begin) + dt > budget_s):
            break
    return times, outs


def cpu_baseline(traj, I, N, budget_s=25.0, hess0=None):
    """The oracle (NumPy restatement of the reference algorithm incl. its SVD-based B-matrix projection) timed on this host's
    cores on the first steps of the same trajectory, at two BLAS thread settings: 1 thread and a capped all-core
    setting (the box's CPU share for one GPU is 16 cores; more OpenBLAS threads than that on 300 x 300 problems only
    oversubscribe).  `value` is the faster of the two medians; both are reported."""
    from threadpoolctl import threadpool_limits, threadpool_info
    ncpu = os.cpu_count() or 1
    cap = int(os.environ.get("MOPT_BENCH_CPU_THREADS", min(ncpu, 16)))
    blas = sorted({f"{p.get('internal_api')}:{p.get('version')}" for p in threadpool_info()})
    runs = {}
    outs = None
    for thr, share in ((cap, 0.6), (1, 0.4)):
        with threadpool_limits(limits=thr):
            times, o = _oracle_steps(traj, I, N, budget_s * share, 20, hess0)
        if outs is None:
            outs = o
        timed = times[1:] if len(times) > 1 else times
        runs[thr] = {"median_ms": 1e3 * float(np.median(timed)), "min_ms": 1e3 * float(np.min(timed)), "samples": len(timed)}
    best = min(runs, key=lambda t: runs[t]["median_ms"])
    return {
        "value": runs[best]["median_ms"], "unit": "ms/step", "cores": int(best), "kind": "port",
        "threads": {str(t): r for t, r in runs.items()}, "host_cpus": ncpu, "blas": blas,
        "sample": f"steps 1..k of the same {I}x{N} band trajectory (step 0 = warm-up), median of k samples per thread setting "
                  f"(k <= 20, bounded by --cpu-budget {budget_s:.0f} s split 60/40); oracle/neb.py RFONEB with the reference's "
                  f"SVD-based BNEB projection; NumPy/BLAS limited to {cap} threads and to 1 thread (threadpoolctl), stdout suppressed, no disk I/O",
    }, outs


FAMILIES = ("hess_update", "project", "eigh_init", "eigh_jacobi", "eigh_tridiag", "eigh_bisect", "eigh_invit", "eigh_backtrans",
            "eigh_final", "rfo", "neb_force", "neb_ayala", "neb_post", "halo", "eigh_dc", "rfo_scalar", "eigh_small", "eigh_early", "apply_qt", "larft",
            "small_step")
# algorithmic bytes per image-step of a family in units of n^2 (80 n^2 for the whole RSIRFO.run, of which the
# eigen-solve's share is input read + eigenvector write; reduction route: the tridiagonalisation reads the matrix once (8 n^2)
# and leaves the reflectors in its strict upper triangle (4 n^2); the back-transformation reads reflectors + eigenvectors and
# writes eigenvectors (20 n^2)), and algorithmic flops in units of n^3
SHARE_BYTES = {"eigh_jacobi": 16.0, "eigh_tridiag": 12.0, "eigh_backtrans": 20.0, "eigh_invit": 8.0, "hess_update": 24.0,
               "project": 16.0, "rfo": 24.0}
SHARE_FLOPS = {"eigh_tridiag": 4.0 / 3.0, "eigh_backtrans": 2.0}
KERNEL_OF = {"eigh_jacobi": "k_jac_step", "eigh_tridiag": "k_tri_persist3", "eigh_backtrans": "k_tri_backtrans_mfma",
             "eigh_invit": "k_tri_invit", "eigh_bisect": "k_tri_bisect", "small_step": "k_small_rsirfo"}
# what the dominant kernel is actually bound by; the HBM and FP64 fractions are reported beside it
BOUND_NOTE = {"small_step": "latency (one workgroup per image runs the whole step: barrier-separated phases of a few us each, matrices in LDS)",
              "eigh_tridiag": "latency (inter-workgroup hand-off chain of n-1 dependent Householder steps)",
              "eigh_invit": "latency (sequential inverse iteration inside eigenvalue clusters)",
              "eigh_bisect": "instruction issue (Sturm recurrences)"}


SIDE_FAMILIES = {"halo", "eigh_dc", "rfo_scalar", "eigh_small", "eigh_early", "apply_qt", "larft"}


def library_stamp():
    """Identify the dominant kernel's code: sha1 over the translation unit of the eigen-solver's tridiagonalisation (k_tridiag.hip) and
    every header of the library (profiles/traffic.json carries the same; kernels of other files do not enter the counter figure)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted([os.path.join(ROOT, "multioptpy_amd", "csrc", "k_tridiag.hip")] + glob.glob(os.path.join(ROOT, "multioptpy_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "multioptpy_amd", "csrc", "*.hpp"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def main_single(args):
    """One geometry through the plugin classes: what optmain calls per iteration (optimizer.py:685-697).  A step = one `run` call incl. the
    PCIe traffic of its host arrays (the plugin boundary hands over host ndarrays); inputs of every step come from the optimizer's own
    trajectory on the LJ surface.  cpu_baseline: the oracle's classes (NumPy restatement) on the same inputs, one BLAS thread."""
    import contextlib
    import io
    import cases
    from threadpoolctl import threadpool_limits
    from multioptpy_amd.Optimizer.rsirfo import RSIRFO
    from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO
    from oracle.rsirfo import RSIRFO as ORSIRFO
    from oracle.rsprfo import EnhancedRSPRFO as ORSPRFO
    N = args.single_geometry
    n = 3 * N
    W, K = max(1, args.warmup), args.steps
    x0 = cases.cluster(N, jitter=0.3) * 0.95
    H0 = cases.lj(x0, hessian=True)[2]

    def trajectory(cls, method, order, nstep, quiet, follow=None):
        """`follow` = the geometries of another run: every step then gets THAT run's inputs (previous geometry / gradient / energy included),
        so the two move vectors of a step are comparable one by one."""
        opt = cls(method=method, saddle_order=order, element_list=["Ar"] * N)
        opt.set_hessian(H0.copy())
        opt.set_bias_hessian(np.zeros((n, n)))
        xs, px, pg, pe = x0.copy(), [], [], 0.0
        times, geoms, moves, flags = [], [], [], []
        for k in range(nstep):
            if follow is not None:
                xs = follow[k]
            geoms.append(xs.copy())
            e, g = cases.lj(xs)[:2]
            t0 = time.perf_counter()
            if quiet:
                with contextlib.redirect_stdout(io.StringIO()):
                    mv = opt.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, pe, [], [], g.reshape(-1, 1), pg)
            else:
                mv = opt.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, pe, [], [], g.reshape(-1, 1), pg)
            times.append(time.perf_counter() - t0)
            moves.append(np.asarray(mv).reshape(-1).copy())
            flags.append(bool(getattr(opt, "trace", {}).get("basis_dependent", False)))
            px, pg, pe = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy(), e
            step = np.asarray(mv).reshape(N, 3)
            xs = xs - step * min(1.0, 0.1 / max(np.linalg.norm(step), 1e-12))
        return times, geoms, moves, flags

    out = {}
    for name, cls, ocls, method, order in (("RSIRFO.run", RSIRFO, ORSIRFO, "rsirfo_bfgs", 0), ("EnhancedRSPRFO.run", EnhancedRSPRFO, ORSPRFO, "rsprfo_bofill", 1)):
        tg, xg, mg, _ = trajectory(cls, method, order, W + K, False)
        with threadpool_limits(limits=1):
            tc, _, mc, fl = trajectory(ocls, method, order, W + K, True, follow=xg)
        # the oracle's mode following can hang on LAPACK's basis inside a group of equal shifted eigenvalues (oracle/rsprfo.py
        # _basis_dependent): from the first such step on the two runs may legitimately follow different modes
        upto = fl.index(True) if True in fl else len(fl)
        dev = max((float(np.abs(a - b).max()) for a, b in zip(mg[:upto], mc[:upto])), default=0.0)
        out[name] = {"gpu_ms": 1e3 * float(np.median(tg[W:])), "gpu_min_ms": 1e3 * float(np.min(tg[W:])),
                     "cpu_ms": 1e3 * float(np.median(tc[W:])), "cpu_min_ms": 1e3 * float(np.min(tc[W:])),
                     "max_abs_dev_move_vector_bohr": dev, "steps_compared": upto,
                     "first_basis_dependent_step": (upto if upto < len(fl) else None)}
        if not dev < 1e-8:
            raise SystemExit(f"{name}: move vectors differ from the oracle's by {dev:.3e} on the same inputs")
    line = {
        "metric": "optimizer-step ms (Hessian update+RFO, QM excluded), single geometry", "value": out["RSIRFO.run"]["gpu_ms"], "unit": "ms/step",
        "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": out["RSIRFO.run"]["gpu_ms"], "higher_is_better": False, "scaling": "replicas only",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"RSIRFO.run (rsirfo_bfgs) through the plugin class, one {N}-atom Ar LJ cluster (n={n}), exact LJ Hessian at the start, "
                               "host ndarrays in and out (PCIe included)", "atoms": N,
                   "second": f"EnhancedRSPRFO.run (rsprfo_bofill, saddle order 1): {out['EnhancedRSPRFO.run']['gpu_ms']:.4f} ms"},
        "roofline": None,
        "cpu_baseline": {"value": out["RSIRFO.run"]["cpu_ms"], "unit": "ms/step", "cores": 1, "kind": "port",
                         "sample": f"the oracle's RSIRFO.run (oracle/rsirfo.py) on the same inputs (the device run's trajectory), 1 BLAS thread, stdout suppressed; "
                                   f"EnhancedRSPRFO.run: {out['EnhancedRSPRFO.run']['cpu_ms']:.4f} ms",
                         "host_cpus": os.cpu_count()},
        "classes": out,
    }
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--atoms", type=int, default=100)
    ap.add_argument("--hessian-init", choices=("identity", "exact"), default="identity",
                    help="initial per-image Hessians of the band: identity (rfo_neb.py:25) or exact LJ (-fc 1, neb.py:794-804)")
    ap.add_argument("--no-second-workload", action="store_true", help="skip the short exact-Hessian-start measurement reported beside the line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=25.0)
    ap.add_argument("--single-geometry", type=int, default=0, metavar="NATOMS",
                    help="BASELINE configs 2 / 3 (optmain on one geometry): time RSIRFO.run and EnhancedRSPRFO.run through the plugin classes "
                         "(host arrays in and out) for an N-atom LJ cluster, with the oracle classes on the host beside them")
    args = ap.parse_args()
    if args.single_geometry > 0:
        return main_single(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    backend = os.environ.get("MOPT_BENCH_BACKEND", "nccl")      # "gloo" + MOPT_BENCH_SHARE_GPU=1: rehearsal on a 1-GPU box
    if os.environ.get("MOPT_BENCH_SHARE_GPU", "0") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import cases
    from multioptpy_amd.band import Band
    I, N = args.images, args.atoms
    n = 3 * N
    W, K = max(1, args.warmup), args.steps
    from multioptpy_amd.parallel import shard_bounds, halo_exchange, step_multi, rccl_comm_for_process
    first, count = shard_bounds(I, world, rank)
    dev = torch.device("cuda", local)
    comm = rccl_comm_for_process(local) if world > 1 else None      # RCCL communicator of the library, id broadcast by torch.distributed

    checks = {}

    def measure(hess_init, W, K, want_families):
        """W warm-up + K timed band steps of the workload; returns (ms per step, family timings, trajectory, Hessians, sweeps)."""
        hess0 = lj_hessians(cases.band(I, N)) if hess_init == "exact" else None
        # ---- inputs for every step, staged in HBM (untimed)
        # (one step more than is timed: traj[W + K] holds the geometry the last timed step must produce -- the check below)
        traj = make_trajectory(I, N, W + K + 1, local, hess0)
        staged = [(torch.from_numpy(X).to(dev), torch.from_numpy(E).to(dev), torch.from_numpy(G).to(dev)) for X, E, G in traj[:W + K]]
        torch.cuda.synchronize()

        def run(band, timing=False):
            # world > 1: every rank feeds ONLY its own images (what its calculator would have produced); neighbour coordinates and
            # gradients, all energies and FIRE's partial sums travel inside the library over RCCL (csrc/comm.hip), on the band's
            # exchange stream, inside the timed region; nothing waits on the host between steps
            def one(k):
                dX, dE, dG = staged[k]
                if world > 1:
                    band.upload_shard_dev(dX[first:first + count].data_ptr(), dE[first:first + count].data_ptr(), dG[first:first + count].data_ptr())
                    halo_exchange([band])
                    step_multi([band], k)
                else:
                    band.upload_dev(dX.data_ptr(), dE.data_ptr(), dG.data_ptr(), traj[k][1] if k < W else None)
                    band.step(k)
            for k in range(W):
                one(k)
            band.sync()
            if timing:
                band.timing(True)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(W, W + K):
                one(k)
            band.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            return time.perf_counter() - t0

        def fresh():
            b = Band(I, N, first=first, count=count, device=local)
            if world > 1:
                b.attach(comm)
            if hess0 is not None:
                for i in range(first, first + count):
                    b.set_hessian(i, hess0[i])
            return b

        band = fresh()
        elapsed = run(band)
        # ---- outside the timed region: (1) no step of the timed loop may have failed (the failure word is sticky on the device; a
        # hand-off time-out or a solver failure in ANY of the K steps raises here instead of being timed as a valid step);
        # (2) the geometry the last timed step produced on this rank's images equals the whole-band trajectory (computed above on this
        # rank's GPU by the unsharded band): for N > 1 this is the proof that the RCCL exchange delivered the right rows
        bits = band.failures()
        if bits:
            raise SystemExit(f"bench: a timed band step failed (failure bits {bits})")
        got = band.download() / cases.BOHR2ANG
        dev_shard = float(np.abs(got - traj[W + K][0][first:first + count]).max())
        if world > 1:
            t = torch.tensor([elapsed, dev_shard], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, dev_shard = float(t[0].item()), float(t[1].item())
        if not dev_shard < 1e-10:
            raise SystemExit(f"bench: the sharded band deviates from the whole-band trajectory by {dev_shard} bohr after the last timed step")
        checks[hess_init] = dev_shard
        sweeps = [band.get_state(first + j).jacobi_sweeps for j in range(count)]
        band.close()
        fam = {}
        if want_families:
            # the same K steps again on a fresh band with HIP-event timing per kernel family (events on the library's stream;
            # every rank runs it -- run() contains barriers -- but only rank 0 records and reports)
            b2 = fresh()
            run(b2, timing=(rank == 0))
            if b2.failures():
                raise SystemExit("bench: a band step of the family-timing pass failed")
            if rank == 0:
                if os.environ.get("MOPT_TIMELINE"):
                    b2.timing_read("timeline")
                for name in FAMILIES:
                    ms, nl = b2.timing_read(name)
                    fam[name] = {"ms_per_step": ms / K, "launches_per_step": nl / K}
                b2.timing(False)
            b2.close()
        return 1e3 * elapsed / K, fam, traj, hess0, sweeps

    ms_per_step, fam, traj, hess0, sweeps = measure(args.hessian_init, W, K, True)

    roofline = None
    if rank == 0:
        # (the dominant kernel is looked for on the main stream: helper-stream families overlap it, and one of them -- eigh_early -- is a
        # watcher that mostly waits)
        dom = max((k for k in fam if k not in SIDE_FAMILIES), key=lambda k: fam[k]["ms_per_step"])
        share = SHARE_BYTES.get(dom, 80.0)
        kname = KERNEL_OF.get(dom, dom)
        bytes_dom = share * n * n * count
        t_dom = fam[dom]["ms_per_step"] * 1e-3
        ach = bytes_dom / t_dom / 1e9 if t_dom > 0 else 0.0
        flops_dom = SHARE_FLOPS.get(dom, 0.0) * n ** 3 * count
        flops_step = ((20.0 / 3.0) * n ** 3 + 40.0 * n * n) * count
        # (families timed on helper streams overlap the main stream and are left out of the sums)
        step_kernel_ms = sum(v["ms_per_step"] for k, v in fam.items() if k not in SIDE_FAMILIES)
        traffic = None
        traffic_note = "no PMC figure for this kernel/workload in profiles/traffic.json"
        try:    # HBM-side bytes per launch of the dominant kernel from the committed PMC passes (tools/profile_bench.sh)
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                tj = json.load(fh).get(kname)
            wl = f"{I}x{N}:{args.hessian_init}"
            if tj and tj.get("workload", "32x100:identity") == wl and world == 1:
                # gfx950: FETCH_SIZE counts wide (16 B per lane) reads at half their bytes (MI355X_MICROARCH.md, HBM section);
                # WRITE_SIZE is exact for 16-byte stores.  Raw counter units are KB.
                traffic = (2.0 * tj["fetch_kb_per_launch"] + tj["write_kb_per_launch"]) * 1024.0
                traffic_note = ("2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE per launch, bytes; separate rocprofv3 --pmc passes "
                                f"of round {tj.get('round')} (profiles/traffic.json); raw: fetch {tj['fetch_kb_per_launch']:.0f} KB, "
                                f"write {tj['write_kb_per_launch']:.0f} KB")
                if tj.get("source_sha1") not in (None, library_stamp()):
                    traffic_note += "; NOTE: kernel sources changed since these passes were taken"
        except (OSError, ValueError, KeyError):
            traffic = None
        roofline = {
            "bound": "hbm", "bound_note": BOUND_NOTE.get(dom, "hbm stream"), "kernel": kname,
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_note": traffic_note,
            "fp64_frac": (flops_dom / t_dom / 1e12 / FP64_PEAK_TFLOPS) if (t_dom > 0 and flops_dom > 0) else None,
            "alg_bytes_per_step_dominant": bytes_dom, "alg_flops_per_step_dominant": flops_dom,
            "kernel_ms_per_step": fam[dom]["ms_per_step"], "launches_per_step": fam[dom]["launches_per_step"],
            # fractions of the whole step are taken over its wall time (ms_per_step of the timed run): the family timers see the
            # main stream only, kernels forked onto the helper streams overlap it and appear in no family
            "whole_step": {"alg_bytes": 80.0 * n * n * count, "alg_flops": flops_step, "kernel_ms": step_kernel_ms, "wall_ms": ms_per_step,
                           "hbm_frac": (80.0 * n * n * count / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_per_step > 0 else 0.0,
                           "fp64_frac": (flops_step / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if ms_per_step > 0 else 0.0},
            "families_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in fam.items()},
            "launches_per_step_all": round(sum(v["launches_per_step"] for v in fam.values()), 1),
            "families_on_helper_streams": sorted(SIDE_FAMILIES),
        }

    # ---- second workload, reported beside the headline line: the same band started from exact LJ Hessians (dense molecular
    # spectra with the six-fold TR/rot zero cluster: the regime the reference enters this path with, neb.py:794-804)
    second = None
    if world == 1 and not args.no_second_workload and args.hessian_init == "identity":
        K2 = min(K, 10)
        ms2, fam2, _, _, _ = measure("exact", W, K2, True)
        if rank == 0:
            second = {"hessian_init": "exact", "steps": K2, "warmup": W, "ms_per_step": ms2,
                      "families_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in fam2.items()}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, outs = cpu_baseline(traj, I, N, args.cpu_budget, hess0)
        # sanity: the oracle fed with the same inputs reproduces the GPU trajectory's next geometries
        dev_max = 0.0
        for k, nx in enumerate(outs):
            if k + 1 < len(traj):
                dev_max = max(dev_max, float(np.abs(nx / cases.BOHR2ANG - traj[k + 1][0]).max()))
        cpu["max_abs_dev_gpu_vs_oracle_bohr"] = dev_max

    if rank == 0:
        out = {
            "metric": "optimizer-step ms (Hessian update+RFO+NEB force, QM excluded)",
            "value": ms_per_step, "unit": "ms/step", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"RFO-NEB band step, {I} images x {N} atoms (n={n}), Ar Lennard-Jones surface, "
                                   "block-Bofill/FSB update + TR/rot projection + eigh + RS-I-RFO + BNEB force + TR/FIRE blend",
                       "images": I, "atoms": N, "images_per_gpu": count, "parallelism": f"image-sharded x{world}",
                       "hessian_init": args.hessian_init,
                       "window": f"band steps {W}..{W + K - 1} of a run started from {args.hessian_init} Hessians"
                                 + (" (identity + low rank: the favourable regime of the eigen-solver; the dense regime is second_workload)"
                                    if args.hessian_init == "identity" else ""),
                       "rccl_ranks": world if world > 1 else 0,
                       "max_abs_dev_sharded_vs_whole_bohr": checks.get(args.hessian_init),
                       "eigen_solver": "reduction route (tridiagonalise + bisection + inverse iteration)" if max(sweeps) == 0
                                       else "one-sided Jacobi",
                       "jacobi_sweeps_last_step": [int(min(sweeps)), int(max(sweeps))]},
            "roofline": roofline, "cpu_baseline": cpu, "second_workload": second,
        }
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
