#!/usr/bin/env python
"""bench.py -- optimizer-step wall time of the RFO-NEB band step on MI355X.

Metric (BASELINE.json): optimizer-step ms (Hessian update + RFO solve + NEB force, QM call excluded) on a
32-image x 100-atom NEB band, at 1/2/4/8 GPUs (images sharded over the ranks, one process per GPU).

One "step" = RFOOptimizer.optimize of the reference (multioptpy/Optimizer/rfo_neb.py:104-208) for the whole band:
BNEB force, Ayala curvature correction, one RSIRFO.run per image (block-Bofill/FSB update, TR/rot projections,
symmetric eigen-solve, RS-I-RFO step), step limit, TR_calc, FIRE, blend.  Inputs (geometries, energies, gradients of
an argon LJ surface, synthetic) for every step are staged in HBM before the timed region; for N > 1 the timed
region also contains the RCCL all-gather of the new geometries (the exchange a real run needs before the next
energy/gradient evaluation).

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


def make_trajectory(I, N, nsteps, device):
    """Untimed: run the band optimizer itself for nsteps to obtain realistic per-step inputs (X, E, G)."""
    import cases
    from multioptpy_amd.band import Band
    band = Band(I, N, device=device)
    X = cases.band(I, N)
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


def cpu_baseline(traj, I, N, budget_s=25.0):
    """The oracle (NumPy restatement of the reference algorithm incl. its SVD-based B-matrix projection) timed on
    this host's cores on the first steps of the same trajectory."""
    import contextlib
    import io
    from oracle import neb as oneb
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        thr = os.cpu_count() or 1
    opt = oneb.RFONEB([None] * I)
    calc = oneb.BNEB(literal=True)
    preX = preG = None
    pre_vel = []
    times = []
    outs = []
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
        if it >= 1 and (time.perf_counter() - t_begin) + dt > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    return {
        "value": 1e3 * float(np.median(timed)), "unit": "ms/step", "cores": int(thr), "kind": "port",
        "sample": f"steps 1..{len(times) - 1} of the same {I}x{N} band trajectory (step 0 = warm-up), median; "
                  f"oracle/neb.py RFONEB with the reference's SVD-based BNEB projection, NumPy/OpenBLAS {thr} threads",
    }, outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--atoms", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=25.0)
    args = ap.parse_args()

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
    from multioptpy_amd.parallel import shard_bounds, gather_band
    first, count = shard_bounds(I, world, rank)

    # ---- inputs for every step, staged in HBM (untimed)
    traj = make_trajectory(I, N, W + K, local)
    dev = torch.device("cuda", local)
    staged = [(torch.from_numpy(X).to(dev), torch.from_numpy(E).to(dev), torch.from_numpy(G).to(dev)) for X, E, G in traj]
    torch.cuda.synchronize()

    def run(band, timing=False):
        newx_t = torch.empty((count, N, 3), dtype=torch.float64, device=dev)
        for k in range(W):
            dX, dE, dG = staged[k]
            band.upload_dev(dX.data_ptr(), dE.data_ptr(), dG.data_ptr(), traj[k][1])
            band.step(k)
        band.sync()
        if timing:
            band.timing(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(W, W + K):
            dX, dE, dG = staged[k]
            band.upload_dev(dX.data_ptr(), dE.data_ptr(), dG.data_ptr(), None)
            band.step(k)
            if world > 1:
                band.download_dev(newx_t.data_ptr())
                band.sync()
                gather_band(newx_t, I)
        band.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0

    band = Band(I, N, first=first, count=count, device=local)
    elapsed = run(band)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / K
    sweeps = [band.get_state(first + j).jacobi_sweeps for j in range(count)]
    band.close()

    roofline = None
    fam = {}
    # ---- roofline leg: the same K steps again on a fresh band with HIP-event timing per kernel family
    # (every rank runs it -- run() contains barriers -- but only rank 0 records and reports)
    b2 = Band(I, N, first=first, count=count, device=local)
    run(b2, timing=(rank == 0))
    if rank == 0:
        for name in ("hess_update", "project", "eigh_init", "eigh_jacobi", "eigh_tridiag", "eigh_bisect", "eigh_invit", "eigh_backtrans",
                     "eigh_final", "rfo", "neb_force", "neb_ayala", "neb_post"):
            ms, nl = b2.timing_read(name)
            fam[name] = {"ms_per_step": ms / K, "launches_per_step": nl / K}
        b2.timing(False)
        dom = max(fam, key=lambda k: fam[k]["ms_per_step"])
        # algorithmic bytes: 80 n^2 per image-step for the whole RSIRFO.run, of which the eigen-solve's share
        # (input read + eigenvector write) is 16 n^2; flops_alg = (20/3) n^3 + 40 n^2 per image-step.
        # reduction route: the tridiagonalisation reads the matrix once (8 n^2) and leaves the reflectors in its strict upper
        # triangle (4 n^2); the back-transformation reads reflectors + eigenvectors and writes eigenvectors (20 n^2).
        share = {"eigh_jacobi": 16.0, "eigh_tridiag": 12.0, "eigh_backtrans": 20.0, "eigh_invit": 8.0, "hess_update": 24.0,
                 "project": 16.0, "rfo": 24.0}.get(dom, 80.0)
        kname = {"eigh_jacobi": "k_jac_step", "eigh_tridiag": "k_tri_persist3", "eigh_backtrans": "k_tri_backtrans",
                 "eigh_invit": "k_tri_invit", "eigh_bisect": "k_tri_bisect"}.get(dom, dom)
        bytes_dom = share * n * n * count
        t_dom = fam[dom]["ms_per_step"] * 1e-3
        ach = bytes_dom / t_dom / 1e9 if t_dom > 0 else 0.0
        flops_step = ((20.0 / 3.0) * n ** 3 + 40.0 * n * n) * count
        step_kernel_ms = sum(v["ms_per_step"] for v in fam.values())
        traffic = None
        try:    # HBM-side bytes per launch of the dominant kernel from the committed PMC passes (tools/profile_bench.sh)
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")) as fh:
                tj = json.load(fh).get(kname)
            if tj and I == 32 and N == 100 and world == 1:
                traffic = (tj["fetch_kb_per_launch"] + tj["write_kb_per_launch"]) * 1024.0
        except (OSError, ValueError, KeyError):
            traffic = None
        roofline = {
            "bound": "hbm", "kernel": kname,
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_note": "FETCH_SIZE + WRITE_SIZE per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), bytes",
            "alg_bytes_per_step_dominant": bytes_dom, "kernel_ms_per_step": fam[dom]["ms_per_step"],
            "launches_per_step": fam[dom]["launches_per_step"],
            # fractions of the whole step are taken over its wall time (ms_per_step of the timed run): the family timers see the
            # main stream only, kernels forked onto the helper streams overlap it and appear in no family
            "whole_step": {"alg_bytes": 80.0 * n * n * count, "alg_flops": flops_step, "kernel_ms": step_kernel_ms, "wall_ms": ms_per_step,
                           "hbm_frac": (80.0 * n * n * count / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_per_step > 0 else 0.0,
                           "fp64_frac": (flops_step / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if ms_per_step > 0 else 0.0},
            "families_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in fam.items()},
        }

    b2.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, outs = cpu_baseline(traj, I, N, args.cpu_budget)
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
                       "eigen_solver": "reduction route (tridiagonalise + bisection + inverse iteration)" if max(sweeps) == 0
                                       else "one-sided Jacobi",
                       "jacobi_sweeps_last_step": [int(min(sweeps)), int(max(sweeps))]},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
