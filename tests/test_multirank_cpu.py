"""world_size-2 gloo tests of the N > 1 path's host protocol (no GPU): shard bounds, the exchange plan, and a sharded band stepped
end to end -- every rank feeds ONLY its own images' energies / gradients, the neighbour rows, all energies and FIRE's partial sums
travel by the plan of multioptpy_amd.parallel.halo_plan (the plan csrc/comm.hip executes with RCCL or peer copies), everything a rank
did not own or receive is filled with garbage, and the oracle's band step on that view must reproduce the whole-band step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, I, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multioptpy_amd.parallel import shard_bounds, gather_band
    first, count = shard_bounds(I, world, rank)
    rng = np.random.default_rng(0)
    full = rng.normal(size=(I, N, 3))                 # what a whole-band step would produce
    mine = torch.from_numpy(full[first:first + count].copy())
    got = gather_band(mine, I)
    ok = bool(np.array_equal(got.numpy(), full))
    q.put((rank, first, count, ok))
    dist.destroy_process_group()


def test_shard_bounds():
    from multioptpy_amd.parallel import shard_bounds
    assert [shard_bounds(32, 8, r) for r in range(8)] == [(4 * r, 4) for r in range(8)]
    assert shard_bounds(32, 1, 0) == (0, 32)
    with pytest.raises(ValueError):
        shard_bounds(30, 4, 0)


def test_halo_plan_is_consistent():
    """Every send has its matching receive on the peer, the boundary images are exactly the neighbours a rank lacks."""
    from multioptpy_amd.parallel import halo_plan, shard_bounds
    for I, W in ((8, 2), (32, 8), (64, 4), (6, 1)):
        plans = [halo_plan(I, W, r) for r in range(W)]
        for r, p in enumerate(plans):
            first, count = shard_bounds(I, W, r)
            want = [i for i in (first - 1, first + count) if 0 <= i < I]
            assert sorted(img for _, img in p["recv"]) == want
            for peer, img in p["recv"]:
                assert (r, img) in plans[peer]["send"]
            for peer, img in p["send"]:
                assert first <= img < first + count and (r, img) in plans[peer]["recv"]


def _spawn(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_gather_band_world2_gloo():
    assert _spawn(_worker, 2, 8, 5) == [(0, 0, 4, True), (1, 4, 4, True)]


def _sharded_worker(rank, world, port, I, N, nsteps, q):
    import contextlib
    import io
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    from multioptpy_amd.parallel import shard_bounds, TorchHalo
    from oracle import neb as oneb
    first, count = shard_bounds(I, world, rank)
    own = slice(first, first + count)
    halo = TorchHalo(I)
    garbage = np.random.default_rng(1000 + rank)
    # reference: the whole band on one "device"
    ref_opt, ref_calc = oneb.RFONEB([None] * I), oneb.BNEB()
    my_opt, my_calc = oneb.RFONEB([None] * I), oneb.BNEB()
    X = cases.band(I, N)
    preX = preG = None
    pre_vel_ref, pre_vel_own = [], None
    worst = 0.0
    for it in range(nsteps):
        E = np.zeros(I)
        G = np.zeros((I, N, 3))
        for i in range(I):
            E[i], G[i] = cases.lj(X[i])
        with contextlib.redirect_stdout(io.StringIO()):
            Fref = ref_calc.calc_force(X, E, G, it)
            ref_new = ref_opt.optimize(X, G, preX, preG, it, E, pre_vel_ref, Fref.copy(), ref_calc)
        # ---- this rank: only its own images' calculator output, everything else garbage until the exchange delivers it
        Xv = garbage.normal(size=(I, 3 * N)) * 50.0
        Gv = garbage.normal(size=(I, 3 * N)) * 50.0
        Ev = garbage.normal(size=I) * 50.0
        Xv[own], Gv[own], Ev[own] = X[own].reshape(count, -1), G[own].reshape(count, -1), E[own]
        halo.exchange(Xv, Gv, Ev)
        Xs, Gs = Xv.reshape(I, N, 3), Gv.reshape(I, N, 3)
        pXs = pGs = None
        if it > 0:
            pXs = garbage.normal(size=(I, N, 3)) * 50.0
            pGs = garbage.normal(size=(I, N, 3)) * 50.0
            pXs[own], pGs[own] = preX[own], preG[own]
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            Fs = my_calc.calc_force(Xs, Ev, Gs, it)
            part = 0.0
            pv = []
            if it > 0:
                part = float(np.sum(pre_vel_own * Fs[own]))
                pv = garbage.normal(size=(I, N, 3)) * 50.0
                pv[own] = pre_vel_own
            parts = halo.gather_scalar(part)
            fire_dot = 0.0
            for p in parts:
                fire_dot += p
            new = my_opt.optimize(Xs, Gs, pXs, pGs, it, Ev, pv, Fs.copy(), my_calc, fire_dot=fire_dot if it > 0 else None)
        worst = max(worst, float(np.abs(np.asarray(new)[own] - np.asarray(ref_new)[own]).max()))
        worst = max(worst, float(np.abs(Fs[own] - Fref[own]).max()))
        pre_vel_own = Fs[own].copy()
        preX, preG, pre_vel_ref = X.copy(), G.copy(), Fref.copy()
        X = np.asarray(ref_new) / cases.BOHR2ANG
    q.put((rank, worst))
    dist.destroy_process_group()


def test_sharded_band_protocol_world2_gloo():
    """Four band steps of an 8-image band over two ranks: own images + exchanged rows suffice (garbage everywhere else)."""
    res = _spawn(_sharded_worker, 2, 8, 7, 4)
    for rank, worst in res:
        assert worst < 1e-12, (rank, worst)


def test_sharded_band_protocol_world4_gloo():
    res = _spawn(_sharded_worker, 4, 8, 5, 3)
    for rank, worst in res:
        assert worst < 1e-12, (rank, worst)
