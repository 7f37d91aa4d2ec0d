"""Image sharding of a NEB band (one image block per GPU) and the exchange a sharded step needs.

Each rank owns a contiguous block of images; their Hessians, RSIRFO state and eigen-solves never move.  Per step the
ranks exchange (reference coupling in brackets):
  * the boundary images' coordinates AND gradients with rank +- 1   [pathopt_bneb_force.py:67-117, rfo_neb.py:63-68,
    trust_radius_neb.py:36-37],
  * all energies                                                     [pathopt_bneb_force.py:6-17,50-55, rfo_neb.py:123],
  * one partial sum per rank of FIRE's band-wide sum v_prev . F      [fire_neb.py:60-65].
`halo_plan` is the single statement of who sends what; the C library (csrc/comm.hip: RCCL or peer copies) and the
torch.distributed rehearsal transport used by the CPU tests (`TorchHalo`) both follow it.

`ShardedBand` drives the shards of ONE process (the reference hands all images to one Python process): one `Band` per
device, exchange and step through the library.  A multi-process job (bench.py under torch.distributed.run) uses one
`Band` per process with `rccl_comm_for_process`.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check


def shard_bounds(n_images, world_size, rank):
    """Contiguous, equal-size blocks: (first, count).  The band must divide evenly (32 -> 16/8/4 per rank)."""
    if n_images % world_size != 0:
        raise ValueError(f"{n_images} images do not divide over {world_size} ranks")
    count = n_images // world_size
    return rank * count, count


def halo_plan(n_images, world_size, rank):
    """What rank `rank` sends and receives per step, as image indices:
    {'send': [(peer, image)], 'recv': [(peer, image)], 'energies': (first, count)}; X and G rows travel for every listed image,
    the energy block is all-gathered, the FIRE partial sum is all-gathered as one scalar per rank."""
    first, count = shard_bounds(n_images, world_size, rank)
    send, recv = [], []
    if rank > 0:
        send.append((rank - 1, first))
        recv.append((rank - 1, first - 1))
    if rank < world_size - 1:
        send.append((rank + 1, first + count - 1))
        recv.append((rank + 1, first + count))
    return {"send": send, "recv": recv, "energies": (first, count)}


class TorchHalo:
    """The same exchange over torch.distributed (gloo on CPU): rehearsal transport for the tests of the host protocol.
    Arrays are whole-band sized (I, n) / (I,) numpy arrays in which only the owned rows are valid on entry."""

    def __init__(self, n_images):
        import torch.distributed as dist
        self.dist = dist
        self.I = n_images
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.plan = halo_plan(n_images, self.world, self.rank)

    def exchange(self, X, G, E):
        import torch
        dist = self.dist
        reqs, bufs = [], []
        for peer, img in self.plan["send"]:
            for arr in (X, G):
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(arr[img])), dst=peer))
        for peer, img in self.plan["recv"]:
            for arr in (X, G):
                t = torch.empty(arr[img].shape, dtype=torch.float64)
                bufs.append((arr, img, t))
                reqs.append(dist.irecv(t, src=peer))
        for r in reqs:
            r.wait()
        for arr, img, t in bufs:
            arr[img] = t.numpy()
        first, count = self.plan["energies"]
        parts = [torch.empty(count, dtype=torch.float64) for _ in range(self.world)]
        dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(E[first:first + count])))
        E[:] = torch.cat(parts).numpy()

    def gather_scalar(self, value):
        """Every rank's partial sum, in rank order (added in that order by every rank)."""
        import torch
        parts = [torch.empty(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(parts, torch.tensor([float(value)], dtype=torch.float64))
        return [float(p.item()) for p in parts]


# ---- communicators of the C library -------------------------------------------------------------------------------
class Comm:
    def __init__(self, handle, rank, nranks):
        self._h, self.rank, self.nranks = handle, rank, nranks

    def close(self):
        if self._h:
            lib().mopt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        # (either order of band / communicator teardown is safe in the library: each side detaches itself from the other)
        try:
            self.close()
        except Exception:
            pass


def _create_group(fn, devices):
    ctxs = [_lib.ctx(d) for d in devices]
    R = len(ctxs)
    arr = (C.c_void_p * R)(*[c.value if isinstance(c, C.c_void_p) else c for c in ctxs])
    outs = (C.c_void_p * R)()
    check(fn(R, arr, outs), ctxs[0])
    return [Comm(C.c_void_p(outs[r]), r, R) for r in range(R)]


def peer_comms(devices):
    """One process, one rank per entry of `devices` (entries may repeat: several shards on one GPU); plain peer copies."""
    return _create_group(lib().mopt_comm_create_peer, devices)


def rccl_comms(devices):
    """One process, one rank per device: ncclCommInitAll over the listed devices (RCCL over xGMI)."""
    if len(set(devices)) != len(devices):
        raise ValueError("RCCL needs distinct devices; use peer_comms for several shards on one GPU")
    return _create_group(lib().mopt_comm_create_rccl_all, devices)


def rccl_comm_for_process(device=None):
    """Multi-process job (one process per GPU, torch.distributed initialised): rank 0 creates the RCCL id, torch.distributed
    broadcasts it, every process creates its communicator."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        rc = lib().mopt_comm_unique_id(buf, 128)
        if rc != 0:
            raise _lib.MoptError(f"mopt_comm_unique_id failed with {rc}")
    t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().tolist())
    idbuf = (C.c_ubyte * 128).from_buffer_copy(raw)
    c = _lib.ctx(device)
    h = C.c_void_p()
    check(lib().mopt_comm_create_rccl_rank(c, world, rank, idbuf, C.byref(h)), c)
    return Comm(h, rank, world)


def _handles(bands):
    return (C.c_void_p * len(bands))(*[b._h.value for b in bands])


def halo_exchange(bands):
    check(lib().mopt_halo_exchange(len(bands), _handles(bands)), bands[0]._ctx)


def step_multi(bands, optimize_num):
    check(lib().mopt_band_step_multi(len(bands), _handles(bands), int(optimize_num)), bands[0]._ctx)


class ShardedBand:
    """All shards of a band in one process: Band-like interface (upload / step / download / set_hessian / get_hessian) on whole-band
    arrays; the per-step neighbour data travel between the devices inside the library."""

    def __init__(self, I, N, devices, transport="rccl", **band_options):
        from .band import Band
        self.I, self.N, self.n = int(I), int(N), 3 * int(N)
        self.devices = list(devices)
        R = len(self.devices)
        if transport == "rccl" and len(set(self.devices)) == R and R > 0:
            self.comms = rccl_comms(self.devices)
        else:
            self.comms = peer_comms(self.devices)
        self.transport = "rccl" if (transport == "rccl" and len(set(self.devices)) == R) else "peer"
        self.bands = []
        for r, dev in enumerate(self.devices):
            first, count = shard_bounds(self.I, R, r)
            b = Band(self.I, self.N, first=first, count=count, device=dev, **band_options)
            b.attach(self.comms[r])
            self.bands.append(b)

    def close(self):
        for b in self.bands:
            b.close()
        for c in self.comms:
            c.close()
        self.bands, self.comms = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def failures(self):
        bits = 0
        for b in self.bands:
            bits |= b.failures()
        return bits

    def _owner(self, image):
        for b in self.bands:
            if b.first <= image < b.first + b.count:
                return b
        raise IndexError(image)

    def set_hessian(self, image, H):
        self._owner(image).set_hessian(image, H)

    def add_hessian(self, image, B):
        self._owner(image).add_hessian(image, B)

    def get_hessian(self, image):
        return self._owner(image).get_hessian(image)

    def get_state(self, image):
        return self._owner(image).get_state(image)

    def upload(self, X, E, G):
        X = np.asarray(X, dtype=np.float64).reshape(self.I, self.n)
        G = np.asarray(G, dtype=np.float64).reshape(self.I, self.n)
        E = np.asarray(E, dtype=np.float64).reshape(self.I)
        for b in self.bands:
            s = slice(b.first, b.first + b.count)
            b.upload_shard(X[s], E[s], G[s])
        halo_exchange(self.bands)

    def model_hessian(self, element_list, approx_hess_type="fischerd3old", coords=None, gradients=None):
        for b in self.bands:
            b.model_hessian(element_list, approx_hess_type, coords=coords, gradients=gradients)      # (whole-band arrays: every shard takes its own images)

    def step(self, optimize_num):
        step_multi(self.bands, optimize_num)

    def sync(self):
        for b in self.bands:
            b.sync()

    def download(self, want_force=False):
        outs = [b.download(want_force=want_force) for b in self.bands]
        if not want_force:
            return np.concatenate(outs, axis=0)
        newX = np.concatenate([o[0] for o in outs], axis=0)
        F = np.concatenate([o[1][b.first:b.first + b.count] for o, b in zip(outs, self.bands)], axis=0)
        return newX, F


def gather_band(local_new, n_images):
    """All-gather the per-rank blocks (count, N, 3) into the whole band (I, N, 3), in image order (torch.distributed; what a
    driver needs before a band-wide host step such as the Kabsch chain, neb.py:993-996)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_new
    if dist.get_backend() == "gloo":
        # CPU rehearsal path (tests, single-GPU boxes): gloo has no all_gather_into_tensor for device tensors
        src = local_new.detach().cpu().contiguous()
        parts = [torch.empty_like(src) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, src)
        return torch.cat(parts, dim=0).to(local_new.device)
    out = torch.empty((n_images,) + tuple(local_new.shape[1:]), dtype=local_new.dtype, device=local_new.device)
    dist.all_gather_into_tensor(out, local_new.contiguous())
    return out
