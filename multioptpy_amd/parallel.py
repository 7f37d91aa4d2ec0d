"""Image sharding of a NEB band over ranks (one process per GPU) and the one exchange the path needs.

Each rank owns a contiguous block of images (their Hessians, RSIRFO state and eigen-solves never move).  The only
data that crosses ranks per step is the new geometry of every image (I x N x 3 doubles, 77 KB at 32 x 100), gathered
before the next energy/gradient evaluation; torch.distributed provides the transport (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(n_images, world_size, rank):
    """Contiguous, equal-size blocks: (first, count).  The band must divide evenly (32 -> 16/8/4 per rank)."""
    if n_images % world_size != 0:
        raise ValueError(f"{n_images} images do not divide over {world_size} ranks")
    count = n_images // world_size
    return rank * count, count


def gather_band(local_new, n_images):
    """All-gather the per-rank blocks (count, N, 3) into the whole band (I, N, 3), in image order."""
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
