"""Image-plane sharding across ranks + the final gather.

Tile t (row-major 16x16 tiles, PathTraceIntegrator.cpp:27-42) goes to rank t % world.  Every rank
renders its own shard with no data-path collective; one gather of the per-rank framebuffers to
rank 0 ends the render (RCCL over xGMI when the tensors live on the GPU, gloo on CPU tensors).
Shards differ in size by at most one tile, so they are padded to the common maximum for the
collective.
"""
import numpy as np

from ._host import Shard


def shard_sizes(res_x, res_y, world, seed=0xBA5EBA11):
    return [Shard(res_x, res_y, seed, r, world).n_pixels for r in range(world)]


def gather_framebuffer(local_rgb, shard, world, rank, dist=None, device=None):
    """Gather shard-ordered rgb tensors (torch, on `device`) to rank 0 and assemble the image.

    local_rgb: torch tensor (n_local, 3) float32.  Returns an (H, W, 3) numpy image on rank 0,
    None elsewhere.  With world == 1 and no process group no collective is issued (a one-rank group still goes through the
    collective: the RCCL rehearsal of a one-GPU box).
    """
    import torch

    res_x, res_y = shard.res_x, shard.res_y
    if world == 1 and dist is None:
        return shard.scatter(local_rgb.detach().cpu().numpy())
    sizes = shard_sizes(res_x, res_y, world)
    n_max = max(sizes)
    # RCCL gathers device tensors in place; gloo (CPU tests, rehearsals) works on host copies
    where = local_rgb.device if dist.get_backend() == "nccl" else torch.device("cpu")
    padded = torch.zeros((n_max, 3), dtype=torch.float32, device=where)
    padded[: local_rgb.shape[0]] = local_rgb.to(where)
    gathered = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, gather_list=gathered, dst=0)
    if rank != 0:
        return None
    image = np.zeros((res_y*res_x, 3), np.float32)
    for r in range(world):
        sh = Shard(res_x, res_y, 0xBA5EBA11, r, world)
        image[sh.pixel_index] = gathered[r][: sizes[r]].cpu().numpy()
    return image.reshape(res_y, res_x, 3)


_INDEX_CACHE = {}


def _shard_indices(res_x, res_y, total, device):
    """Pixel indices (torch int64 on `device`) and sizes of all `total` shards of an image: computed once per (image, split, device) --
    dicing the image into tiles is host work that does not belong between two renders."""
    import torch

    key = (res_x, res_y, total, str(device))
    if key not in _INDEX_CACHE:
        idx = [torch.from_numpy(Shard(res_x, res_y, 0xBA5EBA11, i, total).pixel_index.astype(np.int64)).to(device) for i in range(total)]
        _INDEX_CACHE[key] = (idx, [int(t.numel()) for t in idx])
    return _INDEX_CACHE[key]


def gather_framebuffers(local_rgbs, shards, world, rank, sub, dist=None):
    """Like gather_framebuffer for `sub` image-plane shards per rank (one context each: two pools on separate streams
    fill each other's launch gaps and kernel tails).  Shard j of rank r is shard r*sub + j of world*sub.
    The shards are put into image order where they live (on the GPU with RCCL: one indexed copy per shard), then the image
    goes to the host once.  Returns the (H, W, 3) image on rank 0, None elsewhere."""
    import torch

    res_x, res_y = shards[0].res_x, shards[0].res_y
    nccl = dist is not None and dist.get_backend() == "nccl"
    where = local_rgbs[0].device if (dist is None or nccl) else torch.device("cpu")
    idx, sizes = _shard_indices(res_x, res_y, world*sub, where)
    image = torch.zeros((res_y*res_x, 3), dtype=torch.float32, device=where) if rank == 0 else None
    for j in range(sub):
        local = local_rgbs[j]
        if dist is None: # one rank, no process group (a one-rank group takes the collective below: tests/test_gpu_dist.py)
            image[idx[j]] = local.detach()
            continue
        n_max = max(sizes[r*sub + j] for r in range(world))
        padded = torch.zeros((n_max, 3), dtype=torch.float32, device=where)
        padded[: local.shape[0]] = local.to(where)
        gathered = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        dist.gather(padded, gather_list=gathered, dst=0)
        if rank == 0:
            for r in range(world):
                image[idx[r*sub + j]] = gathered[r][: sizes[r*sub + j]]
    if rank != 0:
        return None
    if image.device.type == "cpu":
        return image.numpy().reshape(res_y, res_x, 3)
    # one device-to-host copy of the image into a page-locked buffer kept between renders (a pageable destination costs several ms for these
    # 11 MB -- a tenth of an 8-rank render); the caller gets a view of that buffer, valid until the next gather of the same image size
    key = ("host", res_x, res_y)
    if key not in _INDEX_CACHE:
        _INDEX_CACHE[key] = torch.empty((res_y*res_x, 3), dtype=torch.float32, pin_memory=True)
    host = _INDEX_CACHE[key]
    host.copy_(image)
    return host.numpy().reshape(res_y, res_x, 3)


def exchange_records(table, dist, device=None):
    """Adaptive sampling's one exchange step: every rank needs the SampleRecord statistics of ALL variance
    tiles to compute the next pass' sample distribution (PathTraceIntegrator.cpp:44-108 reads the whole
    table).  Each 4x4 variance tile lies in exactly one rank's shard and the others hold zeros for it, so a
    SUM all-reduce reassembles the table exactly (x + 0 is exact in float32; counts are integers).
    12 bytes per variance tile (691 KB at 1280x720): latency-bound, one collective per pass.

    table: numpy array of _host.RECORD_DTYPE (whole image); returns the merged table.
    """
    import torch

    from ._host import RECORD_DTYPE
    table = np.ascontiguousarray(table, RECORD_DTYPE)
    counts = torch.from_numpy(table["sample_count"].astype(np.int32))
    stats = torch.from_numpy(np.stack([table["mean"], table["running_variance"]], axis=1).copy())
    if device is not None:
        counts, stats = counts.to(device), stats.to(device)
    dist.all_reduce(counts)
    dist.all_reduce(stats)
    out = np.zeros(table.size, RECORD_DTYPE)
    out["sample_count"] = counts.cpu().numpy().astype(np.uint32)
    stats = stats.cpu().numpy()
    out["mean"], out["running_variance"] = stats[:, 0], stats[:, 1]
    return out
