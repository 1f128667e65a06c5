"""Multi-GPU plumbing: one process per GPU, pixel-tile sharding, ONE collective per render — the sum-reduce of the
accumulation buffers to rank 0 (reduce_framebuffer: north_star's wording, bench.py's default) or the gather of every
rank's own tiles (gather_tiles: the same image from 1/N of the bytes) — RCCL over xGMI through torch.distributed's
"nccl" backend; "gloo" on CPU for tests.

The image is embarrassingly parallel per pixel (main.wgsl:4,16: a pixel depends only on its index, the
frame number, the scene and the view), so ranks never exchange anything while rendering.  Each rank
accumulates into a full-size, zero-initialised framebuffer and touches only its own tiles; the reduce
adds x + 0 + ... + 0, so the N-GPU image is bit-identical to the 1-GPU image.
"""
import os

TILE_PIXELS = 4032  # contiguous pixel runs per shard tile (about two 1080p rows, 63 waves): long enough for coherent
                    # primary-ray waves, short enough (515 tiles at 1080p) that every rank samples every
                    # image region and the per-rank work is balanced.  NOT 4096: a rank's pixel count is the
                    # stride between the frames of a batch in every per-path array, and 64 tiles x 4096 = 2^18
                    # pixels (rank 0 of 8 at 1080p) puts the same pixel of the frames in flight into the same HBM
                    # channels: k_shade +12 %, one rank of eight 7 % slower (profiles/r04_shard_tile.txt)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them)."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def owned_pixel_mask(npix, rank, world, tile=TILE_PIXELS):
    """Boolean mask of the pixels ptmi_set_shard(rank, world, tile) renders."""
    import numpy as np

    p = np.arange(npix, dtype=np.int64)
    return ((p // tile) % world) == rank


def reduce_framebuffer(fb_tensor, dst=0):
    """The single collective of a render: sum every rank's framebuffer into rank `dst`."""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(fb_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return fb_tensor


_tile_index_cache = {}


def owned_pixel_index(npix, rank, world, tile, device):
    """int64 tensor of the pixels `rank` owns, in image order (cached per shape and device)."""
    import torch

    key = (npix, rank, world, tile, str(device))
    if key not in _tile_index_cache:
        p = torch.arange(npix, dtype=torch.int64, device=device)
        _tile_index_cache[key] = p[((p // tile) % world) == rank].contiguous()
    return _tile_index_cache[key]


def gather_tiles(fb_tensor, tile=TILE_PIXELS, dst=0):
    """The same collective with 1/N of the bytes ("or equivalently an all-gather on a tile-major layout"): every pixel is non-zero on
    exactly one rank, so instead of summing N full framebuffers each rank packs the tiles it owns (npix/N pixels) and rank `dst` gathers the packs and
    puts them in place — no arithmetic at all, the N-GPU image is the 1-GPU image bit for bit.  torch.distributed.gather = grouped send / recv over
    RCCL (xGMI point to point: the N-1 packs arrive over N-1 links at once)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return fb_tensor
    rank, world = dist.get_rank(), dist.get_world_size()
    fb = fb_tensor.view(-1, 4)
    npix = fb.shape[0]
    n_max = max(int(owned_pixel_index(npix, r, world, tile, fb.device).numel()) for r in range(world))  # packs are padded to one length
    mine = owned_pixel_index(npix, rank, world, tile, fb.device)
    pack = torch.zeros((n_max, 4), dtype=fb.dtype, device=fb.device)
    pack[: mine.numel()] = fb[mine]
    if rank == dst:
        parts = [torch.empty_like(pack) for _ in range(world)]
        dist.gather(pack, parts, dst=dst)
        for r in range(world):
            if r != dst:
                idx = owned_pixel_index(npix, r, world, tile, fb.device)
                fb[idx] = parts[r][: idx.numel()]
    else:
        dist.gather(pack, None, dst=dst)
    return fb_tensor


def barrier():
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_scalar(value, op="sum"):
    import torch
    import torch.distributed as dist

    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())
