"""Multi-GPU plumbing: one process per GPU, pixel-tile sharding, ONE sum-reduce of the accumulation
buffer to rank 0 (RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" on CPU for tests).

The image is embarrassingly parallel per pixel (main.wgsl:4,16: a pixel depends only on its index, the
frame number, the scene and the view), so ranks never exchange anything while rendering.  Each rank
accumulates into a full-size, zero-initialised framebuffer and touches only its own tiles; the reduce
adds x + 0 + ... + 0, so the N-GPU image is bit-identical to the 1-GPU image.
"""
import os

TILE_PIXELS = 4096  # contiguous pixel runs per shard tile (about two 1080p rows): long enough for coherent
                    # primary-ray waves, short enough (506 tiles at 1080p) that every rank samples every
                    # image region and the per-rank work is balanced


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
