"""N > 1 path on CPU: world_size 2 over gloo.  Each rank renders its pixel tiles (the oracle stands in for
the GPU renderer — same shard rule as ptmi_set_shard), the framebuffers are sum-reduced to rank 0 with
the product's dist helpers, and the result must equal the single-process image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_pkg

W, H, FRAMES, TILE = 64, 48, 2, 256


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch

    pkg = load_pkg()
    from oracle import ptm_oracle
    from webgpu_path_tracer_amd import dist as pdist

    r, w, _ = pdist.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    b = pkg.scenes.golden_buffers("c2m")
    view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS["cornell"])
    fb, st = ptm_oracle.render(b, W, H, view, 1, FRAMES, max_bounces=5, shard=(rank, world, TILE), threads=2)
    mask = pdist.owned_pixel_mask(W * H, rank, world, TILE)
    assert np.array_equal(fb.reshape(-1, 4)[:, 3] == 1.0, mask)  # touched exactly the owned pixels
    t = torch.from_numpy(fb.reshape(-1).copy())
    t2 = t.clone()
    pdist.barrier()
    pdist.reduce_framebuffer(t, 0)
    pdist.gather_tiles(t2, TILE, 0)  # the same collective from 1/N of the bytes: rank 0 must end up with the very same image
    if rank == 0:
        assert torch.equal(t.view(torch.int32), t2.view(torch.int32))
    rays = pdist.all_reduce_scalar(st["rays"], "sum")
    slowest = pdist.all_reduce_scalar(float(rank + 1), "max")
    assert slowest == float(world)
    if rank == 0:
        np.save(os.path.join(outdir, "reduced.npy"), t.numpy().reshape(H, W, 4))
        np.save(os.path.join(outdir, "rays.npy"), np.array([rays]))
    pdist.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_tiles_reduce_to_the_full_image(tmp_path, pkg, oracle):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "reduced.npy")
    b = pkg.scenes.golden_buffers("c2m")
    view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS["cornell"])
    want, st = oracle.render(b, W, H, view, 1, FRAMES, max_bounces=5)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert int(np.load(tmp_path / "rays.npy")[0]) == st["rays"]


def test_owned_pixel_masks_partition_the_image(pkg):
    from webgpu_path_tracer_amd import dist as pdist

    for npix, world, tile in ((1920 * 1080, 8, pdist.TILE_PIXELS), (1000, 3, 64), (64, 4, 64), (5, 2, 1)):
        masks = [pdist.owned_pixel_mask(npix, r, world, tile) for r in range(world)]
        assert (np.sum(masks, axis=0) == 1).all()
    # load balance of the bench's tile size at 1080p over 8 ranks: every rank within 7% of the mean
    counts = [pdist.owned_pixel_mask(1920 * 1080, r, 8).sum() for r in range(8)]
    assert max(counts) / (1920 * 1080 / 8) < 1.07
