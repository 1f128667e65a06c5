#!/usr/bin/env python3
"""Does the stride between the frames of a batch (= the pixels a context owns) decide how fast the wavefront kernels run?
configs[1]'s scene at image sizes whose pixel count is / is not a large power of two, a fresh context per repetition with PTMI_PLACEMENT_TRIES=1
(no placement search): Mrays/s and per-kernel ms per step.  GPU box: python tools/stride_probe.py [WxH ...] -> stdout + gpurun_out/stride_probe.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PTMI_PLACEMENT_TRIES", "1")
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402

pkg = entry._load_pkg()
sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(1920, 1080), (2048, 1024), (2048, 1023), (2048, 1025), (1024, 1024), (1024, 1023)]
reps = int(os.environ.get("REPS", "3"))
shard = os.environ.get("SHARD")  # "world:tile": rank 0 of `world` with tiles of `tile` pixels, spp x world (one rank of a weak-scaled run: profiles/r04_shard_tile.txt)
out = []
for W, H in sizes:
    class A:
        width, height, bounces, bvh, tris, stack_size, frames_in_flight = W, H, 8, "median", 0, 0, 0
    wl = bench.make_workload(pkg, os.environ.get("WL", "c2"), A)
    for rep in range(reps):
        ctx = bench.make_context(pkg, wl, 0, A)
        spp = 64
        if shard:
            world, tile = (int(x) for x in shard.split(":"))
            ctx.set_shard(0, world, tile)
            spp = 64 * world
        ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize()
        ctx.reset_stats(); ctx.set_timing(1)
        ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize()
        split = ctx.stats(); ctx.set_timing(0); ctx.reset_stats()
        t = time.perf_counter()
        n = 5
        for _ in range(n):
            ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize()
        dt = (time.perf_counter() - t) / n
        st = ctx.stats()
        row = {"W": W, "H": H, "npix": W * H, "rep": rep, "mrays_per_s": st["rays"] / n / dt / 1e6, "ms_per_step": dt * 1e3,
               **{k: split[k] for k in ("generate_ms", "bvh_ms", "shade_ms", "tail_ms", "accumulate_ms")}, "placement_sets": st["placement_sets"]}
        out.append(row)
        print("%4dx%-4d npix %8d = 2^%.3f rep %d: %7.0f Mrays/s %7.2f ms/step | gen %.2f bvh %.2f shade %.2f tail %.2f acc %.2f" % (
            W, H, W * H, __import__("math").log2(W * H), rep, row["mrays_per_s"], row["ms_per_step"], row["generate_ms"], row["bvh_ms"], row["shade_ms"], row["tail_ms"], row["accumulate_ms"]), flush=True)
        ctx.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stride_probe.json"), "w"), indent=1)
