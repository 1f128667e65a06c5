#!/usr/bin/env python3
"""Per-kernel times of ONE rank of an N-GPU render (rank 0 of N = 1, 4, 8; weak scaling: spp x N, or MODE=strong: the fixed total), a fresh context per N as a real
rank has it (the placement search then sees the rank's own access pattern).  GPU box: [MODE=strong] [WORLDS=1,8] [WL=c2] python tools/shard_split.py [reuse] -> stdout"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402

pkg = entry._load_pkg()
reuse = "reuse" in sys.argv[1:]


class A:
    width, height, bounces, bvh, tris, stack_size, frames_in_flight = 1920, 1080, 8, "median", 0, 0, 0


TILE = int(os.environ.get("TILE", "4032"))
WORLDS = [int(x) for x in os.environ.get("WORLDS", "1,4,8").split(",")]
for w in os.environ.get("WL", "c2,c3").split(","):
    wl = bench.make_workload(pkg, w, A)
    ctx = bench.make_context(pkg, wl, 0, A) if reuse else None
    for world in WORLDS:
        if not reuse:
            ctx = bench.make_context(pkg, wl, 0, A)
        ctx.set_shard(0, world, TILE)
        spp = bench.SPP[w] * (1 if os.environ.get("MODE") == "strong" else world)
        ctx.set_timing(1)
        ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize(); ctx.reset_stats()
        ctx.clear(); t = time.perf_counter(); ctx.render(wl["view"], 1, spp); ctx.synchronize(); dt = time.perf_counter() - t
        st = ctx.stats()
        print(w, "world", world, "spp", spp, "wall %.2f ms" % (dt * 1e3), {k: round(st[k], 2) for k in ("generate_ms", "bvh_ms", "shade_ms", "tail_ms", "accumulate_ms")},
              "launches", st["generate_launches"], st["intersect_launches"], st["tail_launches"], st["accumulate_launches"], "placement sets", st["placement_sets"], "(%.0f ms)" % st["placement_ms"], flush=True)
        if not reuse:
            ctx.close()
    if reuse:
        ctx.close()
