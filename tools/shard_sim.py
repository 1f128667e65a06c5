#!/usr/bin/env python3
"""One rank of an N-GPU run, timed on ONE GPU: what each rank of bench.py --gpus N would do, without the other ranks.

Every rank of an N-GPU render traces the pixel tiles t with t % N == rank (4032-pixel tiles, ptmi_set_shard) — here rank 0 of N for N = 1, 2, 4, 8 —
in both scaling modes of bench.py: `weak` (spp x N: every rank keeps the rays of the 1-GPU run) and `strong` (fixed total spp: every rank traces 1/N of the
rays).  The step's collective cannot run on one GPU; its size is stated instead (bytes landing on the root per step, for the full-buffer reduce and for
the gather of owned tiles), with the time they need at a stated per-link xGMI rate, so that the projected N-GPU figure = N x rays / (rank time + collective).

usage (GPU box): tools/shard_sim.py [c2 c3 c4 c5]  -> gpurun_out/shard_sim.json (copied to profiles/rNN_shard_sim.json).  c4 / c5 = configs[3] / configs[4], the two
configurations BASELINE.json assigns to 8 GPUs (fixed totals: 512 spp at 1080p, 1024 spp at 4K): N = 1 and rank 0 of 8, strong.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402

XGMI_LINK_GBS = 48.0       # achieved per-direction rate of one xGMI link for large transfers (64 GB/s peak per direction, MI355X_MICROARCH.md: 7 links x 153 GB/s bidirectional per GPU)
RCCL_REDUCE_BUSBW_GBS = 100.0  # assumed bus bandwidth of ncclReduce for a 33-133 MB message over 8 GPUs (ring, several channels); a guess until SCALE runs — stated, not measured
COLLECTIVE_LATENCY_MS = 0.05


def main():
    pkg = entry._load_pkg()
    workloads = sys.argv[1:] or ["c2", "c3"]
    out = {"note": __doc__.strip().splitlines()[0], "xgmi_link_gbs_assumed": XGMI_LINK_GBS, "rccl_reduce_busbw_gbs_assumed": RCCL_REDUCE_BUSBW_GBS,
           "collective_latency_ms_assumed": COLLECTIVE_LATENCY_MS, "tile_pixels": 4032, "workloads": {}}
    for w in workloads:
        heavy = w in ("c4", "c5")  # seconds per step on one GPU: N = 1 and N = 8 strong only, fewer repetitions

        class A:
            width, height = (3840, 2160) if w == "c5" else (1920, 1080)
            bounces, bvh, tris, stack_size, frames_in_flight = 8, "median", 0, 0, 0
        wl = bench.make_workload(pkg, w, A)
        ctx = bench.make_context(pkg, wl, 0, A)
        base_spp = bench.SPP[w]
        fb_bytes = A.width * A.height * 16
        rows = []
        for world in ((1, 8) if heavy else (1, 2, 4, 8)):
            ctx.set_shard(0, world, 4032)
            for mode in ("weak", "strong"):
                if (world == 1 and mode == "strong") or (heavy and world > 1 and mode == "weak"):
                    continue
                spp = base_spp * world if mode == "weak" else base_spp
                ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize(); ctx.reset_stats()
                reps = 3 if w == "c2" else 1 if (heavy and world == 1) else 2
                t = time.perf_counter()
                for _ in range(reps):
                    ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize()
                dt = (time.perf_counter() - t) / reps
                st = ctx.stats()
                rays = st["rays"] / reps
                reduce_bytes = (world - 1) * fb_bytes           # full-buffer reduce: every other rank's buffer reaches the root (through the ring)
                gather_bytes = (world - 1) * fb_bytes // world  # gather: every other rank sends its own tiles
                reduce_ms = 0.0 if world == 1 else fb_bytes / (RCCL_REDUCE_BUSBW_GBS * 1e9) * 1e3 + COLLECTIVE_LATENCY_MS
                gather_ms = 0.0 if world == 1 else (fb_bytes / world) / (XGMI_LINK_GBS * 1e9) * 1e3 + COLLECTIVE_LATENCY_MS  # N-1 links in parallel, one pack each
                row = {"world": world, "scaling": mode, "spp_total": spp, "rank0_ms_per_step": dt * 1e3, "rank0_mrays_per_s": rays / dt / 1e6,
                       "collective_bytes_into_root": {"reduce": reduce_bytes, "gather": gather_bytes},
                       "projected_mrays_per_s": {"no_collective": world * rays / dt / 1e6, "reduce": world * rays / (dt + reduce_ms * 1e-3) / 1e6,
                                                 "gather": world * rays / (dt + gather_ms * 1e-3) / 1e6},
                       "assumed_collective_ms": {"reduce": reduce_ms, "gather": gather_ms}}
                rows.append(row)
                print("%s world %d %-6s spp %4d: rank 0 %8.2f ms/step %7.0f Mrays/s | projected x%d: %7.0f (no collective) %7.0f (reduce) %7.0f (gather)" % (
                    w, world, mode, spp, dt * 1e3, rays / dt / 1e6, world, row["projected_mrays_per_s"]["no_collective"], row["projected_mrays_per_s"]["reduce"],
                    row["projected_mrays_per_s"]["gather"]), flush=True)
        one = rows[0]["rank0_mrays_per_s"]
        for r in rows:
            r["projected_efficiency"] = {k: v / (r["world"] * one) for k, v in r["projected_mrays_per_s"].items()}
        out["workloads"][w] = {"label": wl["label"], "rows": rows}
        ctx.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "shard_sim.json"), "w"), indent=1)
    print("wrote gpurun_out/shard_sim.json")


if __name__ == "__main__":
    main()
