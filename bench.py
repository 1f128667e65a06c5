#!/usr/bin/env python3
"""bench.py — Mrays/s of the HIP wavefront integrator on BASELINE.json's configs[1]
(Cornell + monkey_968.obj, 1920x1080, 64 spp, 8 bounces), one JSON line on rank 0.

A "step" is one full render of the workload: clear the accumulation buffer, trace `spp` progressive
frames (frame numbers 1..spp, resetBuffer = 0), and — for N > 1 — sum-reduce the per-rank framebuffers
to rank 0.  Scene, BVH and path buffers are resident in HBM before the timed region.
Rays are counted exactly (one per hitScene invocation) by the device.
N > 1 is weak scaling: pixels are sharded across ranks in tiles and spp is multiplied by N, so the
rays per GPU stay fixed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def alg_bytes(st):
    """SURVEY.md §8d algorithmic bytes of hitScene in the reference's layouts."""
    return 48 * st["node_visits"] + 96 * st["tri_tests"] + 64 * st["mat_fetches"] + 32 * st["sphere_tests"] + 80 * st["quad_tests"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 = BASELINE configs[1] (default: the config the metric is quoted on); c3 = configs[2] dragon-class 871k tris; "
                         "c4 = configs[3] sponza-class interior; c5 = configs[4] buddha-class + glass + importance sampling (use --width 3840 --height 2160)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=0, help="progressive frames per step and per GPU (0 = the config's: 64 for c2, 256 for c3)")
    ap.add_argument("--stack-size", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--bvh", default="median", choices=["median", "sah"],
                    help="median = the reference's live builder (default, what the metric is quoted on); sah = the reference's "
                         "other, never-called builder (lib/BVH/bvhNode.js:108-283) as an opt-in (not for c2's golden buffers)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 on ONE GPU for rehearsal: ranks share cuda:0, the framebuffer reduce goes through gloo on host copies "
                         "(RCCL wants one device per rank); numbers from this mode are not bench results")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the cpu_baseline (the 1-GPU box's CPU share)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch

    pkg = entry._load_pkg()
    from webgpu_path_tracer_amd import dist as pdist

    rank, world, local = pdist.init_process_group("gloo" if args.rehearse_gloo else None)
    if args.rehearse_gloo:
        local = 0
    if world != max(args.gpus, 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local)

    W, H = args.width, args.height
    cam, extra = "cornell", {}
    if args.spp <= 0:
        args.spp = {"c2": 64, "c3": 256, "c4": 512, "c5": 1024}[args.workload]
    if args.workload == "c5" and args.bounces == 8:
        args.bounces = 16
    spp = args.spp * world
    sah = args.bvh == "sah"
    if sah and args.workload == "c2":
        raise SystemExit("--bvh sah needs a workload built through the Scene API (c3, c4, c5)")
    if args.workload == "c2":
        buffers = pkg.scenes.golden_buffers("c2")  # reference-generated buffers of configs[1] (tests/golden)
        label, stack = "configs[1]: Cornell + monkey_968.obj (967 tris)", args.stack_size or 20
    elif args.workload == "c3":
        buffers = pkg.scenes.c3_scene().buffers(native=pkg.ptmi.NativeHost(), sah=sah)  # procedural stand-in, 871,414 tris
        label, stack = "configs[2]: Cornell + dragon-class mesh (871,414 tris, procedural stand-in for stanfordDragon.obj)", args.stack_size or 24
    elif args.workload == "c4":
        buffers = pkg.scenes.c4_scene().buffers(native=pkg.ptmi.NativeHost(), sah=sah)
        label, stack, cam = "configs[3]: sponza-class interior (262,267 tris, procedural stand-in for sponzaAtrium.obj), camera inside", args.stack_size or 24, "interior"
    else:
        buffers = pkg.scenes.c5_scene().buffers(native=pkg.ptmi.NativeHost(), sah=sah)
        label, stack, extra = "configs[4]: Cornell + buddha-class glass mesh (1,087,716 tris, procedural stand-in for buddha.obj), importance sampling", args.stack_size or 24, dict(importance_sampling=1)
    view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS[cam])
    ctx = pkg.Context(local)
    ctx.upload_scene(buffers)
    ctx.set_params(max_bounces=args.bounces, frames_in_flight=args.frames_in_flight, stack_size=stack, **extra)
    ctx.resize(W, H)
    fb_t = None
    if world > 1:
        fb_t = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
        ctx.bind_framebuffer(fb_t.data_ptr(), fb_t.numel() * 4)
        ctx.set_shard(rank, world, pdist.TILE_PIXELS)

    def step():
        ctx.clear()
        ctx.render(view, 1, spp)
        ctx.synchronize()
        if world > 1 and args.rehearse_gloo:
            host = fb_t.cpu()
            pdist.reduce_framebuffer(host, 0)
            fb_t.copy_(host)
            torch.cuda.synchronize()
        elif world > 1:
            pdist.reduce_framebuffer(fb_t, 0)
            torch.cuda.synchronize()  # the reduce runs on torch's stream; the next clear runs on the context's

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    ctx.reset_stats()
    ctx.set_timing(2)  # HIP events around the dominant kernel (k_bvh) only: every event is a stream marker
    torch.cuda.synchronize()
    pdist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    pdist.barrier()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    ctx.set_timing(0)

    dt_max = pdist.all_reduce_scalar(dt, "max")
    rays_all = pdist.all_reduce_scalar(st["rays"], "sum")
    paths_all = pdist.all_reduce_scalar(st["paths"], "sum")

    # exact algorithmic bytes of one step (counted variant of the same kernels, untimed)
    ctx.reset_stats()
    ctx.set_counters(True)
    ctx.set_timing(1)  # per-kernel split, outside the timed region
    ctx.clear()
    ctx.render(view, 1, spp)
    cst = ctx.stats()
    ctx.set_counters(False)
    ctx.set_timing(0)
    assert cst["rays"] * args.steps == st["rays"], "ray count differs between the counted and the timed pass"
    bytes_total = alg_bytes(cst) * args.steps  # all of hitScene (k_prims + k_bvh)
    # the dominant kernel is the BVH traversal; its share of the algorithmic bytes (reference layouts):
    bvh_bytes = (48 * cst["bvh_node_visits"] + 96 * cst["tri_tests"] + 64 * cst["bvh_mat_fetches"]) * args.steps
    launches = max(st["intersect_launches"], 1)
    bvh_s = st["bvh_ms"] / 1e3
    achieved = bvh_bytes / bvh_s / 1e9 if bvh_s > 0 else 0.0
    # all of hitScene's algorithmic bytes (part 1 runs inside k_generate / k_shade) over the whole render of the counted pass
    hit_scene = (bytes_total / args.steps) / (cst["render_ms"] / 1e3) / 1e9 if cst["render_ms"] > 0 else 0.0
    traffic = None
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get(args.workload, {}).get("k_bvh_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        out = {
            "metric": "Mrays/s at 1080p, 8 bounces; achieved HBM GB/s vs roofline",
            "value": rays_all / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s, %dx%d, %d spp per GPU (%d total), %d bounces, stack_size %d%s" % (label, W, H, args.spp, spp, args.bounces, stack, ", SAH BVH (opt-in)" if sah else ""),
                "rays_per_step": rays_all / args.steps,
                "mpaths_per_s": paths_all / dt_max / 1e6,
                "parallelism": "pixel tiles x%d + 1 RCCL reduce" % world if world > 1 else "1 GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_bvh",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": bvh_bytes / launches,
                "avg_launch_ms": st["bvh_ms"] / launches,
                "launches": launches,
                "hit_scene_algorithmic_gbs_whole_render": hit_scene,
                "note": "k_bvh is bound by VALU issue and fetch latency (rocprofv3 PMC: ~66 % VALU busy at ~54 % active lanes on deep trees), not by HBM: "
                        "algorithmic bytes are reference-layout bytes, most of them served by L2 / Infinity Cache, so frac can exceed 1 (DESIGN.md section 5)",
                "work_per_ray": {k: cst[k] / max(cst["rays"], 1) for k in ("node_visits", "bvh_node_visits", "tri_tests", "quad_tests", "sphere_tests", "mat_fetches")},
                "kernel_ms_one_step_counted_pass": {"prims": cst["prims_ms"], "bvh": cst["bvh_ms"], "shade": cst["shade_ms"], "other": cst["other_ms"], "render": cst["render_ms"]},
            },
        }
        if world == 1 and args.cpu_seconds > 0:
            from oracle import ptm_oracle

            cores = min(ptm_oracle.max_threads(), args.cpu_threads)
            t = time.perf_counter()
            _, ost = ptm_oracle.render(buffers, W, H, view, 1, 1, max_bounces=args.bounces, stack_size=stack, threads=cores, **extra)
            one = time.perf_counter() - t
            frames = int(max(1, min(args.spp, args.cpu_seconds / max(one, 1e-3))))
            t = time.perf_counter()
            _, ost = ptm_oracle.render(buffers, W, H, view, 1, frames, max_bounces=args.bounces, stack_size=stack, threads=cores, **extra)
            cdt = time.perf_counter() - t
            out["cpu_baseline"] = {
                "value": ost["rays"] / cdt / 1e6,
                "unit": "Mrays/s",
                "cores": cores,
                "kind": "port",
                "sample": "same scene and camera, %dx%d, frames 1..%d of %d (%d rays), scalar f32 oracle with OpenMP over pixels" % (W, H, frames, args.spp, ost["rays"]),
            }
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
