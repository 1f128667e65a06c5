#!/usr/bin/env python3
"""Where do k_bvh's wave-cycles go?  A measurement build (-DPTMI_LANE_TALLY: _build.build_variant('lanes', ['-DPTMI_LANE_TALLY'])) keeps a stopwatch per
wave of k_bvh (BT() marks, csrc/ptmi_kernels.h): cycles, marks and lanes per region — flag scan, ray pick-up, the wait for an inner node's record, the
box tests + pops, the wait for a triangle record, the triangle test, retiring stores, votes, carry.

  GPU box:  tools/bvh_regions.py [c2|c3|c4|c5] [spp] [ENV=VAL ...]   -> gpurun_out/bvh_regions_<workload>[_tag].json
            tools/bvh_regions.py c3 lone                             -> gpurun_out/tail_regions_c3.json   (a lone frame: k_tail's regions instead)
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REGIONS = ["scan", "pickup", "inner_fetch_wait", "inner_step", "leaf_fetch_wait", "leaf_test", "retire", "vote", "carry", "start"]
TAIL_REGIONS = ["take_paths", "walk_fetch_wait", "walk_step", "walk_leaf", "shade", "prims", "vote", "-", "-", "-"]
VARIANT = os.path.join(ROOT, "webgpu-path-tracer_amd", "variants", "libptmi_lanes.so")


def main():
    args = [a for a in sys.argv[1:] if "=" not in a]
    envs = [a for a in sys.argv[1:] if "=" in a]
    for e in envs:
        k, v = e.split("=", 1)
        os.environ[k] = v
    workload = args[0] if args else "c2"
    lone = "lone" in args  # one frame per render: k_generate + k_tail — the regions of k_tail then
    args = [a for a in args if a != "lone"]
    os.environ["PTMI_LIB"] = VARIANT
    import __graft_entry__ as entry
    import bench

    pkg = entry._load_pkg()

    class A:
        width, height = (1920, 1080)
        bounces, bvh, tris, stack_size, frames_in_flight = 8, "median", 0, 0, 0
    wl = bench.make_workload(pkg, workload, A)
    ctx = bench.make_context(pkg, wl, 0, A)
    spp = 1 if lone else int(args[1]) if len(args) > 1 else {"c2": 64, "c3": 64, "c4": 16, "c5": 16}[workload]
    regions, which = (TAIL_REGIONS, 2) if lone else (REGIONS, 0)
    lib = pkg.load_library()
    lib.ptmi_bvh_tally.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    buf = (ctypes.c_uint64 * (3 * len(REGIONS)))()
    ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize()
    assert lib.ptmi_bvh_tally(ctx.h, buf, len(REGIONS), 1 | which) == 0
    ctx.reset_stats()
    ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize()
    assert lib.ptmi_bvh_tally(ctx.h, buf, len(REGIONS), 1 | which) == 0
    st = ctx.stats()
    tot = float(sum(buf[3 * k] for k in range(len(REGIONS)))) or 1.0
    out = {"workload": wl["label"], "spp": spp, "env": envs, "rays": st["rays"], "bvh_ms": st["bvh_ms"], "intersect_launches": st["intersect_launches"],
           "bvh_node_visits": st["bvh_node_visits"], "tri_tests": st["tri_tests"], "regions": {}}
    for k, name in enumerate(regions):
        cyc, marks, lanes = int(buf[3 * k]), int(buf[3 * k + 1]), int(buf[3 * k + 2])
        out["regions"][name] = {"wave_cycles": cyc, "share": cyc / tot, "marks": marks, "cycles_per_mark": cyc / marks if marks else None, "lanes_per_mark": lanes / marks if marks else None}
        if marks:
            print("%-18s %5.1f %%  marks %11d  cycles/mark %8.0f  lanes/mark %5.1f" % (name, 100 * cyc / tot, marks, cyc / marks, lanes / marks))
    print("k_bvh %.2f ms over %d launches (tally build)" % (st["bvh_ms"], st["intersect_launches"]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = "".join("_" + e.replace("=", "") for e in envs)
    p = os.path.join(ROOT, "gpurun_out", "%s_regions_%s%s.json" % ("tail" if lone else "bvh", workload, tag))
    json.dump(out, open(p, "w"), indent=1)
    print("wrote", p)


if __name__ == "__main__":
    main()
