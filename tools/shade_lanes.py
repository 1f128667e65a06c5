#!/usr/bin/env python3
"""Where do k_shade's idle lanes go?  

A measurement build of the library (-DPTMI_LANE_TALLY, _build.build_variant('lanes', ...)) counts at ~30 points of k_shade's code how often a
wave enters the region behind the point and with how many lanes of its exec mask set (LT() in csrc/ptmi_device.h).  This tool

  GPU box:   tools/shade_lanes.py run [c2|c3|c4|c5] [spp]   -> gpurun_out/shade_lanes_<workload>.json  ({point: visits, lanes} of one step)
  anywhere:  tools/shade_lanes.py static                     -> profiles/r04_shade_lanes_static.json    (static instructions behind each point)
  anywhere:  tools/shade_lanes.py merge <run.json> <static.json> <out.json>

`static` compiles csrc/ptmi.hip to assembly with the same flag and attributes every instruction of k_shade6<false,false> to the last `; LT_MARK k`
comment in front of it in program order (an approximation: the compiler moves code across the markers' basic blocks, but the regions are large).
`merge` multiplies the two: estimated dynamic VALU instructions per region = visits x static VALU instructions, and the lanes those instructions
ran with — the table VERDICT asked for.
"""
import ctypes
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

POINTS = ["GROUP", "VALID", "MISS", "HIT", "RH_SPHERE", "RH_VOLUME", "RH_QUAD", "RH_TRI", "MS_LAMBERT", "MS_MIRROR", "MS_GLASS", "MS_ISO", "RR", "ACC_CONT", "END_SAMPLE",
          "END_CHANGES", "FLUSH", "QUAD_LOOP", "QUAD_FRONT", "QUAD_DENOM", "QUAD_T", "QUAD_ACCEPT", "ROOT_BOX", "MISS_SHORTCUT", "KEEP", "DIV3_SLOW", "RCP_SLOW", "SQRT_SLOW",
          "SPHERE_LOOP", "IS_LIGHT", "STAGE"]
VARIANT = os.path.join(ROOT, "webgpu-path-tracer_amd", "variants", "libptmi_lanes.so")


def run(workload="c2", spp=0):
    os.environ["PTMI_LIB"] = VARIANT
    import __graft_entry__ as entry
    import bench

    pkg = entry._load_pkg()

    class A:  # the arguments bench.make_workload reads
        width, height = (3840, 2160) if workload == "c5" else (1920, 1080)
        bounces, bvh, tris, stack_size, frames_in_flight = 8, "median", 0, 0, 0
    wl = bench.make_workload(pkg, workload, A)
    ctx = bench.make_context(pkg, wl, 0, A)
    spp = spp or {"c2": 64, "c3": 64, "c4": 32, "c5": 16}[workload]
    lib = pkg.load_library()
    lib.ptmi_lane_tally.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    buf = (ctypes.c_uint64 * (2 * len(POINTS)))()
    ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize()
    assert lib.ptmi_lane_tally(ctx.h, buf, len(POINTS), 1) == 0
    ctx.reset_stats(); ctx.set_timing(1)
    ctx.clear(); ctx.render(wl["view"], 1, spp); ctx.synchronize()
    assert lib.ptmi_lane_tally(ctx.h, buf, len(POINTS), 1) == 0
    st = ctx.stats(); ctx.set_timing(0)
    print("k_shade %.3f ms in this (measurement) build" % st["shade_ms"])
    TT = ["OTHER", "LOAD1_slot_state", "LOAD2_material_and_hit_data", "SHADE_compute", "STAGE", "FLUSH_PRIMS_quads_root_box", "FLUSH_STORE", "RING_READ"]
    tb = (ctypes.c_uint64 * len(TT))()
    times = None
    if hasattr(lib, "ptmi_time_tally"):
        lib.ptmi_time_tally.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        assert lib.ptmi_time_tally(ctx.h, tb, len(TT), 1) == 0
        tot = float(sum(tb)) or 1.0
        times = {TT[k]: {"wave_cycles": int(tb[k]), "share": tb[k] / tot} for k in range(len(TT))}
        for k, v in times.items():
            print("wave-cycles %-28s %5.1f %%" % (k, 100 * v["share"]))
    out = {"workload": wl["label"], "spp": spp, "rays": st["rays"], "paths": st["paths"], "shade_launches": st["shade_launches"], "wave_cycles_by_region": times,
           "points": {POINTS[k]: {"visits": int(buf[2 * k]), "lanes": int(buf[2 * k + 1])} for k in range(len(POINTS))}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    p = os.path.join(ROOT, "gpurun_out", "shade_lanes_%s.json" % workload)
    json.dump(out, open(p, "w"), indent=1)
    g = out["points"]["GROUP"]["visits"] or 1
    for k, v in out["points"].items():
        if v["visits"]:
            print("%-14s visits/group %7.3f  lanes/visit %5.1f" % (k, v["visits"] / g, v["lanes"] / v["visits"]))
    print("wrote", p)


def static_impl(kernel_pat=r"k_shade6ILb0ELb0E", out=None):
    import importlib.util

    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "webgpu-path-tracer_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    asm = "/tmp/ptmi_lanes.s"
    flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
    if not os.path.exists(asm) or os.environ.get("FORCE"):
        subprocess.run(["hipcc"] + flags + ["-DPTMI_LANE_TALLY", "--cuda-device-only", "-S", "-o", asm, os.path.join(b.CSRC, "ptmi.hip")], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    txt = open(asm).read()
    m = re.search(r"^(_ZN4ptmi\w*%s\w*):" % kernel_pat, txt, re.M)
    if not m:
        raise SystemExit("kernel not found in the assembly")
    body = txt[m.end():txt.index(".Lfunc_end", m.end())]
    cur, tab = "PROLOGUE", {}
    for line in body.splitlines():
        t = line.strip()
        mm = re.match(r";\s*LT_MARK (\d+)", t)
        if mm:
            cur = POINTS[int(mm.group(1))] if int(mm.group(1)) < len(POINTS) else "P%s" % mm.group(1)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        e = tab.setdefault(cur, {"instr": 0, "valu": 0, "mov_cndmask": 0, "salu": 0, "branch": 0, "vmem": 0, "lds": 0, "smem": 0, "trans": 0, "f64": 0})
        e["instr"] += 1
        if op.startswith("v_"):
            e["valu"] += 1
            if re.match(r"v_(mov_b32|mov_b64|cndmask_b32|readlane|readfirstlane|writelane|accvgpr)", op):
                e["mov_cndmask"] += 1
            if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(iflag_)?f32", op):
                e["trans"] += 1
            if "_f64" in op:
                e["f64"] += 1
        elif re.match(r"s_(cbranch|branch|setpc|swappc|endpgm)", op):
            e["branch"] += 1
        elif re.match(r"s_(load|buffer_load)", op):
            e["smem"] += 1
        elif op.startswith("s_"):
            e["salu"] += 1
        elif re.match(r"(global|flat|buffer|scratch)_", op):
            e["vmem"] += 1
        elif op.startswith("ds_"):
            e["lds"] += 1
    out = out or os.path.join(ROOT, "profiles", "r04_shade_lanes_static.json")
    json.dump({"kernel": m.group(1), "note": "static instructions of the -DPTMI_LANE_TALLY build attributed to the last LT_MARK in program order (each marker itself costs ~12 "
                                              "instructions of tally code, included)", "regions": tab}, open(out, "w"), indent=1)
    for k, v in tab.items():
        print("%-14s %s" % (k, v))
    print("wrote", out)


def merge(run_json, static_json, out_json):
    r, s = json.load(open(run_json)), json.load(open(static_json))
    g = r["points"]["GROUP"]["visits"] or 1
    rows, tot_dyn, tot_lane = [], 0.0, 0.0
    for k, pt in r["points"].items():
        if not pt["visits"]:
            continue
        st = s["regions"].get(k, {})
        valu = max(0, st.get("valu", 0) - 6)  # the tally's own VALU (ballot compare, mbcnt x2, compare, popcount moves) is not the product's
        dyn = pt["visits"] * valu
        lanes = pt["lanes"] / pt["visits"]
        rows.append({"region": k, "visits_per_group": pt["visits"] / g, "lanes_per_visit": lanes, "static_valu": valu, "static_mov_cndmask": st.get("mov_cndmask", 0),
                     "static_salu": st.get("salu", 0), "dyn_valu_per_group": dyn / g, "idle_lane_instr_per_group": dyn / g * (64 - lanes) / 64})
        tot_dyn += dyn / g
        tot_lane += dyn / g * lanes / 64
    rows.sort(key=lambda e: -e["idle_lane_instr_per_group"])
    out = {"workload": r["workload"], "spp": r["spp"], "groups": g, "est_dyn_valu_per_64_slot_group": tot_dyn, "est_active_lane_frac": tot_lane / tot_dyn if tot_dyn else None,
           "note": "regions in the order of the lane-instructions they waste; dyn = visits x static VALU of the region (program-order attribution, approximate)", "regions": rows}
    json.dump(out, open(out_json, "w"), indent=1)
    print("est. dynamic VALU per 64-slot group %.0f, active lanes %.3f" % (tot_dyn, out["est_active_lane_frac"] or 0))
    for e in rows[:16]:
        print("%-14s visits/group %6.3f lanes %5.1f static VALU %4d dyn/group %6.1f idle-lane instr/group %6.1f" % (
            e["region"], e["visits_per_group"], e["lanes_per_visit"], e["static_valu"], e["dyn_valu_per_group"], e["idle_lane_instr_per_group"]))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "run"
    if cmd == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "c2", int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    elif cmd == "static":
        static_impl()
    else:
        merge(*sys.argv[2:5])
