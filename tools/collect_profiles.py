#!/usr/bin/env python3
"""Files the output of tools/release_pass.sh (gpurun_out/rel/) under profiles/ (tracked).  Usage: collect_profiles.py rNN"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = os.path.join(ROOT, "gpurun_out", "rel")
PROF = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"

for w in ("c2", "c3", "c4", "c5"):
    src = os.path.join(REL, "bench_%s.json" % w)
    if os.path.exists(src):
        line = open(src).read().strip().splitlines()[-1]
        json.dump(json.loads(line), open(os.path.join(PROF, "%s_%s_bench.json" % (tag, w)), "w"), indent=1)
for w in ("c3", "c4", "c5"):  # the opt-in SAH tree on the same box
    src = os.path.join(REL, "bench_%s_sah.json" % w)
    if os.path.exists(src):
        line = open(src).read().strip().splitlines()[-1]
        json.dump(json.loads(line), open(os.path.join(PROF, "%s_%s_sah_bench.json" % (tag, w)), "w"), indent=1)
for w in ("c2", "c3", "c4", "c5"):
    src = os.path.join(REL, "kernel_stats_%s.csv" % w)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(PROF, "%s_%s_kernel_stats.csv" % (tag, w)))
if os.path.exists(os.path.join(REL, "valu_peak.json")) and os.path.getsize(os.path.join(REL, "valu_peak.json")) > 100:
    shutil.copy(os.path.join(REL, "valu_peak.json"), os.path.join(PROF, "valu_peak.json"))
for name in ("gather_probe.json", "gather_probe2.json", "size_sweep.txt", "coherence.txt", "occupancy_sweep.txt", "pytest_gpu.txt", "smoke.txt", "bench_c2_wall.txt",
             "valu_busy_calib.json", "shard_sim.json", "shard_sim.txt", "obj_parse.json", "shade_lanes_c2.json", "shade_lanes_c3.json", "shade_lanes_c5.json"):
    src = os.path.join(REL, name)
    if os.path.exists(src) and os.path.getsize(src) > 20:
        shutil.copy(src, os.path.join(PROF, "%s_%s" % (tag, name)))
for w in ("c2", "c3", "c4", "c5"):
    p = os.path.join(PROF, "%s_%s_bench.json" % (tag, w))
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    r = d["roofline"]
    k = r["kernels"]
    print("%s %6.0f Mrays/s %8.1f ms/step | roofline %s %s frac %.2f (valu at 2.4 GHz %.2f, at the pass clock %.2f, rocprof VALUBusy %.2f; hbm %.2f gather %.2f; lane-weighted %.2f) | " % (
        w, d["value"], d["ms_per_step"], r["kernel"], r["bound"], r["frac"] or 0, r.get("valu_busy_frac_at_2p4_ghz") or 0, r.get("valu_busy_frac_at_pass_clock") or 0,
        r.get("rocprof_valu_busy") or 0, r.get("fabric_frac_of_hbm_peak") or 0, r.get("l1_gather_frac") or 0, r.get("lane_weighted_frac_at_2p4_ghz") or 0)
        + " ".join("%s %.2f" % (n[2:], k[n]["ms_per_step"]) for n in k) + " | cpu %.1f / %.1f" % (
        d.get("cpu_baseline", {}).get("value", 0), d.get("cpu_baseline", {}).get("single_thread", {}).get("value", 0)))
