#!/usr/bin/env python3
"""Files the output of tools/release_pass.sh (gpurun_out/rel/) under profiles/ (tracked).  Usage: collect_profiles.py rNN"""
import json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = os.path.join(ROOT, "gpurun_out", "rel")
PROF = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def counted(name):  # the <..., COUNT = true> instantiations run only in bench.py's untimed counted pass
    if "<" not in name:
        return False
    args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
    base = name.split("<")[0]
    return args[0] == "true" if base in ("k_bvh", "k_generate", "k_prims") else args[-1] == "true"


traffic = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --workload <w> --steps 1 "
           "--warmup 0 --cpu-seconds 0`; timed-pass kernels only (the counted COUNT=true variants are excluded). Counter unit KB. "
           "gfx950 correction per MI355X_MICROARCH.md: hbm = 2*FETCH_SIZE + WRITE_SIZE: exact for coalesced float4 streams (calibrated on "
           "k_accumulate's known bytes), an upper bound for gathers of 16-64 B pieces (k_bvh records). FETCH_SIZE is a fabric-side counter "
           "that includes Infinity-Cache hits."}
for w in ("c2", "c3"):
    f = json.load(open(os.path.join(REL, "pmc_%s_FETCH_SIZE.json" % w)))
    wr = json.load(open(os.path.join(REL, "pmc_%s_WRITE_SIZE.json" % w)))
    out = {}
    for name, v in f.items():
        if not name.startswith("k_") or counted(name):
            continue
        base = name.split("<")[0]
        n = v["launches"]
        fb = v.get("FETCH_SIZE", 0.0) * 1024 / n
        wb = wr.get(name, {}).get("WRITE_SIZE", 0.0) * 1024 / n
        out[base] = {"launches": n, "fetch_size_bytes_per_launch": fb, "write_size_bytes_per_launch": wb, "hbm_bytes_per_launch_corrected": 2 * fb + wb}
    out["k_bvh_hbm_bytes_per_launch"] = out.get("k_bvh", {}).get("hbm_bytes_per_launch_corrected")
    traffic[w] = out
json.dump(traffic, open(os.path.join(PROF, "hbm_traffic.json"), "w"), indent=1)
for w in ("c2", "c3", "c4", "c5"):
    shutil.copy(os.path.join(REL, "bench_%s.json" % w), os.path.join(PROF, "%s_%s_bench.json" % (tag, w)))
for w in ("c2", "c3"):
    shutil.copy(os.path.join(REL, "kernel_stats_%s.csv" % w), os.path.join(PROF, "%s_%s_kernel_stats.csv" % (tag, w)))
for w in ("c2", "c4"):
    src = os.path.join(REL, "pmc_valu_%s.json" % w)
    if os.path.exists(src):
        v = json.load(open(src))
        out = {"_note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU over "
                        "`python bench.py --workload %s%s --steps 1 --warmup 0 --cpu-seconds 0` (timed pass + counted pass). SQ_BUSY_CYCLES is summed over the "
                        "32 shader engines; a wave64 VALU instruction occupies its SIMD for 4 cycles; active lanes = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)."
                        % (w, " --spp 64" if w == "c4" else "")}
        for name, c in v.items():
            if not name.startswith("k_"):
                continue
            rec = dict(c)
            if c.get("SQ_BUSY_CYCLES") and c.get("ms_total"):
                clk = c["SQ_BUSY_CYCLES"] / 32.0 / (c["ms_total"] * 1e-3)
                rec["clock_ghz"] = clk / 1e9
                rec["valu_busy_frac"] = c.get("SQ_INSTS_VALU", 0.0) * 4.0 / (1024.0 * c["SQ_BUSY_CYCLES"] / 32.0)
            if c.get("SQ_ACTIVE_INST_VALU"):
                rec["active_lane_frac"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * c["SQ_ACTIVE_INST_VALU"])
            out[name] = rec
        json.dump(out, open(os.path.join(PROF, "%s_pmc_valu_%s.json" % (tag, w)), "w"), indent=1)
for w in ("c2", "c3", "c4", "c5"):
    d = json.loads(open(os.path.join(PROF, "%s_%s_bench.json" % (tag, w))).read())
    r = d["roofline"]
    print(w, round(d["value"]), "Mrays/s", round(d["ms_per_step"], 1), "ms/step | k_bvh", round(r["avg_launch_ms"], 3), "ms/launch, frac", round(r["frac"], 2),
          "| cpu", round(d.get("cpu_baseline", {}).get("value", 0), 1))
