#!/usr/bin/env python3
"""Regenerates the results table of DESIGN.md §5 (between the rNN-table markers) from profiles/rNN_*_bench.json.  usage: tools/design_table.py [r04]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
D = {w: json.load(open(os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (tag, w)))) for w in ("c2", "c3", "c4", "c5")}
cols = ["c2", "c3", "c4", "c5"]


def k(w, name):
    return D[w]["roofline"]["kernels"][name]


def f(x, n=2):
    return "—" if x is None else ("%." + str(n) + "f") % x


rows = []
rows.append("| | configs[1] 967 tris | configs[2] 871 k tris | configs[3] interior 262 k | configs[4] 1.09 M, glass+IS, 4K |")
rows.append("|---|---|---|---|---|")
rows.append("| Mrays/s (Mpaths/s) | " + " | ".join("**%.0f** (%.0f)" % (D[w]["value"], D[w]["config"]["mpaths_per_s"]) for w in cols) + " |")
spp = {"c2": "64 spp", "c3": "256 spp", "c4": "512 spp", "c5": "128 spp"}
rows.append("| ms per step | " + " | ".join("%.1f (%s)" % (D[w]["ms_per_step"], spp[w]) for w in cols) + " |")
rows.append("| node visits / triangle tests per ray | " + " | ".join("%.1f / %.2f" % (D[w]["roofline"]["work_per_ray"]["node_visits"], D[w]["roofline"]["work_per_ray"]["tri_tests"]) for w in cols) + " |")
rows.append("| ms per step: generate / bvh / shade / tail / accumulate | " + " | ".join(" / ".join(f(k(w, n)["ms_per_step"], 2 if k(w, n)["ms_per_step"] < 100 else 0) for n in ("k_generate", "k_bvh", "k_shade", "k_tail", "k_accumulate")) for w in cols) + " |")
rows.append("| dominant kernel: bound, `frac` | " + " | ".join("`%s`: %s **%.2f**" % (D[w]["roofline"]["kernel"], {"valu_issue": "VALU issue at 2.4 GHz", "l1_gather": "L1 gather at 2.4 GHz", "hbm": "HBM", "vector_memory_path": "vector-memory path (TA busy) at 2.4 GHz"}[D[w]["roofline"]["bound"]], D[w]["roofline"]["frac"]) for w in cols) + " |")
rows.append("| the same kernel: VALU model at the pass clock / all-unclassified-at-4 upper bound / rocprof VALUBusy / lane-weighted at 2.4 GHz | " + " | ".join("%s / %s / %s / %s" % (f(D[w]["roofline"].get("valu_busy_frac_at_pass_clock")), f(D[w]["roofline"].get("valu_busy_frac_upper_bound_at_pass_clock")), f(D[w]["roofline"].get("rocprof_valu_busy")), f(D[w]["roofline"].get("lane_weighted_frac_at_2p4_ghz"))) for w in cols) + " |")
rows.append("| `k_bvh`: L1 gather / TA busy (pass clock) / VALU at 2.4 GHz / fabric bytes ÷ 8 TB/s / active lanes / waves parked on memory | " + " | ".join("%s / %s / %s / %s / %s / %s" % (f(k(w, "k_bvh").get("l1_gather_frac")), f(k(w, "k_bvh").get("ta_busy_frac")), f(k(w, "k_bvh").get("valu_busy_frac_at_2p4_ghz")), f(k(w, "k_bvh").get("fabric_frac_of_hbm_peak")), f(k(w, "k_bvh").get("active_lane_frac")), f(k(w, "k_bvh").get("wave_wait_frac"))) for w in cols) + " |")
rows.append("| `k_shade`: TA busy (pass clock) / VALU at 2.4 GHz / rocprof VALUBusy / fabric bytes ÷ 8 TB/s / active lanes / waves parked | " + " | ".join("%s / %s / %s / %s / %s / %s" % (f(k(w, "k_shade").get("ta_busy_frac")), f(k(w, "k_shade").get("valu_busy_frac_at_2p4_ghz")), f(k(w, "k_shade").get("rocprof_valu_busy")), f(k(w, "k_shade").get("fabric_frac_of_hbm_peak")), f(k(w, "k_shade").get("active_lane_frac")), f(k(w, "k_shade").get("wave_wait_frac"))) for w in cols) + " |")
rows.append("| `k_generate`: VALU at 2.4 GHz / at the pass clock / rocprof VALUBusy / fabric bytes ÷ 8 TB/s | " + " | ".join("%s / %s / %s / %s" % (f(k(w, "k_generate").get("valu_busy_frac_at_2p4_ghz")), f(k(w, "k_generate").get("valu_busy_frac_at_pass_clock")), f(k(w, "k_generate").get("rocprof_valu_busy")), f(k(w, "k_generate").get("fabric_frac_of_hbm_peak"))) for w in cols) + " |")
rows.append("| one step over the fabric: GB / ÷ 8 TB/s / × the reference megakernel's compulsory bytes | " + " | ".join("%.1f / %s / %.1f×" % (D[w]["roofline"]["step_fabric_bytes"] / 1e9, f(D[w]["roofline"].get("step_fabric_frac_of_hbm_peak")), D[w]["roofline"]["step_fabric_bytes"] / D[w]["roofline"]["compulsory_bytes"]) if D[w]["roofline"].get("step_fabric_bytes") else "—" for w in cols) + " |")
S = {}
for w in cols:
    p_sah = os.path.join(ROOT, "profiles", "%s_%s_sah_bench.json" % (tag, w))
    if os.path.exists(p_sah):
        S[w] = json.load(open(p_sah))
if S:
    rows.append("| the same with the opt-in SAH tree (`--bvh sah`, built on the GPU): Mrays/s / node visits per ray / `k_bvh` ms per step / build ms (nodes, depth) | " + " | ".join(
        "—" if w not in S else "**%.0f** / %.1f / %s / %.0f (%d, %d)" % (S[w]["value"], S[w]["roofline"]["work_per_ray"]["node_visits"], f(S[w]["roofline"]["kernels"]["k_bvh"]["ms_per_step"], 2 if S[w]["roofline"]["kernels"]["k_bvh"]["ms_per_step"] < 100 else 0),
                                                        S[w]["setup_ms"]["build_scene_bvh_sah_device_ms"], S[w]["setup_ms"]["sah_tree"]["nodes"], S[w]["setup_ms"]["sah_tree"]["depth"]) for w in cols) + " |")
rows.append("| CPU oracle, 16 threads / 1 thread (Mrays/s) | " + " | ".join("%.1f / %.1f" % (D[w]["cpu_baseline"]["value"], D[w]["cpu_baseline"]["single_thread"]["value"]) for w in cols) + " |")
if any((D[w]["cpu_baseline"].get("js_traversal") or {}).get("value") for w in cols):
    rows.append("| `hitScene` restated in single-threaded JavaScript under Node (`js/hit_scene.mjs`; Mrays/s, synthetic rays) | " + " | ".join(
        ("%.2f" % D[w]["cpu_baseline"]["js_traversal"]["value"]) if (D[w]["cpu_baseline"].get("js_traversal") or {}).get("value") else "—" for w in cols) + " |")
rows.append("| vs. 33.5 Mpaths/s (`benchmarks.txt:18-20`): paths / rays | " + " | ".join("%.0f× / %.0f×" % (D[w]["config"]["mpaths_per_s"] / 33.5, D[w]["value"] / 33.5) for w in cols) + " |")
table = "\n".join("  " + r for r in rows)
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
begin, end = "  <!-- %s-table-begin -->\n" % tag, "  <!-- %s-table-end -->" % tag
i, j = s.find(begin), s.find(end)
n = 1 if (i >= 0 and j > i) else 0
s2 = s[: i + len(begin)] + table + "\n" + s[j:] if n else s
if n != 1:
    sys.exit("markers not found in DESIGN.md")
open(p, "w").write(s2)
print(table)
