#!/usr/bin/env python3
"""Static ISA histogram of one kernel of libptmi (gfx950): instruction counts per class, weighted by the issue cost
measured in profiles/valu_peak.json, per basic block and in total.

  tools/isa_histogram.py 'k_shade<false, false, false>' [--blocks] [--json out.json]

Compiles csrc/ptmi.hip to assembly with the product's flags (device only), cuts the kernel out by its demangled
name and classifies every instruction.  A static count says what a path through the kernel is made of; the dynamic
counts per class come from the SQ_INSTS_VALU_* counters (tools/pmc_hist.sh).
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASSES = [
    ("f64_trans", r"v_(rcp|rsq|sqrt)_f64"),
    ("f64_arith", r"v_(fma|mul|add|min|max|ldexp|frexp\w*|trunc|floor|rndne|fract|div_\w+)_f64"),
    ("cvt", r"v_cvt_"),
    ("f32_trans", r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(iflag_)?f32"),
    ("f32_div_helpers", r"v_div_(scale|fmas|fixup)_f32"),
    ("f32_fma", r"v_(fma|fmac|mad|mac|fmaak|fmamk)_f32"),
    ("f32_pk", r"v_pk_\w+_f32"),
    ("f32_mul_add", r"v_(mul|add|sub|subrev)_f32"),
    ("f32_minmax", r"v_(min|max|min3|max3|med3)_f32"),
    ("cmp", r"v_cmp\w*"),
    ("cndmask_mov", r"v_(cndmask_b32|mov_b32|mov_b64|accvgpr\w+|readlane\w*|readfirstlane\w*|writelane\w*|swap_b32|permlane\w*|bfrev\w*)"),
    ("int_fast", r"v_(and_b32|or_b32|xor_b32|not_b32|add_u32|sub_u32|subrev_u32|add_co_u32|sub_co_u32|subrev_co_u32|addc_co_u32|subb_co_u32|subbrev_co_u32)"),  # 2 cycles (valu_peak: v_add_u32, v_and_b32, v_xor_b32)
    ("int", r"v_(and|or|xor|not|lshl\w*|lshr\w*|ashr\w*|add\w*_u32|add_co\w*|addc_co\w*|sub\w*_u32|sub_co\w*|subb\w*|mul_\w*(u32|i32|u24|i24)\w*|mad_\w*(u32|i32|u24|i24|u64|i64)\w*|bfe\w*|bfi\w*|and_or\w*|or3\w*|xad\w*|lshl_\w+|add3\w*|add_lshl\w*|lshl_add\w*|lshl_or\w*|alignbit\w*|alignbyte\w*|mbcnt\w*|ffb\w*|bcnt\w*|min_\w*[ui]\d+|max_\w*[ui]\d+|perm_b32|sad\w*|cvt_pk\w*)"),
    ("valu_other", r"v_\w+"),
    ("salu", r"s_(?!load|buffer_load|waitcnt|barrier|endpgm|nop|branch|cbranch|setpc|swappc|getpc|sleep|setprio|sendmsg|memtime|memrealtime|dcache|icache|code_end)\w+"),
    ("branch", r"s_(branch|cbranch\w*|setpc\w*|swappc\w*|endpgm)"),
    ("wait_nop", r"s_(waitcnt\w*|nop|barrier|sleep|setprio)"),
    ("smem", r"s_(load|buffer_load|memtime|memrealtime|dcache\w*)\w*"),
    ("vmem_load", r"(global|flat|buffer)_load\w*"),
    ("vmem_store", r"(global|flat|buffer)_store\w*"),
    ("vmem_atomic", r"(global|flat|buffer)_atomic\w*"),
    ("scratch", r"scratch_\w+"),
    ("lds", r"ds_\w+"),
]
CLASS_RE = [(n, re.compile(p + r"$")) for n, p in CLASSES]
VALU = {"f64_trans", "f64_arith", "cvt", "f32_trans", "f32_div_helpers", "f32_fma", "f32_pk", "f32_mul_add", "f32_minmax", "cmp", "cndmask_mov", "int_fast", "int", "valu_other"}
# classes the SQ_INSTS_VALU_* counters can tell apart (bench.py reads them per kernel); the rest is "unclassified" there
COUNTED = {"f32_fma": "fast", "f32_mul_add": "fast", "f32_trans": "trans32", "f64_trans": "trans64", "int_fast": "int32", "int": "int32", "cvt": "cvt"}


def classify(op):
    for n, r in CLASS_RE:
        if r.match(op):
            return n
    return "other"


def issue_weights():
    """Issue cycles per wave-instruction per SIMD (throughput at 8 waves/SIMD, wall clock) from the microbenchmark
    (tools/valu_peak.hip -> profiles/valu_peak.json), rounded to the hardware's classes 2 / 4 / 8 / 16."""
    w = collections.defaultdict(lambda: 4.0)
    p = os.path.join(ROOT, "profiles", "valu_peak.json")
    if not os.path.exists(p):
        return w, None
    d = json.load(open(p))
    cyc = {c["op"]: c["cycles_per_wave_instr_per_simd_wall"] for c in d["cases"] if c["waves_per_simd"] == 8}

    def g(op, dflt):
        v = cyc.get(op, dflt)
        return min((2.0, 4.0, 8.0, 16.0), key=lambda k: abs(k - v))

    w.update({
        "f32_fma": g("v_fma_f32", 2), "f32_mul_add": g("v_mul_f32", 2), "f32_minmax": g("v_min_f32", 4), "cmp": g("v_cmp_gt_f32", 4),
        "cndmask_mov": g("v_mov_b32", 2), "int_fast": g("v_add_u32", 2), "int": g("v_lshlrev_b32", 4), "f32_pk": g("v_pk_fma_f32", 4), "f32_trans": g("v_sqrt_f32", 8),
        "f32_div_helpers": g("v_div_scale_f32", 4), "f64_arith": g("v_fma_f64", 4), "f64_trans": g("v_rcp_f64", 16), "cvt": g("v_cvt_f64_f32", 4), "valu_other": 4.0,
    })
    return w, p


def assembly(src=None, out="/tmp/ptmi_isa.s"):
    src = src or os.path.join(ROOT, "webgpu-path-tracer_amd", "csrc", "ptmi.hip")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-Wno-unused-command-line-argument"]
    subprocess.run(["hipcc"] + flags + ["--cuda-device-only", "-S", "-o", out, src], check=True)
    return open(out).read().splitlines()


def demangle(sym):
    return subprocess.run(["c++filt", sym], stdout=subprocess.PIPE, text=True).stdout.strip()


def kernels(lines):
    """name -> (start, end) line ranges of every function in the assembly."""
    out, cur, start = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|\.L_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L") or (m and l.startswith(".L_Z")):
            if l.startswith(".LBB") or l.startswith(".Ltmp") or l.startswith(".Lfunc"):
                continue
            cur, start = m.group(1), i
        if l.startswith(".Lfunc_end") and cur:
            out[cur] = (start, i)
            cur = None
    return out


def count(lines, b, e):
    total = collections.Counter()
    for l in lines[b + 1:e]:
        t = l.strip()
        m = re.match(r"^([a-z][a-z0-9_]+)(\s|$)", t)
        if not m or t.startswith("."):
            continue
        total[classify(re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1)))] += 1
    return total


def pieces(a):
    lines = assembly(os.path.join(ROOT, "tools", "isa_pieces.hip"), "/tmp/ptmi_pieces.s")
    w, _ = issue_weights()
    res = {}
    for name, (b, e) in kernels(lines).items():
        if not name.startswith("p_"):
            continue
        t = count(lines, b, e)
        valu = sum(v for k, v in t.items() if k in VALU)
        cyc = sum(v * w[k] for k, v in t.items() if k in VALU)
        res[name[2:]] = {"static_valu": valu, "valu_issue_cycles_weighted": cyc, "by_class": dict(t.most_common())}
        print("%-18s VALU %4d  issue cycles %5.0f  | %s" % (name[2:], valu, cyc, ", ".join("%s %d" % kv for kv in t.most_common(6))))
    if a.json:
        json.dump({"_note": "static counts of stand-alone kernels wrapping one building block each (tools/isa_pieces.hip; includes ~6 instructions of load/store scaffolding); "
                            "loops (the quad loop of prims) are counted once; weights as in r02_isa_histogram.json", "pieces": res}, open(a.json, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", nargs="?", default="", help="substring of the demangled kernel name, e.g. 'k_shade<false, false, false>'")
    ap.add_argument("--pieces", action="store_true", help="instead: the building blocks of tools/isa_pieces.hip (norm3, div3, sqrt, rcp, rand, scatter, prims ...), one line each")
    ap.add_argument("--blocks", action="store_true", help="also list the largest basic blocks")
    ap.add_argument("--json", default=None)
    ap.add_argument("--product", action="store_true", help="the kernels bench.py's cost model reads (profiles/rNN_isa_histogram.json): tools/isa_histogram.py --product --json profiles/r04_isa_histogram.json")
    a = ap.parse_args()
    if a.pieces:
        return pieces(a)
    lines = assembly()
    ks = kernels(lines)
    dem = {s: demangle(s.lstrip(".L") if s.startswith(".L_Z") else s).replace(" ", "") for s in ks}
    if a.product:
        pats = ["k_shade6<true,false>(", "k_shade6<false,false>(", "k_shade<true,true,false,false>(", "k_bvh2<false,true,false>(", "k_bvh2<false,false,false>(", "k_accumulate(",
                "k_tail<false,false,false,true>(", "k_tail6<false,true>(", "k_generate<false>("]
        want = [s for s in ks if any(p in dem[s] for p in pats)]
    else:
        want = [s for s in ks if a.kernel.replace(" ", "") in dem[s]]
    if not want:
        sys.exit("no function matches; have: " + ", ".join(sorted(demangle(s) for s in ks))[:4000])
    w, src = issue_weights()
    res = {}
    for s in want:
        b, e = ks[s]
        total = collections.Counter()
        ops = collections.Counter()
        blocks, cur = [], [None, collections.Counter()]
        for l in lines[b + 1:e]:
            t = l.strip()
            mlab = re.match(r"^(\.LBB\w+):", t)
            if mlab:
                blocks.append(cur)
                cur = [mlab.group(1), collections.Counter()]
                continue
            m = re.match(r"^([a-z][a-z0-9_]+)(\s|$)", t)
            if not m or t.startswith("."):
                continue
            op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1))
            c = classify(op)
            total[c] += 1
            ops[op] += 1
            cur[1][c] += 1
        blocks.append(cur)
        valu = sum(v for k, v in total.items() if k in VALU)
        cyc = sum(v * w[k] for k, v in total.items() if k in VALU)
        name = demangle(s.lstrip(".L") if s.startswith(".L_Z") else s)
        # what bench.py's cost model charges the instructions its counters cannot classify, and the integer ones: this kernel's static mix
        unc = {k: v for k, v in total.items() if k in VALU and k not in COUNTED}
        ints = {k: v for k, v in total.items() if COUNTED.get(k) == "int32"}
        res[name] = {"static_instructions": sum(total.values()), "static_valu": valu, "valu_issue_cycles_weighted": cyc, "by_class": dict(total.most_common()),
                     "unclassified_avg_cycles": (sum(v * w[k] for k, v in unc.items()) / sum(unc.values())) if unc else None,
                     "int32_avg_cycles": (sum(v * w[k] for k, v in ints.items()) / sum(ints.values())) if ints else None,
                     "top_opcodes": dict(ops.most_common(40))}
        print("== %s" % name)
        print("   %d instructions, %d VALU (%.0f issue cycles with the measured weights%s)" % (sum(total.values()), valu, cyc, "" if src else " — profiles/valu_peak.json missing, default 2"))
        for k, v in total.most_common():
            print("   %-16s %6d  x %4.1f cyc" % (k, v, w[k]) if k in VALU else "   %-16s %6d" % (k, v))
        if a.blocks:
            big = sorted(blocks, key=lambda bc: -sum(bc[1].values()))[:12]
            for lab, c in big:
                print("   block %-12s %5d instr: %s" % (lab, sum(c.values()), ", ".join("%s %d" % kv for kv in c.most_common(6))))
    if a.json:
        json.dump({"_note": "static counts from `hipcc -S` with the product's flags; weights = cycles per wave-instruction per SIMD at 5 waves/SIMD (profiles/valu_peak.json)",
                   "kernels": res}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
