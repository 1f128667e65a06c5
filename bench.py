#!/usr/bin/env python3
"""bench.py — Mrays/s of the HIP wavefront integrator on BASELINE.json's configs[1]
(Cornell + monkey_968.obj, 1920x1080, 64 spp, 8 bounces), one JSON line on rank 0.

A "step" is one full render of the workload: clear the accumulation buffer, trace `spp` progressive
frames (frame numbers 1..spp, resetBuffer = 0), and — for N > 1 — sum-reduce the per-rank framebuffers
to rank 0.  Scene, BVH and path buffers are resident in HBM before the timed region.
Rays are counted exactly (one per hitScene invocation) by the device.
N > 1: pixels are sharded across ranks in tiles; `--scaling weak` (default) multiplies spp by N so that the rays
per GPU stay fixed, `--scaling strong` keeps the total spp (BASELINE configs[3]/[4] are fixed totals).

What the line carries besides the contract's fields (N = 1):
  roofline      for the kernel with the largest share of the timed step: its launch time is measured live with HIP events
                on the context's stream inside the timed region; VALU-busy cycles and HBM bytes per launch come from
                rocprofv3 --pmc passes that THIS run makes over the same workload before it touches the GPU itself
                (separate passes for the SQ counters, FETCH_SIZE and WRITE_SIZE; gfx950 correction 2*FETCH_SIZE + WRITE_SIZE).
                frac = achieved / peak <= 1 in the unit of the bound it names; the per-kernel table is next to it.
  configs       the other single-GPU configuration north_star sets its target on (configs[2], 871,414 triangles),
                timed by the same procedure (2 steps).
  setup_ms      scene set-up: BVH build (host threads / GPU / the reference's algorithm in single-threaded JavaScript),
                validation + digests + upload.
  cpu_baseline  the CPU oracle on all host cores and on one, plus the JavaScript BVH build.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD = 1024           # 256 CUs x 4 SIMDs
REF_DRAGON_MPATHS = 33.5  # reference/benchmarks.txt:18-20: 62 fps x 900x600 px, 1 path per pixel and frame (derived in BASELINE.md §1)
KERNELS = ("k_generate", "k_tail", "k_bvh", "k_shade", "k_accumulate")
TIMING_MODE = {"k_bvh": 2, "k_shade": 3, "k_generate": 4, "k_accumulate": 5, "k_tail": 6}
MS_KEY = {"k_bvh": "bvh_ms", "k_shade": "shade_ms", "k_generate": "generate_ms", "k_accumulate": "accumulate_ms", "k_tail": "tail_ms"}
LAUNCH_KEY = {"k_bvh": "intersect_launches", "k_shade": "shade_launches", "k_generate": "generate_launches", "k_accumulate": "accumulate_launches",
              "k_tail": "tail_launches"}
SPP = {"c2": 64, "c3": 256, "c4": 512, "c5": 1024}
FETCH_MULT = {"k_bvh": 1.0}  # bytes per FETCH_SIZE byte, calibrated per access pattern (profiles/fetch_calib.json); streams: 2.0
# The rate at which the chip's 256 L1 / texture-addresser paths serve per-lane gathers of 64-byte records (4 x global_load_dwordx4 per lane, every lane
# its own record — k_bvh's fetch), measured by tools/gather_probe.hip (profiles/r03_gather_probe.json): 2.75-2.8 clocks per record per CU whether the table
# sits in L1 (8 KB), in L2 (2 MB), is read lane-wise or quad-cooperatively, at 5 or 8 waves per SIMD; proportional to the lanes taking part
# (profiles/r03_gather_probe2.json).  The same records from the Infinity Cache: 65 G/s, from HBM: 54 G/s.
GATHER_PEAK_RECORDS_PER_S = 223.5e9  # fallback; gather_peak() reads the probe's file
GUIDE_MAX_CLOCK_GHZ = 2.4  # MI355X_MICROARCH.md "Max clock"
N_CUS = 256
# Is there a counter that tells DRAM from Infinity-Cache (MALL) traffic?  `rocprofv3 --list-avail` on the MI355X box (gfx950, ROCm 7.2) was searched in round 5
# (profiles/r05_list_avail_mall.txt): no MALL hit / miss counter is exposed; TCC_EA0_{RD,WR}REQ_DRAM count requests DESTINED for DRAM (as opposed to GMI / IO) at the L2's
# EA interface, upstream of the memory-side cache, where a MALL hit and a DRAM access look the same.
MALL_NOTE = ("not separable on this box: rocprofv3 --list-avail (gfx950, ROCm 7.2; profiles/r05_list_avail_mall.txt) exposes no Infinity-Cache hit / miss counter — "
             "TCC_EA0_*REQ_DRAM count requests destined for DRAM at the L2's EA interface, upstream of the memory-side cache — so the fabric bytes above are an upper bound on DRAM bytes")


def gather_peak():
    """(records/s, clocks per record per CU, source): the per-lane 64-byte-record gather rate from the committed probe run
    (profiles/r03_gather_probe.json, tools/gather_probe.hip), expressed at the guide's 2.4 GHz — the probe measures clocks per record."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r03_gather_probe.json")))
        clk = [c["cycles_per_record_per_cu"] for c in d["cases"] if c.get("mode") == "lane" and c.get("table_kb", 1 << 30) <= 2048]
        cpr = sorted(clk)[len(clk) // 2]
        return N_CUS * GUIDE_MAX_CLOCK_GHZ * 1e9 / cpr, cpr, "profiles/r03_gather_probe.json: median of the lane-wise cases with tables of 8 KB .. 2 MB, i.e. L1- and L2-resident"
    except Exception:
        return GATHER_PEAK_RECORDS_PER_S, N_CUS * GUIDE_MAX_CLOCK_GHZ * 1e9 / GATHER_PEAK_RECORDS_PER_S, "built-in constant (profiles/r03_gather_probe.json unreadable)"


def static_mix():
    """Per kernel: what the cost model charges an instruction its counters cannot classify and an integer one — the kernel's own static mix
    (profiles/r04_isa_histogram.json, tools/isa_histogram.py: moves / selects at 2 cycles, compares / min / max / f64 / division helpers at 4; integer add / logic
    at 2, shifts / multiplies at 4).  Missing file or kernel: 4 and 4, the upper bound round 3 reported."""
    mix = {}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r04_isa_histogram.json")))["kernels"]
        pick = {"k_shade": ("k_shade6<false, false>", "k_shade<true, true, false, false>"), "k_bvh": ("k_bvh2<false, true, false>",), "k_generate": ("k_generate<false>",),
                "k_tail": ("k_tail6<false, true>", "k_tail<false, false, false, true>"), "k_accumulate": ("k_accumulate",)}
        for k, names in pick.items():
            for name in names:
                e = next((v for n, v in d.items() if name in n), None)
                if e and e.get("unclassified_avg_cycles"):
                    mix.setdefault(k, {})[name] = (e["unclassified_avg_cycles"], e.get("int32_avg_cycles") or 4.0)
    except Exception:
        pass
    return mix


def alg_bytes(st):
    """SURVEY.md §8d algorithmic bytes of hitScene in the reference's layouts."""
    return 48 * st["node_visits"] + 96 * st["tri_tests"] + 64 * st["mat_fetches"] + 32 * st["sphere_tests"] + 80 * st["quad_tests"]


class TimedNative:
    """NativeHost that remembers how long the BVH build took and what it was given (for the other builders' timings)."""

    def __init__(self, native):
        self.native, self.bvh_ms, self.boxes = native, 0.0, None

    def build_bvh(self, bmin, bmax, prim_type=2):
        t = time.perf_counter()
        r = self.native.build_bvh(bmin, bmax, prim_type)
        self.bvh_ms += (time.perf_counter() - t) * 1e3
        self.boxes = (np.ascontiguousarray(bmin, np.float64), np.ascontiguousarray(bmax, np.float64))
        return r

    def __getattr__(self, k):
        return getattr(self.native, k)


def make_workload(pkg, name, args):
    """Scene buffers + parameters of one BASELINE configuration."""
    wl = {"name": name, "cam": "cornell", "extra": {}, "W": args.width, "H": args.height, "bounces": args.bounces, "setup": {}}
    sah = args.bvh == "sah"
    tri_kw = {"n_tris": args.tris} if args.tris else {}  # experiments only: the configurations' own triangle counts are the default
    t0 = time.perf_counter()
    native = TimedNative(pkg.ptmi.NativeHost())
    if name == "c2":
        if sah:
            raise SystemExit("--bvh sah needs a workload built through the Scene API (c3, c4, c5)")
        wl["buffers"] = pkg.scenes.golden_buffers("c2")  # reference-generated buffers of configs[1] (tests/golden)
        wl["label"], wl["stack"] = "configs[1]: Cornell + monkey_968.obj (967 tris)", args.stack_size or 20
    else:
        # --bvh sah: the scene goes up unbuilt and the tree is made where the triangles are (ptmi_build_scene_bvh_sah, make_context); its depth is
        # known only then, and the SAH trees of these meshes are 27-31 deep: STACK_SIZE has to exceed that or the reference's stack-full abort (Q7) cuts rays
        def buffers(sc):
            return sc.buffers_unbuilt() if sah else sc.buffers(native=native)
        wl["device_sah"] = sah
        if name == "c3":
            wl["buffers"] = buffers(pkg.scenes.c3_scene(**tri_kw))  # procedural stand-in, 871,414 tris
            wl["label"], wl["stack"] = "configs[2]: Cornell + dragon-class mesh (871,414 tris, procedural stand-in for stanfordDragon.obj)", args.stack_size or (40 if sah else 24)
        elif name == "c4":
            wl["buffers"] = buffers(pkg.scenes.c4_scene(**tri_kw))
            wl["label"], wl["stack"], wl["cam"] = "configs[3]: sponza-class interior (262,267 tris, procedural stand-in for sponzaAtrium.obj), camera inside", args.stack_size or (40 if sah else 24), "interior"
        else:
            wl["buffers"] = buffers(pkg.scenes.c5_scene(**tri_kw))
            wl["label"], wl["stack"], wl["extra"] = "configs[4]: Cornell + buddha-class glass mesh (1,087,716 tris, procedural stand-in for buddha.obj), importance sampling", args.stack_size or (40 if sah else 24), dict(importance_sampling=1)
            if wl["bounces"] == 8:
                wl["bounces"] = 16
    wl["view"] = pkg.scenes.camera_view(*pkg.scenes.CAMERAS[wl["cam"]])
    wl["native"] = native
    if name != "c2" and not sah:
        wl["setup"]["host_scene_and_packing_ms"] = (time.perf_counter() - t0) * 1e3 - native.bvh_ms  # Python mirror of lib/scene.js (+ the procedural mesh): not the product
        wl["setup"]["bvh_build_native_host_ms"] = native.bvh_ms
    return wl


def make_context(pkg, wl, device, args):
    ctx = pkg.Context(device)
    t = time.perf_counter()
    ctx.upload_scene(wl["buffers"])
    ms = (time.perf_counter() - t) * 1e3
    if wl.get("device_sah") and not wl.get("bvh_host"):
        t = time.perf_counter()
        ctx.build_scene_bvh(sah=True)
        wl["setup"]["build_scene_bvh_sah_device_ms"] = (time.perf_counter() - t) * 1e3  # boxes + binned-SAH build + triangles into leaf order, all on the GPU
        info = ctx.scene_bvh_info()
        wl["setup"]["sah_tree"] = {"nodes": info["nodes"], "depth": info["depth"]}
        if info["depth"] >= wl["stack"]:
            raise SystemExit("--stack-size %d does not exceed the SAH tree's depth %d" % (wl["stack"], info["depth"]))
        # the oracle (cpu_baseline), the compulsory-bytes figure and any further context of this run want the tree and the reordered triangles on the host
        n_tri = np.asarray(wl["buffers"]["triangles"]).size // 24
        wl["buffers"] = dict(wl["buffers"], bvh=ctx.read_scene_buffer("bvh", info["nodes"]).reshape(-1), triangles=ctx.read_scene_buffer("triangles", n_tri).reshape(-1))
        wl["bvh_host"] = True
    t = time.perf_counter()
    ctx.set_params(max_bounces=wl["bounces"], frames_in_flight=args.frames_in_flight, stack_size=wl["stack"], **wl["extra"])
    ctx.resize(wl["W"], wl["H"])
    ctx.prepare()  # validation + digests + upload, synchronous
    wl["setup"]["upload_validate_digests_ms"] = ms + (time.perf_counter() - t) * 1e3
    return ctx


# ---------------------------------------------------------------------------------------------------- rocprofv3 passes
PMC_PASSES = (
    ("sq", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64"),
    ("fetch", "FETCH_SIZE SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32"),
    ("write", "WRITE_SIZE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT TA_TA_BUSY_sum GRBM_GUI_ACTIVE"),  # (TA / GRBM: blocks of their own, they ride along)
)
# Issue cost of a wave64 instruction on one SIMD, measured by tools/valu_peak.hip (profiles/valu_peak.json, throughput at 4-8 waves
# per SIMD): 2 cycles for f32 add/mul/fma (and mov, and/or/xor, integer add), 4 for min/max, compares, shifts, conversions, integer
# multiplies, packed f32, f64 arithmetic and the v_div_* helpers, 8 for f32 transcendentals, 16 for v_rcp_f64.  The counters
# can tell the 2-, 8- and 16-cycle classes apart; everything else is charged 4 (moves and integer adds too: an over-estimate).
def issue_cycles(p, L, other_at=4.0, int_at=4.0):
    """Issue cycles per launch of the kernel's own instruction mix: 2 per f32 add / mul / fma, 8 per f32 transcendental, 16 per v_rcp_f64, 4 per conversion
    (counted classes), `int_at` per integer instruction (SQ_INSTS_VALU_INT32: add / logic issue in 2, shifts and multiplies in 4) and `other_at` per
    instruction the counters cannot classify (moves / selects 2; compares, min / max, f64 arithmetic, division helpers 4).  4 and 4 = the upper bound;
    the kernel's static mix (static_mix) = the model."""
    fast = (p.get("SQ_INSTS_VALU_ADD_F32", 0.0) + p.get("SQ_INSTS_VALU_MUL_F32", 0.0) + p.get("SQ_INSTS_VALU_FMA_F32", 0.0)) / max(p.get("launches_fetch", L), 1)
    n, t32, t64 = p["SQ_INSTS_VALU"] / L, p.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / L, p.get("SQ_INSTS_VALU_TRANS_F64", 0.0) / L
    i32 = p.get("SQ_INSTS_VALU_INT32", 0.0) / max(p.get("launches_write", L), 1)
    cvt = p.get("SQ_INSTS_VALU_CVT", 0.0) / max(p.get("launches_write", L), 1)
    fast, i32, cvt = min(fast, n), min(i32, n), min(cvt, n)
    other = max(0.0, n - fast - t32 - t64 - i32 - cvt)
    return 2.0 * fast + 8.0 * t32 + 16.0 * t64 + 4.0 * cvt + int_at * i32 + other_at * other


def pmc_child(args):
    """Runs under rocprofv3 (started by pmc_passes): the same workload, one warm-up step and one profiled-as-everything step,
    timed-pass kernels only.  No torch, no oracle: nothing but the context."""
    pkg = entry._load_pkg()
    wl = make_workload(pkg, args.workload, args)
    ctx = make_context(pkg, wl, 0, args)
    spp = args.spp or SPP[args.workload]
    for _ in range(2):
        ctx.clear()
        ctx.render(wl["view"], 1, spp)
        ctx.synchronize()
    ctx.close()


def pmc_passes(args, workload, log):
    """Per-kernel counter sums of the timed-pass kernels of `workload`, from three rocprofv3 --pmc runs of this script in child mode.
    Returns {kernel: {counter: value per launch, 'launches': n, 'pmc_ms_per_launch': ...}} or (None, reason)."""
    rocprof = shutil.which("rocprofv3")
    if not rocprof:
        return None, "rocprofv3 not on PATH"
    out = {}
    t_all = time.perf_counter()
    for tag, ctrs in PMC_PASSES:
        d = tempfile.mkdtemp(prefix="ptmi_pmc_%s_" % tag, dir="/tmp")
        cmd = [rocprof, "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                                                  "--workload", workload, "--width", str(args.width), "--height", str(args.height), "--bounces", str(args.bounces),
                                                  "--frames-in-flight", str(args.frames_in_flight), "--bvh", args.bvh, "--stack-size", str(args.stack_size), "--tris", str(args.tris),
                                                  "--spp", str(args.spp if workload == args.workload else 0)]
        try:
            # (PTMI_PLACEMENT_TRIES=1: no placement search under the profiler — its dry runs are launches of the very kernels being counted)
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", PTMI_PLACEMENT_TRIES="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=args.pmc_timeout)
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 pass '%s' timed out" % tag
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            log.append(r.stdout[-1500:])
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 pass '%s' failed (rc %d)" % (tag, r.returncode)
        seen = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].replace("ptmi::", "").replace("void ", "").split("(")[0].strip().split("<")[0]
                k = {"k_bvh2": "k_bvh", "k_shade6": "k_shade", "k_tail6": "k_tail"}.get(k, k)  # the traversal kernel's second edition and the 80-VGPR build of k_shade (round 3) are the same pipeline stages
                if k not in KERNELS:
                    continue
                rec = out.setdefault(k, {})
                rec[row["Counter_Name"]] = rec.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                seen.setdefault((tag, k), {})[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        for (t, k), disp in seen.items():
            out[k]["launches_" + t] = len(disp)
            out[k]["pmc_ms_" + t] = sum(disp.values())
        shutil.rmtree(d, ignore_errors=True)
    out["_seconds"] = time.perf_counter() - t_all
    return out, None


def kernel_table(split, steps_in_split, pmc, shade_variant=None):
    """Per-kernel figures: live ms per step (HIP events, uncounted kernels), and — from this run's PMC passes — the VALU-issue fraction by the cost model
    and by rocprof's own definition, HBM bytes per launch, wait fraction, active lanes.  No clamps: a model that says > 1 says so."""
    tab = {}
    total = sum(split[MS_KEY[k]] for k in KERNELS) or 1.0
    mix = static_mix()
    for k in KERNELS:
        ms, n = split[MS_KEY[k]] / steps_in_split, max(split[LAUNCH_KEY[k]], 1) / steps_in_split
        e = {"ms_per_step": ms, "launches_per_step": n, "share_of_kernel_time": split[MS_KEY[k]] / total, "avg_launch_ms": ms / n if n else None}
        p = (pmc or {}).get(k)
        if p and p.get("launches_sq"):
            L = p["launches_sq"]
            cyc = p["SQ_BUSY_CYCLES"] / 32.0 / L                      # kernel cycles per launch (the counter is summed over 8 XCDs x 4 SEs)
            clock = cyc / (p["pmc_ms_sq"] / L * 1e-3) if p["pmc_ms_sq"] else 0.0
            km = mix.get(k, {})
            other_at, int_at = km.get(shade_variant) or (next(iter(km.values())) if km else (4.0, 4.0))
            busy = issue_cycles(p, L, other_at, int_at) / N_SIMD      # issue cycles of this kernel's instruction mix, per SIMD and launch (static-mix model)
            busy_ub = issue_cycles(p, L) / N_SIMD                     # the same with every unclassified / integer instruction at 4 cycles (upper bound)
            busy4 = 4.0 * p["SQ_ACTIVE_INST_VALU"] / N_SIMD / L       # rocprof's VALUBusy numerator: SQ_ACTIVE_INST_VALU counts quad-cycles, at least one per instruction
            live_cyc = (ms / n * 1e-3) * clock if n and clock else cyc
            lanes = p["SQ_THREAD_CYCLES_VALU"] / (64.0 * p["SQ_ACTIVE_INST_VALU"]) if p["SQ_ACTIVE_INST_VALU"] else None
            frac = busy / live_cyc if live_cyc else None
            ghz = clock / 1e9
            e.update({
                "valu_instr_per_launch": p["SQ_INSTS_VALU"] / L,
                # THE fraction: busy issue cycles of the kernel's own mix / (1024 SIMDs x launch time x the guide's 2.4 GHz) — DVFS loss counts against the kernel
                "valu_busy_frac_at_2p4_ghz": (frac * ghz / GUIDE_MAX_CLOCK_GHZ) if (frac is not None and clock) else None,
                "valu_busy_frac_at_pass_clock": frac,
                "valu_busy_frac_upper_bound_at_pass_clock": busy_ub / live_cyc if live_cyc else None,
                # rocprof's own derived metrics for the same launches (its VALUBusy = 4 x SQ_ACTIVE_INST_VALU / SIMDs / cycles; VALUUtilization = active lanes):
                # over-counts every 2-cycle instruction as a 4-cycle quad (tools/valu_busy_calib.sh, profiles/r04_valu_busy_calib.json)
                "rocprof_valu_busy": busy4 / live_cyc if live_cyc else None,
                "rocprof_valu_utilization": lanes,
                "lane_weighted_frac_at_2p4_ghz": (frac * ghz / GUIDE_MAX_CLOCK_GHZ * lanes) if (frac is not None and clock and lanes) else None,
                "avg_cycles_per_valu_instr": issue_cycles(p, L, other_at, int_at) / (p["SQ_INSTS_VALU"] / L) if p["SQ_INSTS_VALU"] else None,
                "cost_model": {"unclassified_at": other_at, "int32_at": int_at},
                "wave_wait_frac": p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"] if p["SQ_WAVE_CYCLES"] else None,
                "active_lane_frac": lanes,
                "clock_ghz_in_pmc_pass": ghz,
                "pmc_pass_avg_launch_ms": p["pmc_ms_sq"] / L,
            })
        if p and p.get("launches_write") and p.get("GRBM_GUI_ACTIVE") and p.get("TA_TA_BUSY_sum") is not None:
            # the CU's vector-memory path (texture addresser / L1 <-> registers: a dwordx4 wave instruction occupies it for 64 clocks, 16 bytes per clock and CU):
            # busy cycles per CU over the kernel's cycles (TA_TA_BUSY is summed over the CUs, GRBM_GUI_ACTIVE over the 8 XCDs)
            e["ta_busy_frac"] = (p["TA_TA_BUSY_sum"] / (N_SIMD / 4.0)) / (p["GRBM_GUI_ACTIVE"] / 8.0)
        if p and p.get("launches_fetch") and p.get("launches_write"):
            fb = p["FETCH_SIZE"] * 1024.0 / p["launches_fetch"]         # counter unit: KB
            wb = p["WRITE_SIZE"] * 1024.0 / p["launches_write"]
            # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) but exactly the bytes of 64-byte
            # record gathers (tools/fetch_calib.hip on an 8 GiB table, profiles/fetch_calib.json: stream 2.000, gather 1.000, stores 1.000).
            # k_bvh's reads are record gathers plus a thin flag scan; the other kernels stream their state.
            mult = FETCH_MULT.get(k, 2.0)
            # FETCH_SIZE / WRITE_SIZE count at the L2 <-> fabric boundary: reads the Infinity Cache (MALL) serves are in them (MI355X_MICROARCH.md, HBM), so these
            # are FABRIC bytes — an upper bound on DRAM bytes; for a scene that fits the 256 MB MALL (every BASELINE configuration's digests) the DRAM share of
            # k_bvh's gathers is far smaller.  `fabric_frac_of_hbm_peak` divides them by the 8 TB/s HBM peak all the same: it can only over-state how HBM-bound a kernel is.
            e["fabric_bytes_per_launch"] = mult * fb + wb
            e["fabric_bytes_per_launch_if_all_reads_were_streams"] = 2.0 * fb + wb
            e["fabric_frac_of_hbm_peak"] = e["fabric_bytes_per_launch"] / (ms / n * 1e-3) / (HBM_PEAK_GBS * 1e9) if n and ms else None
        tab[k] = e
    return tab


def roofline_of(tab, dom, timed_ms_per_launch, timed_launches, gather_records_per_launch=None):
    """The contract's roofline object for the dominant kernel `dom`, in the unit of the bound it comes closest to: VALU issue (busy issue cycles of
    the kernel's own instruction mix against 1024 SIMDs at the guide's 2.4 GHz), HBM (measured fabric bytes / 8 TB/s) or — k_bvh — the L1 gather
    path (64-byte records fetched per second / the rate tools/gather_probe.hip measures for that access pattern, at 2.4 GHz)."""
    e = tab[dom]
    vf, hf = e.get("valu_busy_frac_at_2p4_ghz"), e.get("fabric_frac_of_hbm_peak")
    r = {"kernel": dom, "avg_launch_ms": timed_ms_per_launch, "launches": timed_launches, "share_of_kernel_time": e["share_of_kernel_time"],
         "traffic": e.get("fabric_bytes_per_launch"),
         "traffic_is": "bytes per launch over the fabric (2 x FETCH_SIZE for streams, 1 x for 64-byte gathers, + WRITE_SIZE: profiles/fetch_calib.json), Infinity-Cache hits included"}
    gf = None
    gpeak, gcpr, gsrc = gather_peak()
    if dom == "k_bvh" and gather_records_per_launch and timed_ms_per_launch:
        gf = gather_records_per_launch / (timed_ms_per_launch * 1e-3) / gpeak
    if vf is None and hf is None and gf is None:
        r.update({"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None})
        return r
    # re-base the fractions on the launch time measured inside the timed region
    scale = (e["avg_launch_ms"] / timed_ms_per_launch) if (timed_ms_per_launch and e["avg_launch_ms"]) else 1.0
    vf = vf * scale if vf is not None else None
    hf = hf * scale if hf is not None else None
    tf = e.get("ta_busy_frac") if dom != "k_bvh" else None  # (k_bvh's reading of the same path is the record model below: exact counters against the probe's rate)
    if tf is not None:
        tf = (tf * e["clock_ghz_in_pmc_pass"] / GUIDE_MAX_CLOCK_GHZ if e.get("clock_ghz_in_pmc_pass") else tf) * scale  # against the guide's clock and the timed launch, like the VALU fraction
    best = max((x for x in (vf, hf, gf, tf) if x is not None))
    if tf is not None and tf == best:
        cus = N_SIMD / 4.0
        r.update({"bound": "vector_memory_path", "achieved": tf * cus * GUIDE_MAX_CLOCK_GHZ * 16.0, "peak": cus * GUIDE_MAX_CLOCK_GHZ * 16.0, "unit": "GB/s", "frac": tf,
                  "frac_is": "TA_TA_BUSY per CU / the kernel's cycles, re-based on the guide's 2.4 GHz: how busy the CU's vector-memory path (texture addresser / L1 <-> registers) is. "
                             "A dwordx4 wave instruction occupies it for 64 clocks — 16 bytes per clock and CU, 9.8 TB/s over 256 CUs at 2.4 GHz (tools/gather_probe2.hip: "
                             "65 clocks per contiguous dwordx4 wave load) — and the 120 bytes of path state a ray brings and takes cross it once each per bounce.  `achieved` = frac x "
                             "that rate (a busy fraction in bytes' clothing: partial-lane and gather instructions occupy the path for more clocks than their bytes).  The busiest "
                             "of this kernel's resources; next to it the VALU-issue fraction by the cost model and the fabric bytes",
                  "peak_definition": "256 CUs x 16 B per clock x 2.4 GHz"})
    elif gf is not None and gf == best:
        gbs = gather_records_per_launch * 64.0 / (timed_ms_per_launch * 1e-3) / 1e9
        r.update({"bound": "l1_gather", "achieved": gbs, "peak": gpeak * 64.0 / 1e9, "unit": "GB/s", "frac": min(1.0, gf), "frac_unclamped": gf,
                  "peak_definition": "64-byte BVH / triangle records fetched per lane (pair fetches = reference node visits below the root / 2, plus triangle tests; exact counters) "
                                     "x 64 B / launch time, against the rate the 256 CUs' L1 / texture-addresser paths deliver such per-lane gathers at: %.2f clocks per record per CU "
                                     "at 2.4 GHz = %.1f G records/s (%s) — the same from L1, from L2, lane-wise or quad-cooperative, at 5 or 8 waves per SIMD.  The probe makes every lane "
                                     "fetch a DISTINCT record; lanes that share a record (coherent rays, e.g. primary rays) are served by one fetch and counted per lane here, so the "
                                     "fraction over-counts them and can pass 1 (clamped in `frac`, raw in `frac_unclamped`).  Evidence that this path, not HBM / latency / VALU, bounds "
                                     "k_bvh: its rate per node visit does not move with the scene's size from 1 MB to 134 MB of digests (profiles/r03_size_sweep.txt) nor with "
                                     "occupancy from 18 to 26 waves per CU (DESIGN.md §4)" % (gcpr, gpeak / 1e9, gsrc)})
    elif hf is not None and hf == best:
        gbs = e["fabric_bytes_per_launch"] / (timed_ms_per_launch * 1e-3) / 1e9
        r.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
    else:
        rate = e["valu_instr_per_launch"] / (timed_ms_per_launch * 1e-3) / 1e9
        peak = N_SIMD * GUIDE_MAX_CLOCK_GHZ / e["avg_cycles_per_valu_instr"] if e.get("avg_cycles_per_valu_instr") else None
        r.update({"bound": "valu_issue", "achieved": rate, "peak": peak, "unit": "G wave-instr/s", "frac": vf,
                  "frac_is": "busy issue cycles of the kernel's own instruction mix / (1024 SIMDs x launch time x 2.4 GHz, the guide's clock): cycles per class measured by "
                             "tools/valu_peak.hip (2 f32 add/mul/fma, 8 f32 transcendental, 16 v_rcp_f64, 4 conversions), integer and unclassified instructions at this "
                             "kernel's static mix (cost_model; profiles/r04_isa_histogram.json).  Not clamped.  Next to it: the same at the clock the PMC pass ran at, the "
                             "all-unclassified-at-4 upper bound, rocprof's own VALUBusy (counts every instruction as >= 4 cycles: a kernel of 2-cycle instructions reads "
                             "2x too busy, profiles/r04_valu_busy_calib.json) and the lane-weighted fraction",
                  "peak_definition": "1024 SIMDs x 2.4 GHz / (average issue cycles per wave64 instruction of this kernel = %.2f)" % (e.get("avg_cycles_per_valu_instr") or 0.0)})
    if r.get("bound") in ("valu_issue", "vector_memory_path") and r.get("kernel") == "k_shade":
        r["bound_probes"] = ("round 5 (profiles/NOTES_r05.md §3): a quarter of this kernel's vector instructions removed three bit-exact ways (direction-binned flush passes, "
                             "hit_quad for axis-aligned quads, a quad's Lambertian frame from a table) changes its time by 0 +- 2 %; 16 more bytes per kept ray (+12 % traffic) cost "
                             "6-8 %; most of its traffic removed (paths kept in their lanes) with a fifth more instructions costs 7 %.  `frac` is the busiest resource, not a "
                             "binding bound: the kernel waits on the round trips of each wave's dependent chain at six waves per SIMD")
    if max(vf or 0.0, hf or 0.0, gf or 0.0) < 0.5:
        r["bound_note"] = "no resource is busy half the time: the kernel waits on memory latency (wave_wait_frac %.2f)" % (e.get("wave_wait_frac") or 0.0)
    r["valu_busy_frac_at_2p4_ghz"], r["fabric_frac_of_hbm_peak"], r["l1_gather_frac"] = vf, hf, gf
    if e.get("ta_busy_frac") is not None:
        r["ta_busy_frac"] = e["ta_busy_frac"]
        r["ta_busy_frac_is"] = "TA_TA_BUSY per CU / the kernel's cycles: how busy the CU's vector-memory path (L1 <-> registers, 16 bytes per clock and CU) is with this kernel's loads and stores"
    for k in ("valu_busy_frac_at_pass_clock", "valu_busy_frac_upper_bound_at_pass_clock", "rocprof_valu_busy", "rocprof_valu_utilization", "lane_weighted_frac_at_2p4_ghz",
              "clock_ghz_in_pmc_pass", "cost_model"):
        if e.get(k) is not None:
            r[k] = e[k] * scale if k.startswith(("valu_busy", "rocprof_valu_busy", "lane_weighted")) else e[k]
    return r


# ---------------------------------------------------------------------------------------------------- one GPU measurement
class _NoDist:
    """Stand-in for webgpu_path_tracer_amd.dist when there is one process (no torch import at all: a fresh box pays 1-2 minutes for it)."""
    TILE_PIXELS = 4032

    @staticmethod
    def barrier():
        pass

    @staticmethod
    def all_reduce_scalar(value, op="sum"):
        return value


def measure(pkg, torch, pdist, ctx, wl, spp, steps, warmup, reduce_fn):
    """reduce_fn: the step's collective (None on one GPU) — torch.distributed's reduce of the bound framebuffer tensor when the driver starts one
    process per GPU, ptmi_reduce_framebuffer (ncclReduce inside the library) when one process drives all GPUs."""
    view = wl["view"]
    coll = [0.0]  # seconds this rank spent in the step's collective (the wait for the slowest rank included)

    def step():
        ctx.clear()
        ctx.render(view, 1, spp)
        ctx.synchronize()
        if reduce_fn is not None:
            t = time.perf_counter()
            reduce_fn()
            coll[0] += time.perf_counter() - t

    for _ in range(warmup):
        step()
    # per-kernel split of one step (HIP events around every launch; the kernels are the timed ones, not the counted variants)
    ctx.synchronize()
    ctx.reset_stats()
    ctx.set_timing(1)
    step()
    split = ctx.stats()
    ctx.set_timing(0)
    dom = max(KERNELS, key=lambda k: split[MS_KEY[k]])
    # the timed region: events only around the dominant kernel (every event is a stream marker)
    ctx.reset_stats()
    ctx.set_timing(TIMING_MODE[dom])
    if torch is not None:
        torch.cuda.synchronize()
    pdist.barrier()
    coll[0] = 0.0
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        step()  # (ends with ctx.synchronize(): the step's wall time)
        step_ms.append((time.perf_counter() - ts) * 1e3)
    ctx.synchronize()
    if torch is not None:
        torch.cuda.synchronize()
    pdist.barrier()
    dt = time.perf_counter() - t0
    coll_ms = coll[0] / max(steps, 1) * 1e3
    st = ctx.stats()
    ctx.set_timing(0)
    # exact work counters of one step (counted variants of the same kernels, untimed)
    ctx.reset_stats()
    ctx.set_counters(True)
    ctx.clear()
    ctx.render(view, 1, spp)
    cst = ctx.stats()
    ctx.set_counters(False)
    assert cst["rays"] * steps == st["rays"], "ray count differs between the counted and the timed pass"
    return {"dt": dt, "st": st, "split": split, "cst": cst, "dom": dom, "collective_ms_per_step": coll_ms, "step_ms": step_ms}


def measure_short(torch, pdist, ctx, wl, spp, steps, reduce_fn):
    """The other scaling mode's figure for the same line (N > 1): a few steps, no per-kernel split, no counted pass."""
    def step():
        ctx.clear()
        ctx.render(wl["view"], 1, spp)
        ctx.synchronize()
        if reduce_fn is not None:
            reduce_fn()

    step()
    ctx.synchronize()
    ctx.reset_stats()
    if torch is not None:
        torch.cuda.synchronize()
    pdist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    if torch is not None:
        torch.cuda.synchronize()
    pdist.barrier()
    return time.perf_counter() - t0, ctx.stats()


def cpu_baseline(pkg, wl, spp, args):
    from oracle import ptm_oracle

    W, H, view, buffers = wl["W"], wl["H"], wl["view"], wl["buffers"]
    kw = dict(max_bounces=wl["bounces"], stack_size=wl["stack"], **wl["extra"])
    cores = min(ptm_oracle.max_threads(), args.cpu_threads)

    def sample(threads, seconds, rows):
        # rows = how many image rows the probe frame covers (the single-thread sample is bounded by a pixel window)
        pr = (0, -1) if rows >= H else (0, rows * W)
        t = time.perf_counter()
        _, ost = ptm_oracle.render(buffers, W, H, view, 1, 1, threads=threads, pixel_range=pr, **kw)
        one = time.perf_counter() - t
        frames = int(max(1, min(spp, seconds / max(one, 1e-3))))
        t = time.perf_counter()
        _, ost = ptm_oracle.render(buffers, W, H, view, 1, frames, threads=threads, pixel_range=pr, **kw)
        cdt = time.perf_counter() - t
        what = "same scene and camera, %dx%d%s, frames 1..%d of %d (%d rays), scalar f32 oracle%s" % (
            W, H, "" if rows >= H else ", pixel rows 0..%d" % (rows - 1), frames, spp, ost["rays"],
            " with OpenMP over pixels on %d of the box's %d logical CPUs (--cpu-threads: a 1-GPU box's CPU share)" % (threads, os.cpu_count() or 0) if threads > 1 else ", one thread")
        return ost["rays"] / cdt / 1e6, what

    v, what = sample(cores, args.cpu_seconds, H)
    out = {"value": v, "unit": "Mrays/s", "cores": cores, "kind": "port", "sample": what}
    v1, what1 = sample(1, max(2.0, args.cpu_seconds * 0.6), max(1, H // 16))
    out["single_thread"] = {"value": v1, "unit": "Mrays/s", "cores": 1, "sample": what1}
    return out


def js_traversal(wl, n_rays=300000):
    """north_star's literal CPU baseline — "the reference's single-threaded JS BVH traversal" — does not exist in the reference (its only traversal is WGSL,
    shaders/hitRay.wgsl:42-110).  This is that WGSL restated in JavaScript (js/hit_scene.mjs: hitScene with hit_sphere / hit_quad / hit_triangle / hit_aabb, f32 by
    Math.fround; bit-identical hit records to the oracle's, tests/test_js_host.py), timed single-threaded under Node on this box over a bounded sample of rays
    against the workload's own buffers: half of them leave the camera's neighbourhood towards the scene, half start anywhere in the room in random directions."""
    node = shutil.which("node") or shutil.which("nodejs")
    b = wl["buffers"]
    if not node or len(b.get("bvh", ())) == 0:
        return None
    d = tempfile.mkdtemp(prefix="ptmi_jstrav_", dir="/tmp")
    try:
        for k in ("spheres", "quads", "triangles", "meshes", "transforms", "materials", "bvh"):
            np.ascontiguousarray(b[k]).tofile(os.path.join(d, "w_%s.bin" % k))
        rng = np.random.default_rng(17)
        h = n_rays // 2
        o = rng.uniform(-0.3, 0.3, (h, 3)) + np.array([0, -0.1, 2.4])
        dr = rng.uniform(-1.0, 1.0, (h, 3)) * np.array([1.2, 1.0, 0.9]) - o
        dr /= np.linalg.norm(dr, axis=1, keepdims=True)
        di = rng.normal(0, 1, (h, 3))
        di /= np.linalg.norm(di, axis=1, keepdims=True)
        rays = np.concatenate([np.concatenate([o, dr], 1), np.concatenate([rng.uniform(-0.95, 0.95, (h, 3)), di], 1)]).astype(np.float32)
        rays.tofile(os.path.join(d, "rays.f32"))
        r = subprocess.run([node, "--max-old-space-size=8192", os.path.join(ROOT, "webgpu-path-tracer_amd", "js", "traverse_time.mjs"), d, "w", os.path.join(d, "rays.f32"), "", str(wl["stack"])],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        js = json.loads(r.stdout)
        return {"value": js["mrays_s"], "unit": "Mrays/s", "cores": 1, "rays": js["rays"], "node_visits_per_ray": js["node_visits"] / max(1, js["rays"]), "node": js["node"],
                "sample": "hitScene in single-threaded JavaScript (js/hit_scene.mjs) on %d synthetic rays against the workload's buffers, stack %d" % (js["rays"], wl["stack"]),
                "reference": "shaders/hitRay.wgsl:1-113 restated; the reference itself has no CPU traversal"}
    except Exception as e:
        return {"error": str(e)[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def js_bvh_build(native):
    """The reference's one CPU loop — the median-split BVH build (lib/BVH/bvhNode.js:28-73) — as single-threaded JavaScript under
    Node on this box (js/bvh_time.mjs: the shipped restatement, byte-identical output to the reference's, tests/test_host_buffers.py)."""
    node = shutil.which("node") or shutil.which("nodejs")
    if not node or native.boxes is None:
        return None
    a, b = native.boxes
    with tempfile.NamedTemporaryFile(suffix=".f64", delete=False, dir="/tmp") as f:
        f.write(a.tobytes())
        f.write(b.tobytes())
    try:
        r = subprocess.run([node, "--max-old-space-size=8192", os.path.join(ROOT, "webgpu-path-tracer_amd", "js", "bvh_time.mjs"), f.name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        js = json.loads(r.stdout)
        return {"ms": js["ms"], "primitives": js["n"], "node": js["node"], "threads": 1, "host_cpus": os.cpu_count(),
                "reference": "lib/BVH/bvhNode.js:28-73; benchmarks.txt:19 quotes 4483 ms for the 297,972-triangle dragon in a browser"}
    finally:
        os.remove(f.name)


def obj_parse_times(n_tris):
    """the native OBJ parser (1 and 16 host threads) against the reference's reader in JavaScript under Node on a generated
    file of the workload's triangle count (tools/obj_parse_bench.py: nothing is fetched, the text is written to /tmp).  Host work only."""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "obj_parse_bench.py"), str(n_tris)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": str(e)[:200]}


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
    ap.add_argument("--spp", type=int, default=0, help="progressive frames per step (0 = the config's: 64 for c2, 256 for c3, 512 for c4, 1024 for c5); per GPU with --scaling weak, in total with strong")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="N > 1: weak = spp x N (fixed rays per GPU), strong = fixed total spp (fixed total rays); "
                                                                                 "the line carries the other mode's figure too (config.other_scaling)")
    ap.add_argument("--collective", default="reduce", choices=["reduce", "gather"],
                    help="one process per GPU: reduce = sum-reduce of the full accumulation buffers to rank 0 (north_star's wording); gather = every rank sends the tiles "
                         "it owns, 1/N of the bytes, no arithmetic (SURVEY.md 8e's equivalent)")
    ap.add_argument("--stack-size", type=int, default=0)
    ap.add_argument("--tris", type=int, default=0, help="experiments: tessellate the procedural mesh of c3/c4/c5 to this many triangles (0 = the configuration's count)")
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--bvh", default="median", choices=["median", "sah"],
                    help="median = the reference's live builder (default, what the metric is quoted on); sah = the reference's "
                         "other, never-called builder (lib/BVH/bvhNode.js:108-283) as an opt-in (not for c2's golden buffers)")
    ap.add_argument("--devices", default="", help="comma-separated GPU ids for ONE multi-device context in this process (ptmi_create_multi + ncclReduce inside the library); "
                                                 "implied by --gpus N > 1 without torch.distributed.run (then 0..N-1); `0,0` rehearses the path on one GPU")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 on ONE GPU for rehearsal: ranks share cuda:0, the framebuffer reduce goes through gloo on host copies "
                         "(RCCL wants one device per rank); numbers from this mode are not bench results")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the cpu_baseline (the 1-GPU box's CPU share)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline's all-cores sample (0 = skip the CPU legs)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"], help="auto = rocprofv3 --pmc passes over the workload before the timed run (N = 1 only)")
    ap.add_argument("--pmc-timeout", type=float, default=240.0)
    ap.add_argument("--extra-configs", default="auto", choices=["auto", "off"], help="auto = append the configs[2] (871k-triangle) run to the line (N = 1, default workload only)")
    ap.add_argument("--c3-steps", type=int, default=6, help="timed steps of the appended configs[2] leg (the line carries their median and min-max)")
    ap.add_argument("--c3-warmup", type=int, default=2)
    ap.add_argument("--host-collective", action="store_true",
                    help="one process per GPU: the step's collective over gloo on host copies of the framebuffers instead of RCCL (every rank keeps its own GPU) — "
                         "what the watchdog falls back to when the RCCL run hangs or dies")
    ap.add_argument("--watchdog-seconds", type=float, default=300.0,
                    help="N > 1: the measurement runs in a fresh child process (started before this one touches the GPU); a child that has not delivered within this many "
                         "seconds is killed and ONE more fresh child runs the fall-back (in-library: PTMI_MULTI_REDUCE=copy; one process per GPU: --collective gather "
                         "--host-collective); its line says FALLBACK in config.parallelism.  0 = no watchdog (measure in this process)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--attempt", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--fallback-reason", default="", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.pmc_child:
        pmc_child(args)
        return
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1 or bool(args.devices) or max(args.gpus, 1) > 1
    if multi and not args.worker and args.watchdog_seconds > 0:
        sys.exit(watchdog(args))
    worker(args)


def watchdog(args):
    """N > 1: the first contact with RCCL on a new node must not be able to lose the record.  This process never touches the GPU (no torch, no HIP): it starts the
    measurement as a fresh child — same arguments and environment plus --worker — and relays its line.  A child that dies, or has not finished within
    --watchdog-seconds (the usual first-contact failure is a hang inside communicator initialisation), is killed by its process group and ONE more fresh child
    runs the fall-back: the in-library context with PTMI_MULTI_REDUCE=copy (peer copies + add kernel instead of ncclReduce), the one-process-per-GPU driver with
    --collective gather --host-collective (gloo on host copies, a new rendezvous port).  Under torch.distributed.run every rank's parent does this on its own; the
    limits are the same, so the ranks arrive at the fall-back together."""
    import signal

    rank = int(os.environ.get("RANK", "0"))
    launched = int(os.environ.get("WORLD_SIZE", "1")) > 1
    argv = [a for a in sys.argv[1:]]
    why = ""
    for attempt in (0, 1):
        env = dict(os.environ)
        extra = ["--worker", "--attempt", str(attempt)]
        if attempt == 1:
            extra += ["--fallback-reason", why]
            if launched:
                extra += ["--collective", "gather", "--host-collective"]
                # a rendezvous of its own: under torch.distributed.run the workers are clients of the launcher's store (TORCHELASTIC_USE_AGENT_STORE), which still
                # holds whatever keys the first attempt's ranks wrote — so the fall-back's rank 0 starts a fresh store on another port instead
                env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 17)
                env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
            else:
                env["PTMI_MULTI_REDUCE"] = "copy"
        t0 = time.perf_counter()
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + extra, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=args.watchdog_seconds)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)  # the child's own session: exactly the processes it started
            except ProcessLookupError:
                pass
            out, _ = p.communicate()
            why = "attempt %d produced no line within %.0f s (killed)" % (attempt, args.watchdog_seconds)
            print("bench.py watchdog (rank %d): %s" % (rank, why), file=sys.stderr, flush=True)
            continue
        lines = [l for l in (out or "").splitlines() if l.startswith("{")]
        if p.returncode == 0 and (lines or rank != 0):
            for l in lines:
                print(l, flush=True)
            return 0
        why = "attempt %d exited with code %d after %.0f s" % (attempt, p.returncode, time.perf_counter() - t0)
        print("bench.py watchdog (rank %d): %s" % (rank, why), file=sys.stderr, flush=True)
    return 1


def worker(args):
    if args.attempt == 0 and os.environ.get("PTMI_BENCH_SIMULATE") == "hang":  # (tests/test_bench_dist_gpu.py: the watchdog's rehearsal)
        time.sleep(1e6)
    if args.attempt == 0 and os.environ.get("PTMI_BENCH_SIMULATE") == "crash":
        os._exit(3)
    rank_env, world_env = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # Three ways to run.  (a) one GPU.  (b) N > 1 under torch.distributed.run (WORLD_SIZE set: how the driver launches it): one process per GPU,
    # torch.distributed's reduce over RCCL.  (c) N > 1 WITHOUT a launcher (`python bench.py --gpus N`, or --devices): ONE process, ONE context over
    # the N GPUs (ptmi_create_multi: a host thread and a stream per GPU, pixel tiles dealt to them) and ncclReduce inside the library — the design
    # north_star describes for the Node host.  `--devices 0,0` rehearses (c) on a one-GPU box (the shards share the GPU, summed by a kernel).
    devices = [int(x) for x in args.devices.split(",")] if args.devices else None
    inlib = world_env == 1 and (devices is not None or max(args.gpus, 1) > 1)
    if inlib and devices is None:
        devices = list(range(args.gpus))
    solo = world_env == 1 and not inlib
    extra_c3 = solo and args.extra_configs == "auto" and args.workload == "c2"

    # rocprofv3 passes first: child processes, before this process has touched the GPU
    pmc, pmc_note, pmc_log = {}, {}, []
    if solo and args.pmc == "auto":
        for w in [args.workload] + (["c3"] if extra_c3 else []):
            pmc[w], why = pmc_passes(args, w, pmc_log)
            if why:
                pmc_note[w] = why

    pkg = entry._load_pkg()
    torch = None
    host_coll = False
    if world_env > 1:
        import torch
        from webgpu_path_tracer_amd import dist as pdist

        host_coll = args.rehearse_gloo or args.host_collective
        rank, world, local = pdist.init_process_group("gloo" if host_coll else None)
        if args.rehearse_gloo:
            local = 0
        if world != max(args.gpus, 1):
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        torch.cuda.set_device(local)
    else:
        pdist, rank, local = _NoDist, 0, 0
        world = len(devices) if inlib else 1

    def run_workload(name, steps, warmup, spp_arg, wl_args=None):
        wl_args = wl_args or args
        wl = make_workload(pkg, name, wl_args)
        per = spp_arg or SPP[name]
        spp = per * world if args.scaling == "weak" else per
        ctx = make_context(pkg, wl, devices if inlib else local, wl_args)
        reduce_fn = None
        if inlib:
            reduce_fn = ctx.reduce_framebuffer
        elif world > 1:
            fb_t = torch.zeros(wl["H"] * wl["W"] * 4, dtype=torch.float32, device="cuda")
            ctx.bind_framebuffer(fb_t.data_ptr(), fb_t.numel() * 4)
            ctx.set_shard(rank, world, pdist.TILE_PIXELS)

            def collective(t):
                if args.collective == "gather":
                    pdist.gather_tiles(t, pdist.TILE_PIXELS, 0)
                else:
                    pdist.reduce_framebuffer(t, 0)

            def reduce_fn():
                if host_coll:
                    host = fb_t.cpu()
                    collective(host)
                    fb_t.copy_(host)
                else:
                    collective(fb_t)
                torch.cuda.synchronize()  # the collective runs on torch's stream; the next clear runs on the context's

        m = measure(pkg, torch, pdist, ctx, wl, spp, steps, warmup, reduce_fn)
        if world > 1:  # the other scaling mode, briefly: strong next to the default weak (fixed total spp: the reading of BASELINE configs[3] / [4]), or the reverse
            spp_o = per if args.scaling == "weak" else per * world
            dt_o, st_o = measure_short(torch, pdist, ctx, wl, spp_o, max(1, min(steps, 5)), reduce_fn)
            m["other"] = {"scaling": "strong" if args.scaling == "weak" else "weak", "spp": spp_o, "steps": max(1, min(steps, 5)), "dt": dt_o, "rays": st_o["rays"]}
        m["reduce_info"] = ctx.reduce_info() if inlib else None
        m["reduce_mode"] = ctx.stats().get("reduce_mode") if inlib else None
        return wl, ctx, spp, per, m

    wl, ctx, spp, per, m = run_workload(args.workload, args.steps, args.warmup, args.spp)
    st, cst, dom = m["st"], m["cst"], m["dom"]
    dt_max = pdist.all_reduce_scalar(m["dt"], "max")
    rays_all = pdist.all_reduce_scalar(st["rays"], "sum")
    paths_all = pdist.all_reduce_scalar(st["paths"], "sum")
    o_rays = o_dt = None
    if world > 1:  # (collectives: every rank takes part)
        o_rays, o_dt = pdist.all_reduce_scalar(m["other"]["rays"], "sum"), pdist.all_reduce_scalar(m["other"]["dt"], "max")

    def describe(wl, spp, per, m, steps, rays, paths, dt, pm):
        """value / roofline / kernel table of one measured workload."""
        st, cst, dom = m["st"], m["cst"], m["dom"]
        tab = kernel_table(m["split"], 1, pm, "k_shade<true, true, false, false>" if wl["extra"].get("importance_sampling") else "k_shade6<false, false>")
        n_dom = max(st[LAUNCH_KEY[dom]], 1)
        bvh_launches = max(m["split"][LAUNCH_KEY["k_bvh"]], 1)
        gather = (cst["bvh_node_visits"] / 2.0 + cst["tri_tests"]) / bvh_launches  # 64-byte records k_bvh fetches per launch (pair records + triangle records)
        roof = roofline_of(tab, dom, st[MS_KEY[dom]] / n_dom, n_dom, gather)
        if tab["k_bvh"]["ms_per_step"] > 0:
            tab["k_bvh"]["l1_gather_frac"] = gather * bvh_launches / (tab["k_bvh"]["ms_per_step"] * 1e-3) / gather_peak()[0]  # (not clamped: coherent lanes sharing a record count per lane)
            tab["k_bvh"]["records_per_launch"] = gather
        # What SURVEY §8d's yardstick calls waste: HBM bytes the step moves beyond what the reference's megakernel must move — its framebuffer read-modify-write
        # once per frame (32 B per pixel and frame) and the scene once.  The wavefront design pays the rest as path state crossing HBM between kernels.
        if any(tab[k].get("fabric_bytes_per_launch") is not None for k in KERNELS):  # (a kernel the batch never launched has no counters and moved nothing)
            step_bytes = sum((tab[k].get("fabric_bytes_per_launch") or 0.0) * tab[k]["launches_per_step"] for k in KERNELS)
            scene_bytes = sum(np.asarray(wl["buffers"][k]).nbytes for k in ("spheres", "quads", "triangles", "meshes", "transforms", "materials", "bvh"))
            compulsory = 32.0 * wl["W"] * wl["H"] * spp / max(world, 1) + scene_bytes
            roof["step_fabric_bytes"] = step_bytes
            roof["compulsory_bytes"] = compulsory
            roof["state_traffic_bytes"] = step_bytes - compulsory
            roof["state_over_compulsory"] = (step_bytes - compulsory) / compulsory if compulsory else None
            roof["step_fabric_frac_of_hbm_peak"] = step_bytes / (dt / steps) / (HBM_PEAK_GBS * 1e9)
            roof["dram_vs_mall"] = MALL_NOTE
            roof["traffic_note"] = ("step_fabric_bytes = measured fabric bytes of all kernels of one step (this run's FETCH_SIZE / WRITE_SIZE passes: Infinity-Cache hits count, "
                                    "so DRAM bytes are at most this); compulsory_bytes = the reference "
                                    "megakernel's own HBM need (framebuffer RMW per pixel and frame + the scene once); the difference is wavefront path state (queues, hit "
                                    "records, per-path radiance) that the reference never moves — waste by SURVEY §8d's yardstick, the price of compaction and lane refill")
        hit_scene_gbs = alg_bytes(cst) / (m["split"]["render_ms"] / 1e3) / 1e9 if m["split"]["render_ms"] > 0 else None
        roof["algorithmic_hit_scene_gbs_informational"] = hit_scene_gbs  # SURVEY §8d reference-layout bytes / render time: cache-oblivious, NOT a fraction of HBM peak
        roof["work_per_ray"] = {k: cst[k] / max(cst["rays"], 1) for k in ("node_visits", "bvh_node_visits", "tri_tests", "quad_tests", "sphere_tests", "mat_fetches")}
        roof["kernels"] = tab
        return {
            "value": rays / dt / 1e6, "ms_per_step": dt / steps * 1e3, "rays_per_step": rays / steps, "mpaths_per_s": paths / dt / 1e6,
            "ms_per_step_stats": {"median": float(np.median(m["step_ms"])), "min": float(np.min(m["step_ms"])), "max": float(np.max(m["step_ms"])), "steps": len(m["step_ms"]),
                                  "note": "this rank's wall time of each timed step; `ms_per_step` = the whole timed region / steps"},
            "workload": "%s, %dx%d, %d spp%s, %d bounces, stack_size %d%s" % (
                wl["label"], wl["W"], wl["H"], spp,
                ("" if world == 1 else " (weak scaling: %d per GPU x %d GPUs, every GPU traces 1/%d of the pixels for all of them)" % (per, world, world) if args.scaling == "weak"
                 else " in total (strong scaling: every one of the %d GPUs traces 1/%d of the pixels, %d spp each)" % (world, world, spp)), wl["bounces"], wl["stack"],
                ", SAH BVH (opt-in)" if args.bvh == "sah" else ""),
            "roofline": roof,
        }

    out = None
    if rank == 0:
        d = describe(wl, spp, per, m, args.steps, rays_all, paths_all, dt_max, pmc.get(args.workload))
        out = {
            "metric": "Mrays/s at 1080p, 8 bounces; achieved HBM GB/s vs roofline",
            "value": d["value"],
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": d["ms_per_step"],
            "ms_per_step_stats": d["ms_per_step_stats"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": d["workload"],
                "rays_per_step": d["rays_per_step"],
                "mpaths_per_s": d["mpaths_per_s"],
                "parallelism": ("ptmi_create_multi x%d (one process, one context; devices %s); the step's collective inside the library: %s%s" % (
                    world, ",".join(map(str, devices)), m["reduce_info"], "" if len(set(devices)) == len(devices) else "; REHEARSAL: shards share a GPU") if inlib
                    else ("pixel tiles x%d (one process per GPU) + 1 %s %s per step (torch.distributed%s)" % (
                        world, "gloo (host copies)" if host_coll else "RCCL", "reduce of the full accumulation buffers" if args.collective == "reduce" else "gather of every rank's own tiles",
                        "; REHEARSAL: the ranks share cuda:0" if args.rehearse_gloo else ""))
                    if world > 1 else "1 GPU") + ("; FALLBACK after the first attempt failed: %s" % args.fallback_reason if args.fallback_reason else ""),
            },
            "roofline": d["roofline"],
        }
        if world > 1:
            out["config"]["collective_ms_per_step"] = m["collective_ms_per_step"]  # rank 0's wall time inside the collective, its wait for the slowest rank included
            out["config"]["collective_bytes_into_root"] = (world - 1) * wl["W"] * wl["H"] * 16 // (world if ((inlib and m["reduce_mode"] == 4) or (not inlib and args.collective == "gather")) else 1)
            o = m["other"]
            out["config"]["other_scaling"] = {"scaling": o["scaling"], "spp_total": o["spp"], "steps": o["steps"], "value": o_rays / o_dt / 1e6, "unit": "Mrays/s",
                                              "ms_per_step": o_dt / o["steps"] * 1e3, "note": "informational: the same ranks and collective in the other scaling mode; `value` above is --scaling %s" % args.scaling}
        if args.workload in pmc_note:
            out["roofline"]["pmc_note"] = pmc_note[args.workload]
        elif pmc.get(args.workload):
            out["roofline"]["pmc_source"] = "rocprofv3 --pmc passes made by this run (%s), %.0f s" % (" | ".join(c for _, c in PMC_PASSES), pmc[args.workload]["_seconds"])
    ctx.close()

    if rank == 0 and solo:
        # Informational: the same workload through ONE context with two shards on this GPU (ptmi_create_multi with the device listed twice:
        # two streams, pixel tiles dealt to them, summed at read-back) — the shards fill each other's k_bvh tails.  Not `value`: the kernels
        # of the two streams overlap, so a per-kernel roofline of that mode would not mean what the one above means.
        try:
            ctx2 = pkg.Context([local, local])
            ctx2.upload_scene(wl["buffers"])
            ctx2.set_params(max_bounces=wl["bounces"], frames_in_flight=args.frames_in_flight, stack_size=wl["stack"], **wl["extra"])
            ctx2.resize(wl["W"], wl["H"])

            def step2():
                ctx2.clear()
                ctx2.render(wl["view"], 1, spp)
                ctx2.synchronize()

            for _ in range(max(1, min(args.warmup, 2))):
                step2()
            ctx2.reset_stats()
            t2 = time.perf_counter()
            n2 = max(1, min(args.steps, 10))
            for _ in range(n2):
                step2()
            dt2 = time.perf_counter() - t2
            st2 = ctx2.stats()
            ctx2.close()
            out["config"]["two_shards_on_one_gpu"] = {"value": st2["rays"] / dt2 / 1e6, "unit": "Mrays/s", "ms_per_step": dt2 / n2 * 1e3, "steps": n2,
                                                      "note": "informational: one context, two streams (pixel tiles split in the library, bit-identical image); not the headline value"}
        except Exception as e:  # never let the extra leg cost the line
            out["config"]["two_shards_on_one_gpu"] = {"error": str(e)[:200]}
        setup = {args.workload: wl["setup"]}
        if extra_c3:
            wl3, ctx3, spp3, per3, m3 = run_workload("c3", args.c3_steps, args.c3_warmup, 0)
            d3 = describe(wl3, spp3, per3, m3, args.c3_steps, m3["st"]["rays"], m3["st"]["paths"], m3["dt"], pmc.get("c3"))
            d3["steps"], d3["warmup"] = args.c3_steps, args.c3_warmup
            # the reference's own dragon figure (benchmarks.txt:18-20) is paths/s at its canvas size; both readings of north_star's ">= 10x"
            d3["vs_baseline"] = d3["mpaths_per_s"] / REF_DRAGON_MPATHS
            d3["vs_baseline_basis"] = ("Mpaths/s / 33.5 Mpaths/s = the reference's 62 fps x 900x600 px on its 297,972-triangle dragon (benchmarks.txt:18-20; hardware, "
                                       "bounce cap unstated; derived in BASELINE.md §1, not a published Mrays/s). Rays / 33.5 M (the minimum reading: >= 1 ray per path) = %.0f" % (d3["value"] / REF_DRAGON_MPATHS))
            if "c3" in pmc_note:
                d3["roofline"]["pmc_note"] = pmc_note["c3"]
            # set-up of the 871k-triangle scene, next to the reference's 4,483 ms BVH build of its dragon
            t = time.perf_counter()
            ctx3.build_bvh(*wl3["native"].boxes)
            wl3["setup"]["bvh_build_device_ms"] = (time.perf_counter() - t) * 1e3
            ctx3.close()
            # the device-resident set-up (ptmi_build_scene_bvh): unordered triangles up, boxes + build + reordering + digests on the GPU, nothing back
            raw = pkg.scenes.c3_scene(**({"n_tris": args.tris} if args.tris else {})).buffers_unbuilt()  # (the Python mirror's packing: not timed, not the product)
            ctx4 = pkg.Context(local)
            t0 = time.perf_counter()
            ctx4.upload_scene(raw)
            t1 = time.perf_counter()
            ctx4.build_scene_bvh()
            t2 = time.perf_counter()
            ctx4.set_params(max_bounces=wl3["bounces"], stack_size=wl3["stack"])
            ctx4.resize(wl3["W"], wl3["H"])
            ctx4.prepare()
            t3 = time.perf_counter()
            ctx4.close()
            wl3["setup"]["device_resident"] = {"upload_ms": (t1 - t0) * 1e3, "build_scene_bvh_ms": (t2 - t1) * 1e3, "validate_digests_ms": (t3 - t2) * 1e3, "total_ms": (t3 - t0) * 1e3,
                                               "note": "ptmi_upload x7 (triangles in mesh order, no BVH) + ptmi_build_scene_bvh + ptmi_prepare: the whole scene set-up after the host's packing"}
            if args.cpu_seconds > 0:
                wl3["setup"]["bvh_build_js_single_thread"] = js_bvh_build(wl3["native"])
                wl3["setup"]["obj_parse"] = obj_parse_times(wl3["buffers"]["triangles"].size // 24)
                d3["cpu_js_traversal"] = js_traversal(wl3)  # north_star's "single-threaded JS BVH traversal", on its target scene
            d3["setup_ms"] = wl3["setup"]
            # ... and the same scene from the reference's OTHER builder (lib/BVH/bvhNode.js:108-283, which its renderer never calls), built on the GPU: an opt-in
            # (--bvh sah), so informational here — the configuration's own figure above is on the median tree the reference renders with
            try:
                import copy

                a_sah = copy.copy(args)
                a_sah.bvh = "sah"
                wls, ctxs, spps, pers, ms = run_workload("c3", 3, 1, 0, a_sah)
                ds = describe(wls, spps, pers, ms, 3, ms["st"]["rays"], ms["st"]["paths"], ms["dt"], None)
                ctxs.close()
                d3["with_sah_tree"] = {"value": ds["value"], "unit": "Mrays/s", "ms_per_step": ds["ms_per_step"], "steps": 3, "warmup": 1, "work_per_ray": ds["roofline"]["work_per_ray"],
                                       "k_bvh_ms_per_step": ds["roofline"]["kernels"]["k_bvh"]["ms_per_step"], "setup_ms": wls["setup"], "stack_size": wls["stack"],
                                       "note": "informational: ptmi_build_scene_bvh_sah (binned SAH, byte-identical to the reference's generate_bvh_heirarchy_SAH) instead of the median split"}
            except Exception as e:  # never let the extra leg cost the line
                d3["with_sah_tree"] = {"error": str(e)[:200]}
            out["configs"] = [d3]
            # top-level vs_baseline stays null: BASELINE.md publishes no Mrays/s for configs[1]; the dragon ratio belongs to the configs[2] run above
            out["vs_baseline_note"] = "no published number for this metric/config (BASELINE.json `published` is empty); the ratio to the reference's own dragon figure is in configs[0].vs_baseline"
        out["setup_ms"] = setup[args.workload]
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(pkg, wl, spp, args)
            out["cpu_baseline"]["js_traversal"] = js_traversal(wl)
            if extra_c3 and out["configs"][0]["setup_ms"].get("bvh_build_js_single_thread"):
                out["cpu_baseline"]["bvh_build_js_ms"] = out["configs"][0]["setup_ms"]["bvh_build_js_single_thread"]
            elif args.workload != "c2":
                out["cpu_baseline"]["bvh_build_js_ms"] = js_bvh_build(wl["native"])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world_env > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
