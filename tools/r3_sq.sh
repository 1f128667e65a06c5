#!/bin/bash
# SQ counters of one bench step, per kernel: usage tools/r3_sq.sh <tag> <workload> <spp> [bench args]; environment (PTMI_LIB, PTMI_BVH_KERNEL ...) is inherited
tag=$1; w=$2; spp=$3; shift 3
A="--workload $w --spp $spp --steps 1 --warmup 0 --cpu-seconds 0 --pmc off --extra-configs off $@"
tools/pmc.sh sq_$tag "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" $A > /dev/null || exit 1
python3 - gpurun_out/pmc_sq_$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("ms_total", 0))[:4]:
    if "SQ_INSTS_VALU" not in v: continue
    print("%-10s %-36s ms %8.1f  VALU %.3e  lanes %.3f  wait %.2f  busy(4c) %.2f  VMEM_RD %.3e  LDS %.3e" % (
        sys.argv[2], k[:36], v["ms_total"], v["SQ_INSTS_VALU"], v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
        4.0 * v["SQ_ACTIVE_INST_VALU"] / 1024 / (v["SQ_BUSY_CYCLES"] / 32.0), v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_LDS"]))
PY
python3 - gpurun_out/pmc_sq_$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "k_bvh" in k and "SQ_WAVE_CYCLES" in v:
        cyc = v["SQ_BUSY_CYCLES"] / 32.0
        print("%-10s %-36s avg waves/CU %.1f  VALU/cycle/SIMD %.3f  clock %.2f GHz" % (sys.argv[2], k[:36], 4 * v["SQ_WAVE_CYCLES"] / cyc / 256, v["SQ_INSTS_VALU"] / cyc / 1024, cyc / (v["ms_total"] * 1e-3) / 1e9))
PY
