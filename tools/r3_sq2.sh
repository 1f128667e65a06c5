#!/bin/bash
# scalar-side SQ counters of one bench step, per kernel: usage tools/r3_sq2.sh <tag> <workload> <spp> [bench args]
tag=$1; w=$2; spp=$3; shift 3
A="--workload $w --spp $spp --steps 1 --warmup 0 --cpu-seconds 0 --pmc off --extra-configs off $@"
tools/pmc.sh sq2_$tag "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" $A > /dev/null || exit 1
python3 - gpurun_out/pmc_sq2_$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("ms_total", 0))[:2]:
    cyc = v["SQ_BUSY_CYCLES"] / 32.0   # kernel cycles
    print("%-8s %-30s ms %7.1f  SALU %.3e BRANCH %.3e  per-CU scalar instr/cycle %.2f  ACTIVE_INST_SCA/cyc/CU %.2f  INST_CYCLES_SALU/cyc/CU %.2f  ACTIVE_INST_ANY/WAVE_CYC %.2f  WAIT_INST_ANY/WAVE_CYC %.2f" % (
        sys.argv[2], k[:30], v["ms_total"], v["SQ_INSTS_SALU"], v["SQ_INSTS_BRANCH"], (v["SQ_INSTS_SALU"] + v["SQ_INSTS_BRANCH"]) / 256 / cyc,
        v["SQ_ACTIVE_INST_SCA"] / 256 / cyc, v["SQ_INST_CYCLES_SALU"] / 256 / cyc, v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
