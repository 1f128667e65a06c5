#!/bin/bash
# Hardware counters of one bench step, per kernel (rocprofv3 --pmc through tools/pmc.sh, which cuts every list into passes that fit).
#   tools/counters.sh sq     <tag> <workload> <spp> [bench args]   vector side: instructions, active lanes, waits, rocprof's VALUBusy
#   tools/counters.sh scalar <tag> <workload> <spp> [bench args]   scalar side: SALU / branch instructions, scalar issue per CU
#   tools/counters.sh ta     <tag> <workload> <spp> [bench args]   texture addresser / L1 / texture data: is k_bvh bound by its 16-byte record gathers?
#   tools/counters.sh tabusy                                       texture-addresser busy fraction per kernel, all four workloads -> gpurun_out/ta_busy.json
# The environment (PTMI_LIB for an A/B build, PTMI_* tuning variables) is inherited by the bench processes.
export PTMI_PLACEMENT_TRIES=${PTMI_PLACEMENT_TRIES:-1}  # no placement search under the profiler: its dry runs are launches of the kernels being profiled
kind=$1; shift
bench_args() { echo "--workload $1 --spp $2 --steps 1 --warmup 0 --cpu-seconds 0 --pmc off --extra-configs off"; }
case $kind in
sq)
  tag=$1; w=$2; spp=$3; shift 3
  tools/pmc.sh sq_$tag "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" $(bench_args $w $spp) "$@" > /dev/null || exit 1
  python3 - gpurun_out/pmc_sq_$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("ms_total", 0))[:4]:
    if "SQ_INSTS_VALU" not in v: continue
    print("%-10s %-36s ms %8.1f  VALU %.3e  lanes %.3f  wait %.2f  busy(4c) %.2f  VMEM_RD %.3e  LDS %.3e" % (
        sys.argv[2], k[:36], v["ms_total"], v["SQ_INSTS_VALU"], v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
        4.0 * v["SQ_ACTIVE_INST_VALU"] / 1024.0 / (v["SQ_BUSY_CYCLES"] / 32.0), v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_LDS"]))
PY
  ;;
scalar)
  tag=$1; w=$2; spp=$3; shift 3
  tools/pmc.sh sq2_$tag "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" $(bench_args $w $spp) "$@" > /dev/null || exit 1
  python3 - gpurun_out/pmc_sq2_$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("ms_total", 0))[:2]:
    cyc = v["SQ_BUSY_CYCLES"] / 32.0   # kernel cycles
    print("%-8s %-30s ms %7.1f  SALU %.3e BRANCH %.3e  per-CU scalar instr/cycle %.2f  ACTIVE_INST_SCA/cyc/CU %.2f  INST_CYCLES_SALU/cyc/CU %.2f  ACTIVE_INST_ANY/WAVE_CYC %.2f  WAIT_INST_ANY/WAVE_CYC %.2f" % (
        sys.argv[2], k[:30], v["ms_total"], v["SQ_INSTS_SALU"], v["SQ_INSTS_BRANCH"], (v["SQ_INSTS_SALU"] + v["SQ_INSTS_BRANCH"]) / 256 / cyc,
        v["SQ_ACTIVE_INST_SCA"] / 256 / cyc, v["SQ_INST_CYCLES_SALU"] / 256 / cyc, v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
  ;;
ta)
  tag=$1; w=$2; spp=$3; shift 3
  tools/pmc.sh ta_$tag "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" $(bench_args $w $spp) "$@" || exit 1
  ;;
tabusy)
  for item in "c2 64" "c3 64" "c4 32" "c5 16 --width 3840 --height 2160"; do
    set -- $item; w=$1; spp=$2; shift 2
    tools/pmc.sh tab_$w "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" $(bench_args $w $spp) "$@" > /dev/null || exit 1
  done
  python3 - <<'PY'
import json
out = {"_note": "rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE over one bench step (+ its counted pass), per kernel: ta_busy = TA_TA_BUSY_sum / 256 texture addressers / (GRBM_GUI_ACTIVE / 8 XCDs)"}
for w in ("c2", "c3", "c4", "c5"):
    d = json.load(open("gpurun_out/pmc_tab_%s.json" % w))
    out[w] = {}
    for k, v in d.items():
        if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0 and any(n in k for n in ("k_bvh", "k_shade", "k_generate", "k_accumulate")):
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            out[w][k.split("(")[0]] = {"ta_busy": round(v["TA_TA_BUSY_sum"] / 256.0 / cyc, 3), "ta_addr_stalled_by_tc": round(v["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / 256.0 / cyc, 3), "ms_total": round(v["ms_total"], 2), "launches": v["launches"]}
json.dump(out, open("gpurun_out/ta_busy.json", "w"), indent=1)
for w in ("c2", "c3", "c4", "c5"):
    print(w, {k: v["ta_busy"] for k, v in out[w].items()})
PY
  ;;
*) echo "usage: tools/counters.sh sq|scalar|ta|tabusy ..."; exit 2;;
esac
