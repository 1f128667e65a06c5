#!/bin/bash
# texture-addresser busy fraction per kernel for the four workloads: gpurun_out/r3ab/ta_busy.json
mkdir -p gpurun_out/r3ab
for item in "c2 64" "c3 64" "c4 32" "c5 16 --width 3840 --height 2160"; do
  set -- $item; w=$1; spp=$2; shift 2
  tools/pmc.sh tab_$w "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" --workload $w --spp $spp "$@" --steps 1 --warmup 0 --cpu-seconds 0 --pmc off --extra-configs off > /dev/null || exit 1
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
json.dump(out, open("gpurun_out/r3ab/ta_busy.json", "w"), indent=1)
for w in ("c2", "c3", "c4", "c5"):
    print(w, {k: v["ta_busy"] for k, v in out[w].items()})
PY
