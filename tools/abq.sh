#!/bin/bash
# usage: tools/abq.sh "c2 c3:128 c4:64 c5:32@3840x2160" name1 name2 ...   — short bench lines (workload[:spp][@WxH]) for each A/B build
# webgpu-path-tracer_amd/variants/libptmi_<name>.so ("base" = the in-tree library); an item "VAR=val+name" sets an environment variable
# for that run.  No PMC passes, no CPU legs.  Results: gpurun_out/abq/<name>_<workload>.json and one table line per run.
items=$1; shift
mkdir -p gpurun_out/abq
for v in "$@"; do
  envs=""; name=$v
  while [[ "$name" == *=*+* ]]; do envs="$envs ${name%%+*}"; name=${name#*+}; done
  lib=""; [ "$name" != base ] && lib=$(pwd)/webgpu-path-tracer_amd/variants/libptmi_$name.so
  for item in $items; do
    w=${item%%[:@]*}; spp=""; dims=""
    [[ "$item" == *:* ]] && { s=${item#*:}; spp="--spp ${s%%@*}"; }
    [[ "$item" == *@* ]] && { d=${item##*@}; dims="--width ${d%%x*} --height ${d##*x}"; }
    tag=$(echo "${v}_${item}" | tr '=+:@ /' '______')
    env $envs PTMI_LIB=$lib timeout -k 10 300 python bench.py --workload $w --steps ${STEPS:-2} --warmup 1 --cpu-seconds 0 --pmc off --extra-configs off $spp $dims \
      > gpurun_out/abq/$tag.json 2> gpurun_out/abq/$tag.err || { echo "$v $item FAILED"; tail -3 gpurun_out/abq/$tag.err; continue; }
    python3 - gpurun_out/abq/$tag.json "$v" "$item" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernels"]
print("%-28s %-18s %8.0f Mrays/s %9.2f ms/step | " % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"]) + "  ".join("%s %.2f" % (n[2:], k[n]["ms_per_step"]) for n in k), flush=True)
PY
  done
done
