#!/bin/bash
# usage: tools/ab.sh "<bench args>" name1 name2 ...   — runs bench.py (no PMC, no CPU legs, no extra configs) with each A/B build
# webgpu-path-tracer_amd/variants/libptmi_<name>.so ("base" = the in-tree library) and prints value + per-kernel ms
args=$1; shift
mkdir -p gpurun_out/ab
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$(pwd)/webgpu-path-tracer_amd/variants/libptmi_$v.so
  PTMI_LIB=$lib timeout -k 10 300 python bench.py $args --pmc off --cpu-seconds 0 --extra-configs off > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab/$v.err; continue; }
  python3 - gpurun_out/ab/$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernels"]
print("%-14s %8.0f Mrays/s %8.2f ms/step | " % (sys.argv[2], d["value"], d["ms_per_step"]) + "  ".join("%s %.2f" % (n[2:], k[n]["ms_per_step"]) for n in k))
PY
done
