#!/bin/bash
# Runs on the GPU box: the opt-in SAH tree (lib/BVH/bvhNode.js:108-283) against the reference's live median tree on configs[2..4], same box.
# usage: tools/sah_pass.sh [pmc]   — results in gpurun_out/sah/
set -o pipefail
R=gpurun_out/sah; mkdir -p $R; export TMPDIR=/tmp
pmc="--pmc off"; [ "$1" = pmc ] && pmc="--pmc-timeout 400"
for w in c3 c4 c5; do
  x=""; [ $w = c5 ] && x="--width 3840 --height 2160 --spp 128"
  for b in median sah; do
    p="$pmc"; [ $b = median ] && p="--pmc off"
    timeout -k 10 500 python bench.py --workload $w --bvh $b --steps 2 --warmup 1 --cpu-seconds 0 --extra-configs off $p $x > $R/bench_${w}_$b.json 2> $R/bench_${w}_$b.err || { echo "$w $b FAILED"; tail -5 $R/bench_${w}_$b.err; exit 1; }
    python3 - $R/bench_${w}_$b.json $w $b <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernels"]; wpr = d["roofline"]["work_per_ray"]
print("%-3s %-6s %8.0f Mrays/s %9.2f ms/step | visits/ray %.2f tri/ray %.2f | " % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], wpr["node_visits"], wpr["tri_tests"]) + "  ".join("%s %.2f" % (n[2:], k[n]["ms_per_step"]) for n in k) + " | setup " + json.dumps({a: round(b, 1) for a, b in d.get("setup_ms", {}).items() if isinstance(b, (int, float))}), flush=True)
PY
  done
done
