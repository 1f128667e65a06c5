#!/bin/bash
# usage: envab.sh "<bench args>" "VAR=val VAR2=val" ...
args=$1; shift
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py $args --pmc off --cpu-seconds 0 --extra-configs off > /tmp/envab.json 2>/tmp/envab.err || { echo "$e FAILED"; tail -2 /tmp/envab.err; continue; }
  python3 - "$e" <<'PY'
import json, sys
d = json.loads(open('/tmp/envab.json').read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
print("%-40s %8.0f Mrays/s %8.2f ms/step | " % (sys.argv[1], d["value"], d["ms_per_step"]) + "  ".join("%s %.2f" % (n[2:], k[n]["ms_per_step"]) for n in k))
PY
done
