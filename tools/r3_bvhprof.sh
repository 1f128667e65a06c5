#!/bin/bash
# kernel-level profile of the device BVH builder (rocprofv3 --kernel-trace --stats), summary to gpurun_out/r3ab/bvh_dev_kernel_stats.csv
mkdir -p gpurun_out/r3ab; root=$(pwd); export TMPDIR=/tmp; rm -rf /tmp/bp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o run -- python3 $root/tools/bvh_dev_profile.py 2>&1 | grep "device build\|Error\|error" | head)
f=$(find /tmp/bp -name "*kernel_stats.csv" | head -1); [ -z "$f" ] && exit 1
cp $f gpurun_out/r3ab/bvh_dev_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("total kernel ms %.1f" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6))
for r in rows[:16]:
    print("%-100s calls %5s total %8.2f ms avg %8.1f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
