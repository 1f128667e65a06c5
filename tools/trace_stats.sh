#!/bin/bash
# usage: tools/trace_stats.sh <tag> <python script and args...> — rocprofv3 kernel stats (calls, total, average) of any script
export PTMI_PLACEMENT_TRIES=${PTMI_PLACEMENT_TRIES:-1}  # no placement search under the profiler: its dry runs are launches of the kernels being profiled
tag=$1; shift
out=/tmp/ks_$tag; rm -rf $out; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 "$@" > /tmp/ks_$tag.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print('%-60s calls %6s total %9.2f ms avg %8.1f us' % (r['Name'][:60], r['Calls'], int(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
tail -3 /tmp/ks_$tag.log | grep -v "rocprofv3\|simple_timer"
