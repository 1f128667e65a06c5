#!/bin/bash
# usage: tools/trace_kernel.sh <tag> <kernel substring> [script.py] <args...>   (bench.py when no script is named) — per-launch durations (ms) of one kernel, timed pass only
export PTMI_PLACEMENT_TRIES=${PTMI_PLACEMENT_TRIES:-1}  # no placement search under the profiler: its dry runs are launches of the kernels being profiled
tag=$1; kern=$2; shift 2
out=/tmp/kt_$tag; rm -rf $out; export TMPDIR=/tmp
prog="bench.py"; if [[ "$1" == *.py ]]; then prog=$1; shift; fi
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 $prog "$@" > /dev/null 2>&1
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']: rows.append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r['Kernel_Name'][:40]))
rows.sort()
print(' '.join('%.2f' % d for _, d, _ in rows)); print('sum %.1f ms over %d launches' % (sum(d for _, d, _ in rows), len(rows)))
PY
