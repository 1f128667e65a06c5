#!/bin/bash
# usage: tools/trace_gaps.sh <tag> <python script and args...> — busy vs idle time of the GPU between the first and last kernel
export PTMI_PLACEMENT_TRIES=${PTMI_PLACEMENT_TRIES:-1}  # no placement search under the profiler: its dry runs are launches of the kernels being profiled
tag=$1; shift
out=/tmp/kg_$tag; rm -rf $out; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 "$@" > /tmp/kg_$tag.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]))
rows.sort()
# look at the last 400 kernels (steady state)
rows = rows[-400:]
busy = sum(e - s for s, e, _ in rows); span = rows[-1][1] - rows[0][0]
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
print('last %d kernels: span %.2f ms, busy %.2f ms (%.0f %%), median gap %.1f us, mean gap %.1f us, max gap %.1f us' % (len(rows), span / 1e6, busy / 1e6, 100 * busy / span, sorted(gaps)[len(gaps) // 2], sum(gaps) / len(gaps), max(gaps)))
big = sorted(((g, rows[i][2], rows[i + 1][2]) for i, g in enumerate(gaps)), reverse=True)[:6]
for g, a, b in big: print('  gap %.1f us between %s -> %s' % (g, a, b))
PY
