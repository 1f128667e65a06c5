#!/bin/bash
# usage: tools/trace_seq.sh <tag> <bench args...> — the kernels of the LAST timed step in launch order: start offset, duration, gap to the previous kernel (us)
export PTMI_PLACEMENT_TRIES=${PTMI_PLACEMENT_TRIES:-1}  # no placement search under the profiler: its dry runs are launches of the kernels being profiled
tag=$1; shift
out=/tmp/ks_$tag; rm -rf $out; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 bench.py "$@" > /tmp/ks_$tag.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('ptmi::', '').split('<')[0].split('(')[0]))
rows.sort()
gen = [i for i, r in enumerate(rows) if r[2] == 'k_generate']
acc = [i for i, r in enumerate(rows) if r[2] == 'k_accumulate']
# first render of the run that is followed by an accumulate
i0 = gen[min(1, len(gen) - 1)]
i1 = min(a for a in acc if a > i0)
t0 = rows[i0][0]
for i in range(i0, i1 + 1):
    s, e, n = rows[i]
    print('%9.1f us  %-14s %8.1f us   gap %6.1f' % ((s - t0) / 1e3, n, (e - s) / 1e3, (s - rows[i - 1][1]) / 1e3 if i > i0 else 0.0))
PY
