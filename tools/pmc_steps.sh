#!/bin/bash
# usage: tools/pmc_steps.sh <tag> <kernel substring> "<counters>" [script.py] <args...> — rocprofv3 --pmc, counters PER DISPATCH of one kernel (the steps of a
# render: launch k of a step is bounce k), first 16 (PMC_ROWS) dispatches, in gpurun_out/pmcsteps_<tag>.txt.  Any number of counters: tools/pmc_split.py
# cuts the list into passes that fit the hardware's counter slots (one rocprofv3 run each, merged by dispatch order).
export PTMI_PLACEMENT_TRIES=${PTMI_PLACEMENT_TRIES:-1}  # no placement search under the profiler: its dry runs are launches of the kernels being profiled
tag=$1; kern=$2; ctrs=$3; shift 3
prog="bench.py"; if [[ "$1" == *.py ]]; then prog=$1; shift; fi   # (another script instead of bench.py: name it first)
export TMPDIR=/tmp; mkdir -p gpurun_out
n=0; dirs=""
while read -r pass; do
  [ -z "$pass" ] && continue
  out=/tmp/pmcs_${tag}_$n; rm -rf $out; mkdir -p $out
  timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $pass --output-format csv -d $out -o run -- python3 $prog "$@" > gpurun_out/pmcsteps_${tag}_pass$n.log 2>&1 || { echo "pmc_steps.sh: pass $n ($pass) failed"; tail -3 gpurun_out/pmcsteps_${tag}_pass$n.log; exit 1; }
  dirs="$dirs $out"; n=$((n+1))
done < <(python3 tools/pmc_split.py "$ctrs")
python3 - "$kern" $dirs > gpurun_out/pmcsteps_${tag}.txt <<'PY'
import csv, glob, sys, collections, os
kern, dirs = sys.argv[1], sys.argv[2:]
merged = []
for out in dirs:
    rows = collections.OrderedDict()
    for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kern not in r['Kernel_Name']: continue
            d = rows.setdefault(int(r['Dispatch_Id']), {'ms': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})
            d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for i, (k, d) in enumerate(sorted(rows.items())):  # the same program in every pass: dispatch i of one is dispatch i of the other
        if i >= len(merged): merged.append({})
        for a, b in d.items(): merged[i].setdefault(a, b)
for i, d in enumerate(merged[:int(os.environ.get('PMC_ROWS', '16'))]):
    print(i, ' '.join('%s=%s' % (a, ('%.3f' % b) if a == 'ms' else ('%.4e' % b)) for a, b in d.items()))
PY
cat gpurun_out/pmcsteps_${tag}.txt
