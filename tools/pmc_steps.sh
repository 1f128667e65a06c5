#!/bin/bash
# usage: tools/pmc_steps.sh <tag> <kernel substring> "<counters>" <bench args...> — one rocprofv3 --pmc pass, counters PER DISPATCH of one kernel
# (the steps of a render: launch k of a step is bounce k), first 16 (PMC_ROWS) dispatches, in gpurun_out/pmcsteps_<tag>.txt
tag=$1; kern=$2; ctrs=$3; shift 3
prog="bench.py"; if [[ "$1" == *.py ]]; then prog=$1; shift; fi   # (another script instead of bench.py: name it first)
out=/tmp/pmcs_$tag; rm -rf $out; mkdir -p $out gpurun_out
export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctrs --output-format csv -d $out -o run -- python3 $prog "$@" > gpurun_out/pmcsteps_${tag}.log 2>&1
python3 - "$out" "$kern" > gpurun_out/pmcsteps_${tag}.txt <<'PY'
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r['Kernel_Name']: continue
        d = rows.setdefault(int(r['Dispatch_Id']), {'ms': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for i, (k, d) in enumerate(sorted(rows.items())[:int(__import__('os').environ.get('PMC_ROWS', '16'))]):
    print(i, ' '.join('%s=%s' % (a, ('%.3f' % b) if a == 'ms' else ('%.4e' % b)) for a, b in d.items()))
PY
cat gpurun_out/pmcsteps_${tag}.txt
