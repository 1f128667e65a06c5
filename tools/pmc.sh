#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <bench args...>   — one rocprofv3 --pmc pass, per-kernel sums in gpurun_out/pmc_<tag>.json
tag=$1; ctrs=$2; shift 2
out=/tmp/pmc_$tag; rm -rf $out; mkdir -p $out gpurun_out
export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctrs --output-format csv -d $out -o run -- python3 bench.py "$@" > gpurun_out/pmc_${tag}.log 2>&1
python3 - "$out" "gpurun_out/pmc_${tag}.json" <<'PY'
import csv, glob, json, sys, collections, re
out, dst = sys.argv[1], sys.argv[2]
seen = {}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('ptmi::', '').replace('void ', '').split('(')[0].strip()
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
        seen.setdefault((k, r['Dispatch_Id']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
res = {k: dict(v, launches=max(n[(k, c)] for c in v)) for k, v in agg.items()}
dur = collections.defaultdict(float)
for (k, _), ms in seen.items(): dur[k] += ms
for k in res: res[k]['ms_total'] = dur.get(k, 0.0)
json.dump(res, open(dst, 'w'), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get('ms_total', 0))[:6]:
    print(k, {a: (round(b, 1) if b < 1e4 else f'{b:.3e}') for a, b in v.items()})
PY
