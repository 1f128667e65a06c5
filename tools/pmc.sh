#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <bench args...>   — rocprofv3 --pmc over `python3 bench.py <args>`, per-kernel sums in gpurun_out/pmc_<tag>.json.
# The counter list may be of any length: tools/pmc_split.py cuts it into passes that fit the blocks' counter slots (an over-subscribed pass makes
# rocprofv3 abort at the first HIP call — round 3 lost five leases to that), one rocprofv3 run per pass, results merged per kernel.
export PTMI_PLACEMENT_TRIES=${PTMI_PLACEMENT_TRIES:-1}  # no placement search under the profiler: its dry runs are launches of the kernels being profiled
tag=$1; ctrs=$2; shift 2
export TMPDIR=/tmp; mkdir -p gpurun_out
n=0; dirs=""
while read -r pass; do
  [ -z "$pass" ] && continue
  out=/tmp/pmc_${tag}_$n; rm -rf $out; mkdir -p $out
  timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $pass --output-format csv -d $out -o run -- python3 bench.py "$@" > gpurun_out/pmc_${tag}_pass$n.log 2>&1 || { echo "pmc.sh: pass $n ($pass) failed"; tail -3 gpurun_out/pmc_${tag}_pass$n.log; exit 1; }
  dirs="$dirs $out"; n=$((n+1))
done < <(python3 tools/pmc_split.py "$ctrs")
python3 - "gpurun_out/pmc_${tag}.json" $dirs <<'PY'
import csv, glob, json, sys, collections
dst, dirs = sys.argv[1], sys.argv[2:]
res = {}
for out in dirs:
    seen = {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('ptmi::', '').replace('void ', '').split('(')[0].strip()
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            n[(k, r['Counter_Name'])] += 1
            seen.setdefault((k, r['Dispatch_Id']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    dur = collections.defaultdict(float)
    for (k, _), ms in seen.items(): dur[k] += ms
    for k, v in agg.items():
        e = res.setdefault(k, {})
        e.update(v)
        e['launches'] = max(n[(k, c)] for c in v)
        e.setdefault('ms_total', dur.get(k, 0.0))  # (of the first pass that saw the kernel: passes differ by the profiler's overhead only)
json.dump(res, open(dst, 'w'), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get('ms_total', 0))[:6]:
    print(k, {a: (round(b, 1) if b < 1e4 else f'{b:.3e}') for a, b in v.items()})
PY
