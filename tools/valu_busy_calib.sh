#!/bin/bash
# What does rocprof's VALUBusy (4 x SQ_ACTIVE_INST_VALU / SIMDs / cycles) read for a kernel of nothing but 2-cycle instructions?  (VERDICT round 3, item 5:
# bench.py's cost model says k_shade is 0.75 busy, rocprof's own definition 0.97.)  Runs tools/valu_peak — kernels of ONE opcode each, 8 waves per SIMD, all
# SIMDs saturated — under rocprofv3 --pmc and prints, per opcode, SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU (quad-cycles the counter charges one instruction) next to
# the measured issue cycles per instruction.  usage (GPU box): tools/valu_busy_calib.sh  -> gpurun_out/valu_busy_calib.json
export TMPDIR=/tmp; out=/tmp/vbc; rm -rf $out; mkdir -p $out gpurun_out
[ -x tools/valu_peak ] || hipcc --offload-arch=gfx950 -O2 -o tools/valu_peak tools/valu_peak.hip || exit 1
root=$(pwd)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $out -o run -- $root/tools/valu_peak > $out/valu_peak.json 2> $out/err.txt) || { tail -5 $out/err.txt; exit 1; }
cd $root; python3 - $out gpurun_out/valu_busy_calib.json <<'PY'
import csv, glob, json, sys, collections, re
out, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
order = []
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r['Kernel_Name'], r['Dispatch_Id'])
        if key not in agg: order.append(key)
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        agg[key]['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
peak = json.load(open(out + '/valu_peak.json'))
# one case = one (opcode, waves per SIMD); every case launches its kernel 21 times (1 warm-up + 20): group dispatches by kernel name in order
by_kernel = collections.OrderedDict()
for key in sorted(order, key=lambda k: int(k[1])):
    by_kernel.setdefault(key[0], []).append(agg[key])
res = []
names = list(by_kernel)
for c in peak['cases']:
    res.append(c)
rows = []
src = open('tools/valu_peak.hip').read()
op_names = re.findall(r'"([^"]+)"', src[src.index('kNames[N_OPS]'):src.index('struct WaveRec')])
for name, ds in by_kernel.items():
    m = re.search(r'k_issue<\(?(?:Op\))?(\d+)', name)
    tot = collections.defaultdict(float)
    for d in ds:
        for k, v in d.items(): tot[k] += v
    if tot['SQ_INSTS_VALU'] <= 0: continue
    rows.append({'op': op_names[int(m.group(1))] if m and int(m.group(1)) < len(op_names) else None, 'kernel': name.split('(')[0], 'dispatches': len(ds), 'quad_cycles_charged_per_instr': tot['SQ_ACTIVE_INST_VALU'] / tot['SQ_INSTS_VALU'],
                 'rocprof_valu_busy': 4.0 * tot['SQ_ACTIVE_INST_VALU'] / 1024.0 / (tot['SQ_BUSY_CYCLES'] / 32.0), 'lanes_per_instr': tot['SQ_THREAD_CYCLES_VALU'] / tot['SQ_ACTIVE_INST_VALU'] / 1.0 if tot['SQ_ACTIVE_INST_VALU'] else None})
json.dump({'_note': 'tools/valu_peak under rocprofv3 --pmc: per single-opcode kernel (all its dispatches, 1..8 waves per SIMD), the quad-cycles SQ_ACTIVE_INST_VALU charges per instruction and '
                    'rocprof VALUBusy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / kernel cycles.  Read next to profiles/valu_peak.json (measured issue cycles per instruction)', 'rows': rows,
           'valu_peak_cases_of_this_run': peak['cases']}, open(dst, 'w'), indent=1)
for r in rows: print(r)
PY
