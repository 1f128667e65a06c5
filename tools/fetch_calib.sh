#!/bin/bash
# Calibrates FETCH_SIZE / WRITE_SIZE on known byte counts (tools/fetch_calib.hip); writes gpurun_out/fetch_calib.json
export TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/fc_$c -o run -- $OLDPWD/tools/fetch_calib > /tmp/fc_$c.out 2> /tmp/fc_$c.err) || { tail -3 /tmp/fc_$c.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, json
want = json.loads(open('/tmp/fc_FETCH_SIZE.out').read().strip().splitlines()[-1])
got = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('/tmp/fc_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if k.startswith('calib_'): got.setdefault(k, {})[r['Counter_Name']] = got.get(k, {}).get(r['Counter_Name'], 0.0) + float(r['Counter_Value']) * 1024.0
out = {"_note": "counter values x 1024 (unit KB) against the bytes the kernels asked for; table 8 GiB (no cache can hold it)"}
for k, w in want.items():
    e = dict(w); e.update({n: v for n, v in got.get(k, {}).items()})
    if 'read_bytes' in w and 'FETCH_SIZE' in e: e['true_read_over_FETCH_SIZE'] = w['read_bytes'] / e['FETCH_SIZE']
    if 'write_bytes' in w and 'WRITE_SIZE' in e: e['true_write_over_WRITE_SIZE'] = w['write_bytes'] / e['WRITE_SIZE']
    out[k] = e
json.dump(out, open('gpurun_out/fetch_calib.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
