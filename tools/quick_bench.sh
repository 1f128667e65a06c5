#!/bin/bash
# usage: tools/quick_bench.sh "c2 c3:128 c4:64" -- short bench lines for A/B runs (workload[:spp]); no PMC passes, no CPU legs
for item in $1; do
  w=${item%%:*}; spp=""; [[ "$item" == *:* ]] && spp="--spp ${item##*:}"
  timeout -k 10 300 python bench.py --workload $w --steps ${STEPS:-2} --warmup 1 --cpu-seconds 0 --pmc off --extra-configs off $spp 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w:', round(d['value']), 'Mrays/s', round(d['ms_per_step'],1), 'ms', {k[2:]: round(v['ms_per_step'],2) for k,v in d['roofline']['kernels'].items()})"
done
