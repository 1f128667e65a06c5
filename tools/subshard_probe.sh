#!/bin/bash
# one rank of a strong-scaled N = 8 run of configs[1] (1/8 of the pixels x 64 spp = 16.6 M paths per step) as 1, 2, 4 shards on this one GPU (streams overlap)

for dev in "" "--devices 0,0" "--devices 0,0,0,0"; do
  for sz in "c2:8" "c3:32"; do
    w=${sz%%:*}; spp=${sz##*:}
    timeout -k 10 300 python bench.py --workload $w --spp $spp --scaling strong $dev --steps 10 --warmup 2 --cpu-seconds 0 --pmc off --extra-configs off --watchdog-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w spp $spp [$dev]:', round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step', d['ms_per_step_stats'])"
  done
done
