#!/bin/bash
# repeated, alternating: single stream vs two shards on the same GPU
for rep in 1 2 3; do
for item in "c2 --steps 20" "c3 --steps 4" "c5 --spp 64 --width 3840 --height 2160 --steps 3"; do
  for dev in "" "--devices 0,0 --scaling strong"; do
    timeout -k 10 300 python bench.py --workload $item $dev --warmup 2 --cpu-seconds 0 --pmc off --extra-configs off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-50s %-36s %8.0f Mrays/s %9.2f ms/step' % ('$item', '$dev', d['value'], d['ms_per_step']))"
  done
done; done
