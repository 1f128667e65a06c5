#!/bin/bash
# one context vs one context with two shards on the same GPU (two streams fill each other's k_bvh tails): --devices 0,0 --scaling strong
mkdir -p gpurun_out/r3ab
for item in "c2 --steps 10" "c3 --steps 3" "c4 --spp 128 --steps 2" "c5 --spp 64 --width 3840 --height 2160 --steps 2"; do
  for dev in "" "--devices 0,0 --scaling strong" "--devices 0,0,0 --scaling strong"; do
    timeout -k 10 300 python bench.py --workload $item $dev --warmup 1 --cpu-seconds 0 --pmc off --extra-configs off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-50s %-36s %8.0f Mrays/s %9.2f ms/step' % ('$item', '$dev', d['value'], d['ms_per_step']))"
  done
done 2>&1 | tee gpurun_out/r3ab/twin.txt
