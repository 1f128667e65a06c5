#!/bin/bash
# Parameter sweeps of bench.py (short lines: no PMC passes, no CPU legs), one table each under gpurun_out/sweep/.
#   tools/sweep.sh size        k_bvh's rate per node visit against the size of the scene's digests (configs[3] tessellated to 8 k .. 1.05 M triangles)
#   tools/sweep.sh coherence   how much ray coherence buys k_bvh: primary rays only (--bounces 1), two bounces, the full path
#   tools/sweep.sh twin [reps] one context against one context with two / three shards on the same GPU (--devices 0,0: two streams fill each other's tails)
mkdir -p gpurun_out/sweep
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']; w=r['work_per_ray']
rays=d['config']['rays_per_step']; bv=k['k_bvh']['ms_per_step']
print('$1: %.0f Mrays/s  %.2f ms/step  bvh %.1f ms  shade %.1f ms  visits/ray %.1f tri/ray %.2f -> %.1f G bvh-visits/s' % (d['value'], d['ms_per_step'], bv, k['k_shade']['ms_per_step'], w['bvh_node_visits'], w['tri_tests'], rays*w['bvh_node_visits']/max(bv,1e-9)/1e6))"; }
B="--steps 2 --warmup 1 --cpu-seconds 0 --pmc off --extra-configs off"
case $1 in
size)
  for t in 8000 26000 65000 131000 262267 524000 1048000; do
    timeout -k 10 300 python bench.py --workload c4 --spp 64 --tris $t $B 2>/dev/null | line "c4 tris $t (digests $((t*128/1000000)) MB)"
  done 2>&1 | tee gpurun_out/sweep/size_sweep.txt ;;
coherence)
  for w in c4:128 c3:256 c5:64@3840x2160; do for b in 1 2 8; do
    wl=${w%%[:@]*}; s=${w#*:}; spp=${s%%@*}; dims=""
    [[ "$w" == *@* ]] && { d=${w##*@}; dims="--width ${d%%x*} --height ${d##*x}"; }
    timeout -k 10 300 python bench.py --workload $wl --spp $spp $dims --bounces $b $B 2>/dev/null | line "$w bounces $b"
  done; done 2>&1 | tee gpurun_out/sweep/coherence.txt ;;
twin)
  for rep in $(seq 1 ${2:-1}); do
  for item in "c2 --steps 10" "c3 --steps 3" "c5 --spp 64 --width 3840 --height 2160 --steps 2"; do
    for dev in "" "--devices 0,0 --scaling strong"; do
      timeout -k 10 300 python bench.py --workload $item $dev --warmup 1 --cpu-seconds 0 --pmc off --extra-configs off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-50s %-36s %8.0f Mrays/s %9.2f ms/step' % ('$item', '$dev', d['value'], d['ms_per_step']))"
    done
  done; done 2>&1 | tee gpurun_out/sweep/twin.txt ;;
*) echo "usage: tools/sweep.sh size|coherence|twin [reps]"; exit 2;;
esac
