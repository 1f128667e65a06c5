#!/bin/bash
# how much does ray coherence buy k_bvh?  primary rays only (bounces 1: pixel order, coherent) against the full path (incoherent)
mkdir -p gpurun_out/r3ab
for w in c4:128 c3:256 c5:64@3840x2160; do
for b in 1 2 8; do
  item=$w; wl=${item%%[:@]*}; s=${item#*:}; spp=${s%%@*}; dims=""
  [[ "$item" == *@* ]] && { d=${item##*@}; dims="--width ${d%%x*} --height ${d##*x}"; }
  PTMI_BVH_KERNEL=1 timeout -k 10 300 python bench.py --workload $wl --spp $spp $dims --bounces $b --steps 2 --warmup 1 --cpu-seconds 0 --pmc off --extra-configs off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']; w=r['work_per_ray']
rays=d['config']['rays_per_step']
bv=k['k_bvh']['ms_per_step']
print('$w bounces $b: rays/step %.3e  bvh %.2f ms  bvh_node_visits/ray %.1f tri/ray %.2f  -> %.2f G bvh-visits/s, %.1f ns*CU per 64 visits' % (rays, bv, w['bvh_node_visits'], w['tri_tests'], rays*w['bvh_node_visits']/bv/1e6, bv*1e6*256/(rays*w['bvh_node_visits']/64)))"
done; done 2>&1 | tee gpurun_out/r3ab/coherence.txt
