#!/bin/bash
# k_bvh's rate per node visit against the size of the scene's digests (does the traversal run at the speed of the memory system?)
mkdir -p gpurun_out/r3ab
for t in 8000 26000 65000 131000 262267 524000 1048000; do
  timeout -k 10 300 python bench.py --workload c4 --spp 64 --tris $t --steps 2 --warmup 1 --cpu-seconds 0 --pmc off --extra-configs off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']; w=r['work_per_ray']
rays=d['config']['rays_per_step']; bv=k['k_bvh']['ms_per_step']
print('c4 tris $t (digests %.1f MB): %.0f Mrays/s  bvh %.1f ms  visits/ray %.1f tri/ray %.2f -> %.1f G bvh-visits/s' % ($t*128/1e6, d['value'], bv, w['bvh_node_visits'], w['tri_tests'], rays*w['bvh_node_visits']/bv/1e6))"
done 2>&1 | tee gpurun_out/r3ab/size_sweep.txt
