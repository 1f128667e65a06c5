#!/bin/bash
# Runs on the GPU box (through gpurun): parity tests, smoke, the bench line as the driver runs it (with its own rocprofv3 --pmc passes), the other
# workloads, rocprofv3 kernel stats of the bench command, the microbenchmarks the rooflines rest on, round 4's measurements (lane / stopwatch tallies of
# k_shade, rocprof's VALUBusy on single-opcode kernels, one rank of 1 / 2 / 4 / 8, the OBJ parser against the reference's reader).  Everything lands in
# gpurun_out/rel/; tools/collect_profiles.py rNN files it under profiles/.   usage: tools/release_pass.sh [quick | extras]
set -o pipefail
R=gpurun_out/rel; mkdir -p $R; export TMPDIR=/tmp
root=$(pwd)
if [ "$1" != quick ] && [ "$1" != extras ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $R/pytest_gpu.txt || exit 1
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $R/smoke.txt || exit 1
fi
if [ "$1" != extras ]; then
SECONDS=0
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $R/bench_c2.json 2> $R/bench_c2.err || exit 1
echo "bench.py --steps 20 --warmup 5: wall ${SECONDS} s" | tee $R/bench_c2_wall.txt
timeout -k 10 600 python bench.py --workload c3 --steps 2 --cpu-seconds 10 --extra-configs off > $R/bench_c3.json 2> $R/bench_c3.err || exit 1
timeout -k 10 600 python bench.py --workload c4 --steps 1 --cpu-seconds 10 --pmc-timeout 400 > $R/bench_c4.json 2> $R/bench_c4.err || exit 1
timeout -k 10 600 python bench.py --workload c5 --width 3840 --height 2160 --spp 128 --steps 1 --cpu-seconds 10 --pmc-timeout 400 > $R/bench_c5.json 2> $R/bench_c5.err || exit 1
# the opt-in SAH tree (lib/BVH/bvhNode.js:108-283, built on the GPU: ptmi_build_scene_bvh_sah) on the same box, with its own counter passes
timeout -k 10 600 python bench.py --workload c3 --bvh sah --steps 2 --cpu-seconds 0 --extra-configs off > $R/bench_c3_sah.json 2> $R/bench_c3_sah.err || exit 1
timeout -k 10 600 python bench.py --workload c4 --bvh sah --steps 1 --cpu-seconds 0 --pmc-timeout 400 > $R/bench_c4_sah.json 2> $R/bench_c4_sah.err || exit 1
timeout -k 10 600 python bench.py --workload c5 --bvh sah --width 3840 --height 2160 --spp 128 --steps 1 --cpu-seconds 0 --pmc-timeout 400 > $R/bench_c5_sah.json 2> $R/bench_c5_sah.err || exit 1
# rocprofv3 kernel stats of the bench command itself (its own --pmc child passes off: one profiler at a time)
export PTMI_PLACEMENT_TRIES=1  # (the placement search's dry runs are launches of the same kernels: kept out of the averages)
for w in c2 c3 c4 c5; do
  rm -rf /tmp/ks_$w
  x=""; [ $w = c4 ] && x="--spp 128"; [ $w = c5 ] && x="--width 3840 --height 2160 --spp 64"
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$w -o run -- python3 $root/bench.py --workload $w $x --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --extra-configs off > $root/$R/ks_$w.log 2>&1) || exit 1
  cp $(find /tmp/ks_$w -name '*kernel_stats.csv' | head -1) $R/kernel_stats_$w.csv || exit 1
done
unset PTMI_PLACEMENT_TRIES
fi
if [ "$1" != quick ]; then
[ -x tools/valu_peak ] && timeout -k 10 300 tools/valu_peak > $R/valu_peak.json 2> $R/valu_peak.err
bash tools/valu_busy_calib.sh > $R/valu_busy_calib.txt 2>&1 && cp gpurun_out/valu_busy_calib.json $R/
# the gather path's ceiling (k_bvh's roofline) and the sweeps that show k_bvh sits on it
[ -x tools/gather_probe ] && timeout -k 10 300 tools/gather_probe > $R/gather_probe.json 2> $R/gather_probe.err
bash tools/sweep.sh size > /dev/null 2>&1; cp gpurun_out/sweep/size_sweep.txt $R/ 2>/dev/null
timeout -k 10 900 python tools/shard_sim.py c2 c3 c4 c5 > $R/shard_sim.txt 2>&1 && cp gpurun_out/shard_sim.json $R/
timeout -k 10 400 python tools/obj_parse_bench.py > $R/obj_parse.json 2> $R/obj_parse.err
if [ -f webgpu-path-tracer_amd/variants/libptmi_lanes.so ]; then
  for w in c2 c3 c5; do timeout -k 10 200 python tools/shade_lanes.py run $w > $R/shade_lanes_$w.txt 2>&1 && cp gpurun_out/shade_lanes_$w.json $R/; done
fi
fi
echo release pass done
