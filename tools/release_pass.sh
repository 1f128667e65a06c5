#!/bin/bash
# Runs on the GPU box (through gpurun): parity tests, smoke, the four bench workloads, rocprofv3 kernel stats and the
# FETCH_SIZE / WRITE_SIZE passes.  Everything lands in gpurun_out/rel/; tools/collect_profiles.py files it under profiles/.
set -o pipefail
R=gpurun_out/rel; mkdir -p $R; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $R/pytest_gpu.txt || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $R/smoke.txt || exit 1
timeout -k 10 600 python bench.py > $R/bench_c2.json 2> $R/bench_c2.err || exit 1
timeout -k 10 600 python bench.py --workload c3 --steps 2 > $R/bench_c3.json 2> $R/bench_c3.err || exit 1
timeout -k 10 600 python bench.py --workload c4 --steps 1 --cpu-seconds 10 > $R/bench_c4.json 2> $R/bench_c4.err || exit 1
timeout -k 10 600 python bench.py --workload c5 --width 3840 --height 2160 --spp 128 --steps 1 --cpu-seconds 10 > $R/bench_c5.json 2> $R/bench_c5.err || exit 1
for w in c2 c3; do
  extra=""
  rm -rf /tmp/ks_$w; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$w -o run -- python3 $OLDPWD/bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 $extra > $OLDPWD/$R/ks_$w.log 2>&1) || exit 1
  cp $(find /tmp/ks_$w -name '*kernel_stats.csv' | head -1) $R/kernel_stats_$w.csv || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    tools/pmc.sh ${w}_$c $c --workload $w --steps 1 --warmup 0 --cpu-seconds 0 $extra > /dev/null || exit 1
    cp gpurun_out/pmc_${w}_$c.json $R/
  done
done
# what bounds the kernels: VALU instruction counts, busy cycles, active lanes (cited in DESIGN.md section 5)
tools/pmc.sh valu_c2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" --workload c2 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null || exit 1
tools/pmc.sh valu_c4 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" --workload c4 --spp 64 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null || exit 1
cp gpurun_out/pmc_valu_c2.json gpurun_out/pmc_valu_c4.json $R/
echo release pass done
