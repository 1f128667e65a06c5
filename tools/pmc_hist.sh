#!/bin/bash
# usage: tools/pmc_hist.sh <tag> <bench args...> — dynamic instruction mix and wait breakdown per kernel, three rocprofv3 --pmc passes
# (8 SQ counters each) over one bench step; per-kernel sums land in gpurun_out/pmc_<tag>_{mix32,mix64,wait}.json
export PTMI_PLACEMENT_TRIES=${PTMI_PLACEMENT_TRIES:-1}  # no placement search under the profiler: its dry runs are launches of the kernels being profiled
tag=$1; shift
tools/pmc.sh ${tag}_mix32 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "$@" --steps 1 --warmup 0 --cpu-seconds 0 --pmc off --extra-configs off > /dev/null || exit 1
tools/pmc.sh ${tag}_mix64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "$@" --steps 1 --warmup 0 --cpu-seconds 0 --pmc off --extra-configs off > /dev/null || exit 1
tools/pmc.sh ${tag}_wait "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" "$@" --steps 1 --warmup 0 --cpu-seconds 0 --pmc off --extra-configs off > /dev/null || exit 1
echo pmc_hist $tag done
