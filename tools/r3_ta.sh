#!/bin/bash
# is k_bvh bound by the texture-addresser / L1 lookup rate of its 16-byte record gathers?  TA / TCP / TD counters of one bench step
# (few counters per pass: these blocks have two to four counter slots)
w=${1:-c4}; spp=${2:-64}; shift 2
A="--workload $w --spp $spp --steps 1 --warmup 0 --cpu-seconds 0 --pmc off --extra-configs off $@"
tools/pmc.sh ta1_$w "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" $A || exit 1
tools/pmc.sh ta2_$w "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" $A || exit 1
tools/pmc.sh tcp1_$w "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" $A || exit 1
tools/pmc.sh tcp2_$w "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" $A || exit 1
tools/pmc.sh td_$w "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" $A || exit 1
