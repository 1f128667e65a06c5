#!/bin/bash
# k_tail's parked walkers (PTMI_TAIL_PARK lanes; trees of >= 12 levels): lone frames and 8-frame batches.  usage: tools/park_probe.sh ["0 8 16"]
# (profiles/r05_tail_park_shallow.txt was taken with a build whose depth gate could be lowered: PTMI_TAIL_PARK_MIN_DEPTH=0, since removed)
for p in ${1:-0 8 12 16 24}; do echo "== PTMI_TAIL_PARK=$p"; PTMI_TAIL_PARK=$p python tools/tail_probe.py -1 2>&1 | grep limit; done
