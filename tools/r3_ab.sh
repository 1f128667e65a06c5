#!/bin/bash
# round-3 A/B on the GPU box: parity tests with the in-tree library, then short bench lines per variant.  usage: tools/r3_ab.sh "<items>" variant...
set -o pipefail
mkdir -p gpurun_out/r3ab
items=$1; shift
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3ab/pytest.txt; rc=$?
  cat gpurun_out/r3ab/pytest.txt
  [ $rc -ne 0 ] && exit 1
fi
STEPS=${STEPS:-2} tools/abq.sh "$items" "$@" 2>&1 | tee gpurun_out/r3ab/table_$(date +%H%M%S).txt
