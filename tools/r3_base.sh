#!/bin/bash
# round-3 baseline on the GPU box: parity tests + short bench lines of the four workloads with the in-tree library
set -o pipefail
mkdir -p gpurun_out/r3base
(timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3base/pytest.txt) && \
STEPS=2 tools/abq.sh "c2 c3 c4:128 c5:64@3840x2160" base > gpurun_out/r3base/quick.txt 2>&1
cat gpurun_out/r3base/pytest.txt gpurun_out/r3base/quick.txt
