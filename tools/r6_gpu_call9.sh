#!/bin/bash
# round 6, GPU call 9: store merging in the factor / sweeps on top of the own-pair forward sweep; then the GPU suite on the product library
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=r6_levers_ab2.jsonl bash tools/variants_ab.sh base lxu merged > /dev/null 2>&1
cat gpurun_out/r6_levers_ab2.jsonl
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_suite2.log 2>&1
tail -4 gpurun_out/r6_gpu_suite2.log
