#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=r6_levers_ab3.jsonl bash tools/variants_ab.sh lxu2 mf mw mb c2 > /dev/null 2>&1
cat gpurun_out/r6_levers_ab3.jsonl
