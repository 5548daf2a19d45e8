#!/bin/bash
# one-wave kernels: stage sums pre-reduced over a stage's three lanes by wave shifts (wshl3) vs base3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
OUT=r6_wshl3_ab.jsonl bash tools/variants_ab.sh base3 wshl3 > /dev/null 2>&1; cat $O/r6_wshl3_ab.jsonl
TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_wshl3.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact2.py tests/test_gpu_layout.py tests/test_gpu_lds_poison.py -m gpu -x -q 2>&1 | tail -4
