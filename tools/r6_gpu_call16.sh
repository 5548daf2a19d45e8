#!/bin/bash
# four-wave tick variant for 21 <= N <= 31: parity tests, then ticks per variant at the shipped horizon
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_quad.py tests/test_gpu_lds_poison.py tests/test_gpu_gaussian.py tests/test_gpu_abi_contracts.py -q -x 2>&1 | tail -15
timeout 600 python tools/tick_shapes.py 100 | tee $O/round6_tick_shipped_horizon.jsonl
