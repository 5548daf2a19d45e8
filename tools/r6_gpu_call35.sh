#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_compact2.py tests/test_gpu_quad.py tests/test_gpu_parity.py tests/test_gpu_lds_poison.py tests/test_gpu_tight.py tests/test_gpu_layout.py -m gpu -q -x 2>&1 | tail -4
python tools/n30_throughput.py 2>/dev/null | tee $O/round6_n30_one_wave_ab.jsonl
TMPC_HIP_LIBRARY=$R/mpc_planner_amd/libtmpc_hip_lab.so TMPC_NO_ONE_WAVE_N30=1 python tools/n30_throughput.py 2>/dev/null | tee -a $O/round6_n30_one_wave_ab.jsonl
