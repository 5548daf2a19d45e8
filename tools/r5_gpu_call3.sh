#!/bin/bash
# round 5, GPU call 3: GPU suite with the SecondOrderUnicycleModel stack and the projection scenes; non-temporal traffic A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -s > $O/r5_call3_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/r5_call3_pytest.log | cut -c1-400; grep "\[parity\]" $O/r5_call3_pytest.log | tail -1
timeout 1500 bash tools/traffic_ab.sh measure 2>&1 | tail -8
