#!/bin/bash
# eight lanes per stage (four-wave kernel, 21 <= N <= 31): the second quad folded by row_shl:4 (q8) vs tsum6
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O; : > $O/r6_q8_ab.jsonl
timeout 900 python -m pytest tests/test_gpu_quad.py tests/test_gpu_gaussian.py tests/test_gpu_lds_poison.py -q -x 2>&1 | tail -3
for pass in 1 2; do for v in tsum6 q8; do
  TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so python tools/tick_shapes.py 100 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    if 'N 30' in d['shape'] or 'curvature' in d['shape'] or 'shipped' in d['shape']:
        print(json.dumps({'variant':'$v','pass':$pass,'tick':d['shape'][:40],'planners':d['planners'],'mode_3':[d['by_mode']['mode_3'][k] for k in ('p50_ms','kernel_ms')]+[sum(d['by_mode']['mode_3'][k] for k in ('exit_code_mismatch','sqp_iter_mismatch','ipm_iter_mismatch'))]}))" | tee -a $O/r6_q8_ab.jsonl
done; done
