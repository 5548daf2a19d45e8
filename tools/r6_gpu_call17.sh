#!/bin/bash
# MIRROR rotation A/B: base (rsqrt-rcp-rsqrt chain), rotA (two rsqrt), rotB (rotA + paired round-robin 4 x 4 sweep in the latency kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
OUT=r6_rot_ab.jsonl bash tools/variants_ab.sh base rotA rotB > /dev/null 2>&1; cat $O/r6_rot_ab.jsonl
for v in base rotA rotB; do
  TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so python tools/tick_shapes.py 100 > $O/r6_rot_tick_$v.jsonl 2>/dev/null
  python - $v <<'PY'
import json, sys
for l in open(f"gpurun_out/r6_rot_tick_{sys.argv[1]}.jsonl"):
    d = json.loads(l)
    print(sys.argv[1], d["shape"][:40], d["planners"], {m: (v["p50_ms"], v["kernel_ms"], v["exit_code_mismatch"] + v["sqp_iter_mismatch"] + v["ipm_iter_mismatch"], "%.1e" % v["max_rel"]) for m, v in d["by_mode"].items()})
PY
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
