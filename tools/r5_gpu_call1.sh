#!/bin/bash
# round 5, GPU call 1: chain-floor microbenchmark, the GPU suite with the tightened regression line, the starting bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 build/tools/chain_floor 4096 > gpurun_out/round5_chain_floor_raw.json 2> gpurun_out/round5_chain_floor.err
echo "chain_floor rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r5_call1_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r5_call1_pytest.log; grep "\[parity\]" gpurun_out/r5_call1_pytest.log | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round5_a_baseline_bench.json 2> gpurun_out/r5_call1_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/round5_a_baseline_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity'], d['cpu_baseline']['value'], d.get('value_end_to_end'))
print(json.dumps(d['end_to_end']['vs_resident_step']), json.dumps(d['end_to_end']['input_rotation']))
PY
