#!/bin/bash
# quick check of a kernel change: driver-form throughput + parity block (cfg 2), cfg 4, cfg 3 as named, the solve parity tests
mkdir -p gpurun_out; O=gpurun_out/quick_ab.jsonl; : > $O
for wl in cfg2 cfg4 cfg3; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-end-to-end --no-tight --no-cpu-baseline --latency-reps 20 \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'wl': '$wl', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'parity': d.get('parity'), 'lat': (d.get('latency_b64') or {}).get('p50_ms')}))" >> $O
done
cat $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact2.py tests/test_gpu_iterations.py -m gpu -x -q 2>&1 | tail -3
