cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for i in 1 2; do
python bench.py --workload cfg4 --steps 50 --warmup 5 --no-cpu-baseline --no-tight --no-end-to-end --parity-check 64 --index-check-sets 0 --latency-reps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); p=d.get('parity') or {}
print('cfg4', round(d['value']), round(d['ms_per_step'],3), p.get('exit_code_mismatch'), p.get('ipm_iter_mismatch'), p.get('parity_max_rel'), d['roofline']['kernel'][:110])"
done
timeout 900 python -m pytest tests/test_gpu_layout.py tests/test_gpu_parity.py tests/test_gpu_compact2.py tests/test_gpu_iterations.py -m gpu -x -q 2>&1 | tail -3
