cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for pass in 1 2; do
for spec in "cur:0" "consths:0" "cur:"; do
  name=${spec%%:*}; hp=${spec#*:}
  ( export TMPC_HIP_LIBRARY=$PWD/build/exp/libtmpc_hip_$name.so; if [ -n "$hp" ]; then export TMPC_EXP_HPAD=$hp; fi
    python bench.py --workload jackal --steps 20 --warmup 3 --no-cpu-baseline --no-tight --no-end-to-end --parity-check 0 --index-check-sets 0 --latency-reps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$name hpad=${hp:-auto}', round(d['value']), round(d['ms_per_step'],3))" )
done; done
