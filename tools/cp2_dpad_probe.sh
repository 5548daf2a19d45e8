cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
# (the TMPC_* kernel-selection switches exist in the lab build of the library only: round 6)
export TMPC_HIP_LIBRARY=${TMPC_HIP_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/mpc_planner_amd/libtmpc_hip_lab.so}
for v in 0 1 2 3 ""; do
  for wl in "cfg3 --sets 8 --steps 20 --warmup 3" "jackal --steps 8 --warmup 2"; do
  ( if [ -n "$v" ]; then export TMPC_EXP_DPAD=$v; fi
    python bench.py --workload $wl --no-cpu-baseline --no-tight --no-end-to-end --parity-check 0 --index-check-sets 0 --latency-reps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('dpad=${v:-model}', '$wl'.split()[0], round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernel'][:80])" )
  done
done
