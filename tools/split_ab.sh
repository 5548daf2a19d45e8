cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for sc in 64 96 128 192 256; do
for st in prio0_0 prio3_2; do
  export TMPC_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/exp/libtmpc_hip_$st.so
  python bench.py --scenes $sc --steps 30 --warmup 5 --no-cpu-baseline --latency-reps 0 --no-tight --no-end-to-end --parity-check 0 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'variant':'$st','scenes':$sc,'rounds':$sc*64/2048.0,'value':d['value'],'kernel_ms_avg':d['roofline']['kernel_ms_avg']}))"
done; done | tee gpurun_out/round5_k_prio_vs_launch_size.jsonl
