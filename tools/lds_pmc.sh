#!/bin/bash
# LDS unit counters of the dominant kernel (cfg 2, bench.py's default launch): instructions, active cycles, bank-conflict cycles
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
BENCH="python bench.py --no-cpu-baseline --latency-reps 0 --no-tight --no-end-to-end --parity-check 0 --index-check-sets 0 --gen-workers 1 --scene-cache /tmp/sc_pmc.npz"
$BENCH --steps 1 --warmup 0 > /dev/null 2>&1      # (fills the scene cache)
for c in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_BUSY_CYCLES"; do
  d=gpurun_out/lds_pmc; rm -rf $d
  timeout 300 rocprofv3 --pmc $c -d $d --output-format csv -- $BENCH --steps 3 --warmup 1 > /dev/null 2>&1
  python - "$c" <<'PY'
import csv, glob, sys, json
rows = []
for f in glob.glob('gpurun_out/lds_pmc/**/*counter_collection.csv', recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if 'tmpc_solve_compact' in r.get('Kernel_Name', '')]
acc = {}
for r in rows:
    acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print(json.dumps({'counters': sys.argv[1], 'mean_per_launch': {k: sum(v) / len(v) for k, v in acc.items()}, 'launches': {k: len(v) for k, v in acc.items()}}))
PY
  rm -rf $d
done | tee gpurun_out/${OUT:-round6_lds_pmc.jsonl}
