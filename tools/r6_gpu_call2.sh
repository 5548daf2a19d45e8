#!/bin/bash
# round 6, GPU call 2: tight-tolerance disagreements by Riccati form and tolerance
export TMPDIR=/tmp
mkdir -p gpurun_out
for spec in "0 1e-9 cfg2" "1 1e-9 cfg2" "0 1e-8 cfg2" "1 1e-8 cfg2" "0 1e-7 cfg2" "0 1e-9 cfg3" "1 1e-9 cfg3" "0 1e-8 cfg3" "0 1e-9 cfg4" "1 1e-9 cfg4" "0 1e-9 cfg5" "1 1e-9 cfg5"; do
  set -- $spec
  sc=512; [ "$3" != "cfg2" ] && sc=32
  timeout 900 python tools/tight_flip.py --form $1 --qp-tol $2 --workload $3 --scenes $sc --max-cases 6 --out gpurun_out/round6_tight_flip_$3_form$1_$2.json > gpurun_out/r6_tf_$3_$1_$2.log 2>&1
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/round6_tight_flip_$3_form$1_$2.json"))
    print("$3 form $1 tol $2:", d["trajectories"], "trajectories,", d["mismatching"], "mismatching", d["mismatch_kinds"], "rel where counts agree", d["parity_max_rel_where_counts_agree"])
except Exception as e:
    print("$3 form $1 tol $2: FAILED", e)
PY
done
