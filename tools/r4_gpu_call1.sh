#!/bin/bash
# (the TMPC_* kernel-selection switches exist in the lab build of the library only: round 6)
export TMPC_HIP_LIBRARY=${TMPC_HIP_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/mpc_planner_amd/libtmpc_hip_lab.so}
# round 4, GPU call 1: one-wave vs two-wave parallel-in-time kernels on saturated launches + SQ counters (tools/scan_one_wave_slope.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
python $R/tools/scan_one_wave_slope.py 0 1 2 > $O/r4_slope_two_wave.jsonl 2> $O/r4_slope_two_wave.err
TMPC_SCAN_WAVES=1 python $R/tools/scan_one_wave_slope.py 2 > $O/r4_slope_one_wave.jsonl 2> $O/r4_slope_one_wave.err
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  TMPC_SCAN_WAVES=1 timeout 300 rocprofv3 --pmc $set -d $O/r4_pmc_one_wave_$tag --output-format csv -- python $R/tools/scan_one_wave_slope.py 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $set -d $O/r4_pmc_modes_$tag --output-format csv -- python $R/tools/scan_one_wave_slope.py 0 2 > /dev/null 2>&1
done
cd $R
timeout 600 python bench.py --no-lanes --no-tight --latency-reps 20 --cpu-scenes 4 > $O/r4_baseline_bench.json 2> $O/r4_baseline_bench.err
tail -c 600 $O/r4_slope_one_wave.jsonl; tail -c 300 $O/r4_baseline_bench.err
