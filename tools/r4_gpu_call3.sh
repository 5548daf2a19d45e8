#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 900 python tools/twofront_ab.py > $O/r4_twofront_ab.jsonl 2> $O/r4_twofront_ab.err
TMPC_SCAN_TWOFRONT=1 TMPC_SCAN_WAVES=1 timeout 300 python tools/scan_one_wave_slope.py 2 > $O/r4_slope_twofront_one_wave.jsonl 2>&1
TMPC_SCAN_TWOFRONT=1 timeout 300 python tools/scan_one_wave_slope.py 2 > $O/r4_slope_twofront_two_wave.jsonl 2>&1
tail -c 1500 $O/r4_twofront_ab.jsonl
