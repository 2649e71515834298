#!/bin/bash
# Predict end to end with wide pools at scale: random 300-nt records, c=nobpp (G1 + G2 + N + E + H), device pools
# against the host-driven loop (SQ_NO_POOL=1), sha256 of the printed text.  usage: bash tools/pool_scale_probe.sh > out.txt
run() { echo "## records=$1 poollim=$2 ${3:+($3)}"; env $3 PROBE_SHA=1 PROBE_N=$1 python3 tools/predict_probe.py S300 $4 nobpp $2 2>&1 | grep -E "Predict|sha256|rror"; }
run 500 1000 "" 3
run 500 1000 SQ_NO_POOL=1 2
run 3000 1000 "" 2
run 3000 1000 SQ_NO_POOL=1 1
run 3000 100 "" 2
run 3000 100 SQ_NO_POOL=1 1
run 3000 10 "" 2
run 3000 10 SQ_NO_POOL=1 1
echo "## fold phases, 500 records, poollim 1000 (SQ_TIMING)"
SQ_TIMING=1 PROBE_N=500 python3 tools/predict_probe.py S300 3 nobpp 2>&1 | grep -E "rounds=|\[sq_fold\] total|tail:" | tail -3 | cut -c1-200
