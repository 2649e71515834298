#!/bin/bash
t() { timeout 300 python tools/big_batch_probe.py $1 2>&1 | grep -E "^R=|^\{.bits" | tail -2 | cut -c1-200; }
t 1; t 24
run() { echo "## $*"; env "$@" 2>&1 | grep -E "mismatch|MISMATCH|rror|mwm verify" | tail -4 | cut -c1-400; }
run python3 tools/fuzz_parity.py 4000 nobpp 403
