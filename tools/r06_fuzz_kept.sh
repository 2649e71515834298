#!/bin/bash
# parity soak of the list form (CPU oracle vs HIP engine): summary lines
cd $GRAFT_REPO_ROOT
run() { echo "## $*"; env "$@" 2>&1 | grep -E "mismatch|MISMATCH|rror|records" | tail -4 | cut -c1-400; }
run FUZZ_NMIN=257 FUZZ_NMAX=700 FUZZ_POOLLIM=100 python3 tools/fuzz_parity.py 120 greedynobpp 110
run FUZZ_NMIN=257 FUZZ_NMAX=450 python3 tools/fuzz_parity.py 160 nobpp 111
run FUZZ_NMIN=257 FUZZ_NMAX=380 SQ_KEPT_GB=0.05 python3 tools/fuzz_parity.py 100 alt 112
run FUZZ_NMIN=200 FUZZ_NMAX=600 python3 tools/fuzz_parity.py 160 500nobpp 105
