#!/bin/bash
# A last parity soak on the final tree (CPU oracle vs HIP engine); summary lines only.  usage: bash tools/fuzz_final.sh > out.txt
run() { echo "## $*"; env "$@" 2>&1 | grep -E "mismatch|MISMATCH|rror" | tail -4 | cut -c1-400; }
run python3 tools/fuzz_parity.py 6000 nobpp 101
run FUZZ_POOLLIM=1 python3 tools/fuzz_parity.py 4000 fastest 102
run FUZZ_POOLLIM=7 python3 tools/fuzz_parity.py 4000 alt 103
run FUZZ_POOLLIM=100 python3 tools/fuzz_parity.py 3000 greedynobpp 104
run FUZZ_NMIN=200 FUZZ_NMAX=600 python3 tools/fuzz_parity.py 160 500nobpp 105
run FUZZ_NMIN=300 FUZZ_NMAX=900 FUZZ_POOLLIM=1 python3 tools/fuzz_parity.py 120 fastest 106
run python3 tools/fuzz_options.py 200 32 201
run python3 tools/fuzz_align.py 80 301
# long sequences: the scoring kernel's bound, the context tables (forced on for every length: the default starts at 800 nt) with and without crossing stems
run SQ_CTX_MIN_N=0 FUZZ_NMIN=256 FUZZ_NMAX=700 FUZZ_POOLLIM=1 python3 tools/fuzz_parity.py 400 fastest 107
run SQ_CTX_MIN_N=0 FUZZ_NMIN=256 FUZZ_NMAX=520 FUZZ_POOLLIM=25 python3 tools/fuzz_parity.py 200 nobpp 108
run SQ_CTX_MIN_N=0 FUZZ_NMIN=256 FUZZ_NMAX=420 FUZZ_POOLLIM=1000 python3 tools/fuzz_parity.py 120 alt 109
# the list form of the pool round kernel (257-1,024 nt, pools wider than one): every structure reads the list its parent left
run FUZZ_NMIN=257 FUZZ_NMAX=700 FUZZ_POOLLIM=100 python3 tools/fuzz_parity.py 120 greedynobpp 110
run FUZZ_NMIN=257 FUZZ_NMAX=450 python3 tools/fuzz_parity.py 160 nobpp 111
run FUZZ_NMIN=257 FUZZ_NMAX=380 SQ_KEPT_GB=0.05 python3 tools/fuzz_parity.py 100 alt 112
