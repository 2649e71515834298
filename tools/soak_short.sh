#!/bin/bash
# A shorter parity / stability soak for late changes (summary lines only).  usage: bash tools/soak_short.sh > out.txt
run() { echo "## $*"; env "$@" 2>&1 | grep -E "mismatch|MISMATCH|rror|differ|flaky|equal|bad" | tail -8 | cut -c1-300; }
run python3 tools/fuzz_parity.py 2000 nobpp 401
run FUZZ_POOLLIM=1 python3 tools/fuzz_parity.py 1500 fastest 402
run FUZZ_POOLLIM=7 python3 tools/fuzz_parity.py 1000 alt 403
run SQ_CTX_MIN_N=0 FUZZ_NMIN=256 FUZZ_NMAX=600 FUZZ_POOLLIM=1 python3 tools/fuzz_parity.py 150 fastest 404
run SQ_CTX_MIN_N=0 FUZZ_POOLLIM=1 python3 tools/fuzz_parity.py 800 fastest 407
run python3 tools/fuzz_options.py 60 32 405
run python3 tools/fuzz_align.py 30 406
run python3 tools/repeat_soak.py 300
run python3 tools/concurrency_soak.py 6 15 nobpp 1000
run python3 tools/fresh_batch_check.py 40
run python3 tools/scale_soak.py
