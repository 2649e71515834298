#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag8; mkdir -p $o
{
for n in "1000 1024 0" "300 10000 0" "2000 1000 1" "1000 128 0" "2000 125 1"; do python tools/rounds_probe.py $n 5 2>&1 | grep "^rounds\|^launched\|identical"; done
python tools/a5000_phases.py 512 5000 2>&1 | grep "steps 1-3\|step 2"
} > $o/probe.txt 2>&1
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_WALK_CHECK python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
{ for n in "1000 64 0" "2000 32 1" "300 500 0"; do SQ_NO_LAUNCHED=1 python tools/rounds_probe.py $n 1 2>&1 | grep -c "WALK MISMATCH"; done; SQ_NO_LAUNCHED=1 python tools/rounds_probe.py 2000 32 1 1 2>&1 | grep "WALK MISMATCH" | head -5; } > $o/walkcheck.txt 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
FUZZ_POOLLIM=1 python tools/fuzz_parity.py 400 fastest > $o/fuzz_chain.txt 2>&1
FUZZ_POOLLIM=1 FUZZ_NMIN=300 FUZZ_NMAX=700 python tools/fuzz_parity.py 60 fastest > $o/fuzz_chain_long.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/gputest.txt
cat $o/probe.txt $o/walkcheck.txt $o/gputest.txt; tail -n 1 $o/fuzz_chain.txt $o/fuzz_chain_long.txt
