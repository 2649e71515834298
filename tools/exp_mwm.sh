#!/bin/bash
bld() { SQ_DEFS="$1" python -c "from squarna_amd import build; build.build_library(force=True)" > /dev/null 2>&1; }
bld "-DSQ_MWM_PROF2"; echo "== PROF2"; timeout 300 python tools/concurrent_probe.py 1 1 2>&1 | grep "mwm2 n=148" | sort | uniq -c | head -2
bld "-DSQ_MWM_PROF"; echo "== PROF"; timeout 300 python tools/concurrent_probe.py 1 1 2>&1 | grep "mwm n=148" | sort | uniq -c | head -2
