{ echo "== device pools (poollim > 1, sq_pool.hip): randomised parity against the CPU oracle";
  python tools/fuzz_parity.py 3000 nobpp 41 2>&1 | grep -E "records|MISMATCH|oracle";
  FUZZ_POOLLIM=3 python tools/fuzz_parity.py 2000 greedynobpp 42 2>&1 | grep -E "records|MISMATCH|oracle";
  FUZZ_POOLLIM=25 python tools/fuzz_parity.py 1500 alt 43 2>&1 | grep -E "records|MISMATCH|oracle";
  FUZZ_NMIN=200 FUZZ_NMAX=520 python tools/fuzz_parity.py 100 nobpp 44 2>&1 | grep -E "records|MISMATCH|oracle";
  FUZZ_NMIN=500 FUZZ_NMAX=800 python tools/fuzz_parity.py 16 500nobpp 45 2>&1 | grep -E "records|MISMATCH|oracle"; } > gpurun_out/fuzz_pools.txt 2>&1
cat gpurun_out/fuzz_pools.txt
