out=gpurun_out/r02_fuzz.txt; : > $out
run() { echo "## $*" >> $out; "$@" 2>&1 | grep -v "amdgpu.ids" | tail -3 >> $out; }
run python tools/fuzz_parity.py 3000 nobpp 21
run python tools/fuzz_parity.py 2000 alt 22
run python tools/fuzz_parity.py 2000 greedynobpp 23
run python tools/fuzz_parity.py 1500 fastest 24
run python tools/fuzz_parity.py 800 edmondsnobpp 25
run python tools/fuzz_parity.py 800 hungariannobpp 26
run python tools/fuzz_parity.py 800 nussinovnobpp 27
FUZZ_NMIN=200 FUZZ_NMAX=520 run python tools/fuzz_parity.py 120 nobpp 28
FUZZ_NMIN=500 FUZZ_NMAX=900 run python tools/fuzz_parity.py 24 500nobpp 29
run python tools/fuzz_align.py 60 31
cat $out
