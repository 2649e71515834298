run() { echo "== $*"; env "$@" python3 bench.py --no-stream --no-cpu --no-roofline --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['single_batch']['ms_per_fold'])"; }
run A=1
run SQ_STATE_SHORT_THREADS=256
run SQ_STATE_SHORT_THREADS=128
run A=2
