run() { echo "== K=$1 R=$2"; PROBE_REPLICAS=$2 PROBE_MAX_STRUCTS=$((4096*$2)) python tools/concurrent_probe.py $1 12 2>&1 | grep -E "^K=|CPU" | cut -c1-100; }
run 8 3
run 12 2
run 12 3
run 16 2
run 6 4
