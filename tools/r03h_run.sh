for cap in 153600 65536 32768 16384; do echo "== mwm lds cap $cap"
SQ_MWM_LDS_CAP=$cap python tools/mwm_one.py 217 4 2 2>&1 | tail -1
for k in 1 8; do SQ_MWM_LDS_CAP=$cap python tools/concurrent_probe.py $k 20 2>&1 | tail -1 | cut -c1-120; done
done
