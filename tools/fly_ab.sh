cd $GRAFT_REPO_ROOT
for a in "300 10000" "300 1250" "150 20000" "1000 128" "1000 1024"; do
for rep in 1 2; do
for e in nofly fly; do
if [ $e = nofly ]; then export SQ_NO_FLY_BITS=1; unset SQ_FLY_MIN_N; else unset SQ_NO_FLY_BITS; export SQ_FLY_MIN_N=0; fi
echo -n "$e "; SQ_NO_LAUNCHED=1 timeout 300 python tools/rounds_probe.py $a 0 7 2>&1 | grep "^rounds" | sed 's/fold ms.*(min/(min/'
done; done; done
