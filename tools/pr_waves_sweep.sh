# the pool round kernel's LDS (survivors kept in LDS x staging buffer) on the headline step: bash tools/pr_waves_sweep.sh
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
run() { env "$@" python bench.py --steps 20 --warmup 3 --no-stream --no-roofline --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"value\"], d[\"ms_per_step\"])"; }
for st in 128 64; do
SQ_DEFS=-DSQ_PR_STAGE=$st python -c "from squarna_amd.build import build_library; build_library(force=True)"
for ns in 64 48 32 16; do echo "stage $st nsurv $ns"; run SQ_POOL_ROUND_NSURV=$ns; done
done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
