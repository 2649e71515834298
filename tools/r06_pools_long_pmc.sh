#!/bin/bash
# pools_long (one sub-batch of 500 records of 500 nt, 500nobpp): counters of the launched round's kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pmc_r06_pl; mkdir -p $o
cat > /tmp/pl_pmc.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
names, psets = ParseConfig(builtin_config("500nobpp"))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), 500)), None, None, None, psets, None) for _ in range(500)]
HipEngine().fold_records_packed(recs, poollim=1000)
PY
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $o/p1 -- python3 /tmp/pl_pmc.py > $o/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/p2 -- python3 /tmp/pl_pmc.py > $o/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/p3 -- python3 /tmp/pl_pmc.py > $o/p3.log 2>&1
for k in sq_pool_round_root_kernel sq_score_kernel sq_scan6_kernel sq_pool_scan_kernel sq_mwm_kernel; do echo "== $k"; python3 tools/pmc_summary.py $o $k | grep total; done
