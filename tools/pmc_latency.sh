#!/bin/bash
# Where the pool round kernel's waves wait, on the headline step: average latency of its LDS / vector-memory / scalar-memory
# instructions (SQ_INST_LEVEL_x accumulates the instructions in flight per cycle; / SQ_INSTS_x = cycles per instruction) and the
# busy / wait cycle counters.  bash tools/pmc_latency.sh  (on the GPU box) -> gpurun_out/pmc_latency/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pmc_latency; mkdir -p $o
B="python3 bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline"
rocprofv3 --list-avail > $o/avail.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $o/p1 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --output-format csv -d $o/p2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $o/p3 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVE_DEP_WAIT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $o/p4 -- $B > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3", "p4"):
    tot = collections.Counter()
    for f in glob.glob("gpurun_out/pmc_latency/%s/*/*counter_collection.csv" % p):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("sq_pool_round_kernel"): tot[r["Counter_Name"]] += float(r["Counter_Value"])
    print(p, " ".join("%s=%.4g" % kv for kv in sorted(tot.items())))
PY
rm -rf $o/p1 $o/p2 $o/p3 $o/p4
