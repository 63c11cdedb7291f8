#!/bin/bash
# Runs on the GPU box: SQ / cache counters of the contact kernels (scripts/bench_contact.py, config 3).
# usage: scripts/contact_counters.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-ct}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/bench_contact.py --survey-config3 --device-pairs --steps 8"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.log || tail -3 $OUT/sq.log
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/tc -- $CMD > /dev/null 2> $OUT/tc.log || tail -3 $OUT/tc.log
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.log || tail -3 $OUT/sq2.log
cd $R
python3 - <<PY
import csv, glob, collections
for sub in ("sq", "tc", "sq2"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % sub):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_ct_" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc):
            print(sub, k, {n: round(sum(v) / len(v)) for n, v in acc[k].items()}, "launches", len(next(iter(acc[k].values()))))
PY
