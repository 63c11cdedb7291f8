#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-contact-leg"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 40 --warmup 10 $B > /dev/null 2> $OUT/pmc.log
echo rc=$?
cd $R
python3 - <<'P'
import csv,glob,collections
f=glob.glob('gpurun_out/lds/pmc/*/*counter_collection.csv')
print(f)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    if 'k_g2p' in k or 'k_p2g' in k or 'k_ct' in k:
        print(k, {c: round(x) for c,x in v.items()})
P
