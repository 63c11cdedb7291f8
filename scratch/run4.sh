cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d_prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4d_prof -- python3 $R/scratch/chain_prof.py > $R/gpurun_out/r4d_chain.log 2>&1
cd $R
grep "us/substep" gpurun_out/r4d_chain.log
python3 scripts/kstats.py $(find gpurun_out/r4d_prof -name "*kernel_stats.csv" | head -1) 25
# gaps: sum over the kernel trace of (start - previous end) inside the timed region
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r4d_prof/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 40 substeps: find k_g2p launches
g2p=[i for i,r in enumerate(rows) if "k_g2p" in r["Kernel_Name"]]
i0,i1=g2p[-41],g2p[-1]
seg=rows[i0+1:i1+1]
tot=(int(seg[-1]["End_Timestamp"])-int(rows[i0]["End_Timestamp"]))/40e3
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in seg)/40e3
print("per substep: wall %.1f us, kernels busy %.1f us, gaps %.1f us, launches %.1f"%(tot,busy,tot-busy,len(seg)/40))
import collections
names=collections.OrderedDict()
for r in seg[:int(len(seg)/40)]:
    print("   ", r["Kernel_Name"].split("(")[0][:60], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
