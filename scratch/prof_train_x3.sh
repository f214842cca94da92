#!/bin/bash
# kernel-trace of the x3 training step (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/ptrain; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 $R/bench.py --mode train $PROF_MLP --steps 5 --warmup 2 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/**/r_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-64s calls %4s avg_us %9.1f pct %5s"%(r['Name'][:64],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
PY
