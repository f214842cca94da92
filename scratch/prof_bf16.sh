#!/bin/bash
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/pbf16; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 $R/bench.py --mlp bf16 --steps 20 --warmup 3 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/**/r_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    print("%-64s calls %4s avg_us %9.1f"%(r['Name'][:64],r['Calls'],float(r['AverageNs'])/1e3))
PY
grep -o "ms_per_step[^,]*" $O/log.txt
