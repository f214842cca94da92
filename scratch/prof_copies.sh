#!/bin/bash
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/pcopy; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o r -- python3 $R/bench.py --mlp bf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/**/r_memory_copy_trace.csv",recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
print(rows[0].keys())
c=collections.Counter((r.get('Direction'),r.get('Size',r.get('Bytes'))) for r in rows[-60:])
for k,v in c.most_common(): print(k,v)
PY
