#!/bin/bash
# run on the GPU box from the repo root: the fp32 training step with each record format of the fp32 tier, interleaved in one call
# (DDNERF_WGRAD=x3words: hi/lo words split by the recording kernels; x3: blocked fp32 values split by the weight-gradient kernel)
# usage: tools/train_ab.sh OUTDIR [modes...]
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/$1; shift; mkdir -p $O
MODES=${@:-x3words x3}
cd $R
for i in 1 2 3; do
  for m in $MODES; do
    DDNERF_WGRAD=$m timeout -k 10 200 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/train_${m}_$i.json 2> $O/train_${m}_$i.err || exit 1
    python - $O/train_${m}_$i.json $m $i <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("train", sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], flush=True)
PY
  done
done
