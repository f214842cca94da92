#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}
for v in 65536 300000 -1 65536 300000 -1; do
  DDNERF_BF16_G2_MIN=$v timeout -k 10 200 python bench.py --mlp bf16 --no-cpu-baseline --no-clock 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('G2_MIN=$v', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
done
