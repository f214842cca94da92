#!/bin/bash
# round 4, GPU call 11: encoder with the one-fma remainder on 16-bit rows: suite, tier errors, bf16 bench + kernel list
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c11; mkdir -p $O; cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout -k 10 600 python3 tools/tier_errors.py 2>&1 | grep -v amdgpu.ids > $O/tier_errors.log
timeout -k 10 300 python3 bench.py --mlp bf16 --no-cpu-baseline --no-clock > $O/bench_bf16.json 2> $O/bench_bf16.err
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o p -- python3 $R/bench.py --mlp bf16 --no-cpu-baseline --no-clock --steps 80 --warmup 5 > $O/prof_bf16.json 2> $O/prof_bf16.err
find $O/prof_bf16 -name "*kernel_trace.csv" -size +20M -delete
echo finished >> $O/pytest.log
