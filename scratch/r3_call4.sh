#!/bin/bash
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r3c4; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -8 $O/pytest.log
timeout -k 10 200 python bench.py --mlp bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bf16 rc $?"; cat $O/bench_bf16.json
DDNERF_FUSE_RENDER=0 timeout -k 10 200 python bench.py --mlp bf16 --no-cpu-baseline > $O/bench_bf16_nofuse.json 2> $O/bench_bf16_nofuse.err; echo "bf16 nofuse rc $?"; cat $O/bench_bf16_nofuse.json
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bf16 -o r -- python3 $R/bench.py --mlp bf16 --steps 10 --warmup 2 --no-cpu-baseline > $O/p_bf16.log 2>&1; echo "prof rc $?"
cut -d, -f1-4 $O/p_bf16/r_kernel_stats.csv | cut -c1-120 | head -24
