#!/bin/bash
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r3c3; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
timeout -k 10 200 python bench.py --mode train --mlp x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_x3.json 2> $O/bench_train_x3.err; echo "train x3 rc $?"; cat $O/bench_train_x3.json
timeout -k 10 200 python bench.py --mode train --mlp fp32 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_fp32.json 2> $O/bench_train_fp32.err; echo "train fp32 rc $?"; cat $O/bench_train_fp32.json
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train_x3 -o r -- python3 $R/bench.py --mode train --mlp x3 --steps 5 --warmup 2 --no-cpu-baseline > $O/p_train_x3.log 2>&1; echo "prof rc $?"
head -12 $O/p_train_x3/*/r_kernel_stats.csv 2>/dev/null | cut -c1-200
