#!/bin/bash
# run on the GPU box from the repo root: refreshes the numbers under gpurun_out/refresh (copied into profiles/ by hand)
set -e
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench_fp32.json 2> $O/bench_fp32.err; cat $O/bench_fp32.json
timeout -k 10 300 python bench.py --mlp bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; cat $O/bench_bf16.json
timeout -k 10 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; cat $O/bench_train.json
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_fp32 -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-bf16-tier > $O/p_fp32.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bf16 -o r -- python3 $R/bench.py --mlp bf16 --steps 10 --warmup 2 --no-cpu-baseline > $O/p_bf16.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train -o r -- python3 $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/p_train.log 2>&1
echo done1
# x3 (split-precision bf16 MFMA) tiers
cd $R
timeout -k 10 300 python bench.py --mlp x3 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; cat $O/bench_x3.json
timeout -k 10 300 python bench.py --mode train --mlp x3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_x3.json 2> $O/bench_train_x3.err; cat $O/bench_train_x3.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_x3 -o r -- python3 $R/bench.py --mlp x3 --steps 10 --warmup 2 --no-cpu-baseline > $O/p_x3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train_x3 -o r -- python3 $R/bench.py --mode train --mlp x3 --steps 5 --warmup 2 --no-cpu-baseline > $O/p_train_x3.log 2>&1
echo done2
