#!/bin/bash
# run on the GPU box from the repo root: refreshes the numbers under gpurun_out/refresh (copied into profiles/ as rNN_* by
# tools/collect_profiles.py)
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench_fp32.json 2> $O/bench_fp32.err; tail -c 600 $O/bench_fp32.json; echo
timeout -k 10 200 python bench.py --mlp bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
DDNERF_FUSE_ENCODER=0 timeout -k 10 200 python bench.py --mlp bf16 --no-cpu-baseline > $O/bench_bf16_unfused.json 2> $O/bench_bf16_unfused.err
timeout -k 10 200 python bench.py --mlp x3 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err
timeout -k 10 200 python bench.py --mlp fp16 --no-cpu-baseline --no-clock > $O/bench_fp16.json 2> $O/bench_fp16.err
timeout -k 10 200 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_fp32.json 2> $O/bench_train_fp32.err
timeout -k 10 200 python bench.py --mode train --mlp x3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_x3.json 2> $O/bench_train_x3.err
for cfg in config_ff.yml config_360.yml config_blender_mipnerf.yml; do
  timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --steps 40 > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
echo benches done
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_fp32 -o r -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-bf16-tier --no-clock > $O/p_fp32.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bf16 -o r -- python3 $R/bench.py --mlp bf16 --steps 80 --warmup 2 --no-cpu-baseline --no-clock > $O/p_bf16.log 2>&1
export DDNERF_FUSE_ENCODER=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bf16u -o r -- python3 $R/bench.py --mlp bf16 --steps 80 --warmup 2 --no-cpu-baseline --no-clock > $O/p_bf16u.log 2>&1
unset DDNERF_FUSE_ENCODER
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_fp16 -o r -- python3 $R/bench.py --mlp fp16 --steps 80 --warmup 2 --no-cpu-baseline --no-clock > $O/p_fp16.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_x3 -o r -- python3 $R/bench.py --mlp x3 --steps 10 --warmup 2 --no-cpu-baseline > $O/p_x3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train_fp32 -o r -- python3 $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/p_train_fp32.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train_x3 -o r -- python3 $R/bench.py --mode train --mlp x3 --steps 5 --warmup 2 --no-cpu-baseline > $O/p_train_x3.log 2>&1
echo profiles done
find $O -name "*kernel_trace.csv" -size +20M -delete
