#!/bin/bash
# round 4, GPU call 6: the coarse launch with the sampler folded in; the whole GPU suite; A/B of nopf on equal rows; render step kernel list
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c6; mkdir -p $O; cd $R
L=tools/lib
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu -k "not 1500" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout -k 10 200 python3 tools/g2_ab.py ddnerf_amd/csrc/libddnerf_hip.so $L/g2_abase.so $L/g2_anopf.so 2>&1 | grep -v amdgpu.ids > $O/ab_nopf.log
timeout -k 10 300 python3 bench.py --mlp bf16 --no-cpu-baseline --no-clock > $O/bench_bf16.json 2> $O/bench_bf16.err
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o p -- python3 $R/bench.py --mlp bf16 --no-cpu-baseline --no-clock --steps 80 --warmup 5 > $O/prof_bf16.json 2> $O/prof_bf16.err
find $O/prof_bf16 -name "*kernel_trace.csv" -size +20M -delete
echo finished >> $O/pytest.log
