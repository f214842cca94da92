#!/bin/bash
# round 4, GPU call 2: per-period stamps without DMA / without feature fetches; the GPU test suite; the default bench line; image-level numbers
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c2; mkdir -p $O; cd $R
L=tools/lib
for v in nodmap noxp; do
  timeout -k 10 120 python3 tools/g2_clock.py $L/g2_$v.so 2>&1 | grep -v amdgpu.ids >> $O/clock.log || { echo "FAILED $v" >> $O/clock.log; exit 1; }
done
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || echo "bench failed" >> $O/pytest.log
timeout -k 10 200 python3 bench.py --image 800x800 --rays 16384 --no-cpu-baseline --no-bf16-tier > $O/bench_image_fp32.json 2> $O/bench_image_fp32.err
timeout -k 10 200 python3 bench.py --image 800x800 --rays 16384 --no-cpu-baseline --mlp bf16 --no-clock > $O/bench_image_bf16.json 2> $O/bench_image_bf16.err
echo finished >> $O/clock.log
