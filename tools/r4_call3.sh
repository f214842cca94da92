#!/bin/bash
# round 4, GPU call 3: what about the feature fetches makes a code-page crossing expensive?  cache-policy bits, the L2 prefetch touches,
# touching the first lines of the next code page; then the GPU test suite and the default bench line
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c3; mkdir -p $O; cd $R
L=tools/lib
for v in base xnone xsc1 xsc01 xntsc1 xntsc01 nopf touch7 base touch7p nopfp; do
  timeout -k 10 120 python3 tools/g2_clock.py $L/g2_$v.so 2>&1 | grep -v amdgpu.ids >> $O/clock.log || { echo "FAILED $v" >> $O/clock.log; exit 1; }
done
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || echo "bench failed" >> $O/pytest.log
echo finished >> $O/clock.log
