#!/bin/bash
# round 4, GPU call 5: the new training modes (x3 exact records, fp32 pair records), A/B of the body without prefetch touches, bench tiers
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c5; mkdir -p $O; cd $R
L=tools/lib
timeout -k 10 200 python3 tools/g2_ab.py $L/g2_abase.so $L/g2_anopf.so 2>&1 | grep -v amdgpu.ids > $O/ab_nopf.log || exit 1
timeout -k 10 200 python3 tools/g2_ab.py $L/g2_anopf.so $L/g2_abase.so 2>&1 | grep -v amdgpu.ids >> $O/ab_nopf.log || exit 1
timeout -k 10 900 python3 -m pytest tests/test_hip_backward.py tests/test_hip_f16.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || echo "bench failed" >> $O/pytest.log
echo finished >> $O/pytest.log
