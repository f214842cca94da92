#!/bin/bash
# round 4, GPU call 4: feature-fetch footprint / prefetch policy / glc code touches; A/B of the body without the prefetch touches; the fp16
# tier's errors and tests; the bench line with the re-formed staged ceiling
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c4; mkdir -p $O; cd $R
L=tools/lib
for v in xsame xsamep pfntp touch7glcp; do
  timeout -k 10 120 python3 tools/g2_clock.py $L/g2_$v.so 2>&1 | grep -v amdgpu.ids >> $O/clock.log || { echo "FAILED $v" >> $O/clock.log; exit 1; }
done
timeout -k 10 200 python3 tools/g2_ab.py $L/g2_abase.so $L/g2_anopf.so 2>&1 | grep -v amdgpu.ids > $O/ab_nopf.log || exit 1
timeout -k 10 600 python3 tools/tier_errors.py 2>&1 | grep -v amdgpu.ids > $O/tier_errors.log || echo "tier_errors failed" >> $O/clock.log
timeout -k 10 600 python3 -m pytest tests/test_hip_f16.py tests/test_hip_bf16_g2.py tests/test_hip_stages.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout -k 10 300 python3 bench.py --mlp bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout -k 10 300 python3 bench.py --mlp fp16 --no-cpu-baseline --no-clock > $O/bench_fp16.json 2> $O/bench_fp16.err
echo finished >> $O/clock.log
