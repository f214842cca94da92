#!/bin/bash
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r3c2; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cat $O/bench.json
timeout -k 10 200 python bench.py --config config_ff.yml --steps 10 > $O/bench_ff.json 2> $O/bench_ff.err; echo "bench ff rc $?"; cat $O/bench_ff.json
