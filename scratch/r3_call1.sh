#!/bin/bash
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r3c1; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
L=scratch/ab/lib
timeout -k 10 200 python scratch/ab/ab.py bf16 $L/bf16_old.so $L/bf16_new.so $L/bf16_-DBF16_NO_DMA.so $L/bf16_-DBF16_HALF_DMA.so > $O/ab_bf16.log 2>&1; cat $O/ab_bf16.log
timeout -k 10 200 python scratch/ab/ab.py x3 $L/x3_old.so $L/x3_new.so > $O/ab_x3.log 2>&1; cat $O/ab_x3.log
timeout -k 10 100 python scratch/bf16_clock.py $L/bf16_-DBF16_STAMP.so > $O/clock.log 2>&1; cat $O/clock.log
