#!/bin/bash
# round 4, GPU call 8: how far the training curves of the three tiers are from the reference's (bars of the 1500-iteration test)
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c8; mkdir -p $O; cd $R
timeout -k 10 900 python3 tools/train_curve_stats.py 2>&1 | grep -v amdgpu.ids > $O/curves.log
echo finished >> $O/curves.log
