#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
timeout -k 10 250 python scratch/ab/ab.py bf16 $L/bf16_new.so $L/g2_new.so:bf16g2 $L/g2_skip0.so:bf16g2 $L/g2_skip560.so:bf16g2 2>&1 | grep -v amdgpu.ids
for v in "g2_-DBF16_STAMP" g2_skip0_stamp g2_skip560_stamp; do timeout -k 10 100 python scratch/g2_clock.py "$L/$v.so" 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-200; done
