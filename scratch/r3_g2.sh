#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
timeout -k 10 240 python scratch/g2_check.py 2>&1 | grep -v amdgpu.ids || exit 1
timeout -k 10 200 python scratch/ab/ab.py bf16 $L/bf16_new.so $L/g2_new.so:bf16g2 2>&1 | grep -v amdgpu.ids
G2_NOTE=1 timeout -k 10 100 python scratch/g2_clock.py "$L/g2_-DBF16_STAMP.so" 2>&1 | grep -v amdgpu.ids
