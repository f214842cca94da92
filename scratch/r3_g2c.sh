#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
timeout -k 10 250 python scratch/ab/ab.py bf16 $L/bf16_new.so $L/g2_prev.so:bf16g2 $L/g2_new.so:bf16g2 2>&1 | grep -v amdgpu.ids
G2_STAMP_KSTEPS="46:0,40:0,40:1,47:0" timeout -k 10 100 python scratch/g2_clock.py "$L/g2_-DBF16_STAMP.so" 2>&1 | grep -v amdgpu.ids | grep -v "^period [0-9]* (" | cut -c1-160
