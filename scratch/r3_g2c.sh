#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
timeout -k 10 250 python scratch/ab/ab.py bf16 $L/bf16_new.so $L/g2_new.so:bf16g2 $L/g2_dummy_8_2.so:bf16g2 $L/g2_dummy_15_3.so:bf16g2 2>&1 | grep -v amdgpu.ids
