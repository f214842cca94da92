#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
timeout -k 10 250 python scratch/ab/ab.py bf16 $L/bf16_new.so $L/g2_new.so:bf16g2 $L/g2_span_1_1.so:bf16g2 $L/g2_span_3_4.so:bf16g2 $L/g2_span_1_3.so:bf16g2 2>&1 | grep -v amdgpu.ids
