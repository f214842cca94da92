#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
timeout -k 10 300 python scratch/ab/ab.py bf16 $L/bf16_new.so $L/g2_new.so:bf16g2 $L/g2_sk_6_560.so:bf16g2 $L/g2_sk_6_128.so:bf16g2 $L/g2_sk_3_560.so:bf16g2 $L/g2_sk_5_1024.so:bf16g2 2>&1 | grep -v amdgpu.ids
