#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
timeout -k 10 200 python scratch/ab/ab.py bf16 $L/bf16_new.so "$L/bf16_-DBF16_FINE.so" 2>&1 | grep -v amdgpu.ids
for v in "bf16_-DBF16_STAMP" "bf16_-DBF16_FINE,-DBF16_STAMP"; do echo "$v"; timeout -k 10 100 python scratch/bf16_clock.py "$L/$v.so" 2>&1 | grep -v amdgpu.ids; done
