#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
for v in "bf16_-DBF16_NO_DMA,-DBF16_STAMP" "bf16_-DBF16_DEPTH=5,-DBF16_NO_DMA,-DBF16_STAMP" "bf16_-DBF16_DEPTH=4,-DBF16_NO_DMA,-DBF16_STAMP" "bf16_-DBF16_DEPTH=4,-DBF16_STAMP" "bf16_-DBF16_STAMP"; do echo "$v"; timeout -k 10 100 python scratch/bf16_clock.py "$L/$v.so" 2>&1 | grep -v amdgpu.ids; done
