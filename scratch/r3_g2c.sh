#!/bin/bash
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib
for v in g2_skip0p_stamp; do echo $v; timeout -k 10 100 python scratch/g2_clock.py "$L/$v.so" 2>&1 | grep -v amdgpu.ids | grep "^period\|^g2" | head -14; done
