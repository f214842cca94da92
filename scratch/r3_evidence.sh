#!/bin/bash
# round-3 evidence behind DESIGN.md 2.1 / 3, one call on one box -> gpurun_out/evidence/*.log (copied into profiles/r03_*.log)
cd ${GRAFT_REPO_ROOT:?}; L=scratch/ab/lib; O=gpurun_out/evidence; mkdir -p $O
(cd scratch/power && timeout -k 10 100 ./mfma_power 0 && timeout -k 10 100 ./mfma_power 1) > $O/mfma_power.log 2>&1
timeout -k 10 250 python scratch/ab/ab.py bf16 $L/bf16_new.so $L/g2_new.so:bf16g2 "$L/bf16_-DBF16_FINE.so" 2>&1 | grep -v amdgpu.ids > $O/bf16_ab.log
for v in "bf16_-DBF16_STAMP" "bf16_-DBF16_FINE,-DBF16_STAMP"; do echo "$v"; timeout -k 10 100 python scratch/bf16_clock.py "$L/$v.so" 2>&1 | grep -v amdgpu.ids; done >> $O/bf16_ab.log
timeout -k 10 100 python scratch/g2_clock.py "$L/g2_-DBF16_STAMP.so" 2>&1 | grep -v amdgpu.ids > $O/g2_clock.log
timeout -k 10 100 python scratch/enc_time.py $L/enc_old.so $L/enc_new2.so 2>&1 | grep -v amdgpu.ids > $O/encode_ab.log
tail -n 3 $O/*.log
