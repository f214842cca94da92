#!/bin/bash
# round 4, GPU call 1: where do the two-group bf16 kernel's 4-KiB steps come from?  (experiment builds of tools/g2_variant.py)
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c1; mkdir -p $O; cd $R
L=tools/lib
for v in base periods sgpr sgprb touch600 touch1800 touch600p nodma nox nodmax nodmaxp pad2048p align12 base; do
  timeout -k 10 120 python3 tools/g2_clock.py $L/g2_$v.so 2>&1 | grep -v amdgpu.ids >> $O/clock.log || { echo "FAILED $v" >> $O/clock.log; exit 1; }
done
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_MFMA" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  for v in base nodmax touch600; do
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc${i}_$v -o c -- python3 $R/tools/g2_run.py $R/$L/g2_$v.so 40 > $O/pmc${i}_$v.log 2>&1 || echo "pmc $i $v failed" >> $O/clock.log
  done
done
# thread trace attempt (the image ships no decoder library: what does the tool do?)
timeout -k 10 150 rocprofv3 --att --att-target-cu 1 --kernel-trace -d $O/att -o a -- python3 $R/tools/g2_run.py $R/$L/g2_base.so 3 > $O/att.log 2>&1; echo "att rc $?" >> $O/att.log
ls -laR $O/att 2>/dev/null | head -40 >> $O/att.log
find $O -name "*.att" -size +1M -delete 2>/dev/null
find $O -name "*.csv" -size +8M -delete 2>/dev/null
echo finished >> $O/clock.log
