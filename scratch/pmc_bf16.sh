#!/bin/bash
# SQ / GRBM counters of the bf16 MLP kernel alone (scratch/bench_mlp.py), one rocprofv3 pass per counter set.
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/${1:-pmc_bf16}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/s$i -o c -- python3 $R/scratch/bench_mlp.py 524288 bf16 > $O/s$i.log 2>&1 || echo "set $i failed"
done
python3 $R/scratch/pmc_summary.py $O
