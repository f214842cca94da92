#!/bin/bash
# MFMA busy / clock of variant builds (one rocprofv3 run per lib): usage pmc_ab.sh kind lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:?}; kind=$1; shift; cd /tmp; export TMPDIR=/tmp
for so in "$@"; do
  n=$(basename $so .so); O=$R/gpurun_out/pmcab/$n; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O -o c -- python3 $R/scratch/ab/ab.py $kind $R/$so > $O/log.txt 2>&1
done
echo done
