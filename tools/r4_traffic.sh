#!/bin/bash
# round 4: HBM-side bytes of the two-group bf16 kernel split by ingredient (FETCH_SIZE / WRITE_SIZE of builds without feature fetches, without
# LDS-DMA, with a one-tile feature footprint)
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4traffic; mkdir -p $O
[ -f $R/tools/calib/calib.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $R/tools/calib/calib.hip -o $R/tools/calib/calib.so
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o c -- python3 $R/tools/calib/calib.py > $O/calib.log 2>&1
for v in base4 nox2 nodma2 xsame2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${c}_$v -o c -- python3 $R/tools/g2_run.py $R/tools/lib/g2_$v.so 20 > $O/${c}_$v.log 2>&1
  done
done
cd $R
python3 tools/pmc_table.py "g2_fwd" $O/FETCH_SIZE_base4 $O/WRITE_SIZE_base4 $O/FETCH_SIZE_nox2 $O/WRITE_SIZE_nox2 $O/FETCH_SIZE_nodma2 $O/WRITE_SIZE_nodma2 $O/FETCH_SIZE_xsame2 $O/WRITE_SIZE_xsame2 > $O/table.txt 2>&1
python3 tools/pmc_table.py "calib" $O/calib >> $O/table.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
cat $O/table.txt
