#!/bin/bash
# the fp32 kernel's passes of tools/pmc_all.sh alone (after a change to the fp32 forward: its traffic file records the source digest)
set -e
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/pmc; mkdir -p $O
[ -f $R/tools/calib/calib.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $R/tools/calib/calib.hip -o $R/tools/calib/calib.so
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o c -- python3 $R/tools/calib/calib.py > $O/calib.log 2>&1
mlp=fp32
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$mlp -o c -- python3 $R/bench.py --mlp $mlp --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-tier --no-clock --ramp 0 > $O/fetch_$mlp.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$mlp -o c -- python3 $R/bench.py --mlp $mlp --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-tier --no-clock --ramp 0 > $O/write_$mlp.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_$mlp -o c -- python3 $R/bench.py --mlp $mlp --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-tier --no-clock --ramp 0 > $O/mfma_$mlp.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_WAVES --output-format csv -d $O/icache_$mlp -o c -- python3 $R/bench.py --mlp $mlp --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-tier --no-clock --ramp 0 > $O/icache_$mlp.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
echo done
