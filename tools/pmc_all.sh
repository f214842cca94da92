#!/bin/bash
# PMC evidence (MFMA utilisation, clocks, HBM bytes) for the MLP kernels; run on the GPU box from the repo root
set -e
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/pmc; mkdir -p $O
[ -f $R/tools/calib/calib.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $R/tools/calib/calib.hip -o $R/tools/calib/calib.so
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o c -- python3 $R/tools/calib/calib.py > $O/calib.log 2>&1
for mlp in fp32 bf16 bf16u fp16 x3; do
  # (bf16u: the bf16 step with the encoder OUTSIDE the MLP kernel, DDNERF_FUSE_ENCODER=0 -- exported here, not passed through `env`, which
  # would be an exec between the profiler's preloaded library and the program)
  if [ $mlp = bf16u ]; then export DDNERF_FUSE_ENCODER=0; m=bf16; else unset DDNERF_FUSE_ENCODER; m=$mlp; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$mlp -o c -- python3 $R/bench.py --mlp $m --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-tier --no-clock --ramp 0 > $O/fetch_$mlp.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$mlp -o c -- python3 $R/bench.py --mlp $m --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-tier --no-clock --ramp 0 > $O/write_$mlp.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_$mlp -o c -- python3 $R/bench.py --mlp $m --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-tier --no-clock --ramp 0 > $O/mfma_$mlp.log 2>&1
done
unset DDNERF_FUSE_ENCODER
echo done
# instruction-cache behaviour of the MLP kernels whose body exceeds the 64-KB instruction cache (one demand miss per 4-KiB code page)
for mlp in bf16 x3 fp32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_WAVES --output-format csv -d $O/icache_$mlp -o c -- python3 $R/bench.py --mlp $mlp --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-tier --no-clock --ramp 0 > $O/icache_$mlp.log 2>&1
done
cd $R
python3 tools/pmc_table.py "mlp_" $O/icache_bf16 $O/icache_x3 $O/icache_fp32 > $O/icache_table.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
