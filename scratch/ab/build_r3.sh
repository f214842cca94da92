#!/bin/bash
# r03 variant builds for scratch/ab/ab.py: the round-2 kernel (sources exported from HEAD into $OLD), the current one, ablations
cd "$(dirname "$0")/../.." || exit 1
C=ddnerf_amd/csrc; O=scratch/ab/lib; mkdir -p $O; OLD=${OLD:-/tmp/asm/oldsrc/ddnerf_amd/csrc}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -Wno-unused-const-variable -mllvm -amdgpu-mfma-vgpr-form -shared"
b() { out=$1; shift; hipcc $FL "$@" -o $O/$out.so 2> $O/$out.err || echo "FAILED $out"; }
for v in "$@"; do
  case $v in
    bf16_old) b bf16_old $OLD/mlp_bf16.hip $OLD/api.hip & ;;
    x3_old) b x3_old -fno-slp-vectorize $OLD/mlp_x3_fwd.hip $OLD/api.hip & ;;
    bf16_new) b bf16_new $C/mlp_bf16.hip $C/api.hip & ;;
    x3_new) b x3_new -fno-slp-vectorize $C/mlp_x3_fwd.hip $C/api.hip & ;;
    g2_*) name=$v; d=${v#g2_}; [ "$d" = "new" ] && d=""; b $name $(echo "$d" | tr ',' ' ') $C/mlp_bf16_g2.hip $C/api.hip & ;;
    bf16_*) name=$v; d=${v#bf16_}; b $name $(echo "$d" | tr ',' ' ') $C/mlp_bf16.hip $C/api.hip & ;;
    x3_*) name=$v; d=${v#x3_}; b $name -fno-slp-vectorize $(echo "$d" | tr ',' ' ') $C/mlp_x3_fwd.hip $C/api.hip & ;;
  esac
done
wait
ls -la $O/*.so
