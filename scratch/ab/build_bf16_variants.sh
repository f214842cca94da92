#!/bin/bash
# Variant builds of the bf16 MLP kernel for scratch/ab/ab.py (each its own .so; ablation builds give wrong results on purpose)
cd "$(dirname "$0")/../.." || exit 1
C=ddnerf_amd/csrc; O=scratch/ab/lib; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -Wno-unused-const-variable -mllvm -amdgpu-mfma-vgpr-form -shared"
build() { name=$1; shift; hipcc $FL "$@" $C/mlp_bf16.hip $C/api.hip -o $O/bf16_$name.so 2> $O/bf16_$name.err || echo "FAILED $name"; }
for v in "$@"; do
  case $v in
    base) build base & ;;
    nodma) build nodma -DBF16_NO_DMA & ;;
    nobar) build nobar -DBF16_NO_BARRIER & ;;
    norepack) build norepack -DBF16_NO_REPACK & ;;
    pfd8) build pfd8 -DBF16_PFD=8 & ;;
    pfd2) build pfd2 -DBF16_PFD=2 & ;;
    stamp) build stamp -DBF16_STAMP & ;;
    v1) hipcc $FL scratch/ab/v1/mlp_bf16_32x32.hip $C/api.hip -o $O/bf16_v1.so 2> $O/bf16_v1.err || echo "FAILED v1" & ;;
    *) build "$v" $(echo "$v" | tr ',' ' ') & ;;
  esac
done
wait
ls -la $O/*.so
