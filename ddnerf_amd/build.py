"""Builds libddnerf_hip.so (gfx950 only) in-tree with hipcc.  `python -m ddnerf_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repository snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO = os.path.join(CSRC, "libddnerf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# kernels whose arithmetic must round exactly like the reference's scalar op chain: no FMA contraction
EXACT = ["-ffp-contract=off"]
SOURCES = {
    "api.hip": [],
    "rays_encode.hip": EXACT,
    "composite.hip": EXACT,
    "samplers.hip": EXACT,
    "dp_loss.hip": EXACT,
    "raygen.hip": EXACT,
    "mlp_f32.hip": [],
    "mlp_f32_train.hip": [],
    "mlp_f32_wgrad.hip": [],
    "mlp_x3_wgrad.hip": [],
    # accumulators in arch VGPRs (the VALU re-pack reads them), B files in the accumulator half: see mlp_bf16.hip
    "mlp_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    "mlp_x3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    "mlp_x3_train.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
}


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mlp_f32_common.h"), os.path.join(CSRC, "wgrad_reduce.h"), os.path.join(CSRC, "mlp_bf16_common.h"), os.path.join(CSRC, "mlp_x3_common.h"), os.path.join(CSRC, "..", "..", "include", "ddnerf_hip.h"), __file__]
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(SO, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
