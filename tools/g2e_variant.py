#!/usr/bin/env python3
"""Experiment builds of the FUSED two-group bf16 kernel (mlp_bf16_g2e.hip: the encoder inside the MLP kernel); CPU only, the .so files
travel to the GPU box with the snapshot.

    python tools/g2e_variant.py NAME [--experiment key=value ...]

generates the tile bodies with the given generator switches (gen_bf16_g2.py: enc_drain, enc_store, enc_valu, enc_prologue, ...) into
tools/lib/NAME/ and compiles mlp_bf16_g2e.hip against them with -DBF16_STAMP (tile-loop clock stamps) into tools/lib/g2e_NAME.so.  The
library holds the fused kernel only: tools/g2e_ab.py loads the product library first (RTLD_GLOBAL) for everything else."""
import argparse
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddnerf_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "lib")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function", "-ffp-contract=off", "-DBF16_STAMP"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--experiment", action="append", default=[])
    a = ap.parse_args()
    d = os.path.join(LIB, a.name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    cmd = [sys.executable, os.path.join(CSRC, "gen_bf16_g2.py"), d]
    for e in a.experiment:
        cmd += ["--experiment", e]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    for f in ("mlp_bf16_g2e.hip", "mlp_bf16_common.h", "common.h"):
        shutil.copy(os.path.join(CSRC, f), d)
    so = os.path.join(LIB, "g2e_%s.so" % a.name)
    # (the fused entry point finds the two-group image inside the common weight image behind the one-group kernel's: that size is a
    # host-side constant of the product library, baked into a stub here -- the variant must NOT see the product library's symbols at
    # load time: a kernel template's handle is a weak default-visibility symbol, and the product's would be launched instead)
    sys.path.insert(0, ROOT)
    from ddnerf_amd import _lib
    L = _lib.lib()
    stub = os.path.join(d, "stub.cpp")
    open(stub, "w").write('#include <cstddef>\nextern "C" size_t ddnerf_mlp_bf16g1_packed_bytes(int d) { return d ? %dul : %dul; }\n'
                          % (L.ddnerf_mlp_bf16g1_packed_bytes(1), L.ddnerf_mlp_bf16g1_packed_bytes(0)))
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + CSRC, "-shared", os.path.join(d, "mlp_bf16_g2e.hip"), stub, "-o", so])
    shutil.rmtree(d)
    print(so)


if __name__ == "__main__":
    main()
