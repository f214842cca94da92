#!/usr/bin/env python3
"""Experiment builds of the two-group bf16 MLP kernel (CPU only; the .so files travel to the GPU box with the snapshot).

    python tools/g2_variant.py NAME [--stamp] [--experiment key=value ...] [--touch LEAD_BYTES]

generates the tile bodies with the given generator switches into tools/lib/NAME/, compiles mlp_bf16_g2.hip + api.hip against them into
tools/lib/g2_NAME.so (-DBF16_STAMP with --stamp: tile-level clock stamps, plus whatever stamps the switches add) and leaves the
disassembly of mlp_bf16g2_fwd_kernel<false> beside it.  --touch runs the generator twice: the first pass (fillers of the same size)
gives every block's address, the second makes every block load one word from the first 4-KiB code boundary at least LEAD_BYTES
ahead of it (the last blocks: from the body's top, where the next tile continues)."""
import argparse
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddnerf_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "lib")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z21mlp_bf16g2_fwd_kernelILb0EEvPKcS1_Pfll"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-DBF16_DISPATCH"]


def build(name, stamp, exps):
    d = os.path.join(LIB, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    cmd = [sys.executable, os.path.join(CSRC, "gen_bf16_g2.py"), d]
    for e in exps:
        cmd += ["--experiment", e]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    for f in ("mlp_bf16_g2.hip", "mlp_bf16.hip", "mlp_mfma16.inc", "mlp_bf16_common.h", "common.h", "api.hip"):
        shutil.copy(os.path.join(CSRC, f), d)
    so = os.path.join(LIB, "g2_%s.so" % name)
    fl = FLAGS + (["-DBF16_STAMP"] if stamp else []) + ["-I" + CSRC]   # (common.h includes the public header by a path relative to csrc/)
    src = [os.path.join(d, "mlp_bf16_g2.hip"), os.path.join(d, "mlp_bf16.hip"), os.path.join(d, "api.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + fl + ["-shared"] + src + ["-o", so])
    co = os.path.join(d, "g2.co")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + fl + ["--cuda-device-only", "-c", src[0], "-o", co])
    elf = os.path.join(d, "g2.elf")
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + co,
                           "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
    dis = os.path.join(LIB, "g2_%s.dis" % name)
    with open(dis, "w") as f:
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--disassemble-symbols=" + KERNEL, elf], stdout=f)
    shutil.rmtree(d)
    return so, dis


def instructions(dis):
    ins = []
    for ln in open(dis):
        m = re.search(r"^\s+(\S+)(.*?)//\s*([0-9A-F]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2).strip()))
    return ins


def block_addresses(dis):
    """address of the first instruction of every block's 8-byte touch slot (the instruction in front of the block's first MFMA run)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import g2_block_addresses as ba

    ins = instructions(dis)
    g = ba.gen.Gen(0, 0)
    blocks, NK = g.build_blocks()
    mf = [i for i, x in enumerate(ins) if x[1].startswith("v_mfma")]
    k, out = 0, []
    for blk in blocks:
        out.append(ins[mf[k]][0])
        k += 4 * len(blk["order"])
    pc = next(a for a, op, _ in ins if op == "s_getpc_b64") + 4       # s_getpc_b64 returns the address of the instruction behind it
    return out, pc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--stamp", action="store_true")
    ap.add_argument("--experiment", action="append", default=[])
    ap.add_argument("--touch", type=int, default=None, metavar="LEAD_BYTES")
    ap.add_argument("--touch-lines", type=int, default=1, help="the blocks that touch one boundary touch this many successive 64-byte lines behind it (one per block)")
    a = ap.parse_args()
    os.makedirs(LIB, exist_ok=True)
    exps = list(a.experiment)
    if a.touch is not None:
        tf = os.path.join(LIB, "g2_%s.touch" % a.name)
        open(tf, "w").write("nop\n" * 308)
        so, dis = build(a.name, a.stamp, exps + ["touch=" + tf])
        addr, pc = block_addresses(dis)
        offs, seen = [], {}
        for i, x in enumerate(addr):
            p = (x + a.touch + 4095) // 4096 * 4096
            k = seen.get(p, 0)
            seen[p] = k + 1
            line = 64 * (k % a.touch_lines)
            offs.append(p + line - pc if p < addr[-1] else line)
        open(tf, "w").write("\n".join(str(o) for o in offs) + "\n")
        so, dis = build(a.name, a.stamp, exps + ["touch=" + tf])
        addr2, pc2 = block_addresses(dis)
        assert (addr2, pc2) == (addr, pc), "the second pass moved code"
        n = sum(1 for _, op, arg in instructions(dis) if op == "s_load_dword" and "s[98:99]" in arg)
        print("touch: %d loads, pc base 0x%x, first boundaries %s" % (n, pc, [hex(pc + o) for o in offs[30:40]]))
    else:
        so, dis = build(a.name, a.stamp, exps)
    print(so)


if __name__ == "__main__":
    main()
