#!/usr/bin/env python3
"""Byte addresses of the two-group bf16 kernel's blocks in a built code object (experiment tool, CPU only).

usage: g2_block_addresses.py <disassembly of mlp_bf16g2_fwd_kernel<false>> [first period] [last period]
The disassembly comes from `llvm-objdump -d --no-show-raw-insn --disassemble-symbols=<kernel> <code object>`.  Blocks are found by
counting MFMAs (the generator knows how many every block issues); prints, per block, its first address, its size and whether a 4-KiB
address boundary falls inside it."""
import importlib.util
import os
import re
import sys

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("gen", os.path.join(here, "..", "ddnerf_amd", "csrc", "gen_bf16_g2.py"))
argv, sys.argv = sys.argv, ["gen"]
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
sys.argv = argv


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    dis = open(sys.argv[1]).read().splitlines()
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    ins = []
    for ln in dis:
        m = re.search(r"^\s+(\S+).*//\s*([0-9A-F]+):", ln)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
    g = gen.Gen(0, 0)
    blocks, NK = g.build_blocks()
    mf = [i for i, (a, op) in enumerate(ins) if op.startswith("v_mfma")]
    assert len(mf) == 4 * NK, (len(mf), NK)
    k = 0
    out = []
    for bi, blk in enumerate(blocks):
        n = 4 * len(blk["order"])
        first, last = ins[mf[k]][0], ins[mf[k + n - 1]][0]
        nxt = ins[mf[k + n]][0] if k + n < len(mf) else last + 8
        out.append((bi, blk, first, nxt))
        k += n
    for bi, blk, first, nxt in out:
        if lo <= blk["period"] < hi:
            cross = (first // 4096) != ((nxt - 1) // 4096)
            inper = [x for x in blocks if x["period"] == blk["period"]].index(blk)
            print("period %2d block %d (abs %3d, L%d b%2d g%d): 0x%06x .. 0x%06x  %4d bytes  %s" % (
                blk["period"], inper, bi, blk["l"], blk["b"], blk["g"], first, nxt, nxt - first, "<-- crosses 4 KiB at 0x%x" % (nxt // 4096 * 4096) if cross else ""))


if __name__ == "__main__":
    main()
