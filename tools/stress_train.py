#!/usr/bin/env python3
"""Stress of the fp32 tier's values-record training kernels (round 5): the sign record travels through the SCALAR data cache (s_store in the
forward, s_load in the backward), which no other path of this library uses -- so: many forward + backward pairs that REUSE the same record
buffers (what the caching allocator gives a training loop) with fresh inputs every time, sizes from one ragged tile to the fine pass, both
heads, optionally a second stream hammering HBM beside them; every delta record is compared bit for bit with the hi/lo-word path's (masks from
the recorded activations: vector loads), every sign record with the activations it was taken from.

    python tools/stress_train.py [pairs] [--hammer]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ddnerf_amd import ops, synthetic  # noqa: E402


def run(n, hammer=False, sizes=(1, 127, 128, 129, 200, 4096, 33333, 65536, 262144, 524288), max_random=70000, log=print):
    g = torch.Generator(device="cuda").manual_seed(7)
    rs = np.random.RandomState(3)
    bufs = {}

    def alloc(shape, dtype, device):   # the same storage for every record of a given shape: addresses repeat, contents do not
        key = (tuple(shape), dtype)
        if key not in bufs:
            if len(bufs) >= 24:   # (the random sizes: their records come and go through the caching allocator, like a training loop's)
                return torch.empty(shape, dtype=dtype, device=device)
            bufs[key] = [torch.empty(shape, dtype=dtype, device=device) for _ in range(4)]   # (a pair allocates four records of one shape)
        lst = bufs[key]
        lst.append(lst.pop(0))
        return lst[-1]

    ops.RECORD_ALLOC = alloc
    side = torch.cuda.Stream()
    junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda") if hammer else None
    packs = {}
    bad = 0
    prev_alloc = ops.RECORD_ALLOC
    for it in range(n):
        depth = bool(it & 1)
        M = sizes[rs.randint(len(sizes))] if it % 3 else int(rs.randint(1, max_random))
        if it % 8 == 0 or depth not in packs:   # fresh weights now and then
            sd = synthetic.make_state_dict(depth, 100 + it, float(rs.choice([1.0, 3.0, 20.0])))
            names = [nm for nm, _, _ in synthetic.layer_table(depth)]
            flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
            packs[depth] = (ops.mlp_f32_pack(flat, depth), ops.mlp_f32_pack_t(flat, depth))
        pk, pt = packs[depth]
        feat = torch.zeros(M, 128, device="cuda")
        feat[:, :123] = torch.rand(M, 123, device="cuda", generator=g) * 2 - 1
        G = torch.randn(M, 6 if depth else 4, device="cuda", generator=g)
        if hammer:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        raw_v, a_v, signs = ops.mlp_f32_forward_train(feat, pk, depth, rec="values")
        d_v = ops.mlp_f32_backward_data(G, pt, a_v, depth, rec="values", signs=signs)
        raw_w, a_w = ops.mlp_f32_forward_train(feat, pk, depth, rec="hilo")
        d_w = ops.mlp_f32_backward_data(G, pt, a_w, depth, rec="hilo")
        ld = a_v.shape[1]
        ok = torch.equal(raw_v, raw_w)
        # deltas: the word record is the exact split of the value record (rows 0 .. 2437: everything the backward writes)
        split = ops.x3_split(ops.x3_unblock(d_v)[:, :ld].contiguous())
        ok = ok and torch.equal(split.view(torch.int32).view(-1, 2560, 16)[:, :2438], d_w.view(torch.int32).view(-1, 2560, 16)[:, :2438])
        # signs against the recorded activations (every tile the record covers)
        A = ops.x3_unblock(a_v)[:2048]
        pos = (A.view(64, 32, ld // 128, 4, 32) > 0).permute(2, 0, 3, 1, 4)
        r = torch.arange(16, device="cuda")
        sh = torch.arange(32, device="cuda", dtype=torch.int64)
        want = torch.zeros(ld // 128, 64, 4, 16, dtype=torch.int64, device="cuda")
        for h in range(2):
            want |= (pos[:, :, :, (r & 3) + 8 * (r >> 2) + 4 * h, :].to(torch.int64) << (sh + 32 * h)).sum(-1)
        ok = ok and torch.equal(signs.view(torch.int64).view(ld // 128, 64, 4, 16), want)
        if not ok:
            bad += 1
            log("MISMATCH at pair %d (M = %d, depth head %s)" % (it, M, depth))
        if it % 50 == 49:
            log("%d pairs, %d mismatches" % (it + 1, bad))
        del feat, G, raw_v, a_v, signs, d_v, raw_w, a_w, d_w, split, A, pos, want
    torch.cuda.synchronize()
    ops.RECORD_ALLOC = prev_alloc
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
    hammer = "--hammer" in sys.argv
    bad = run(n, hammer, log=lambda m: print(m, flush=True))
    print("stress_train: %d forward + backward pairs%s, %d mismatches" % (n, " beside an HBM-hammering stream" if hammer else "", bad))
    sys.exit(1 if bad else 0)
