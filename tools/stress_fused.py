#!/usr/bin/env python3
"""Repeated-launch stress of the fused bf16 encoder + MLP kernel (mlp_bf16_g2e.hip).  The body relies on three properties of the memory
pipeline that no wait instruction enforces -- a wave's vector-memory loads AND stores retire in issue order (the books of vmcnt), a wave's
store and its later load of the same address stay in order (no drain between the encoder's stores and the fetches of the rows), and the
fetches bypass the L1 (sc1) -- so a violation would be a rare, timing-dependent wrong row.  N launches at random sizes / sample counts /
ray kinds on fresh fenceposts, every output compared BIT FOR BIT with ddnerf_encode + ddnerf_mlp_bf16_forward; other work (a second
stream hammering HBM) runs beside them in half of the rounds.  GPU box: python3 tools/stress_fused.py [launches] [bf16|fp16]
(fp16: the kernel's twin mlp_f16_g2e.hip against ddnerf_encode(fp16 rows) + ddnerf_mlp_f16_forward)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddnerf_amd import ops, synthetic  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    tier = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    pack, plain = {"bf16": (ops.mlp_bf16_pack, ops.mlp_bf16_forward), "fp16": (ops.mlp_f16_pack, ops.mlp_f16_forward)}[tier]
    rng = np.random.Generator(np.random.PCG64(5))
    packs = {}
    for depth in (False, True):
        sd = synthetic.make_state_dict(depth, 12 + depth, 20.0)
        names = [k for k, _, _ in synthetic.layer_table(depth)]
        flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
        packs[depth] = pack(flat, depth)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    bad = done = 0
    t0 = time.time()
    while done < total:
        kind = ("blender", "llff", "real360")[int(rng.integers(3))]
        S = int(rng.choice([64, 128, 128, 192]))
        n = int(rng.choice([1, 7, 300, 1024, 2048, 4096, 4096, 5000])) if S != 192 else int(rng.integers(1, 600))
        depth = bool(rng.integers(2))
        o, d, rad, _ = synthetic.make_rays(kind, n, int(rng.integers(1 << 30)))
        near, far = synthetic.NEAR_FAR[kind]
        rays = ops.pack_rays(*(torch.from_numpy(x).cuda() for x in (o, d, rad)), near, far)
        tab = ops.ray_table(rays, tier)
        noisy = done % 2 == 1
        for rep in range(8):
            t = (near + (far - near) * torch.sort(torch.rand(n, S + 1, device="cuda") ** 2, dim=1).values).contiguous()
            if noisy:
                with torch.cuda.stream(side):
                    junk.mul_(1.0001)
            got = ops.encode_mlp_bf16_forward(tab, t, packs[depth], depth, kind=tier)
            want = plain(ops.encode(rays, t, kind=tier), packs[depth], depth)
            same = bool(((got == want) | (torch.isnan(got) & torch.isnan(want))).all())
            bad += int(not same)
            done += 1
            if not same:
                print("MISMATCH launch %d: kind %s n %d S %d depth %s rows %d" % (done, kind, n, S, depth, int((got != want).any(dim=1).sum())), flush=True)
        if done % 400 < 8:
            print("%d launches, %d mismatches, %.0f s" % (done, bad, time.time() - t0), flush=True)
    torch.cuda.synchronize()
    print("stress_fused (%s): %d launches of the fused kernel against the two-launch path, %d mismatches" % (tier, done, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
