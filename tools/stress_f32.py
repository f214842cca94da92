#!/usr/bin/env python3
"""Ad-hoc stress of the fp32 forward (LDS-DMA weight stream, LDS hand-overs): N launches at several sizes, every output compared with the first."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
depth = True
sd = synthetic.make_state_dict(depth, 5, 4.0)
names = [n for n, _, _ in synthetic.layer_table(depth)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
packed = ops.mlp_f32_pack(flat, depth)
g = torch.Generator(device="cuda").manual_seed(3)
for M, n in ((4096 * 128, 300), (128 * 515 + 77, 300), (4096 * 64, 300), (1000, 300)):
    feat = torch.zeros(M, 128, device="cuda")
    feat[:, :123] = torch.rand(M, 123, device="cuda", generator=g) * 2 - 1
    raw0 = ops.mlp_f32_forward(feat, packed, depth).clone()
    bad = 0
    for it in range(n):
        if not torch.equal(ops.mlp_f32_forward(feat, packed, depth), raw0):
            bad += 1
    print("M = %d: %d launches, %d differ from the first" % (M, n, bad), flush=True)
