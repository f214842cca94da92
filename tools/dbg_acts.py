import sys, numpy as np, torch
sys.path.insert(0, '.')
from ddnerf_amd import ops
from ddnerf_amd import synthetic
depth = True
sd = synthetic.make_state_dict(depth, 11, 4.0)
names = [n for n, _, _ in synthetic.layer_table(depth)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
for M in (1024, 4096 * 8, 4096 * 64):
    g = torch.Generator(device="cuda").manual_seed(1)
    feat = torch.zeros(M, 128, device="cuda")
    feat[:, :123] = torch.rand(M, 123, device="cuda", generator=g) * 2 - 1
    packed = ops.mlp_f32_pack(flat, depth)
    raw_f, acts_f = ops.mlp_f32_forward_train(feat, packed, depth)
    raw_r, acts_r = ops.mlp_f32_forward_train(feat, packed, depth, rec=True)
    ref = ops.x3_split(acts_f).view(torch.int32)[:, :M].reshape(-1, 2560, 16)
    got = acts_r.view(torch.int32)[:, :M].reshape(-1, 2560, 16)
    bad = (ref != got).any(dim=2)          # [blocks, rows]
    print(M, "raw equal", torch.equal(raw_f, raw_r), "bad (block,row) pairs", int(bad.sum()), "rows bad anywhere", bad.any(dim=0).nonzero().flatten()[:40].tolist(), "blocks bad", bad.any(dim=1).nonzero().flatten()[:20].tolist())
    raw_i = ops.mlp_f32_forward(feat, packed, depth)
    print("   raw vs inference kernel", torch.equal(raw_i, raw_f))
