import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from ddnerf_amd import ops, synthetic
depth = True
sd = synthetic.make_state_dict(depth, 9, 3.0)
names = [n for n, _, _ in synthetic.layer_table(depth)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
M = 200
torch.manual_seed(0)
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
G = torch.randn(M, 6, device="cuda")
raw_f, acts_f = ops.mlp_f32_forward_train(feat, ops.mlp_f32_pack(flat, depth), depth)
raw_x, acts_x, bits = ops.mlp_x3_forward_train(feat, ops.mlp_x3_train_pack(flat, depth), depth)
print("raw err", float((raw_f - raw_x).abs().max()))
for r0, r1, nm in [(0, 256, "l0"), (256, 512, "l1"), (1280, 1536, "l5"), (1792, 2048, "l7"), (2048, 2304, "feat"), (2304, 2432, "dir"), (2432, 2560, "x")]:
    a, b = acts_f[r0:r1, :M], acts_x[r0:r1, :M]
    print("acts", nm, "max err %.3g  scale %.3g" % (float((a - b).abs().max()), float(a.abs().max())))
# bits check: bit = value > 0
bits_u = bits.to(torch.int32) & 0xffff
ok = True
for tile in (0, 7, 8, 40, 63, 72, 75):
    for h in (0, 1):
        w = bits_u[tile * 2 + h, :M]
        for r in range(16):
            row = 32 * tile + (r & 3) + 8 * (r >> 2) + 4 * h
            exp = (acts_f[row, :M] > 0).to(torch.int32)
            got = (w >> r) & 1
            bad = int((exp != got).sum())
            # values that are ~0 may differ; count only clear ones
            clear = (acts_f[row, :M].abs() > 1e-5)
            bad = int(((exp != got) & clear).sum())
            if bad: ok = False; print("bits mismatch tile", tile, "h", h, "r", r, "bad", bad)
print("bits ok", ok)
d_f = ops.mlp_f32_backward_data(G, ops.mlp_f32_pack_t(flat, depth), acts_f, depth)
d_x = ops.mlp_x3_backward_data(G, ops.mlp_x3_pack_t(flat, depth), bits, depth)
for r0, r1, nm in [(2432, 2464, "draw"), (2304, 2432, "d dir"), (2048, 2304, "d feat"), (1792, 2048, "d h7"), (1536, 1792, "d h6"), (1280, 1536, "d h5"), (256, 512, "d h1"), (0, 256, "d h0")]:
    a, b = d_f[r0:r1, :M], d_x[r0:r1, :M]
    print("deltas", nm, "max err %.3g  scale %.3g" % (float((a - b).abs().max()), float(a.abs().max())))
print("pad cols of deltas all zero:", float(d_x[:, M:].abs().max()))
for r0, nm in [(1024, "d h4"), (768, "d h3"), (512, "d h2"), (256, "d h1")]:
    a, b = d_f[r0:r0 + 256, :M], d_x[r0:r0 + 256, :M]
    bad = ((a - b).abs() > 1e-5 * a.abs().max()).nonzero()
    print(nm, "elements off:", bad.shape[0], "of", a.numel())
    for rr, cc in bad[:5].tolist():
        print("   row", r0 + rr, "col", cc, "d_f %.3e d_x %.3e act_f %.3e act_x %.3e" % (float(a[rr, cc]), float(b[rr, cc]), float(acts_f[r0 + rr, cc]), float(acts_x[r0 + rr, cc])))
