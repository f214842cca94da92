"""where do the two-group kernel's outputs differ from the one-group kernel's?"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
KO = [(p & ~31) | (16 * ((p >> 2) & 1) + 4 * ((p >> 3) & 3) + (p & 3)) for p in range(128)]
depth = False
sd = synthetic.make_state_dict(depth, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(depth)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
p1, p2 = ops.mlp_bf16g1_pack(flat, depth), ops.mlp_bf16g2_pack(flat, depth)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
fb = feat[:, KO].to(torch.bfloat16).contiguous()
a = ops.mlp_bf16g1_forward(fb, p1, depth)
from collections import Counter
for rep in range(4):
    b = ops.mlp_bf16g2_forward(fb, p2, depth)
    torch.cuda.synchronize()
    bad = (a != b).any(dim=1).nonzero().flatten().tolist()
    cnt = Counter()
    for r in bad:
        t, o = divmod(r, 512)
        w, o2 = divmod(o, 128)
        g, o3 = divmod(o2, 64)
        c, j = divmod(o3, 16)
        cnt[("round", t // 256, "g", g)] += 1
    print("rep %d: %d rows differ; %s" % (rep, len(bad), dict(cnt)))
    for r in bad[:12]:
        t, o = divmod(r, 512); w, o2 = divmod(o, 128); g, o3 = divmod(o2, 64); c, j = divmod(o3, 16)
        print("  row %7d tile %4d wave %d g %d c %d j %2d  diff %s" % (r, t, w, g, c, j, (a[r] - b[r]).tolist()))
