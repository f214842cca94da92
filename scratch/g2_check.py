"""bring-up of the two-group bf16 kernel: bit-exact against the one-group kernel on the same feature rows / weights, then timing"""
import sys, os, statistics, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
KO = [(p & ~31) | (16 * ((p >> 2) & 1) + 4 * ((p >> 3) & 3) + (p & 3)) for p in range(128)]
for depth in (False, True):
    sd = synthetic.make_state_dict(depth, 12, 20.0)
    names = [n for n, _, _ in synthetic.layer_table(depth)]
    flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
    p1, p2 = ops.mlp_bf16g1_pack(flat, depth), ops.mlp_bf16g2_pack(flat, depth)
    for M in (512, 1, 37, 513, 4096 * 64 + 77, 524288):
        feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
        fb = feat[:, KO].to(torch.bfloat16).contiguous()
        a = ops.mlp_bf16g1_forward(fb, p1, depth)
        b = ops.mlp_bf16g2_forward(fb, p2, depth)
        torch.cuda.synchronize()
        bad = (a != b).any(dim=1)
        print("depth %d M %7d: max |diff| %.3g, rows differing %d (first %s)" % (depth, M, float((a - b).abs().max()), int(bad.sum()), bad.nonzero()[:5].flatten().tolist()), flush=True)
M = 524288
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
fb = feat[:, KO].to(torch.bfloat16).contiguous()
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
p1, p2 = ops.mlp_bf16g1_pack(flat, False), ops.mlp_bf16g2_pack(flat, False)
runs = {"one group": lambda: ops.mlp_bf16g1_forward(fb, p1, False), "two groups": lambda: ops.mlp_bf16g2_forward(fb, p2, False)}
times = {k: [] for k in runs}
for _ in range(3):
    for f in runs.values(): f()
for rnd in range(12):
    for k, f in runs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 20)
for k, ts in times.items():
    med = statistics.median(ts)
    print("%-12s median %.4f ms  min %.4f  frac %.4f" % (k, med, min(ts), 1220608 * M / med / 1e9 / 2500))
