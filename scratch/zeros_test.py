"""Is the x3 / bf16 kernel's run time data dependent (the signature of a power limit)?  Same launch, random vs all-zero operands."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
M = 524288
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, fl, ft in (("random", flat, feat), ("zero weights", torch.zeros_like(flat), feat), ("zero everything", torch.zeros_like(flat), torch.zeros_like(feat)), ("random again", flat, feat)):
    px = ops.mlp_x3_pack(fl, False); pb = ops.mlp_bf16_pack(fl, False); pf = ops.mlp_f32_pack(fl, False)
    fb = ft[:, ops.K_ORDER].to(torch.bfloat16).contiguous()
    print("%-16s x3 %.4f ms   bf16 %.4f ms   fp32 %.4f ms" % (name, timeit(lambda: ops.mlp_x3_forward(ft, px, False)), timeit(lambda: ops.mlp_bf16_forward(fb, pb, False), 40), timeit(lambda: ops.mlp_f32_forward(ft, pf, False), 6)), flush=True)
