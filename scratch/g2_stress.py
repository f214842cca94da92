"""stress: the two bf16 kernels bit for bit on fresh random rows, many launches back to back (a race in the assembly kernel's
memory-counter bookkeeping would show up as rare mismatches)"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
KO = torch.as_tensor(ops.K_ORDER, device="cuda")
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 500
bad = 0
t0 = time.time()
for depth in (False, True):
    sd = synthetic.make_state_dict(depth, 12, 20.0)
    names = [n for n, _, _ in synthetic.layer_table(depth)]
    flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
    p1, p2 = ops.mlp_bf16g1_pack(flat, depth), ops.mlp_bf16g2_pack(flat, depth)
    g = torch.Generator(device="cuda").manual_seed(7 + depth)
    for it in range(n_iter):
        M = (524288, 262144, 400000 + 977 * (it % 600), 131072 + (it % 4000))[it % 4]
        feat = torch.zeros(M, 128, device="cuda")
        feat[:, :123] = torch.rand(M, 123, device="cuda", generator=g) * 2 - 1
        fb = feat[:, KO].to(torch.bfloat16).contiguous()
        a = ops.mlp_bf16g1_forward(fb, p1, depth)
        # several launches of the assembly kernel back to back on the same rows: all must equal the one-group kernel's output
        for rep in range(3):
            b = ops.mlp_bf16g2_forward(fb, p2, depth)
            if not torch.equal(a, b):
                bad += 1
                print("MISMATCH depth %d it %d rep %d M %d rows %d" % (depth, it, rep, M, int((a != b).any(dim=1).sum())), flush=True)
        if it % 100 == 0:
            print("depth %d it %d ok so far (%.0f s)" % (depth, it, time.time() - t0), flush=True)
print("done: %d mismatching launches of %d" % (bad, 2 * n_iter * 3))
sys.exit(1 if bad else 0)
