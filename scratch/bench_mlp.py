"""micro-benchmark of the fused MLP kernels alone (fine-pass size), HIP-event timed"""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
which = sys.argv[2] if len(sys.argv) > 2 else "both"
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
fb = feat[:, ops.K_ORDER].to(torch.bfloat16).contiguous()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
FL = 1220608 * M
if which in ("both", "fp32"):
    p = ops.mlp_f32_pack(flat, False)
    t = timeit(lambda: ops.mlp_f32_forward(feat, p, False))
    print("fp32 fwd  %.4f ms  %.1f TFLOP/s  frac %.4f" % (t, FL / t / 1e9, FL / t / 1e9 / 157.3))
if which in ("both", "bf16"):
    p = ops.mlp_bf16_pack(flat, False)
    t = timeit(lambda: ops.mlp_bf16_forward(fb, p, False), 30)
    print("bf16 fwd  %.4f ms  %.1f TFLOP/s  frac %.4f" % (t, FL / t / 1e9, FL / t / 1e9 / 2500))
if which in ("both", "x3"):
    p = ops.mlp_x3_pack(flat, False)
    t = timeit(lambda: ops.mlp_x3_forward(feat, p, False), 20)
    print("x3   fwd  %.4f ms  %.1f TFLOP/s (fp32-equivalent)  x%.2f the fp32 MFMA peak; bf16 MFMA work %.1f TFLOP/s = frac %.4f" % (t, FL / t / 1e9, FL / t / 1e9 / 157.3, 3 * FL / t / 1e9, 3 * FL / t / 1e9 / 2500))
if which in ("train",):
    p = ops.mlp_f32_pack(flat, False); pt = ops.mlp_f32_pack_t(flat, False)
    t = timeit(lambda: ops.mlp_f32_forward_train(feat, p, False), 5)
    print("fp32 fwd_train %.4f ms" % t)
    raw, acts = ops.mlp_f32_forward_train(feat, p, False)
    g = torch.randn_like(raw)
    t = timeit(lambda: ops.mlp_f32_backward_data(g, pt, acts, False), 5)
    print("fp32 bwd_data %.4f ms" % t)
