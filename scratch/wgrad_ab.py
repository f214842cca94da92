"""new packed-operand x3 wgrad vs the fp32-operand one: bit-identity of the weights, bias agreement, timing"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 128
ld = (M + 127) // 128 * 128
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.zeros(2560, ld, device="cuda"); acts[:, :M] = torch.randn(2560, M, device="cuda", generator=g)
deltas = torch.zeros(2560, ld, device="cuda"); deltas[:, :M] = torch.randn(2560, M, device="cuda", generator=g) * 1e-3
pa, pd = ops.x3_split(acts), ops.x3_split(deltas)
assert float((ops.x3_unsplit(pa) - acts).abs().max()) <= 2 ** -16 * float(acts.abs().max())
ws = torch.empty(ops._lib.lib().ddnerf_mlp_f32_wgrad_workspace_floats(M), dtype=torch.float32, device="cuda")
jobs = ((512, 256, 256, 256, 256, 0, 256), (0, 256, 2432, 96, 96, 0, 96), (1280, 256, 1024, 256, 256, 96, 352),
        (2432, 3, 2304, 128, 128, 0, 128), (2304, 128, 2048, 256, 256, 0, 283), (2304, 128, 2528, 32, 27, 256, 283),
        (2435, 1, 2048, 256, 256, 0, 256))
for drow0, n_out, arow0, n_in, used, col0, dld in jobs:
    res = {}
    for mode, D, A in (("x3", deltas, acts), ("x3p", pd, pa)):
        w = torch.zeros(n_out, dld, device="cuda"); b = torch.zeros(n_out, device="cuda")
        f = lambda: ops.mlp_f32_wgrad_job(D, drow0, n_out, A, arow0, n_in, used, M, w, dld, col0, b, ws, mode=mode)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res[mode] = (w.clone(), b.clone(), e0.elapsed_time(e1) / 10)
    ident = torch.equal(res["x3"][0], res["x3p"][0])
    berr = float((res["x3"][1] - res["x3p"][1]).abs().max() / (res["x3"][1].abs().max() + 1e-30))
    gb = (n_out + n_in) * M * 4 / 1e9
    print("job d%d/%d a%d/%d: identical=%s bias_rel=%.2e  x3 %.3f ms (%.2f TB/s)  x3p %.3f ms (%.2f TB/s)" % (
        drow0, n_out, arow0, n_in, ident, berr, res["x3"][2], gb / res["x3"][2], res["x3p"][2], gb / res["x3p"][2]), flush=True)
