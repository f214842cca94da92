"""fine bf16 MLP launch time with features hot in the Infinity Cache (one buffer re-read) vs cold (eight buffers in rotation)"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
M = 524288
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
packed = ops.mlp_bf16_pack(flat, False)
bufs = []
for i in range(8):
    f = torch.zeros(M, 128, device="cuda"); f[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
    bufs.append(f[:, ops.K_ORDER].to(torch.bfloat16).contiguous())
def t(pick, reps=40):
    for i in range(5): ops.mlp_bf16_forward(bufs[pick(i)], packed, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): ops.mlp_bf16_forward(bufs[pick(i)], packed, False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(3):
    print("hot  (one buffer)    %.4f ms" % t(lambda i: 0), flush=True)
    print("cold (eight buffers) %.4f ms" % t(lambda i: i % 8), flush=True)
# per-launch event brackets, as bench.py's KernelTimer does, back to back and with other work in between
def per_launch(between, reps=30):
    ts = []
    for i in range(reps + 5):
        between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.mlp_bf16_forward(bufs[i % 8], packed, False); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = [a.elapsed_time(b) for a, b in ts[5:]]
    return sum(v) / len(v)
x = torch.rand(4096, 129, device="cuda")
rays = torch.rand(4096, 12, device="cuda"); tv = torch.sort(torch.rand(4096, 129, device="cuda") * 4 + 2, dim=1)[0].contiguous()
print("events, back to back         %.4f ms" % per_launch(lambda: None))
print("events, tiny kernel between  %.4f ms" % per_launch(lambda: x.add_(1.0)))
print("events, encode between       %.4f ms" % per_launch(lambda: ops.encode(rays, tv, cylinder=False, bf16=True)))
def idle():
    torch.cuda.synchronize()
print("events, sync (idle) between  %.4f ms" % per_launch(idle))
# the MLP reading what the encoder has just written (as in a render step)
rays4 = torch.rand(4096, 12, device="cuda"); rays4[:, 3:6] = torch.nn.functional.normalize(torch.rand(4096, 3, device="cuda") - 0.5, dim=1)
rays4[:, 9:12] = rays4[:, 3:6]; rays4[:, 6] = 1e-3
tv4 = torch.sort(torch.rand(4096, 129, device="cuda") * 4 + 2, dim=1)[0].contiguous()
def chained(reps=30):
    ts = []
    for i in range(reps + 5):
        f = ops.encode(rays4, tv4, cylinder=False, bf16=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.mlp_bf16_forward(f, packed, False); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = [a.elapsed_time(b) for a, b in ts[5:]]
    return sum(v) / len(v)
print("events, MLP on the fresh encoder output  %.4f ms" % chained())
print("events, encode between (other buffer)    %.4f ms" % per_launch(lambda: ops.encode(rays, tv, cylinder=False, bf16=True)))
print("events, MLP on the fresh encoder output  %.4f ms" % chained())
