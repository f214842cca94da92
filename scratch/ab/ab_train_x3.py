"""interleaved A/B timing of the x3 training forward / backward-data kernels of variant builds (mlp_x3.hip + mlp_x3_train.hip + api.hip)"""
import ctypes as C, sys, os, statistics, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddnerf_amd import synthetic
libs = sys.argv[1:]
M = 524288
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
raw = torch.empty(M, 4, device="cuda"); graw = torch.randn(M, 4, device="cuda")
acts = torch.zeros(2560, M, device="cuda"); deltas = torch.zeros(2560, M, device="cuda"); bits = torch.zeros(160, M, dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
V = C.c_void_p
runs = []
for so in libs:
    L = C.CDLL(so)
    L.ddnerf_mlp_x3_packed_bytes.restype = C.c_size_t; L.ddnerf_mlp_x3_packed_t_bytes.restype = C.c_size_t
    packed = torch.empty(L.ddnerf_mlp_x3_packed_bytes(0), dtype=torch.uint8, device="cuda")
    packed_t = torch.empty(L.ddnerf_mlp_x3_packed_t_bytes(0), dtype=torch.uint8, device="cuda")
    L.ddnerf_mlp_x3_pack.argtypes = [V, C.c_int, V, V]; L.ddnerf_mlp_x3_pack_t.argtypes = [V, C.c_int, V, V]
    assert L.ddnerf_mlp_x3_pack(flat.data_ptr(), 0, packed.data_ptr(), st) == 0
    assert L.ddnerf_mlp_x3_pack_t(flat.data_ptr(), 0, packed_t.data_ptr(), st) == 0
    f = L.ddnerf_mlp_x3_forward_train; f.argtypes = [V, V, C.c_int, V, V, V, C.c_long, C.c_long, V]
    b = L.ddnerf_mlp_x3_backward_data; b.argtypes = [V, V, V, C.c_int, V, C.c_long, C.c_long, V]
    i = L.ddnerf_mlp_x3_forward; i.argtypes = [V, V, C.c_int, V, C.c_long, V]
    runs.append((so, "inf", lambda i=i, packed=packed: i(feat.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)))
    runs.append((so, "fwd", lambda f=f, packed=packed: f(feat.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), acts.data_ptr(), bits.data_ptr(), M, M, st)))
    runs.append((so, "bwd", lambda b=b, packed_t=packed_t: b(graw.data_ptr(), packed_t.data_ptr(), bits.data_ptr(), 0, deltas.data_ptr(), M, M, st)))
for so, k, fn in runs:
    assert fn() == 0
torch.cuda.synchronize()
times = {(so, k): [] for so, k, _ in runs}
for rnd in range(8):
    for so, k, fn in runs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        times[(so, k)].append(e0.elapsed_time(e1) / 5)
for (so, k), ts in times.items():
    print("%-30s %s median %.4f ms  min %.4f" % (os.path.basename(so), k, statistics.median(ts), min(ts)))
