"""timing of ddnerf_mlp_x3_forward_train / ddnerf_mlp_x3_forward in variant libraries (results not checked)"""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import synthetic
M = 4096 * 128
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
raw = torch.empty(M, 4, device="cuda"); acts = torch.empty(2560, M, device="cuda")
bits = torch.empty(160, M, dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
V = C.c_void_p
for so in sys.argv[1:]:
    L = C.CDLL(so)
    L.ddnerf_mlp_x3_packed_bytes.restype = C.c_size_t
    pk = torch.empty(L.ddnerf_mlp_x3_packed_bytes(0), dtype=torch.uint8, device="cuda")
    L.ddnerf_mlp_x3_pack.argtypes = [V, C.c_int, V, V]
    assert L.ddnerf_mlp_x3_pack(flat.data_ptr(), 0, pk.data_ptr(), st) == 0
    f = L.ddnerf_mlp_x3_forward_train; f.argtypes = [V, V, C.c_int, V, V, V, C.c_long, C.c_long, V]
    fi = L.ddnerf_mlp_x3_forward; fi.argtypes = [V, V, C.c_int, V, C.c_long, V]
    fw = lambda: f(feat.data_ptr(), pk.data_ptr(), 0, raw.data_ptr(), acts.data_ptr(), bits.data_ptr(), M, M, st)
    fin = lambda: fi(feat.data_ptr(), pk.data_ptr(), 0, raw.data_ptr(), M, st)
    for name, fn in (("train", fw), ("infer", fin)):
        assert fn() == 0; torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
        print("%-24s %s median %.3f ms" % (os.path.basename(so), name, sorted(ts)[2]), flush=True)
