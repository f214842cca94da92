"""Compare two builds of the bf16 MLP kernel sample by sample (debug aid): dbg_bf16_diff.py ref.so test.so [M]"""
import ctypes as C, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import synthetic
M = int(sys.argv[3]) if len(sys.argv) > 3 else 512
depth = 1
sd = synthetic.make_state_dict(True, 12, 1.0)
names = [n for n, _, _ in synthetic.layer_table(True)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
fb = (torch.rand(M, 128, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16).contiguous()
st = torch.cuda.current_stream().cuda_stream
V = C.c_void_p
outs = []
for so in sys.argv[1:3]:
    L = C.CDLL(so)
    L.ddnerf_mlp_bf16_packed_bytes.restype = C.c_size_t
    packed = torch.zeros(L.ddnerf_mlp_bf16_packed_bytes(depth), dtype=torch.uint8, device="cuda")
    L.ddnerf_mlp_bf16_pack.argtypes = [V, C.c_int, V, V]
    assert L.ddnerf_mlp_bf16_pack(flat.data_ptr(), depth, packed.data_ptr(), st) == 0
    f = L.ddnerf_mlp_bf16_forward; f.argtypes = [V, V, C.c_int, V, C.c_long, V]
    raw = torch.zeros(M, 6, device="cuda")
    for rep in range(3):
        assert f(fb.data_ptr(), packed.data_ptr(), depth, raw.data_ptr(), M, st) == 0
        torch.cuda.synchronize()
        outs.append(raw.clone())
ref = outs[0]
for i, o in enumerate(outs):
    d = (o - ref).abs()
    bad = (d.max(1).values > 1e-6).nonzero().flatten()
    print("run %d: max diff %.3g, samples off %d / %d; per column max %s" % (i, float(d.max()), bad.numel(), M, [float("%.2g" % x) for x in d.max(0).values]))
    if bad.numel():
        b = bad.cpu().numpy()
        print("   first bad samples", b[:40], " lane&15 hist", np.bincount(b % 16, minlength=16), " colblock hist", np.bincount((b // 16) % 4, minlength=4), "wave hist", np.bincount((b // 64) % 4, minlength=4))
