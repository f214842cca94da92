import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
n, S = 4096, 64
V = C.c_void_p
st = torch.cuda.current_stream().cuda_stream
ro, rd, rad, tgt = synthetic.make_rays("blender", n, 1)
rays = ops.pack_rays(torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda(), torch.from_numpy(rad).cuda(), 2.0, 6.0)
g = torch.Generator(device="cuda").manual_seed(S)
t = (2 + 4 * torch.rand(n, S + 1, device="cuda", generator=g)).sort(dim=1).values.contiguous()
res = []
for so in sys.argv[1:]:
    L = C.CDLL(so); L.ddnerf_encode.argtypes = [V, V, V, C.c_int, C.c_int, C.c_int, C.c_int, V]
    feat = torch.full((n * S, 128), float("nan"), dtype=torch.float32, device="cuda")
    L.ddnerf_encode(rays.data_ptr(), t.data_ptr(), feat.data_ptr(), n, S, 0, 0, st); torch.cuda.synchronize()
    res.append(feat.cpu())
a, b = res
d = (a.view(torch.int32) != b.view(torch.int32))
print("differing elements", int(d.sum()), "columns", d.any(0).nonzero().flatten().tolist()[:40], "rows", d.any(1).nonzero().flatten().tolist()[:20])
r = d.any(1).nonzero().flatten()[0].item(); c = d[r].nonzero().flatten()[0].item()
print(r, c, a[r, c].item(), b[r, c].item(), a[r, 96:100], b[r, 96:100])
