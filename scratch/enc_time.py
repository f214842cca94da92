"""A/B of encode kernels: usage enc_time.py lib1.so lib2.so ... (each exports ddnerf_encode); rays of the bench's blender batch, fenceposts of a
stratified coarse pass (S = 64) and of a sorted fine pass (S = 128)"""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops, synthetic
n = 4096
V = C.c_void_p
st = torch.cuda.current_stream().cuda_stream
ro, rd, rad, tgt = synthetic.make_rays("blender", n, 1)
rays = ops.pack_rays(torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda(), torch.from_numpy(rad).cuda(), 2.0, 6.0)
outs = {}
for so in sys.argv[1:]:
    L = C.CDLL(so)
    L.ddnerf_encode.argtypes = [V, V, V, C.c_int, C.c_int, C.c_int, C.c_int, V]
    for S in (64, 128):
        g = torch.Generator(device="cuda").manual_seed(S)
        t = (2 + 4 * torch.rand(n, S + 1, device="cuda", generator=g)).sort(dim=1).values.contiguous()
        for bf in (1, 0):
            feat = torch.empty(n * S, 128, dtype=torch.bfloat16 if bf else torch.float32, device="cuda")
            f = lambda: L.ddnerf_encode(rays.data_ptr(), t.data_ptr(), feat.data_ptr(), n, S, 0, bf, st)
            for _ in range(30): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300): f()
            e1.record(); torch.cuda.synchronize()
            key = (S, bf)
            same = ""
            if key in outs:
                same = "  identical to the first lib: %s" % bool(torch.equal(outs[key].view(torch.int16 if bf else torch.int32), feat.view(torch.int16 if bf else torch.int32)))
            else:
                outs[key] = feat.clone()
            print("%-28s S %3d %s: %.1f us%s" % (os.path.basename(so), S, "bf16" if bf else "fp32", e0.elapsed_time(e1) / 300 * 1e3, same))
