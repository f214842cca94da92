"""the encode kernel alone at the fine-pass size (4096 rays x 128 samples), bf16 and fp32 feature rows"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import _lib
if len(sys.argv) > 1: _lib.SO_PATH = os.path.abspath(sys.argv[1])
from ddnerf_amd import ops, synthetic
ro, rd, rad, _ = (torch.from_numpy(x).cuda() for x in synthetic.make_rays("blender", 4096, 1))
rays = ops.pack_rays(ro, rd, rad, 2.0, 6.0)
t = torch.sort(torch.rand(4096, 129, device="cuda") * 4 + 2, dim=1).values
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("encode 4096 x 128: bf16 rows %.1f us, fp32 rows %.1f us" % (timeit(lambda: ops.encode(rays, t, bf16=True)), timeit(lambda: ops.encode(rays, t))))
