"""where the Python-side time of a render step goes (the bf16 path is CPU-launch-bound)"""
import sys, os, cProfile, pstats, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
from ddnerf_amd import synthetic
class A: pass
args = A(); args.coarse, args.fine, args.mlp, args.rays = 64, 128, "bf16", 4096
dev = torch.device("cuda", 0)
model, _, _, _ = bench.build_model(args, dev, mlp="bf16"); model.eval()
ro, rd, rad, tgt = (torch.from_numpy(x).to(dev) for x in synthetic.make_rays("blender", 4096, 1))
def step():
    with torch.no_grad(): return model.run_iter(ro, rd, rad, mode="validation", rgb_target=tgt)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
