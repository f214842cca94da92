import sys, os, time, torch, gc
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
from ddnerf_amd import synthetic, ops
class A: pass
args = A(); args.coarse, args.fine, args.mlp, args.rays, args.warmup, args.steps = 64, 128, "fp32", 4096, 3, 20
dev = torch.device("cuda", 0)
ro, rd, rad, tgt = (torch.from_numpy(x).to(dev) for x in synthetic.make_rays("blender", 4096, 1))
def run(mlp, tag):
    model, _, _, _ = bench.build_model(args, dev, mlp=mlp); model.eval()
    def step():
        with torch.no_grad(): return model.run_iter(ro, rd, rad, mode="validation", rgb_target=tgt)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(tag, mlp, "cpu issue ms/step %.3f  total ms/step %.3f  reserved MB %.0f" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, torch.cuda.memory_reserved() / 1e6), flush=True)
run("bf16", "first")
run("fp32", "second")
run("bf16", "after fp32")
gc.collect(); torch.cuda.empty_cache()
run("bf16", "after empty_cache")
