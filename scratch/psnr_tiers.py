"""render-parity PSNR of the three MLP kernels against the reference's fp32 outputs on the golden fixtures"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from _cases import load_runiter, runiter_names
from test_hip_run_iter import build_model
rows = []
for name in runiter_names():
    if not name.endswith("_validation"): continue
    c = load_runiter(name); g = c["g"]
    d = lambda x: torch.from_numpy(x).cuda()
    line = [name]
    for mlp in ("fp32", "x3", "bf16"):
        model = build_model(c); model.cfg.nerf["mlp_dtype"] = mlp; model._set_mlp_dtype(); model.eval()
        with torch.no_grad():
            out = model.run_iter(d(g["ro"]), d(g["rd"]), d(g["rad"]), mode="validation", rgb_target=d(g["tgt"]))
        rgb, ref = out[1]["rgb"].cpu().numpy().astype(np.float64), g["o1_rgb"].astype(np.float64)
        mse = float(np.mean((rgb - ref) ** 2)); dep = float(np.abs(out[1]["depth"].cpu().numpy() - g["o1_depth"]).max())
        line.append("%s: %.1f dB (max|drgb| %.1e, max|ddepth| %.1e)" % (mlp, -10 * np.log10(max(mse, 1e-30)), float(np.abs(rgb - ref).max()), dep))
    print(" | ".join(line))
