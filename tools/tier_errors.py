#!/usr/bin/env python3
"""Render errors of the MLP tiers (fp32, x3, fp16, bf16) against the reference's own outputs at BASELINE sizes (tests/golden/fullsize_*):
max |d rgb|, |d depth|, |d acc|, |d weights| over the stored rays, both levels.  GPU box: python3 tools/tier_errors.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import fullsize_names, load_fullsize  # noqa: E402
from ddnerf_amd import synthetic  # noqa: E402
from test_hip_run_iter import build_model  # noqa: E402


def main():
    print("%-44s %-5s %s" % ("fixture", "tier", "  ".join("%-12s" % ("L%d %s" % (l, k)) for l in (0, 1) for k in ("rgb", "depth", "acc", "weights"))))
    for name in fullsize_names():
        c = load_fullsize(name)
        g, st = c["g"], c["stride"]
        ro, rd, rad, tgt = (torch.from_numpy(x).cuda() for x in synthetic.make_rays(c["kind"], c["n"], 1))
        for tier in ("fp32", "x3", "fp16", "bf16"):
            model = build_model(c)
            model.cfg.nerf["mlp_dtype"] = tier
            model._set_mlp_dtype()
            model.eval()
            with torch.no_grad():
                out = model.run_iter(ro, rd, rad, mode="validation", rgb_target=tgt)
            errs = []
            for lvl in (0, 1):
                for k in ("rgb", "depth", "acc", "weights"):
                    errs.append(float(np.abs(out[lvl][k][::st].cpu().numpy() - g["o%d_%s" % (lvl, k)]).max()))
            print("%-44s %-5s %s" % (name, tier, "  ".join("%-12.3e" % e for e in errs)), flush=True)


if __name__ == "__main__":
    main()
