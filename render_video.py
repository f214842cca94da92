#!/usr/bin/env python3
"""Fly-through renderer with the reference's entry point (render_video.py:16-106): loads `<logdir>/config.yml` and
`checkpoint.ckpt`, walks the dataset's render path (spiral / circle / turntable), renders every pose through the HIP path
and writes the frames.  The reference muxes an .avi with OpenCV; neither cv2 nor imageio exists here, so the frames
are written as `video/frames/NNNN.png` (RGB | disparity side by side, the reference's frame layout) -- any encoder can
assemble them -- plus, with --save_images, the separate `images/` and `disparity/` PNGs the reference writes."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ddnerf_amd import data  # noqa: E402
from ddnerf_amd.cfgnode import CfgNode  # noqa: E402
from models import models  # noqa: E402


def disparity_image(disp: torch.Tensor) -> np.ndarray:
    """validation_utils/visualization.py cast_to_disparity_image: min-max normalised to uint8"""
    d = disp.float()
    d = (d - d.min()) / (d.max() - d.min()).clamp_min(1e-12)
    return (d.clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()


def render_model_video(logdir, save_images=False, max_frames=None):
    from PIL import Image

    cfg = CfgNode.load(os.path.join(logdir, "config.yml"))
    if not torch.cuda.is_available():
        raise SystemExit("render_video.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cuda", 0)
    model = getattr(models, cfg.nerf.type)(cfg)
    if cfg.train_params.max_pdf_pad_iters < cfg.experiment.train_iters:  # render_video.py:38-40
        cfg.train_params.pdf_padding = False
        cfg.train_params.gaussian_smooth_factor = cfg.train_params.final_smooth
    model.load_weights_from_checkpoint(torch.load(os.path.join(logdir, "checkpoint.ckpt"), map_location=device))
    model.to(device)
    model.eval()
    _, val_dataset = data.get_datasets(cfg, device)
    savedir = os.path.join(logdir, "video")
    for sub in ("frames", "images", "disparity"):
        os.makedirs(os.path.join(savedir, sub), exist_ok=True)
    n = val_dataset.render_poses.shape[0]
    if max_frames:
        n = min(n, int(max_frames))
    times = []
    for i in range(n):
        torch.cuda.synchronize()
        start = time.time()
        with torch.no_grad():
            o, d, r = val_dataset.get_next_render_pose(device)
            out = model.run_iter(o, d, r, mode="validation", depth_analysis_validation=False)
        torch.cuda.synchronize()
        times.append(time.time() - start)
        rgb = (out[1]["rgb"][..., :3].clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
        disp = disparity_image(out[1]["disp"])
        frame = np.concatenate([rgb, np.repeat(disp[..., None], 3, -1)], 1)  # [H, 2W, 3]
        Image.fromarray(frame).save(os.path.join(savedir, "frames", "%04d.png" % i))
        if save_images:
            Image.fromarray(rgb).save(os.path.join(savedir, "images", "%04d.png" % i))
            Image.fromarray(disp).save(os.path.join(savedir, "disparity", "%04d.png" % i))
        print("Avg time per image: %s" % (sum(times) / (i + 1)))
    return savedir


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--logdir", type=str, required=True)
    ap.add_argument("--save_images", action="store_true")
    ap.add_argument("--max_frames", type=int, default=0, help="render only the first N poses of the path (0: all)")
    a = ap.parse_args()
    render_model_video(a.logdir, a.save_images, a.max_frames)
