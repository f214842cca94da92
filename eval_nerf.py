#!/usr/bin/env python3
"""Evaluation entry point with the reference's CLI (eval_nerf.py:173-181): `--logdir L [--checkpoint NAME]
[--save_images] [--extract_ptc]`.  Loads L/config.yml + L/NAME.ckpt, renders up to 10 validation images through the
HIP path, reports PSNR (coarse/fine) and seconds per image in L/validation/results.txt.  LPIPS / SSIM need packages
(and AlexNet weights) that are not available offline and are outside the hot path; PNGs are written with PIL."""
import argparse
import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ddnerf_amd import data, schedules  # noqa: E402
from ddnerf_amd import metrics  # noqa: E402
from ddnerf_amd.cfgnode import CfgNode  # noqa: E402
from models import models  # noqa: E402

MAX_VALIDATION_IMAGES = 10


def eval_model(basedir, checkpoint_name="checkpoint", extract_ptc=False, save_images=True):
    cfg = CfgNode.load(os.path.join(basedir, "config.yml"))
    savedir = os.path.join(basedir, "validation")
    os.makedirs(savedir, exist_ok=True)
    if not torch.cuda.is_available():
        raise SystemExit("eval_nerf.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cuda", 0)
    _, val_dataset = data.get_datasets(cfg, device)
    model = getattr(models, cfg.nerf.type)(cfg)
    if cfg.train_params.max_pdf_pad_iters < cfg.experiment.train_iters:  # eval_nerf.py:53-55
        cfg.train_params.pdf_padding = False
        cfg.train_params.gaussian_smooth_factor = cfg.train_params.final_smooth
    model.load_weights_from_checkpoint(torch.load(os.path.join(basedir, checkpoint_name + ".ckpt"), map_location=device))
    model.to(device)
    model.eval()
    results, summary, times = defaultdict(dict), defaultdict(list), []
    for i in range(min(len(val_dataset.poses), MAX_VALIDATION_IMAGES)):
        save_path = os.path.join(savedir, "val_image_%d" % (i + 1))
        os.makedirs(save_path, exist_ok=True)
        np.save(os.path.join(save_path, "pose.npy"), val_dataset.poses[i].numpy())
        torch.cuda.synchronize()
        start = time.time()
        with torch.no_grad():
            o, d, r, img = val_dataset.get_next_validation_rays(device)
            out = model.run_iter(o, d, r, mode="validation", depth_analysis_validation=False, rgb_target=img)
        torch.cuda.synchronize()
        times.append(time.time() - start)
        if extract_ptc:
            np.save(os.path.join(save_path, "xyz.npy"), (d * out[1]["depth"].unsqueeze(-1) + o).cpu().numpy())
        if save_images:
            from PIL import Image

            for lvl, name in ((0, "coarse"), (1, "fine")):
                im = (out[lvl]["rgb"].clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
                Image.fromarray(im).save(os.path.join(save_path, "rgb_%s.png" % name))
        for lvl, name in ((0, "psnr_coarse"), (1, "psnr_fine")):
            v = schedules.mse2psnr(float(torch.nn.functional.mse_loss(out[lvl]["rgb"], img)))
            results[i][name] = v
            summary[name].append(v)
        if min(img.shape[0], img.shape[1]) >= 7:  # eval_nerf.py:153-160 (SSIM on the grey images; LPIPS needs AlexNet weights)
            for lvl, name in ((0, "coarse"), (1, "fine")):
                v1, v2 = metrics.calc_ssim(out[lvl]["rgb"], img)
                for key, v in (("ssim_%s_v1" % name, v1), ("ssim_%s_v2" % name, v2)):
                    results[i][key] = v
                    summary[key].append(v)
        print("Avg time per image: %s" % (sum(times) / (i + 1)))
    write_dicts_to_a_file(summary, results, os.path.join(savedir, "results.txt"))
    return summary


def write_dicts_to_a_file(summary_dict, results_dict, results_file):
    """The reference's results.txt, line for line (validation_utils/visualization.py:137-150): the overall averages, then
    one line per image and metric, every number formatted `:.4`."""
    with open(results_file, "w") as f:
        print("average overall results:\n", file=f)
        for key in summary_dict.keys():
            score = sum(summary_dict[key]) / len(summary_dict[key])
            print(f"{key}: \t {score:.4}", file=f)
        print("\nper image results:\n", file=f)
        for key1 in results_dict.keys():
            for key2 in results_dict[key1].keys():
                print(f"image {key1} , {key2}: \t {results_dict[key1][key2]:.4}", file=f)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--logdir", type=str, required=True)
    ap.add_argument("--checkpoint", type=str, default="checkpoint")
    ap.add_argument("--save_images", action="store_true")
    ap.add_argument("--extract_ptc", action="store_true")
    a = ap.parse_args()
    eval_model(a.logdir, a.checkpoint, a.extract_ptc, a.save_images)
