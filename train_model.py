#!/usr/bin/env python3
"""Training entry point with the reference's CLI (train_model.py:19-264): `--config X.yml [--load-checkpoint P]`.

Same config schema, schedules, loss, two Adam optimisers and checkpoint dict
({iter, model_1_state_dict, optimizer_1_state_dict, loss, psnr[, model_2_state_dict, optimizer_2_state_dict]} ->
logdir/checkpoint.ckpt); the per-ray work runs on the HIP path.  Data-parallel when launched with
`python -m torch.distributed.run --nproc-per-node N train_model.py ...` (one process per GPU, RCCL all-reduce of
the two flat gradient buffers).  TensorBoard is not available here: scalars go to logdir/train_log.jsonl."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ddnerf_amd import data, ops, schedules  # noqa: E402
from ddnerf_amd.cfgnode import CfgNode  # noqa: E402
from ddnerf_amd.train_step import TrainStepper  # noqa: E402
from models import models  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True, help="Path to (.yml) config file.")
    ap.add_argument("--load-checkpoint", type=str, default="", help="Path to load saved checkpoint from.")
    args = ap.parse_args()
    cfg = CfgNode.load(args.config)

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if not torch.cuda.is_available():
        raise SystemExit("train_model.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=device)

    logdir = os.path.join(cfg.experiment.logdir, cfg.experiment.id)
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        with open(os.path.join(logdir, "config.yml"), "w") as f:
            f.write(cfg.dump())
    seed = cfg.experiment.randomseed
    np.random.seed(seed + rank)   # every rank draws its own ray shard
    torch.manual_seed(seed + rank)

    train_dataset, val_dataset = data.get_datasets(cfg, device)
    model = getattr(models, cfg.nerf.type)(cfg)
    model.to(device)
    stepper = TrainStepper(model, cfg, dist=world > 1)

    start_iter = 0
    if os.path.exists(args.load_checkpoint):  # train_model.py:77-81, 111-118
        ckpt = torch.load(args.load_checkpoint, map_location=device)
        model.load_weights_from_checkpoint(ckpt)
        start_iter = ckpt["iter"] + 1
        val_dataset.current_idx = (ckpt["iter"] // cfg.experiment.validate_every) % val_dataset.images.shape[0]
        for k, o in enumerate(stepper.optims):
            o.load_state_dict(ckpt["optimizer_%d_state_dict" % (k + 1)])
        if start_iter > cfg.train_params.max_pdf_pad_iters:
            model.cfg.train_params.pdf_padding = False
    stepper.iter = start_iter
    print("set dist_reg_coeficient to - %s" % cfg.train_params.dist_reg_coeficient)
    log = open(os.path.join(logdir, "train_log.jsonl"), "a") if rank == 0 else None
    rays_per_rank = max(1, cfg.nerf.train.num_random_rays // world)
    psnr_fine, t_last = float("nan"), time.time()

    for i in range(start_iter, cfg.experiment.train_iters):
        o, d, r, tgt = train_dataset.get_training_rays_for_next_iter(rays_per_rank, device)
        loss, parts, _ = stepper.step(o, d, r, tgt)
        last = i == cfg.experiment.train_iters - 1
        if rank == 0 and (i % cfg.experiment.print_every == 0 or last):  # the only host sync of the train loop
            vals = [float(x) for x in parts]
            psnr_fine = schedules.mse2psnr(vals[1])
            rec = {"iter": i, "loss": float(loss), "psnr_coarse": schedules.mse2psnr(vals[0]), "psnr_fine": psnr_fine,
                   "lr": schedules.lr_at(i, cfg.experiment.train_iters, cfg.get("scheduler", None) if hasattr(cfg, "get") else None), "s_per_iter":
                   (time.time() - t_last) / max(1, cfg.experiment.print_every)}
            t_last = time.time()
            print(cfg.experiment.id + "\n[TRAIN] Iter: %d Loss: %s PSNR: %s dp coef: %s"
                  % (i, rec["loss"], psnr_fine, cfg.train_params.dp_coeficient))
            log.write(json.dumps(rec) + "\n")
            log.flush()
        if rank == 0 and (i % cfg.experiment.validate_every == 0 or last):  # train_model.py:197-245
            model.eval()
            t0 = time.time()
            with torch.no_grad():
                vo, vd, vr, img = val_dataset.get_next_validation_rays(device)
                out = model.run_iter(vo, vd, vr, mode="validation", rgb_target=img)
                mses = [float(torch.nn.functional.mse_loss(out[j]["rgb"], img)) for j in range(2)]
                if cfg.dataset.ndc_rays:  # train_model.py:226-228: NDC depth maps back in camera-space units (device op)
                    ro_reg, rd_reg, _ = val_dataset.get_current_regular_validation_rays(device)
                    for j in range(2):
                        out[j]["depth"] = ops.ndc_depth_to_regular(out[j]["depth"], ro_reg, rd_reg)
            rec = {"iter": i, "val_psnr_coarse": schedules.mse2psnr(mses[0]), "val_psnr_fine": schedules.mse2psnr(mses[1]),
                   "val_depth_fine_mean": float(out[1]["depth"].mean()), "val_time_s": time.time() - t0}
            print("[VAL] =======> Iter: %d Validation PSNR: %s Time: %s" % (i, rec["val_psnr_fine"], rec["val_time_s"]))
            log.write(json.dumps(rec) + "\n")
            log.flush()
        if rank == 0 and i > 0 and (i % cfg.experiment.save_every == 0 or last):  # train_model.py:248-263
            ckpt = {"iter": i, "model_1_state_dict": model.coarse.state_dict(),
                    "optimizer_1_state_dict": stepper.optims[0].state_dict(), "loss": loss, "psnr": psnr_fine}
            if cfg.nerf.type != "GeneralMipNerfModel":
                ckpt["model_2_state_dict"] = model.fine.state_dict()
                ckpt["optimizer_2_state_dict"] = stepper.optims[1].state_dict()
            torch.save(ckpt, os.path.join(logdir, "checkpoint.ckpt"))
    print("Done!")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
