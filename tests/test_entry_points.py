"""The reference's entry points (train_model.py / eval_nerf.py) end to end on the GPU:
a short training run on the procedural scene must learn (PSNR rises), write a checkpoint with the reference's
keys, resume from it, and eval must read it back and write results.txt."""
import json
import os
import subprocess
import sys

import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return p.stdout


def test_train_checkpoint_resume_eval(tmp_path):
    import torch

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "config_procedural.yml")))
    cfg["experiment"].update(logdir=str(tmp_path / "logs"), id="t", train_iters=81, validate_every=40, save_every=40, print_every=10)
    cfg["dataset"]["procedural"].update(resolution=24, train_views=5, val_views=1)
    for m in ("train", "validation"):
        cfg["nerf"][m].update(num_coarse=16, num_fine=16)
    cfg["nerf"]["train"]["num_random_rays"] = 512
    cpath = tmp_path / "cfg.yml"
    cpath.write_text(yaml.safe_dump(cfg))

    _run([os.path.join(ROOT, "train_model.py"), "--config", str(cpath)], str(tmp_path))
    logdir = tmp_path / "logs" / "t"
    recs = [json.loads(l) for l in open(logdir / "train_log.jsonl")]
    train = [r for r in recs if "psnr_fine" in r]
    assert train[0]["iter"] == 0 and train[-1]["iter"] == 80
    assert train[-1]["psnr_fine"] > train[0]["psnr_fine"] + 3.0, (train[0], train[-1])      # it learns
    assert all(r["loss"] == r["loss"] for r in train)                                        # no NaN
    ck = torch.load(logdir / "checkpoint.ckpt", map_location="cpu")
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "checkpoint_manifest.json")))
    assert set(ck) == set(man["checkpoint_keys_always"] + man["checkpoint_keys_two_networks"])   # train_model.py:248-263, exactly
    for key, tag in (("model_1_state_dict", "DDNerfModel.coarse"), ("model_2_state_dict", "DDNerfModel.fine")):
        assert [[k, list(v.shape), str(v.dtype)] for k, v in ck[key].items()] == man[tag]        # the reference's own manifest
    assert ck["iter"] == 80 and "fc_mu_sigma.weight" in ck["model_1_state_dict"] and "fc_rgb.weight" in ck["model_2_state_dict"]

    # resume: starts after the stored iteration, keeps going to the new horizon
    cfg["experiment"]["train_iters"] = 101
    cpath.write_text(yaml.safe_dump(cfg))
    out = _run([os.path.join(ROOT, "train_model.py"), "--config", str(cpath), "--load-checkpoint", str(logdir / "checkpoint.ckpt")], str(tmp_path))
    assert "Iter: 100" in out and "Iter: 40 " not in out
    assert torch.load(logdir / "checkpoint.ckpt", map_location="cpu")["iter"] == 100

    _run([os.path.join(ROOT, "eval_nerf.py"), "--logdir", str(logdir)], str(tmp_path))
    found = [os.path.join(d, f) for d, _, fs in os.walk(logdir) for f in fs if f == "results.txt"]
    assert found, "eval_nerf.py wrote no results.txt"
    lines = open(found[0]).read().splitlines()                                           # validation_utils/visualization.py:137-150
    assert lines[0] == "average overall results:" and lines[1] == "" and "per image results:" in lines
    head = lines[2:lines.index("per image results:") - 1]
    assert [l.split(":")[0] for l in head][:2] == ["psnr_coarse", "psnr_fine"] and any(l.startswith("ssim_fine_v2: \t ") for l in head)
    per = lines[lines.index("per image results:") + 2:]
    assert per and all(l.startswith("image 0 , ") and ": \t " in l for l in per)

    _run([os.path.join(ROOT, "render_video.py"), "--logdir", str(logdir), "--max_frames", "2", "--save_images"], str(tmp_path))
    frames = sorted(os.listdir(logdir / "video" / "frames"))
    assert frames == ["0000.png", "0001.png"] and os.path.exists(logdir / "video" / "disparity" / "0001.png")
    from PIL import Image
    assert Image.open(logdir / "video" / "frames" / "0000.png").size == (48, 24)        # RGB | disparity side by side
