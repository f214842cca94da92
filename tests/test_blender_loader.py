"""Blender-format scenes (transforms_{train,val,test}.json + RGBA PNGs): ddnerf_amd.data.load_blender against the
conventions of the reference's loader (data_utils/load_blender.py:68-145), and the dataset objects on the device."""
import json
import math
import os

import numpy as np
import pytest


def make_scene(root, n=(5, 2, 4), hw=(16, 16), fov=0.6911):
    from PIL import Image

    rng = np.random.default_rng(0)
    os.makedirs(root, exist_ok=True)
    for split, k in zip(("train", "val", "test"), n):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i in range(k):
            img = rng.integers(0, 256, (hw[0], hw[1], 4), dtype=np.uint8)
            img[: hw[0] // 2, :, 3] = 0          # transparent upper half: exercises the white-background compositing
            Image.fromarray(img, "RGBA").save(os.path.join(root, split, "r_%d.png" % i))
            pose = np.eye(4)
            pose[:3, 3] = rng.standard_normal(3) * 2
            frames.append({"file_path": "./%s/r_%d" % (split, i), "transform_matrix": pose.tolist()})
        json.dump({"camera_angle_x": fov, "frames": frames}, open(os.path.join(root, "transforms_%s.json" % split), "w"))


def test_load_blender_conventions(tmp_path):
    from ddnerf_amd import data

    make_scene(str(tmp_path))
    imgs, poses, render_poses, (H, W, focal), (i_tr, i_va, i_te) = data.load_blender(str(tmp_path), half_res=False, testskip=2)
    assert imgs.shape == (5 + 1 + 2, 16, 16, 4) and imgs.dtype == np.float32 and imgs.max() <= 1.0     # val / test every 2nd frame
    assert list(i_tr) == [0, 1, 2, 3, 4] and list(i_va) == [5] and list(i_te) == [6, 7]
    assert (H, W) == (16, 16) and math.isclose(focal, 0.5 * 16 / math.tan(0.5 * 0.6911), rel_tol=1e-12)
    assert poses.shape == (8, 4, 4) and render_poses.shape == (180, 4, 4)
    assert np.allclose(np.linalg.norm(render_poses[:, :3, 3], axis=-1), 4.0, atol=1e-5)                # turntable radius 4
    h_imgs, _, _, (h, w, f2), _ = data.load_blender(str(tmp_path), half_res=True, testskip=0)
    assert (h, w) == (8, 8) and math.isclose(f2, focal / 2, rel_tol=1e-12) and h_imgs.shape[0] == 11
    assert np.allclose(h_imgs[0, 0, 0], imgs[0, :2, :2].mean((0, 1)), atol=1 / 255 + 1e-6)            # area averaging


@pytest.mark.gpu
def test_get_datasets_blender_on_device(tmp_path):
    import torch
    from ddnerf_amd import data
    from ddnerf_amd.cfgnode import CfgNode

    make_scene(str(tmp_path))
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = CfgNode.load(os.path.join(here, "configs", "config_blender.yml"))
    cfg.dataset.basedir = str(tmp_path)
    cfg.dataset.testskip = 1
    for white in (False, True):
        cfg.nerf.train.white_background = white
        train, val = data.get_datasets(cfg, device="cuda")
        o, d, r, t = train.get_training_rays_for_next_iter(32)
        assert o.shape == (32, 3) and r.shape == (32, 1) and t.shape == (32, 3) and t.is_cuda
        assert val.images.shape == (2, 16, 16, 3)
        top = val.images[0, :8]                     # alpha = 0 there
        assert torch.all(top == (1.0 if white else 0.0))      # data_utils/data_utils.py:37-41
        assert float(r.min()) > 0 and torch.allclose(d.norm(dim=-1) > 0, torch.ones(32, dtype=torch.bool, device="cuda"))
