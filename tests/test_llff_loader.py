"""LLFF / REAL360 scene loader (ddnerf_amd/llff.py; reference data_utils/load_llff.py:63-368).  The reference loader
cannot be imported here (imageio / cv2 are absent), so these are property tests on a synthetic scene: the conventions
the loader promises, each of which the reference's code implies."""
import os
import types

import numpy as np
import pytest

from ddnerf_amd import llff


def _rot(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return q * np.sign(np.linalg.det(q))


def make_scene(root, n=9, hw=(12, 16), seed=0, inward=False, shrunk_folder=None):
    """cameras roughly on a patch (forward-facing) or a ring (inward) -> poses_bounds.npy + images/"""
    from PIL import Image

    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    rows = []
    for i in range(n):
        if inward:
            th = 2 * np.pi * i / n
            c = np.array([3 * np.cos(th), 3 * np.sin(th), 0.4 + 0.1 * rng.standard_normal()])
            back = c / np.linalg.norm(c)  # camera looks at the origin along -back
        else:
            c = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), 0.1 * rng.standard_normal()])
            back = np.array([0.05 * rng.standard_normal(), 0.05 * rng.standard_normal(), 1.0])
            back /= np.linalg.norm(back)
        right = np.cross([0.0, 1.0, 0.0] if not inward else [0.0, 0.0, 1.0], back)
        right /= np.linalg.norm(right)
        up = np.cross(back, right)
        # file convention: columns [down, right, back]
        m = np.stack([-up, right, back, c, np.array([hw[0] * 4, hw[1] * 4, 50.0])], 1)
        rows.append(np.concatenate([m.reshape(-1), [2.0 + rng.uniform(0, 0.5), 9.0 + rng.uniform(0, 2)]]))
        img = (rng.uniform(0, 255, (hw[0] * 4, hw[1] * 4, 3))).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, "images", "im_%03d.png" % i))
        if shrunk_folder:
            os.makedirs(os.path.join(root, shrunk_folder), exist_ok=True)
            Image.fromarray(img[::4, ::4]).save(os.path.join(root, shrunk_folder, "im_%03d.png" % i))
    np.save(os.path.join(root, "poses_bounds.npy"), np.stack(rows))
    return np.stack(rows)


def cfg_for(root, kind="LLFF", factor=4, spherify=False, bd_factor=0.75, llffhold=4):
    ds = types.SimpleNamespace(type=kind, basedir=str(root), downsample_factor=factor, bd_factor=bd_factor,
                               spherify=spherify, llffhold=llffhold)
    return types.SimpleNamespace(dataset=ds)


def test_forward_facing_conventions(tmp_path):
    raw = make_scene(tmp_path, n=9, shrunk_folder="images_4")
    images, poses, bds, render, i_test = llff.load_llff(cfg_for(tmp_path))
    assert images.shape == (9, 12, 16, 3) and images.dtype == np.float32 and 0.0 <= images.min() and images.max() <= 1.0
    assert poses.shape == (9, 3, 5) and bds.shape == (9, 2)
    # intrinsics column: loaded image size, focal divided by the factor
    assert np.allclose(poses[:, 0, 4], 12) and np.allclose(poses[:, 1, 4], 16) and np.allclose(poses[:, 2, 4], 50.0 / 4)
    # nearest bound sits at 1 / bd_factor
    assert np.isclose(bds.min(), 1.0 / 0.75, rtol=1e-5)
    assert np.isclose(bds.max() / bds.min(), raw[:, -1].max() / raw[:, -2].min(), rtol=1e-5)
    # rotations stay orthonormal, right-handed
    R = poses[:, :3, :3]
    assert np.allclose(R @ np.transpose(R, (0, 2, 1)), np.eye(3), atol=1e-5)
    assert np.allclose(np.linalg.det(R), 1.0, atol=1e-5)
    # recentred: the average pose is the identity
    avg = llff.average_pose(poses)
    assert np.allclose(avg[:3, :3], np.eye(3), atol=1e-5) and np.allclose(avg[:3, 3], 0.0, atol=1e-5)
    # hold-out view: the camera closest to the average centre
    assert i_test == int(np.argmin(np.sum(poses[:, :3, 3] ** 2, -1)))
    # spiral: 120 poses, orthonormal, every camera looks at the focus point ahead of the average pose
    assert render.shape == (120, 3, 5) and render.dtype == np.float32
    Rr = render[:, :3, :3]
    assert np.allclose(Rr @ np.transpose(Rr, (0, 2, 1)), np.eye(3), atol=1e-5)
    close, far = bds.min() * 0.9, bds.max() * 5.0
    focus = 1.0 / (0.25 / close + 0.75 / far)
    target = np.array([0.0, 0.0, -focus])
    to_target = target - render[:, :3, 3]
    cosang = np.sum(-render[:, :3, 2] * to_target, -1) / np.linalg.norm(to_target, axis=-1)
    assert np.allclose(cosang, 1.0, atol=1e-4)
    # first spiral pose sits at the 90th-percentile x radius
    assert np.isclose(render[0, 0, 3], np.percentile(np.abs(poses[:, 0, 3]), 90), rtol=1e-4)


def test_in_memory_shrink_matches_folder_geometry(tmp_path):
    make_scene(tmp_path, n=5)  # no images_4 folder: the loader box-filters the full-size images
    images, poses, _, _, _ = llff.load_llff(cfg_for(tmp_path))
    assert images.shape == (5, 12, 16, 3)
    assert np.allclose(poses[:, 2, 4], 50.0 / 4)


def test_pose_image_mismatch_is_an_error(tmp_path):
    make_scene(tmp_path, n=5)
    os.remove(os.path.join(tmp_path, "images", "im_004.png"))
    with pytest.raises(ValueError):
        llff.load_llff(cfg_for(tmp_path))


def test_spherify_real360(tmp_path):
    root = tmp_path / "garden"
    make_scene(root, n=12, inward=True, shrunk_folder="images_4")
    images, poses, bds, render, i_test = llff.load_llff(cfg_for(root, kind="REAL360", spherify=True))
    c = poses[:, :3, 3]
    assert np.isclose(np.sqrt(np.mean(np.sum(c * c, -1))), 1.0, atol=1e-5)  # mean camera distance 1
    # cameras look inward: viewing direction (-z column) points towards the origin
    view = -poses[:, :3, 2]
    assert np.all(np.sum(view * (-c), -1) / np.linalg.norm(c, axis=-1) > 0.9)
    # REAL360 render path: 180 4x4 poses on a circle of radius 0.89 tilted by -10 degrees
    assert render.shape == (180, 4, 4)
    assert np.allclose(np.linalg.norm(render[:, :3, 3], axis=-1), 0.89, atol=1e-5)
    assert 0 <= i_test < 12


def test_spherify_ring_path(tmp_path):
    make_scene(tmp_path, n=12, inward=True, shrunk_folder="images_4")
    poses, bds, _ = llff.read_scene(str(tmp_path), factor=4)
    p = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    p = np.moveaxis(p, -1, 0)
    out, ring, b2 = llff.spherify(llff.recenter(p), np.moveaxis(bds, -1, 0))
    assert ring.shape == (120, 3, 5)
    assert np.allclose(np.linalg.norm(ring[:, :3, 3], axis=-1), 1.0, atol=1e-6)
    assert np.allclose(ring[:, 2, 3], np.mean(out[:, 2, 3]), atol=1e-6)  # at the cameras' mean height


def test_real360_beta_adjustment():
    a = llff.real360_pose(90.0, -10, 0.89, "beta")
    b = llff.real360_pose(90.0, -10, 0.89, "other")
    assert a.shape == (4, 4) and not np.allclose(a, b)
    # at theta = 90 the "beta" radius shrinks to 0.7 x
    plain = llff.real360_pose(90.0, -10, 0.7 * 0.89, "other")
    undo = np.linalg.inv(np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]]))
    assert np.isclose(np.linalg.norm((undo @ plain)[:3, 3]), 0.7 * 0.89, atol=1e-6)


@pytest.mark.gpu
def test_get_datasets_llff_on_device(tmp_path):
    import torch
    from ddnerf_amd import data
    from ddnerf_amd.cfgnode import CfgNode
    import yaml

    make_scene(tmp_path, n=9, shrunk_folder="images_4")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = CfgNode(yaml.safe_load(open(os.path.join(here, "configs", "config_ff.yml"))))
    cfg.dataset.basedir = str(tmp_path)
    cfg.dataset.llffhold = 4
    train, val = data.get_datasets(cfg, device="cuda")
    o, d, r, t = train.get_training_rays_for_next_iter(64)
    assert o.shape == (64, 3) and d.shape == (64, 3) and r.shape == (64, 1) and t.shape == (64, 3) and o.is_cuda
    assert torch.allclose(o[:, 2], torch.full_like(o[:, 2], -1.0))  # NDC rays start on the near plane
    vo, vd, vr, gt = val.get_next_validation_rays()
    assert vo.shape == (12, 16, 3) and gt.shape == (12, 16, 3)
    assert val.images.shape[0] == 3 and len(train.poses) == 6  # llffhold 4 of 9 -> views 0, 4, 8 held out
    ro, rd, rr = val.get_next_render_pose()
    assert ro.shape == (12, 16, 3)
