"""Forward-facing (LLFF) and real-world 360 scenes: `poses_bounds.npy` + an image folder -> images, camera-to-world
poses, depth bounds, a render path and the hold-out view.

Behaviour follows the reference's loader (data_utils/load_llff.py:63-368, itself the LLFF / nerf-pytorch convention):
  * `poses_bounds.npy` rows are 15 pose numbers (a 3x5 matrix: [R | t | (H, W, focal)]) + near/far depth bounds;
  * the rotation columns arrive as [down, right, back] and are re-ordered to [right, up, back]       (:289-291)
  * translations and bounds are rescaled so that the nearest bound sits at 1 / bd_factor             (:298-301)
  * poses are re-centred on their average pose                                                       (:180-193)
  * `spherify` re-centres on the point closest to all optical axes and scales the mean camera
    distance to 1                                                                                    (:196-279)
  * render path: a 2-turn spiral (LLFF, :163-177) or a tilted circle (REAL360, load_blender.py:45-65)
  * hold-out view = the camera closest to the average pose                                           (:357-360)
Pure host-side numpy; images are read with PIL (the reference uses imageio, absent here).  Unlike the reference,
a missing `images_<factor>` folder is not produced by shelling out to ImageMagick: the full-size images are
box-filtered down in memory.  Parity: the whole pose pipeline (everything behind the image decoding) is pinned against the
reference's own outputs on synthetic poses_bounds arrays (tests/golden/loaders.npz, tests/test_loader_parity.py); image
decoding and folder handling by property tests (tests/test_llff_loader.py).  COLMAP models -> poses_bounds: colmap.py."""
from __future__ import annotations

import math
import os

import numpy as np

_IMG_EXT = ("JPG", "jpg", "png")


def _unit(v):
    return v / np.linalg.norm(v)


def look_along(z, up, pos):
    """3x4 camera-to-world with viewing axis z, approximate up vector `up`, centre `pos` (load_llff.py:143-149)"""
    z = _unit(z)
    x = _unit(np.cross(up, z))
    y = _unit(np.cross(z, x))
    return np.stack([x, y, z, pos], 1)


def average_pose(poses):
    """3x5 pose: mean centre, summed viewing / up axes, intrinsics column of the first pose (load_llff.py:157-167)"""
    centre = poses[:, :3, 3].mean(0)
    z = poses[:, :3, 2].sum(0)
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([look_along(z, up, centre), poses[0, :3, -1:]], 1)


def _homogeneous(p34):
    """[..., 3, 4] -> [..., 4, 4]"""
    last = np.broadcast_to(np.array([0.0, 0.0, 0.0, 1.0], dtype=p34.dtype), p34.shape[:-2] + (1, 4))
    return np.concatenate([p34, last], -2)


def recenter(poses):
    """express every pose in the frame of the average pose (load_llff.py:180-193); the intrinsics column is kept"""
    out = poses.copy()
    ref = _homogeneous(average_pose(poses)[:3, :4])
    out[:, :3, :4] = (np.linalg.inv(ref) @ _homogeneous(poses[:, :3, :4]))[:, :3, :4]
    return out


def spherify(poses, bds, n_render=120):
    """inward-facing captures (load_llff.py:196-279): origin = least-squares intersection of the optical axes, z = mean
    offset of the cameras from it, mean camera distance scaled to 1; also returns the circular render path at the
    cameras' mean height.  -> (poses [N,3,5], render_poses [n_render,3,5], bds)"""
    axes = poses[:, :3, 2:3]
    centres = poses[:, :3, 3:4]
    proj = np.eye(3) - axes * np.transpose(axes, [0, 2, 1])  # projector orthogonal to each axis
    focus = np.squeeze(-np.linalg.inv((np.transpose(proj, [0, 2, 1]) @ proj).mean(0)) @ (-proj @ centres).mean(0))
    up = _unit((poses[:, :3, 3] - focus).mean(0))
    e1 = _unit(np.cross([0.1, 0.2, 0.3], up))
    e2 = _unit(np.cross(up, e1))
    frame = np.stack([e1, e2, up, focus], 1)
    reset = np.linalg.inv(_homogeneous(frame[None])) @ _homogeneous(poses[:, :3, :4])
    radius = math.sqrt(float(np.mean(np.sum(np.square(reset[:, :3, 3]), -1))))
    reset[:, :3, 3] *= 1.0 / radius
    bds = bds * (1.0 / radius)
    height = float(np.mean(reset[:, :3, 3], 0)[2])
    ring = math.sqrt(max(1.0 - height * height, 0.0))
    hwf = poses[0, :3, -1:]
    path = []
    for th in np.linspace(0.0, 2.0 * np.pi, n_render):
        c = np.array([ring * np.cos(th), ring * np.sin(th), height])
        z = _unit(c)
        x = _unit(np.cross(z, np.array([0.0, 0.0, -1.0])))
        y = _unit(np.cross(z, x))
        path.append(np.concatenate([np.stack([x, y, z, c], 1), hwf], 1))
    out = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, (reset.shape[0], 3, 1))], -1)
    return out, np.stack(path, 0), bds


def spiral_path(c2w, up, rads, focal, zrate=0.5, rots=2, n=120):
    """the LLFF fly-through (load_llff.py:163-177): a spiral around the average pose looking at a point `focal` ahead"""
    rads = np.array(list(rads) + [1.0])
    hwf = c2w[:, 4:5]
    look_at = c2w[:3, :4] @ np.array([0.0, 0.0, -focal, 1.0])
    path = []
    for th in np.linspace(0.0, 2.0 * np.pi * rots, n + 1)[:-1]:
        c = c2w[:3, :4] @ (np.array([np.cos(th), -np.sin(th), -np.sin(th * zrate), 1.0]) * rads)
        path.append(np.concatenate([look_along(c - look_at, up, c), hwf], 1))
    return np.array(path)


def real360_pose(theta_deg, phi_deg, radius, dataset_name=""):
    """camera on a tilted circle for the REAL360 render path (data_utils/load_blender.py:45-65), including the
    reference's hand-tuned adjustments for its scene named "beta" """
    beta = dataset_name == "beta"
    if beta:
        a = 0.7
        pivot = 90.0 if theta_deg <= 180 else 270.0
        radius = a * radius + (abs(pivot - theta_deg) / 90.0) * (1.0 - a) * radius

    def rot_x(ang):
        m = np.eye(4, dtype=np.float32)
        m[1, 1] = m[2, 2] = np.cos(ang)
        m[1, 2] = -np.sin(ang)
        m[2, 1] = np.sin(ang)
        return m

    def rot_y(ang):
        m = np.eye(4, dtype=np.float32)
        m[0, 0] = m[2, 2] = np.cos(ang)
        m[0, 2] = -np.sin(ang)
        m[2, 0] = np.sin(ang)
        return m

    def shift(axis, t):
        m = np.eye(4, dtype=np.float32)
        m[axis, 3] = t
        return m

    c2w = rot_y(theta_deg / 180.0 * np.pi) @ (rot_x(phi_deg / 180.0 * np.pi) @ shift(2, radius))
    if beta:
        c2w = shift(2, -0.03) @ (shift(1, -0.30) @ (rot_x(10 / 180.0 * np.pi) @ c2w))
    return np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w


def _image_files(folder):
    return [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if f.endswith(_IMG_EXT)]


def read_scene(basedir, factor=None):
    """-> poses [3,5,N] (intrinsics column already set to the loaded image size / scaled focal), bds [2,N],
    images [H,W,3,N] in [0,1]          (load_llff.py:63-136)"""
    from PIL import Image

    arr = np.load(os.path.join(basedir, "poses_bounds.npy"))
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0]).copy()
    bds = arr[:, -2:].transpose([1, 0]).copy()
    full = _image_files(os.path.join(basedir, "images"))
    folder = os.path.join(basedir, "images" + ("_%d" % factor if factor else ""))
    files = _image_files(folder) if os.path.isdir(folder) else None
    scale = float(factor) if factor else 1.0
    if files is None:  # no pre-shrunk folder: shrink in memory
        files = full
    if poses.shape[-1] != len(files):
        raise ValueError("mismatch between %d images and %d poses in %s" % (len(files), poses.shape[-1], basedir))
    imgs = []
    for f in files:
        im = Image.open(f).convert("RGB")
        if files is full and factor and factor != 1:
            im = im.resize((max(1, round(im.width / scale)), max(1, round(im.height / scale))), Image.BOX)
        imgs.append(np.asarray(im, dtype=np.float32)[..., :3] / 255.0)
    imgs = np.stack(imgs, -1)
    poses[:2, 4, :] = np.array(imgs.shape[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] / scale
    return poses, bds, imgs


def load_llff(cfg, do_recenter=True):
    """data_utils/load_llff.py:282-368 -> (images [N,H,W,3] f32, poses [N,3,5] f32, bds [N,2] f32,
    render_poses [R,3,5] (LLFF) or [R,4,4] (REAL360), i_test)"""
    kind = str(cfg.dataset.type).lower()
    if kind not in ("llff", "real360"):
        raise ValueError("dataset type is not supported by the LLFF loader: %r" % cfg.dataset.type)
    poses, bds, imgs = read_scene(cfg.dataset.basedir, factor=cfg.dataset.downsample_factor)
    # [down, right, back] -> [right, up, back]; move the image index to axis 0
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    bd_factor = cfg.dataset.bd_factor
    sc = 1.0 if bd_factor is False or bd_factor is None else 1.0 / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds = bds * sc
    if do_recenter:
        poses = recenter(poses)
    sphere_path = None
    if cfg.dataset.spherify:
        poses, sphere_path, bds = spherify(poses, bds)
    if kind == "llff":
        if sphere_path is not None:
            # the reference leaves its spiral inputs undefined on this combination (NameError, load_llff.py:327-333);
            # the spherified circle is the sensible path
            render_poses = sphere_path.astype(np.float32)
        else:
            c2w = average_pose(poses)
            up = _unit(poses[:, :3, 1].sum(0))
            close_depth, inf_depth = bds.min() * 0.9, bds.max() * 5.0
            dt = 0.75
            focus = 1.0 / ((1.0 - dt) / close_depth + dt / inf_depth)
            rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
            render_poses = spiral_path(c2w, up, rads, focus, zrate=0.5, rots=2, n=120).astype(np.float32)
    else:
        name = str(cfg.dataset.basedir).split("/")[-1]
        render_poses = np.stack([real360_pose(a, -10, 0.89, name) for a in np.linspace(0, 360, 180 + 1)[:-1]], 0)
    c2w = average_pose(poses)
    i_test = int(np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1)))
    return images.astype(np.float32), poses.astype(np.float32), bds.astype(np.float32), render_poses, i_test
