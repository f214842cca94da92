"""Ray datasets for the entry points (the reference's data_utils/ is host-side I/O outside the hot path; this is
the thin part the training / eval loops need):

  * `TrainDataset` / `ValDataset` with the reference's getters (data_utils/dataset.py:8-59, 63-167); the ray bundles
    are generated ON DEVICE by the HIP ray-generation kernels and stay resident in HBM (100 x 800 x 800 rays x 28 B
    = 1.8 GB -- nothing on a 288 GB card), so a training iteration does no host->device copy of rays;
  * sources: `blender` (transforms_*.json + PNG through PIL, data_utils/load_blender.py:68-145) and `procedural`
    (no files: camera ring around a seeded "teacher" DDNeRF whose fine render is the target; used where no dataset
    exists -- smoke tests, training-parity curves); `llff` / `real360` (poses_bounds.npy scenes) in llff.py.  Running
    COLMAP itself (data_utils/poses/) is out of scope: the scene must already carry poses_bounds.npy."""
from __future__ import annotations

import json
import math
import os

import numpy as np
import torch

from . import ops


def _pose_spherical(theta_deg, phi_deg, radius):
    """camera-to-world of a camera on a sphere looking at the origin (the Blender convention the reference uses for
    its render path, data_utils/load_blender.py:38-65)"""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    t = np.eye(4, dtype=np.float64)
    t[2, 3] = radius
    rp = np.array([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1.0]])
    c2w = rt @ rp @ t
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]]) @ c2w
    return c2w.astype(np.float32)


def load_blender(basedir, half_res=False, testskip=1):
    from PIL import Image

    imgs, poses, counts = [], [], [0]
    meta = None
    for s in ("train", "val", "test"):
        with open(os.path.join(basedir, "transforms_%s.json" % s)) as fp:
            meta = json.load(fp)
        skip = 1 if (s == "train" or testskip == 0) else testskip
        n0 = len(imgs)
        for frame in meta["frames"][::skip]:
            im = np.asarray(Image.open(os.path.join(basedir, frame["file_path"] + ".png")).convert("RGBA"), dtype=np.float32) / 255.0
            if half_res:
                # the reference halves with cv2.INTER_AREA on the float RGBA array (load_blender.py:133-140): a plain 2x2 box
                # mean per channel (PIL's own RGBA resize would premultiply by alpha)
                h2, w2 = im.shape[0] // 2, im.shape[1] // 2
                im = im[: 2 * h2, : 2 * w2].reshape(h2, 2, w2, 2, 4).mean((1, 3), dtype=np.float32)
            imgs.append(im)
            poses.append(np.array(frame["transform_matrix"], dtype=np.float32))
        counts.append(counts[-1] + len(imgs) - n0)
    imgs, poses = np.stack(imgs), np.stack(poses)
    H, W = imgs.shape[1:3]
    # data_utils/load_blender.py:104-105, 128-131: focal from the FULL-resolution width, halved with the images
    full_w = W * 2 if half_res else W
    focal = 0.5 * full_w / math.tan(0.5 * float(meta["camera_angle_x"]))
    if half_res:
        focal = focal / 2.0
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    render_poses = np.stack([_pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 181)[:-1]])
    return imgs, poses, render_poses, (H, W, focal), i_split


class _RayDatasetBase:
    def __init__(self, poses, images, focal, ndc_rays, device):
        self.poses = torch.as_tensor(poses, dtype=torch.float32)
        self.images = torch.as_tensor(images, dtype=torch.float32)
        self.H, self.W, self.focal = int(self.images.shape[1]), int(self.images.shape[2]), float(focal)
        self.ndc, self.near, self.device = bool(ndc_rays), 1, device

    def bundle(self, pose):
        o, d, r = ops.ray_bundle(self.H, self.W, self.focal, pose, device=self.device)
        if self.ndc:
            o, d, r = ops.ndc_rays(self.H, self.W, self.focal, o, d, self.near)
            r = r.unsqueeze(-1)
        return o, d, r


class TrainDataset(_RayDatasetBase):
    """data_utils/dataset.py:8-59"""

    def __init__(self, poses, images, focal, ndc_rays=False, single_image_mode=False, device="cuda"):
        super().__init__(poses, images, focal, ndc_rays, device)
        self.single_image_mode = single_image_mode
        o, d, r = zip(*((b[0].reshape(-1, 3), b[1].reshape(-1, 3), b[2].reshape(-1, 1)) for b in map(self.bundle, self.poses)))
        tgt = [im.reshape(-1, 3).to(device) for im in self.images]
        if single_image_mode:
            self.origins, self.directions, self.radii, self.target = list(o), list(d), list(r), tgt
        else:
            self.origins, self.directions, self.radii, self.target = (torch.cat(x) for x in (o, d, r, tgt))
        n = len(self.poses) * self.H * self.W
        print("training set init finnished, %d rays in the dataset (resident on %s)" % (n, device))

    def get_training_rays_for_next_iter(self, number_of_rays, device=None):
        if not self.single_image_mode:
            idx = torch.from_numpy(np.random.choice(self.origins.shape[0], number_of_rays)).to(self.origins.device)
            return self.origins[idx], self.directions[idx], self.radii[idx], self.target[idx]
        k = int(np.random.choice(len(self.origins), 1)[0])
        idx = torch.from_numpy(np.random.choice(self.origins[k].shape[0], number_of_rays)).to(self.origins[k].device)
        return self.origins[k][idx], self.directions[k][idx], self.radii[k][idx], self.target[k][idx]


class ValDataset(_RayDatasetBase):
    """data_utils/dataset.py:63-167 (without the depth-analysis plots)"""

    def __init__(self, poses, images, focal, ndc_rays=False, cfg=None, render_poses=None, device="cuda"):
        super().__init__(poses, images, focal, ndc_rays, device)
        self.current_idx, self.render_poses, self.render_idx = 0, render_poses, 0
        print("validation set init finnished, %d images in the dataset" % self.images.shape[0])

    def get_next_validation_rays(self, device=None):
        o, d, r = self.bundle(self.poses[self.current_idx])
        gt = self.images[self.current_idx].to(o.device)
        self.current_idx = (self.current_idx + 1) % self.images.shape[0]
        return o, d, r, gt

    def get_current_regular_validation_rays(self, device=None):
        return ops.ray_bundle(self.H, self.W, self.focal, self.poses[self.current_idx], device=self.device)

    def get_next_render_pose(self, device=None):
        o, d, r = self.bundle(self.render_poses[self.render_idx])
        self.render_idx += 1
        return o, d, r


def _procedural(cfg, device):
    """camera ring + targets rendered by a seeded teacher model through the HIP path (no files needed)"""
    import copy

    from . import synthetic
    from .models import DDNerfModel

    pc = cfg.dataset.get("procedural", {}) or {}
    H = W = int(pc.get("resolution", 64))
    n_train, n_val = int(pc.get("train_views", 12)), int(pc.get("val_views", 2))
    focal = 0.5 * W / math.tan(0.5 * 0.6911)
    tcfg = copy.deepcopy(cfg)
    tcfg.nerf.type = "DDNerfModel"
    for mode in ("train", "validation"):
        tcfg.nerf[mode]["num_coarse"], tcfg.nerf[mode]["num_fine"] = 64, 64
        tcfg.nerf[mode]["radiance_field_noise_std"], tcfg.nerf[mode]["perturb"] = 0.0, False
    teacher = DDNerfModel(tcfg)
    teacher.coarse.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(True, 101, 20.0).items()})
    teacher.fine.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(False, 102, 20.0).items()})
    teacher.to(device)
    teacher.eval()
    poses = np.stack([_pose_spherical(a, -25.0 - 10.0 * (i % 3), 4.0)
                      for i, a in enumerate(np.linspace(-180, 180, n_train + n_val, endpoint=False))])
    images = []
    with torch.no_grad():
        for p in poses:
            o, d, r = ops.ray_bundle(H, W, focal, p, device=device)
            images.append(teacher.run_iter(o, d, r, mode="validation")[1]["rgb"].clamp(0, 1).cpu())
    images = torch.stack(images)
    idx = np.arange(len(poses))
    return images, torch.from_numpy(poses), torch.from_numpy(poses), (H, W, focal), [idx[n_val:], idx[:n_val], idx[:n_val]]


def get_datasets(cfg, device="cuda"):
    """data_utils/data_utils.py:10-81"""
    kind = str(cfg.dataset.type).lower()
    if kind == "procedural":
        images, poses, render_poses, (H, W, focal), (i_train, i_val, _) = _procedural(cfg, device)
    elif kind == "blender":
        imgs, poses, render_poses, (H, W, focal), (i_train, i_val, _) = load_blender(
            cfg.dataset.basedir, half_res=cfg.dataset.half_res, testskip=cfg.dataset.testskip)
        a = imgs[..., -1:]
        images = imgs[..., :3] * a + (1.0 - a) if cfg.nerf.train.white_background else imgs[..., :3] * a  # :37-41
        images, poses = torch.from_numpy(images), torch.from_numpy(poses)
    elif kind in ("llff", "real360"):  # data_utils/data_utils.py:43-63
        from .llff import load_llff

        imgs, poses5, _bds, render_poses, i_test = load_llff(cfg)
        H, W, focal = (float(v) for v in poses5[0, :3, -1])
        H, W = int(H), int(W)
        n_img = imgs.shape[0]
        i_val = np.arange(n_img)[:: cfg.dataset.llffhold] if cfg.dataset.llffhold > 0 else np.array([i_test])
        i_train = np.array([i for i in range(n_img) if i not in i_val])
        images, poses = torch.from_numpy(imgs), torch.from_numpy(np.ascontiguousarray(poses5[:, :3, :4]))
        render_poses = torch.from_numpy(np.ascontiguousarray(render_poses[:, :3, :4]).astype(np.float32))
    else:
        raise SystemExit("unknown dataset type %r" % cfg.dataset.type)  # data_utils/data_utils.py:15-16
    if cfg.dataset.normalize_poses:  # data_utils/data_utils.py:65-74: the model reads the rescaled near/far live
        poses = poses.clone()
        poses[:, :3, 3] = poses[:, :3, 3] / cfg.dataset.normalize_factor
        cfg.dataset.near = cfg.dataset.near / cfg.dataset.normalize_factor
        cfg.dataset.far = cfg.dataset.far / cfg.dataset.normalize_factor
        cfg.dataset.combined_split = cfg.dataset.combined_split / cfg.dataset.normalize_factor
    train = TrainDataset(poses[i_train], images[i_train], focal, ndc_rays=cfg.dataset.ndc_rays,
                         single_image_mode=cfg.dataset.single_image_mode, device=device)
    val = ValDataset(poses[i_val], images[i_val], focal, ndc_rays=cfg.dataset.ndc_rays, cfg=cfg,
                     render_poses=render_poses, device=device)
    return train, val
