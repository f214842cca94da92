"""COLMAP sparse models (cameras / images / points3D, binary and text) and the camera poses + depth bounds the LLFF tool chain
derives from them (replaces data_utils/poses/colmap_read_model.py:82-296 and the numpy half of
data_utils/poses/pose_utils.py:10-90).  Host-side I/O only: numpy + struct.

File formats (COLMAP's published layout, little endian):
  cameras.bin   u64 n | n x { i32 camera_id, i32 model_id, u64 width, u64 height, f64 params[num_params(model_id)] }
  images.bin    u64 n | n x { i32 image_id, f64 qvec[4] (w x y z), f64 tvec[3], i32 camera_id, name bytes + NUL,
                              u64 m, m x { f64 x, f64 y, i64 point3D_id } }
  points3D.bin  u64 n | n x { u64 point3D_id, f64 xyz[3], u8 rgb[3], f64 error, u64 t, t x { i32 image_id, i32 point2D_idx } }
The text files carry the same records, one per line ('#' comments; images.txt: two lines per image).

The readers return dicts keyed by id, in file order, of records with the reference's field names.  The writers exist for
the fixtures and tests (tests/golden/make_golden.py writes a tiny model with them; the reference's readers parse it and what
they parsed is the golden vector)."""
from __future__ import annotations

import collections
import os
import struct

import numpy as np

Camera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])
_ImageBase = collections.namedtuple("Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])
Point3D = collections.namedtuple("Point3D", ["id", "xyz", "rgb", "error", "image_ids", "point2D_idxs"])


class Image(_ImageBase):
    def qvec2rotmat(self):
        return qvec2rotmat(self.qvec)


# model_id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
_MODEL_ID = {name: (mid, n) for mid, (name, n) in CAMERA_MODELS.items()}


def _read(fid, fmt):
    data = fid.read(struct.calcsize("<" + fmt))
    if len(data) != struct.calcsize("<" + fmt):
        raise ValueError("truncated COLMAP file")
    return struct.unpack("<" + fmt, data)


# ---- binary ------------------------------------------------------------------------------------------------------
def read_cameras_binary(path):
    cameras = {}
    with open(path, "rb") as fid:
        (n,) = _read(fid, "Q")
        for _ in range(n):
            cid, mid, w, h = _read(fid, "iiQQ")
            if mid not in CAMERA_MODELS:
                raise ValueError("unknown COLMAP camera model id %d" % mid)
            name, npar = CAMERA_MODELS[mid]
            cameras[cid] = Camera(id=cid, model=name, width=w, height=h, params=np.array(_read(fid, "d" * npar)))
    return cameras


def read_images_binary(path):
    images = {}
    with open(path, "rb") as fid:
        (n,) = _read(fid, "Q")
        for _ in range(n):
            rec = _read(fid, "idddddddi")
            name = bytearray()
            while True:
                (c,) = _read(fid, "c")
                if c == b"\x00":
                    break
                name += c
            (m,) = _read(fid, "Q")
            flat = _read(fid, "ddq" * m)
            xys = np.column_stack([np.array(flat[0::3], dtype=np.float64), np.array(flat[1::3], dtype=np.float64)]) if m else np.zeros((0, 2))
            images[rec[0]] = Image(id=rec[0], qvec=np.array(rec[1:5]), tvec=np.array(rec[5:8]), camera_id=rec[8], name=name.decode("utf-8"),
                                   xys=xys, point3D_ids=np.array(flat[2::3], dtype=np.int64))
    return images


def read_points3d_binary(path):
    points = {}
    with open(path, "rb") as fid:
        (n,) = _read(fid, "Q")
        for _ in range(n):
            rec = _read(fid, "QdddBBBd")
            (t,) = _read(fid, "Q")
            track = _read(fid, "ii" * t)
            points[rec[0]] = Point3D(id=rec[0], xyz=np.array(rec[1:4]), rgb=np.array(rec[4:7]), error=rec[7],
                                     image_ids=np.array(track[0::2], dtype=np.int64), point2D_idxs=np.array(track[1::2], dtype=np.int64))
    return points


def write_cameras_binary(cameras, path):
    with open(path, "wb") as fid:
        fid.write(struct.pack("<Q", len(cameras)))
        for cam in cameras.values():
            mid, npar = _MODEL_ID[cam.model]
            assert len(cam.params) == npar
            fid.write(struct.pack("<iiQQ", cam.id, mid, int(cam.width), int(cam.height)))
            fid.write(struct.pack("<" + "d" * npar, *[float(p) for p in cam.params]))


def write_images_binary(images, path):
    with open(path, "wb") as fid:
        fid.write(struct.pack("<Q", len(images)))
        for im in images.values():
            fid.write(struct.pack("<idddddddi", im.id, *[float(q) for q in im.qvec], *[float(t) for t in im.tvec], im.camera_id))
            fid.write(im.name.encode("utf-8") + b"\x00")
            fid.write(struct.pack("<Q", len(im.point3D_ids)))
            for (x, y), pid in zip(im.xys, im.point3D_ids):
                fid.write(struct.pack("<ddq", float(x), float(y), int(pid)))


def write_points3d_binary(points, path):
    with open(path, "wb") as fid:
        fid.write(struct.pack("<Q", len(points)))
        for pt in points.values():
            fid.write(struct.pack("<QdddBBBd", pt.id, *[float(v) for v in pt.xyz], *[int(c) for c in pt.rgb], float(pt.error)))
            fid.write(struct.pack("<Q", len(pt.image_ids)))
            for iid, idx in zip(pt.image_ids, pt.point2D_idxs):
                fid.write(struct.pack("<ii", int(iid), int(idx)))


# ---- text --------------------------------------------------------------------------------------------------------
def _records(path):
    with open(path, "r") as fid:
        for line in fid:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_cameras_text(path):
    cameras = {}
    for line in _records(path):
        el = line.split()
        cameras[int(el[0])] = Camera(id=int(el[0]), model=el[1], width=int(el[2]), height=int(el[3]), params=np.array([float(v) for v in el[4:]]))
    return cameras


def read_images_text(path):
    images = {}
    lines = []
    with open(path, "r") as fid:  # (the second line of an image may be empty: no observations)
        for line in fid:
            if not line.lstrip().startswith("#"):
                lines.append(line.strip())
    while lines and not lines[-1]:
        lines.pop()
    i = 0
    while i < len(lines):
        if not lines[i]:
            i += 1
            continue
        el = lines[i].split()
        obs = lines[i + 1].split() if i + 1 < len(lines) else []
        i += 2
        xys = np.column_stack([np.array(obs[0::3], dtype=np.float64), np.array(obs[1::3], dtype=np.float64)]) if obs else np.zeros((0, 2))
        images[int(el[0])] = Image(id=int(el[0]), qvec=np.array([float(v) for v in el[1:5]]), tvec=np.array([float(v) for v in el[5:8]]),
                                   camera_id=int(el[8]), name=el[9], xys=xys, point3D_ids=np.array([int(v) for v in obs[2::3]], dtype=np.int64))
    return images


def read_points3D_text(path):
    points = {}
    for line in _records(path):
        el = line.split()
        points[int(el[0])] = Point3D(id=int(el[0]), xyz=np.array([float(v) for v in el[1:4]]), rgb=np.array([int(v) for v in el[4:7]]),
                                     error=float(el[7]), image_ids=np.array([int(v) for v in el[8::2]], dtype=np.int64),
                                     point2D_idxs=np.array([int(v) for v in el[9::2]], dtype=np.int64))
    return points


def write_model_text(cameras, images, points, folder):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        for c in cameras.values():
            f.write(" ".join([str(c.id), c.model, str(int(c.width)), str(int(c.height))] + [repr(float(p)) for p in c.params]) + "\n")
    with open(os.path.join(folder, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n")
        for im in images.values():
            f.write(" ".join([str(im.id)] + [repr(float(v)) for v in im.qvec] + [repr(float(v)) for v in im.tvec] + [str(im.camera_id), im.name]) + "\n")
            f.write(" ".join("%r %r %d" % (float(x), float(y), int(p)) for (x, y), p in zip(im.xys, im.point3D_ids)) + "\n")
    with open(os.path.join(folder, "points3D.txt"), "w") as f:
        f.write("# 3D point list with one line of data per point:\n")
        for pt in points.values():
            track = " ".join("%d %d" % (int(a), int(b)) for a, b in zip(pt.image_ids, pt.point2D_idxs))
            f.write(" ".join([str(pt.id)] + [repr(float(v)) for v in pt.xyz] + [str(int(c)) for c in pt.rgb] + [repr(float(pt.error))]) + " " + track + "\n")


def read_model(path, ext):
    """(cameras, images, points3D) of the model in folder `path`; ext = ".bin" or ".txt" (colmap_read_model.py:260-269)"""
    if ext == ".txt":
        return (read_cameras_text(os.path.join(path, "cameras" + ext)), read_images_text(os.path.join(path, "images" + ext)),
                read_points3D_text(os.path.join(path, "points3D") + ext))
    return (read_cameras_binary(os.path.join(path, "cameras" + ext)), read_images_binary(os.path.join(path, "images" + ext)),
            read_points3d_binary(os.path.join(path, "points3D") + ext))


# ---- rotations ---------------------------------------------------------------------------------------------------
def qvec2rotmat(qvec):
    w, x, y, z = (float(v) for v in qvec)
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def rotmat2qvec(R):
    """unit quaternion (w, x, y, z), w >= 0, of a rotation matrix: dominant eigenvector of the symmetric 4x4 matrix built from R
    (Bar-Itzhack's form, as COLMAP does it)"""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = np.asarray(R, dtype=np.float64).flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0], [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0],
                  [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


# ---- poses and depth bounds of a model (what LLFF's imgs2poses writes into poses_bounds.npy) -------------------------------
def poses_from_model(realdir):
    """data_utils/poses/pose_utils.py:10-52 -> (poses [3,5,N]: camera-to-world in LLFF's [down, right, back] column order, then
    the (H, W, focal) column of the FIRST camera; the points3D dict; perm = image order sorted by file name)"""
    cams = read_cameras_binary(os.path.join(realdir, "sparse/0/cameras.bin"))
    cam = cams[list(cams.keys())[0]]
    hwf = np.array([cam.height, cam.width, cam.params[0]]).reshape([3, 1])
    imdata = read_images_binary(os.path.join(realdir, "sparse/0/images.bin"))
    perm = np.argsort([imdata[k].name for k in imdata])
    bottom = np.array([0, 0, 0, 1.0]).reshape([1, 4])
    w2c = np.stack([np.concatenate([np.concatenate([im.qvec2rotmat(), im.tvec.reshape([3, 1])], 1), bottom], 0) for im in imdata.values()], 0)
    c2w = np.linalg.inv(w2c)
    poses = c2w[:, :3, :4].transpose([1, 2, 0])
    poses = np.concatenate([poses, np.tile(hwf[..., np.newaxis], [1, 1, poses.shape[-1]])], 1)
    pts3d = read_points3d_binary(os.path.join(realdir, "sparse/0/points3D.bin"))
    # [r, -u, t] -> [-u, r, -t]
    poses = np.concatenate([poses[:, 1:2, :], poses[:, 0:1, :], -poses[:, 2:3, :], poses[:, 3:4, :], poses[:, 4:5, :]], 1)
    return poses, pts3d, perm


def poses_bounds(poses, pts3d, perm):
    """data_utils/poses/pose_utils.py:55-89 -> the [N, 17] array of poses_bounds.npy: 15 pose numbers + the 0.1 / 99.9 percentile
    depths of the points each image sees (a point's track lists 1-based image ids)"""
    n = poses.shape[-1]
    pts = np.array([pts3d[k].xyz for k in pts3d])
    vis = np.zeros((len(pts3d), n), dtype=np.int64)
    for r, k in enumerate(pts3d):
        for ind in pts3d[k].image_ids:
            if n < ind - 1:
                raise ValueError("a point's track names image %d of %d" % (ind, n))
            vis[r, ind - 1] = 1
    zvals = np.sum(-(pts[:, np.newaxis, :].transpose([2, 0, 1]) - poses[:3, 3:4, :]) * poses[:3, 2:3, :], 0)
    rows = []
    for i in perm:
        zs = zvals[:, i][vis[:, i] == 1]
        rows.append(np.concatenate([poses[..., i].ravel(), np.array([np.percentile(zs, 0.1), np.percentile(zs, 99.9)])], 0))
    return np.array(rows)


def save_poses(basedir, poses, pts3d, perm):
    arr = poses_bounds(poses, pts3d, perm)
    np.save(os.path.join(basedir, "poses_bounds.npy"), arr)
    return arr


# ---- a tiny synthetic model (fixtures / tests) ----------------------------------------------------------------------------
def synthetic_model(rng, n_images=5, n_points=40):
    """cameras on a rough arc looking at a point cloud; two camera models, unsorted image names, tracks of varying length"""
    cameras = collections.OrderedDict()
    cameras[3] = Camera(id=3, model="SIMPLE_RADIAL", width=640, height=480, params=np.array([525.5, 320.0, 240.0, 0.013]))
    cameras[7] = Camera(id=7, model="PINHOLE", width=320, height=200, params=np.array([300.25, 301.5, 160.0, 100.0]))
    names = ["img_%02d.png" % k for k in rng.permutation(n_images)]
    pts_xyz = rng.standard_normal((n_points, 3)) * 0.5 + np.array([0.0, 0.0, 4.0])
    images = collections.OrderedDict()
    obs = {i + 1: [] for i in range(n_images)}
    tracks = {}
    for p in range(n_points):
        seen = sorted(rng.choice(n_images, size=int(rng.integers(1, n_images + 1)), replace=False) + 1)
        tracks[p] = [(int(i), len(obs[int(i)])) for i in seen]
        for i in seen:
            obs[int(i)].append(p)
    for i in range(n_images):
        q = rng.standard_normal(4) * np.array([1.0, 0.2, 0.2, 0.2])
        q = q / np.linalg.norm(q)
        q = -q if q[0] < 0 else q
        tvec = rng.standard_normal(3) * 0.3
        ids = obs[i + 1]
        xys = rng.uniform(0, 300, (len(ids), 2))
        images[i + 1] = Image(id=i + 1, qvec=q, tvec=tvec, camera_id=3 if i % 2 == 0 else 7, name=names[i], xys=xys,
                              point3D_ids=np.array([100 + p for p in ids], dtype=np.int64))
    points = collections.OrderedDict()
    for p in range(n_points):
        points[100 + p] = Point3D(id=100 + p, xyz=pts_xyz[p], rgb=rng.integers(0, 256, 3), error=float(rng.random()),
                                  image_ids=np.array([t[0] for t in tracks[p]], dtype=np.int64), point2D_idxs=np.array([t[1] for t in tracks[p]], dtype=np.int64))
    return cameras, images, points
