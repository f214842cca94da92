"""ctypes/numpy front end of the CPU parity oracle (``libddnerf_oracle.so``).

TEST INFRASTRUCTURE ONLY -- see the header of ``ddnerf_oracle.c``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this package; the
product package ``ddnerf_amd`` never does.

``run_iter`` below chains the stage functions exactly the way the reference's
``DDNerfModel.predict`` / ``GeneralMipNerfModel.predict`` do (models/models.py:75-114, 207-322),
taking every random tensor as an explicit input.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libddnerf_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("ddnerf_oracle.c", "ddnerf_oracle_mlp.c", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None
_f = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ddo_dp_loss.restype = C.c_double
        _lib.ddo_mse.restype = C.c_double
        _lib.ddo_mse2psnr.restype = C.c_double
        _lib.ddo_aten_sum.restype = C.c_float
        _lib.ddo_erfinv.restype = C.c_float
        _lib.ddo_get_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_f)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n: int):
    lib().ddo_set_threads(C.c_int(n))


def get_threads() -> int:
    return lib().ddo_get_threads()


def aten_sum(x):
    x = _c(x)
    return np.array([lib().ddo_aten_sum(_p(r), C.c_int(r.shape[0])) for r in x.reshape(-1, x.shape[-1])],
                    np.float32).reshape(x.shape[:-1])


def aten_cumsum(x, prod=False):
    x = _c(x)
    out = np.empty_like(x)
    fn = lib().ddo_aten_cumprod if prod else lib().ddo_aten_cumsum
    for r, o in zip(x.reshape(-1, x.shape[-1]), out.reshape(-1, x.shape[-1])):
        fn(_p(r), _p(o), C.c_int(r.shape[0]))
    return out


def pack_rays(ro, rd, rad, near, far):
    ro, rd, rad = _c(ro).reshape(-1, 3), _c(rd).reshape(-1, 3), _c(rad).reshape(-1)
    n = ro.shape[0]
    rays = np.empty((n, 12), np.float32)
    lib().ddo_pack_rays(_p(ro), _p(rd), _p(rad), C.c_float(near), C.c_float(far), _p(rays), C.c_int(n))
    return rays


def sample_first_cycle(rays, t_lin, t_rand, lindisp=False):
    rays, t_lin, t_rand = _c(rays), _c(t_lin), _c(t_rand)
    n, nc = rays.shape[0], t_lin.shape[0] - 1
    t = np.empty((n, nc + 1), np.float32)
    lib().ddo_sample_first_cycle(_p(rays), _p(t_lin), _p(t_rand), _p(t), C.c_int(n), C.c_int(nc), C.c_int(int(lindisp)))
    return t


def cast_rays(rays, t_vals, cylinder=False):
    rays, t_vals = _c(rays), _c(t_vals)
    n, S = t_vals.shape[0], t_vals.shape[1] - 1
    means = np.empty((n, S, 3), np.float32)
    covs = np.empty((n, S, 3), np.float32)
    lib().ddo_cast_rays(_p(rays), _p(t_vals), _p(means), _p(covs), C.c_int(n), C.c_int(S), C.c_int(int(cylinder)))
    return means, covs


def ipe(means, covs):
    means, covs = _c(means), _c(covs)
    out = np.empty(means.shape[:-1] + (96,), np.float32)
    lib().ddo_ipe(_p(means), _p(covs), _p(out), C.c_int(int(np.prod(means.shape[:-1]))))
    return out


def dir_enc(viewdirs):
    v = _c(viewdirs)
    out = np.empty((v.shape[0], 27), np.float32)
    lib().ddo_dir_enc(_p(v), _p(out), C.c_int(v.shape[0]))
    return out


def encode(rays, t_vals, cylinder=False, ld=123):
    rays, t_vals = _c(rays), _c(t_vals)
    n, S = t_vals.shape[0], t_vals.shape[1] - 1
    feat = np.empty((n * S, ld), np.float32)
    lib().ddo_encode(_p(rays), _p(t_vals), _p(feat), C.c_int(n), C.c_int(S), C.c_int(int(cylinder)), C.c_int(ld))
    return feat


_LAYER_NAMES = (["layers_xyz.%d" % i for i in range(8)] + ["fc_feat", "fc_alpha", "layers_dir.0", "fc_rgb", "fc_mu_sigma"])


def mlp_forward(feat, state_dict, depth_head):
    feat = _c(feat)
    names = _LAYER_NAMES[: 13 if depth_head else 12]
    keep = [_c(state_dict[n + s]) for n in names for s in (".weight", ".bias")]
    arr = (_f * len(keep))(*[_p(k) for k in keep])
    M = feat.shape[0]
    out = np.empty((M, 6 if depth_head else 4), np.float32)
    lib().ddo_mlp_forward(_p(feat), C.c_int(feat.shape[1]), arr, _p(out), C.c_long(M), C.c_int(int(depth_head)))
    return out


def dd_head(raw6, smooth, dist_reg):
    raw6 = _c(raw6)
    n, nc = raw6.shape[:2]
    outs = [np.empty((n, nc), np.float32) for _ in range(7)]
    scal = np.zeros(4, np.float64)
    lib().ddo_dd_head(_p(raw6), C.c_int(n), C.c_int(nc), C.c_float(smooth), C.c_float(dist_reg), *[_p(o) for o in outs],
                      scal.ctypes.data_as(C.POINTER(C.c_double)))
    keys = ("mus", "sigmas", "left", "part", "ssig", "sleft", "spart")
    d = dict(zip(keys, outs))
    d.update(mus_loss=scal[0], sig_loss=scal[1], mus_reg=scal[2], sig_reg=scal[3])
    return d


def composite(raw, t_vals, rays, noise=None, mus=None, white_bkgd=False, blender=True):
    raw, t_vals, rays, noise, mus = _c(raw), _c(t_vals), _c(rays), _c(noise), _c(mus)
    n, S, ldr = raw.shape
    o = dict(rgb_map=np.empty((n, 3), np.float32), disp=np.empty(n, np.float32), acc=np.empty(n, np.float32),
             weights=np.empty((n, S), np.float32), depth=np.empty(n, np.float32),
             cdisp=np.empty(n, np.float32) if mus is not None else None, rgb=np.empty((n, S, 3), np.float32))
    lib().ddo_composite(_p(raw), C.c_int(ldr), _p(t_vals), _p(rays), _p(noise), _p(mus), C.c_int(n), C.c_int(S),
                        C.c_int(int(white_bkgd)), C.c_int(int(blender)), _p(o["rgb_map"]), _p(o["disp"]), _p(o["acc"]),
                        _p(o["weights"]), _p(o["depth"]), _p(o["cdisp"]), _p(o["rgb"]))
    return o


def sample_pdf(bins, weights, u_base, rnd, pdf_padding):
    bins, weights, u_base, rnd = _c(bins), _c(weights), _c(u_base), _c(rnd)
    n, nc = weights.shape
    ns = u_base.shape[0]
    out = np.empty((n, ns), np.float32)
    lib().ddo_sample_pdf(_p(bins), _p(weights), _p(u_base), _p(rnd), C.c_float(np.float32(ns + 1e-5)), _p(out), C.c_int(n),
                         C.c_int(nc), C.c_int(ns), C.c_int(int(pdf_padding)))
    return out


def sample_pdf_mu_sigma(bins, weights, mus, sigmas, part, left, u_base, rnd, near, far, pdf_padding):
    bins, weights, mus, sigmas, part, left, u_base, rnd = map(_c, (bins, weights, mus, sigmas, part, left, u_base, rnd))
    n, nc = weights.shape
    ns = u_base.shape[0]
    out = np.empty((n, ns), np.float32)
    ind = np.empty((n, ns), np.int32)
    lib().ddo_sample_pdf_mu_sigma(_p(bins), _p(weights), _p(mus), _p(sigmas), _p(part), _p(left), _p(u_base), _p(rnd),
                                  C.c_float(np.float32(ns + 1e-5)), C.c_float(near), C.c_float(far), _p(out),
                                  ind.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(n), C.c_int(nc), C.c_int(ns),
                                  C.c_int(int(pdf_padding)))
    return out, ind


def dp_loss(t1, t0, w1, w0, mus0, sig0, left0, part0, blender):
    t1, t0, w1, w0, mus0, sig0, left0, part0 = map(_c, (t1, t0, w1, w0, mus0, sig0, left0, part0))
    n, nc = w0.shape
    nf = w1.shape[1]
    rows = C.c_int(0)
    v = lib().ddo_dp_loss(_p(t1), _p(t0), _p(w1), _p(w0), _p(mus0), _p(sig0), _p(left0), _p(part0), C.c_int(n),
                          C.c_int(nc), C.c_int(nf), C.c_int(int(blender)), C.byref(rows))
    return v, rows.value


def mse(a, b):
    a, b = _c(a), _c(b)
    return lib().ddo_mse(_p(a), _p(b), C.c_size_t(a.size))


def mse2psnr(m):
    return lib().ddo_mse2psnr(C.c_double(m))


# ------------------------------------------------------------------------------------------------
# whole-path restatement (models/models.py:40-73 run_iter, :75-114 / :207-322 predict) for ONE chunk
# ------------------------------------------------------------------------------------------------
def u_base(ns, det, dd):
    """Host-side `u` rows exactly as the reference builds them with torch (float32 semantics).
    det: linspace(0, 0.9999|1, ns) must come from torch; callers pass it in.  non-det: arange*s."""
    s = 1.0 / (ns - 1) if dd else 1.0 / ns
    return (np.arange(ns, dtype=np.float32) * np.float32(s)).astype(np.float32)


def run_iter(ro, rd, rad, sd_coarse, sd_fine, *, model="dd", nc, nf, near, far, blender, white_bkgd=False,
             lindisp=False, cylinder=False, pdf_padding=True, smooth=1.7, dist_reg=0.02, t_lin=None, t_rand=None,
             noise0=None, u_det=None, u_rand=None, noise1=None, want_dp_loss=True):
    """One ray chunk through coarse + fine.  Random/linspace tensors are explicit inputs:
    t_lin = torch.linspace(0,1,nc+1); t_rand = rand(n,nc+1)|None; noise0/1 = randn*std|None;
    u_det = torch.linspace(0, 0.9999 (dd) | 1.0 (mip), nf+1) when perturb is off, else u_rand = rand(n,nf+1)."""
    dd = model == "dd"
    rays = pack_rays(ro, rd, rad, near, far)
    t0 = sample_first_cycle(rays, t_lin, t_rand, lindisp)
    feat0 = encode(rays, t0, cylinder)
    raw0 = mlp_forward(feat0, sd_coarse, dd).reshape(rays.shape[0], nc, -1)
    out = {}
    head = None
    if dd:
        head = dd_head(raw0, smooth, dist_reg)
    c0 = composite(raw0, t0, rays, noise0, head["mus"] if dd else None, white_bkgd, blender)
    out[0] = dict(rgb=c0["rgb_map"], disp=c0["disp"], acc=c0["acc"], weights=c0["weights"], depth=c0["depth"],
                  t_vals=t0, raw=raw0)
    ns = nf + 1
    det = u_rand is None
    ub = u_det if det else u_base(ns, det, dd)
    if dd:
        out[0]["corrected_disp_map"] = c0["cdisp"]
        out[0].update({k: head[k] for k in ("mus", "sigmas", "left", "part", "ssig", "sleft", "spart", "mus_loss",
                                            "sig_loss", "mus_reg", "sig_reg")})
        t1, ind = sample_pdf_mu_sigma(t0, c0["weights"], head["mus"], head["ssig"], head["spart"], head["sleft"], ub,
                                      u_rand, near, far, pdf_padding)
        out[1] = dict(bins_ind=ind)
    else:
        t1 = sample_pdf(t0, c0["weights"], ub, u_rand, pdf_padding)
        out[1] = {}
    feat1 = encode(rays, t1, cylinder)
    raw1 = mlp_forward(feat1, sd_fine if dd else sd_coarse, False).reshape(rays.shape[0], nf, -1)
    c1 = composite(raw1, t1, rays, noise1, None, white_bkgd, blender)
    out[1].update(rgb=c1["rgb_map"], disp=c1["disp"], acc=c1["acc"], weights=c1["weights"], depth=c1["depth"],
                  t_vals=t1, raw=raw1)
    if dd and want_dp_loss:
        v, rows = dp_loss(t1, t0, c1["weights"], c0["weights"], head["mus"], head["sigmas"], head["left"], head["part"],
                          blender)
        # models/models.py:288-289: *(nf) + mus_reg + sig_reg   (integer 0 if every row was filtered)
        out[1]["dp_loss"] = np.float32(np.float32(v) * np.float32(nf)) + np.float32(head["mus_reg"]) + np.float32(head["sig_reg"])
        out[1]["dp_rows"] = rows
    return out
