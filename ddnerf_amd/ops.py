"""Tensor-level wrappers of the HIP kernels (one function per C-ABI entry point).

Every function takes/returns torch tensors that live on the GPU, allocates outputs with torch, and
enqueues the kernel on torch's current stream.  No function has a CPU path."""
from __future__ import annotations

import os

import torch

from . import _lib

FEAT_LD = 128
# optional instrumentation: callable(M, launch) -> launch(); bench.py installs a HIP-event timer here
MLP_LAUNCH_HOOK = None


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.DDNerfHipError("%s must be a GPU tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def pack_rays(origins, directions, radii, near, far):
    """models/models.py:144-162 -> rays [n,12]"""
    o = _f32c(origins, "origins").reshape(-1, 3)
    d = _f32c(directions, "directions").reshape(-1, 3)
    r = _f32c(radii, "radii").reshape(-1)
    n = o.shape[0]
    rays = torch.empty((n, 12), dtype=torch.float32, device=o.device)
    _lib.check(_lib.lib().ddnerf_pack_rays(_ptr(o), _ptr(d), _ptr(r), float(near), float(far), _ptr(rays), n, _stream()),
               "ddnerf_pack_rays")
    return rays


def sample_first_cycle(rays, t_lin, t_rand=None, lindisp=False):
    """models/samplers.py:30-62 -> t_vals [n,nc+1]; lindisp=2: t_lin holds the absolute depths of get_combined_samples"""
    rays, t_lin, t_rand = _f32c(rays, "rays"), _f32c(t_lin, "t_lin"), _f32c(t_rand, "t_rand")
    n, nc = rays.shape[0], t_lin.shape[0] - 1
    t = torch.empty((n, nc + 1), dtype=torch.float32, device=rays.device)
    _lib.check(_lib.lib().ddnerf_sample_first_cycle(_ptr(rays), _ptr(t_lin), _ptr(t_rand), _ptr(t), n, nc, int(lindisp),
                                                    _stream()), "ddnerf_sample_first_cycle")
    return t


FEAT_KINDS = {"fp32": (0, torch.float32), "bf16": (1, torch.bfloat16), "fp16": (2, torch.float16)}


def encode(rays, t_vals, cylinder=False, bf16=False, kind=None):
    """cast_rays + integrated_pos_enc + view-dir encoding -> feat [n*S,128]: fp32 (natural column order), or bf16 / fp16 rows in the
    MFMA k-order of the bf16 / fp16 MLP kernels (kind "fp32" | "bf16" | "fp16"; bf16=True is kind "bf16")"""
    rays, t_vals = _f32c(rays, "rays"), _f32c(t_vals, "t_vals")
    n, S = t_vals.shape[0], t_vals.shape[1] - 1
    code, dtype = FEAT_KINDS[kind or ("bf16" if bf16 else "fp32")]
    feat = torch.empty((n * S, FEAT_LD), dtype=dtype, device=rays.device)
    _lib.check(_lib.lib().ddnerf_encode(_ptr(rays), _ptr(t_vals), _ptr(feat), n, S, int(cylinder), code, _stream()),
               "ddnerf_encode")
    return feat


def encode_rays(rays, t_vals, cylinder=False):
    """encode with the view-direction columns once per RAY (fp32 rows) -> (feat [n*S,128], columns 96..127 NOT written; dirs [n,32])"""
    rays, t_vals = _f32c(rays, "rays"), _f32c(t_vals, "t_vals")
    n, S = t_vals.shape[0], t_vals.shape[1] - 1
    feat = torch.empty((n * S, FEAT_LD), dtype=torch.float32, device=rays.device)
    dirs = torch.empty((n, 32), dtype=torch.float32, device=rays.device)
    _lib.check(_lib.lib().ddnerf_encode_rays(_ptr(rays), _ptr(t_vals), _ptr(feat), _ptr(dirs), n, S, int(cylinder), _stream()), "ddnerf_encode_rays")
    return feat, dirs


def encode_first_cycle_rays(ray_origins, ray_directions, ray_rad, near, far, t_lin, lindisp=False, cylinder=False, out=None):
    """encode_first_cycle (fp32 rows) with the view-direction columns once per ray -> (rays, t_vals, feat, dirs)"""
    o = _f32c(ray_origins.reshape(-1, 3), "origins")
    d = _f32c(ray_directions.reshape(-1, 3), "directions")
    r = _f32c(ray_rad.reshape(-1), "radii")
    t_lin = _f32c(t_lin, "t_lin")
    n, nc = o.shape[0], t_lin.shape[0] - 1
    if out is None:
        out = (torch.empty((n, 12), dtype=torch.float32, device=o.device), torch.empty((n, nc + 1), dtype=torch.float32, device=o.device))
    rays, t_vals = out
    feat = torch.empty((n * nc, FEAT_LD), dtype=torch.float32, device=o.device)
    dirs = torch.empty((n, 32), dtype=torch.float32, device=o.device)
    _lib.check(_lib.lib().ddnerf_encode_first_cycle_rays(_ptr(o), _ptr(d), _ptr(r), float(near), float(far), _ptr(t_lin), int(lindisp), _ptr(rays),
                                                         _ptr(t_vals), _ptr(feat), _ptr(dirs), n, nc, int(cylinder), _stream()),
               "ddnerf_encode_first_cycle_rays")
    return rays, t_vals, feat, dirs


def mlp_rays_supported(S, M):
    return S > 1 and M % S == 0 and M * S < (1 << 32)


def mlp_f32_forward_rays(feat, dirs, S, packed, depth_head):
    """mlp_f32_forward with the view-direction columns from the per-ray table of encode_rays"""
    feat, dirs = _f32c(feat, "feat"), _f32c(dirs, "dirs")
    M = feat.shape[0]
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)

    def launch():
        _lib.check(_lib.lib().ddnerf_mlp_f32_forward_rays(_ptr(feat), _ptr(dirs), int(S), _ptr(packed), int(depth_head), _ptr(raw), M, _stream()),
                   "ddnerf_mlp_f32_forward_rays")

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw


def mlp_x3_forward_rays(feat, dirs, S, packed, depth_head):
    """mlp_x3_forward with the view-direction columns from the per-ray table of encode_rays"""
    feat, dirs = _f32c(feat, "feat"), _f32c(dirs, "dirs")
    M = feat.shape[0]
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)

    def launch():
        _lib.check(_lib.lib().ddnerf_mlp_x3_forward_rays(_ptr(feat), _ptr(dirs), int(S), _ptr(packed), int(depth_head), _ptr(raw), M, _stream()),
                   "ddnerf_mlp_x3_forward_rays")

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw


def mlp_f32_pack(params_flat, depth_head):
    params_flat = _f32c(params_flat, "params")
    nfl = _lib.lib().ddnerf_mlp_f32_packed_floats(int(depth_head))
    packed = torch.empty(nfl, dtype=torch.float32, device=params_flat.device)
    _lib.check(_lib.lib().ddnerf_mlp_f32_pack(_ptr(params_flat), int(depth_head), _ptr(packed), _stream()),
               "ddnerf_mlp_f32_pack")
    return packed


def mlp_f32_forward(feat, packed, depth_head):
    """models/base_architectures.py:40-61 / 103-126: feat [M,128] -> raw [M,4|6]"""
    feat = _f32c(feat, "feat")
    M = feat.shape[0]
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)

    def launch():
        _lib.check(_lib.lib().ddnerf_mlp_f32_forward(_ptr(feat), _ptr(packed), int(depth_head), _ptr(raw), M, _stream()),
                   "ddnerf_mlp_f32_forward")

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw


def mlp_bf16_pack(params_flat, depth_head):
    params_flat = _f32c(params_flat, "params")
    nbytes = _lib.lib().ddnerf_mlp_bf16_packed_bytes(int(depth_head))
    packed = torch.empty(nbytes, dtype=torch.uint8, device=params_flat.device)
    _lib.check(_lib.lib().ddnerf_mlp_bf16_pack(_ptr(params_flat), int(depth_head), _ptr(packed), _stream()),
               "ddnerf_mlp_bf16_pack")
    return packed


def mlp_bf16_forward(feat, packed, depth_head):
    """bf16-MFMA MLP: feat bf16 [M,128] in k-order (as written by encode(bf16=True)) -> raw fp32 [M,4|6]"""
    if not (feat.is_cuda and feat.dtype == torch.bfloat16 and feat.is_contiguous()):
        raise _lib.DDNerfHipError("mlp_bf16_forward wants a contiguous bf16 GPU feature tensor")
    M = feat.shape[0]
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)

    def launch():
        _lib.check(_lib.lib().ddnerf_mlp_bf16_forward(_ptr(feat), _ptr(packed), int(depth_head), _ptr(raw), M, _stream()),
                   "ddnerf_mlp_bf16_forward")

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw


def _bf16_variant_pack(which, params_flat, depth_head):
    params_flat = _f32c(params_flat, "params")
    L = _lib.lib()
    packed = torch.empty(getattr(L, "ddnerf_mlp_%s_packed_bytes" % which)(int(depth_head)), dtype=torch.uint8, device=params_flat.device)
    _lib.check(getattr(L, "ddnerf_mlp_%s_pack" % which)(_ptr(params_flat), int(depth_head), _ptr(packed), _stream()), "ddnerf_mlp_%s_pack" % which)
    return packed


def _bf16_variant_forward(which, feat, packed, depth_head):
    want = torch.float16 if which.startswith("f16") else torch.bfloat16
    if not (feat.is_cuda and feat.dtype == want and feat.is_contiguous()):
        raise _lib.DDNerfHipError("mlp_%s_forward wants a contiguous %s GPU feature tensor" % (which, want))
    M = feat.shape[0]
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)

    def launch():
        _lib.check(getattr(_lib.lib(), "ddnerf_mlp_%s_forward" % which)(_ptr(feat), _ptr(packed), int(depth_head), _ptr(raw), M, _stream()),
                   "ddnerf_mlp_%s_forward" % which)

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw


def train_loss_forward(rgb0, rgb1, target, dp, c0, c1, c_dp):
    """train_model.py:156-172 in one launch -> out [4] = (loss, mse0, mse1, mean dp)"""
    rgb0, rgb1, target, dp = _f32c(rgb0, "rgb0"), _f32c(rgb1, "rgb1"), _f32c(target, "target"), _f32c(dp, "dp")
    out = torch.empty(4, dtype=torch.float32, device=rgb0.device)
    _lib.check(_lib.lib().ddnerf_train_loss_forward(_ptr(rgb0), _ptr(rgb1), _ptr(target), rgb0.numel(), _ptr(dp), 0 if dp is None else dp.numel(),
                                                    float(c0), float(c1), float(c_dp), _ptr(out), _stream()), "ddnerf_train_loss_forward")
    return out


def train_loss_backward(rgb0, rgb1, target, n_dp, c0, c1, c_dp, g):
    rgb0, rgb1, target, g = _f32c(rgb0, "rgb0"), _f32c(rgb1, "rgb1"), _f32c(target, "target"), _f32c(g, "g")
    g0 = torch.empty_like(rgb0)
    g1 = None if rgb1 is None else torch.empty_like(rgb1)
    gd = torch.empty(n_dp, dtype=torch.float32, device=rgb0.device) if n_dp else None
    _lib.check(_lib.lib().ddnerf_train_loss_backward(_ptr(rgb0), _ptr(rgb1), _ptr(target), rgb0.numel(), int(n_dp), float(c0), float(c1), float(c_dp),
                                                     _ptr(g), _ptr(g0), _ptr(g1), _ptr(gd), _stream()), "ddnerf_train_loss_backward")
    return g0, g1, gd


def ray_table(rays, kind="bf16"):
    """rays [n,12] -> the fused kernels' per-ray table [n,32] (fp32 words: 16 floats, then the ray's view-direction row as 32 bf16 / fp16)"""
    rays = _f32c(rays, "rays")
    n = rays.shape[0]
    table = torch.empty((n, 32), dtype=torch.float32, device=rays.device)
    _lib.check(_lib.lib().ddnerf_ray_table(_ptr(rays), n, FEAT_KINDS[kind][0], _ptr(table), _stream()), "ddnerf_ray_table")
    return table


def pack_rays_first_cycle_table(ray_origins, ray_directions, ray_rad, near, far, t_lin, lindisp=False, out=None, kind="bf16"):
    """pack_rays_first_cycle (no jitter) + ray_table in ONE launch -> (rays [n,12], t_vals [n,nc+1], table [n,32]); `out` = (rays, t_vals)
    tensors to fill (handed out earlier by GeneralMipNerfModel.get_rays_batches)"""
    o = _f32c(ray_origins.reshape(-1, 3), "origins")
    d = _f32c(ray_directions.reshape(-1, 3), "directions")
    r = _f32c(ray_rad.reshape(-1), "radii")
    t_lin = _f32c(t_lin, "t_lin")
    n, nc = o.shape[0], t_lin.shape[0] - 1
    if out is None:
        out = (torch.empty((n, 12), dtype=torch.float32, device=o.device), torch.empty((n, nc + 1), dtype=torch.float32, device=o.device))
    rays, t_vals = out
    table = torch.empty((n, 32), dtype=torch.float32, device=o.device)
    _lib.check(_lib.lib().ddnerf_pack_rays_first_cycle_table(_ptr(o), _ptr(d), _ptr(r), float(near), float(far), _ptr(t_lin), _ptr(rays),
                                                             _ptr(t_vals), FEAT_KINDS[kind][0], _ptr(table), n, nc, int(lindisp), _stream()),
               "ddnerf_pack_rays_first_cycle_table")
    return rays, t_vals, table


_FUSED_SCRATCH = {}


def encode_mlp_bf16_supported(S, M, cylinder=False):
    return (not cylinder) and S % 64 == 0 and M <= (1 << 22)


def encode_mlp_bf16_forward(table, t_vals, packed, depth_head, kind="bf16"):
    """models/models.py:117-142 as one launch (bf16 tier; kind="fp16": the fp16 tier's twin): cast_rays + integrated_pos_enc + view directions +
    the MLP; t_vals [n,S+1] -> raw [n*S, 4|6], bit for bit encode(kind=kind) + mlp_bf16_forward / mlp_f16_forward"""
    t_vals = _f32c(t_vals, "t_vals")
    n, S = t_vals.shape[0], t_vals.shape[1] - 1
    M = n * S
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=t_vals.device)
    key = (t_vals.device.index, torch.cuda.current_stream().cuda_stream)
    scratch = _FUSED_SCRATCH.get(key)
    if scratch is None:     # (one area per device and stream: launches on one stream run one after the other)
        scratch = _FUSED_SCRATCH[key] = torch.empty(_lib.lib().ddnerf_encode_mlp_bf16_scratch_bytes(), dtype=torch.uint8, device=t_vals.device)

    def launch():
        fn = "ddnerf_encode_mlp_%s_forward" % {"bf16": "bf16", "fp16": "f16"}[kind]
        _lib.check(getattr(_lib.lib(), fn)(_ptr(table), _ptr(t_vals), _ptr(packed), int(depth_head), _ptr(raw), n, S, _ptr(scratch), _stream()), fn)

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw


# The two kernels behind mlp_bf16_forward by name (it picks by launch size; they produce the same bits for the same sample):
# "g1" a workgroup owns 256 samples (mlp_bf16.hip), "g2" 512 samples as two groups per wave (mlp_bf16_g2.hip); each has its own image.
def mlp_bf16g1_pack(params_flat, depth_head):
    return _bf16_variant_pack("bf16g1", params_flat, depth_head)


def mlp_bf16g1_forward(feat, packed, depth_head):
    return _bf16_variant_forward("bf16g1", feat, packed, depth_head)


def mlp_bf16g2_pack(params_flat, depth_head):
    return _bf16_variant_pack("bf16g2", params_flat, depth_head)


def mlp_bf16g2_forward(feat, packed, depth_head):
    return _bf16_variant_forward("bf16g2", feat, packed, depth_head)


# The fp16 tier: the same two kernels on the fp16 forms of the MFMA and of the re-pack conversion (feat: fp16 rows in k-order, as written
# by encode(kind="fp16")); "f16" picks between them by launch size like "bf16".
def mlp_f16_pack(params_flat, depth_head):
    return _bf16_variant_pack("f16", params_flat, depth_head)


def mlp_f16_forward(feat, packed, depth_head):
    return _bf16_variant_forward("f16", feat, packed, depth_head)


def mlp_f16g1_pack(params_flat, depth_head):
    return _bf16_variant_pack("f16g1", params_flat, depth_head)


def mlp_f16g1_forward(feat, packed, depth_head):
    return _bf16_variant_forward("f16g1", feat, packed, depth_head)


def mlp_f16g2_pack(params_flat, depth_head):
    return _bf16_variant_pack("f16g2", params_flat, depth_head)


def mlp_f16g2_forward(feat, packed, depth_head):
    return _bf16_variant_forward("f16g2", feat, packed, depth_head)


def mlp_x3_pack(params_flat, depth_head):
    params_flat = _f32c(params_flat, "params")
    nbytes = _lib.lib().ddnerf_mlp_x3_packed_bytes(int(depth_head))
    packed = torch.empty(nbytes, dtype=torch.uint8, device=params_flat.device)
    _lib.check(_lib.lib().ddnerf_mlp_x3_pack(_ptr(params_flat), int(depth_head), _ptr(packed), _stream()), "ddnerf_mlp_x3_pack")
    return packed


def mlp_x3_forward(feat, packed, depth_head):
    """the MLP on the bf16 matrix cores with exact hi/lo operand splits (fp32-class accuracy): feat fp32 [M,128] -> raw"""
    feat = _f32c(feat, "feat")
    M = feat.shape[0]
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)

    def launch():
        _lib.check(_lib.lib().ddnerf_mlp_x3_forward(_ptr(feat), _ptr(packed), int(depth_head), _ptr(raw), M, _stream()),
                   "ddnerf_mlp_x3_forward")

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw


# packed position -> original feature column of the bf16 kernel's "k-order" (csrc/mlp_bf16_common.h korder32)
K_ORDER = [(p & ~31) | (16 * ((p >> 2) & 1) + 4 * ((p >> 3) & 3) + (p & 3)) for p in range(FEAT_LD)]


def dd_head(raw6, smooth, dist_reg):
    """models/models.py:242-260, 266-273.  raw6 [n,nc,6]"""
    raw6 = _f32c(raw6, "raw6")
    n, nc = raw6.shape[0], raw6.shape[1]
    dev = raw6.device
    outs = [torch.empty((n, nc), dtype=torch.float32, device=dev) for _ in range(7)]
    scal = torch.empty(4, dtype=torch.float32, device=dev)
    ws = torch.empty(max(1, _lib.lib().ddnerf_dd_head_workspace_floats(n, nc)), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().ddnerf_dd_head(_ptr(raw6), n, nc, float(smooth), float(dist_reg), *[_ptr(o) for o in outs],
                                         _ptr(scal), _ptr(ws), _stream()), "ddnerf_dd_head")
    d = dict(zip(("mus", "sigmas", "left", "part", "ssig", "sleft", "spart"), outs))
    d["scal"] = scal  # mus_loss, sig_loss, mus_reg, sig_reg
    return d


_PINNED = {}


def dd_records_launch(weights, mus, sigmas, ssig):
    """models/models.py:292-295, first half: enqueue the stream compaction of (mus, sigmas, smoothed_sigmas) over the bins with
    pdf = w / sum(w) > 0.1 and an asynchronous copy of the data-dependent length into pinned host memory.  Returns a
    ticket for dd_records_finish.  Enqueue it as EARLY as its inputs exist: the host then learns the length while the GPU
    still works on what was enqueued behind it, instead of draining the queue at the end of the chunk."""
    weights, mus, sigmas, ssig = (_f32c(t, "records") for t in (weights, mus, sigmas, ssig))
    n, nc = weights.shape
    dev = weights.device
    outs = [torch.empty(n * nc, dtype=torch.float32, device=dev) for _ in range(3)]
    total = _records_slot(dev)
    ws = torch.empty(_lib.lib().ddnerf_dd_records_workspace_bytes(n, nc), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().ddnerf_dd_records(_ptr(weights), _ptr(mus), _ptr(sigmas), _ptr(ssig), n, nc, *[_ptr(o) for o in outs],
                                            _ptr(total), _ptr(ws), _stream()), "ddnerf_dd_records")
    return _records_ticket(outs, total)


def _records_slot(dev):
    """where the compaction kernel writes the records' length: one int32 of PINNED host memory (the device writes it in place --
    pinned memory is mapped into the device's address space -- so no copy is enqueued behind the kernel)"""
    # a free list: a slot is taken by a launch and handed back by the finish that read it, so any number of chunks may be in flight
    # (run_iter over a whole image enqueues all of them before it reads the first length)
    free = _PINNED.setdefault(str(dev), [])
    slot = free.pop() if free else torch.empty(1, dtype=torch.int32).pin_memory()
    slot[0] = -1          # (a host store: the kernel's store of the length, >= 0, is what dd_records_finish waits for)
    return slot


# Round 5: NO event behind the compaction.  The host learns the length by reading the pinned slot until the device's store shows (pinned
# memory is coherent: a device write becomes visible to the CPU without a flush).  An event recorded in the stream was a barrier packet
# between the compaction and the fine pass's MLP launch: 11 us of an empty GPU per 0.69-ms chunk (profiles/r05_render_bf16_kernel_stats:
# the gap in front of every fine launch).  DDNERF_RECORDS_EVENT=1: the event, as before.
RECORDS_EVENT = os.environ.get("DDNERF_RECORDS_EVENT", "0") == "1"


def _records_ticket(outs, host):
    """what dd_records_finish waits for: the pinned slot itself (or, DDNERF_RECORDS_EVENT=1, an event recorded right behind the compaction)"""
    ev = None
    if RECORDS_EVENT:
        ev = torch.cuda.Event()
        ev.record()
    return [outs, host, ev, torch.cuda.current_stream()]     # (a list: dd_records_finish empties it -- a ticket is good for ONE finish)


def dd_records_finish(ticket):
    """second half: wait for the length (the ONE host sync of a chunk, on an event recorded right behind the compaction) and
    slice the three records"""
    if not ticket:
        # finishing a ticket twice would hand its pinned length slot to the free list twice: two later launches would then share it
        raise _lib.DDNerfHipError("dd_records_finish: this ticket has been finished already")
    outs, host, ev, stream = ticket
    del ticket[:]
    if ev is not None:
        ev.synchronize()
    else:
        import time

        t_end = time.monotonic() + 5.0
        while int(host[0]) < 0:
            if time.monotonic() > t_end:      # (a launch that never ran: let the stream report it)
                stream.synchronize()
                if int(host[0]) < 0:
                    raise _lib.DDNerfHipError("dd_records_finish: the compaction kernel did not deliver the records' length")
    k = int(host[0])
    _PINNED.setdefault(str(outs[0].device), []).append(host)
    return tuple(o[:k] for o in outs)


def dd_records(weights, mus, sigmas, ssig):
    """models/models.py:292-295 -> (mus, sigmas, smoothed_sigmas) of the bins with pdf = w / sum(w) > 0.1, flat row-major"""
    return dd_records_finish(dd_records_launch(weights, mus, sigmas, ssig))


def composite_forward(raw, t_vals, rays, noise=None, mus=None, white_bkgd=False, blender=True, want_rgb=False):
    """general_utils/volume_rendering_utils.py:6-85.  raw [n,S,4|6]"""
    raw, t_vals, rays = _f32c(raw, "raw"), _f32c(t_vals, "t_vals"), _f32c(rays, "rays")
    noise, mus = _f32c(noise, "noise"), _f32c(mus, "mus")
    n, S, ldr = raw.shape
    dev = raw.device
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    o = dict(rgb_map=e(n, 3), disp=e(n), acc=e(n), weights=e(n, S), depth=e(n), cdisp=e(n) if mus is not None else None,
             rgb=e(n, S, 3) if want_rgb else None)
    flags = (1 if white_bkgd else 0) | (2 if blender else 0)
    _lib.check(_lib.lib().ddnerf_composite_forward(_ptr(raw), ldr, _ptr(t_vals), _ptr(rays), _ptr(noise), _ptr(mus), n, S,
                                                   flags, _ptr(o["rgb_map"]), _ptr(o["disp"]), _ptr(o["acc"]),
                                                   _ptr(o["weights"]), _ptr(o["depth"]), _ptr(o["cdisp"]), _ptr(o["rgb"]),
                                                   _stream()), "ddnerf_composite_forward")
    return o


def pack_rays_first_cycle(ray_origins, ray_directions, ray_rad, near, far, t_lin, t_rand=None, lindisp=False):
    """models/models.py:144-162 + models/samplers.py:30-62 in one launch (a ray batch that is one chunk) -> rays [n,12], t_vals [n,nc+1]"""
    o = _f32c(ray_origins.reshape(-1, 3), "origins")
    d = _f32c(ray_directions.reshape(-1, 3), "directions")
    r = _f32c(ray_rad.reshape(-1), "radii")
    t_lin, t_rand = _f32c(t_lin, "t_lin"), _f32c(t_rand, "t_rand")
    n, nc = o.shape[0], t_lin.shape[0] - 1
    rays = torch.empty((n, 12), dtype=torch.float32, device=o.device)
    t_vals = torch.empty((n, nc + 1), dtype=torch.float32, device=o.device)
    _lib.check(_lib.lib().ddnerf_pack_rays_first_cycle(_ptr(o), _ptr(d), _ptr(r), float(near), float(far), _ptr(t_lin), _ptr(t_rand),
                                                       _ptr(rays), _ptr(t_vals), n, nc, int(lindisp), _stream()), "ddnerf_pack_rays_first_cycle")
    return rays, t_vals


def encode_first_cycle(ray_origins, ray_directions, ray_rad, near, far, t_lin, lindisp=False, cylinder=False, kind="fp32", out=None):
    """pack_rays_first_cycle (no jitter) + encode of the coarse fenceposts in ONE launch -> (rays [n,12], t_vals [n,nc+1], feat [n*nc,128]);
    `out` = (rays, t_vals) tensors to fill (the caller handed them out earlier, see GeneralMipNerfModel.get_rays_batches)"""
    o = _f32c(ray_origins.reshape(-1, 3), "origins")
    d = _f32c(ray_directions.reshape(-1, 3), "directions")
    r = _f32c(ray_rad.reshape(-1), "radii")
    t_lin = _f32c(t_lin, "t_lin")
    n, nc = o.shape[0], t_lin.shape[0] - 1
    if out is None:
        out = (torch.empty((n, 12), dtype=torch.float32, device=o.device), torch.empty((n, nc + 1), dtype=torch.float32, device=o.device))
    rays, t_vals = out
    code, dtype = FEAT_KINDS[kind]
    feat = torch.empty((n * nc, FEAT_LD), dtype=dtype, device=o.device)
    _lib.check(_lib.lib().ddnerf_encode_first_cycle(_ptr(o), _ptr(d), _ptr(r), float(near), float(far), _ptr(t_lin), int(lindisp), _ptr(rays),
                                                    _ptr(t_vals), _ptr(feat), n, nc, int(cylinder), code, _stream()), "ddnerf_encode_first_cycle")
    return rays, t_vals, feat


class KernelNoise:
    """density noise the compositing kernels draw themselves (volume_rendering_utils.py:29-37's randn * std without a generator launch or a
    noise tensor): (seed, offset) of the Philox stream, `base` = the first element index of this pass, `std`"""

    def __init__(self, seed, offset, base, std):
        self.seed, self.offset, self.base, self.std = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(base), float(std)

    def at(self, base):
        return KernelNoise(self.seed, self.offset, base, self.std)

    def materialise(self, count, device):
        """the values themselves ([count] fp32): what the kernels add to the densities of elements base .. base + count"""
        out = torch.empty(count, dtype=torch.float32, device=device)
        _lib.check(_lib.lib().ddnerf_debug_philox_normal(_ptr(out), count, self.seed, self.offset, self.base, self.std, _stream()),
                   "ddnerf_debug_philox_normal")
        return out


def _noise_args(noise):
    """(noise tensor or None, seed, offset, base, std) for the *_rng entry points"""
    if isinstance(noise, KernelNoise):
        return None, noise.seed, noise.offset, noise.base, noise.std
    return _f32c(noise, "noise"), 0, 0, 0, 0.0


def dd_coarse_forward(raw6, t_vals, rays, noise, smooth, dist_reg, white_bkgd, blender, sample=None):
    """The coarse pass of DDNerfModel behind the MLP, render path (models/models.py:242-295), in two launches: DD head + compositing
    + level-0 records.  -> (composite dict, head dict, records ticket for dd_records_finish).  noise: a tensor, None, or a KernelNoise
    (the kernel draws the density noise itself).  sample = (u_base [ns], rnd [n,ns] | None,
    near, far, pdf_padding): the first launch also draws the fine pass's fenceposts (sample_pdf_mu_sigma on this pass's weights, mus and
    smoothed head values, bit for bit) -> a fourth result, samples [n, ns]."""
    raw6, t_vals, rays = _f32c(raw6, "raw"), _f32c(t_vals, "t_vals"), _f32c(rays, "rays")
    noise, nseed, noff, nbase, nstd = _noise_args(noise)
    n, nc, ldr = raw6.shape
    assert ldr == 6
    dev = raw6.device
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    head = {k: e(n, nc) for k in ("mus", "sigmas", "left", "part", "ssig", "sleft", "spart")}
    head["scal"] = e(4)
    c = dict(rgb_map=e(n, 3), disp=e(n), acc=e(n), weights=e(n, nc), depth=e(n), cdisp=e(n), rgb=None)
    outs = [e(n * nc) for _ in range(3)]
    total = _records_slot(dev)
    ws = torch.empty(_lib.lib().ddnerf_dd_coarse_workspace_bytes(n, nc), dtype=torch.uint8, device=dev)
    flags = (1 if white_bkgd else 0) | (2 if blender else 0)
    if sample is None:
        u_base = rnd = samples = None
        near_ = far_ = 0.0
        ns = pad = 0
    else:
        u_base, rnd, near_, far_, pad = sample
        u_base, rnd = _f32c(u_base, "u_base"), _f32c(rnd, "rnd")
        ns = u_base.shape[0]
        samples = e(n, ns)
    _lib.check(_lib.lib().ddnerf_dd_coarse_sample_forward(
        _ptr(raw6), _ptr(t_vals), _ptr(rays), _ptr(noise), n, nc, flags, float(smooth), float(dist_reg),
        *[_ptr(head[k]) for k in ("mus", "sigmas", "left", "part", "ssig", "sleft", "spart", "scal")],
        *[_ptr(c[k]) for k in ("rgb_map", "disp", "acc", "weights", "depth", "cdisp")], *[_ptr(o) for o in outs], _ptr(total), _ptr(ws),
        _ptr(u_base), _ptr(rnd), float(near_), float(far_), _ptr(samples), int(ns), int(bool(pad)), nseed, noff, nbase, nstd, _stream()),
        "ddnerf_dd_coarse_sample_forward")
    if sample is None:
        return c, head, _records_ticket(outs, total)
    return c, head, _records_ticket(outs, total), samples


def composite_forward_keep(raw, t_vals, rays, noise, mus, white_bkgd, blender, dp_filter):
    """compositing of the fine pass + the dp loss's row filter in one launch -> (composite dict, dp-loss workspace holding keep[n])"""
    raw, t_vals, rays = _f32c(raw, "raw"), _f32c(t_vals, "t_vals"), _f32c(rays, "rays")
    noise, nseed, noff, nbase, nstd = _noise_args(noise)
    mus = _f32c(mus, "mus")
    n, S, ldr = raw.shape
    dev = raw.device
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    o = dict(rgb_map=e(n, 3), disp=e(n), acc=e(n), weights=e(n, S), depth=e(n), cdisp=e(n) if mus is not None else None, rgb=None)
    ws = torch.empty(_lib.lib().ddnerf_dp_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
    flags = (1 if white_bkgd else 0) | (2 if blender else 0) | (4 if dp_filter else 0)
    _lib.check(_lib.lib().ddnerf_composite_forward_keep_rng(_ptr(raw), ldr, _ptr(t_vals), _ptr(rays), _ptr(noise), _ptr(mus), n, S, flags,
                                                            _ptr(o["rgb_map"]), _ptr(o["disp"]), _ptr(o["acc"]), _ptr(o["weights"]),
                                                            _ptr(o["depth"]), _ptr(o["cdisp"]), _ptr(ws), nseed, noff, nbase, nstd, _stream()),
               "ddnerf_composite_forward_keep_rng")
    return o, ws


def dp_loss_forward_kept(t1, t0, w1, w0, mus0, sig0, left0, part0, ws, reg_scal):
    """dp_loss_forward behind composite_forward_keep (the row filter is in `ws` already) -> (loss, total [1]): the rows kernel and the
    finish kernel.  (Round 4 built a ONE-launch form -- the workgroup taking the last of 1024 tickets finished -- and measured it SLOWER,
    37.8 us against 20.2 + 4.9: one ticket counter takes ~88 atomics per microsecond; profiles/r04_render_bf16_kernel_stats_dp_one_launch.csv.
    Round 5 removed it.)"""
    t1, t0, w1, w0, mus0, sig0, left0, part0 = (_f32c(t, "dp_loss arg") for t in (t1, t0, w1, w0, mus0, sig0, left0, part0))
    n, nc = w0.shape
    nf = w1.shape[1]
    loss = torch.empty((), dtype=torch.float32, device=w0.device)
    total = torch.empty(1, dtype=torch.float32, device=w0.device)
    _lib.check(_lib.lib().ddnerf_dp_loss_forward_kept(_ptr(t1), _ptr(t0), _ptr(w1), _ptr(w0), _ptr(mus0), _ptr(sig0), _ptr(left0), _ptr(part0),
                                                      n, nc, nf, _ptr(loss), _ptr(_f32c(reg_scal, "reg_scal")), _ptr(total), _ptr(ws), _stream()),
               "ddnerf_dp_loss_forward_kept")
    return loss, total


def sample_pdf(bins, weights, u_base, rnd, pdf_padding):
    """models/samplers.py:64-121 -> samples [n,ns]"""
    bins, weights, u_base, rnd = _f32c(bins, "bins"), _f32c(weights, "weights"), _f32c(u_base, "u_base"), _f32c(rnd, "rnd")
    n, nc = weights.shape
    ns = u_base.shape[0]
    out = torch.empty((n, ns), dtype=torch.float32, device=bins.device)
    _lib.check(_lib.lib().ddnerf_sample_pdf(_ptr(bins), _ptr(weights), _ptr(u_base), _ptr(rnd), _ptr(out), n, nc, ns,
                                            int(pdf_padding), _stream()), "ddnerf_sample_pdf")
    return out


def sample_pdf_mu_sigma(bins, weights, mus, sigmas, part, left, u_base, rnd, near, far, pdf_padding, want_ind=False):
    """models/samplers.py:124-215 -> samples [n,ns] (sorted) [, bins_ind int32 [n,ns]]"""
    bins, weights, mus, sigmas, part, left, u_base, rnd = (
        _f32c(t, k) for t, k in ((bins, "bins"), (weights, "weights"), (mus, "mus"), (sigmas, "sigmas"), (part, "part"),
                                 (left, "left"), (u_base, "u_base"), (rnd, "rnd")))
    n, nc = weights.shape
    ns = u_base.shape[0]
    out = torch.empty((n, ns), dtype=torch.float32, device=bins.device)
    ind = torch.empty((n, ns), dtype=torch.int32, device=bins.device) if want_ind else None
    _lib.check(_lib.lib().ddnerf_sample_pdf_mu_sigma(_ptr(bins), _ptr(weights), _ptr(mus), _ptr(sigmas), _ptr(part),
                                                     _ptr(left), _ptr(u_base), _ptr(rnd), float(near), float(far),
                                                     _ptr(out), _ptr(ind), n, nc, ns, int(pdf_padding), _stream()),
               "ddnerf_sample_pdf_mu_sigma")
    return (out, ind) if want_ind else out


def dp_loss_forward(t1, t0, w1, w0, mus0, sig0, left0, part0, blender, reg_scal=None):
    """models/dd_utils.py:6-78 -> 0-dim fp32 tensor (kl_div mean over kept rows; 0 if none kept).
    With reg_scal (the DD head's scal[4]) -> (loss, total) where total [1] = loss * nf + mus_reg + sig_reg, the level-1
    `dp_loss` record of models/models.py:287-289, written by the same launch."""
    t1, t0, w1, w0, mus0, sig0, left0, part0 = (_f32c(t, "dp_loss arg") for t in (t1, t0, w1, w0, mus0, sig0, left0, part0))
    n, nc = w0.shape
    nf = w1.shape[1]
    loss = torch.empty((), dtype=torch.float32, device=w0.device)
    total = torch.empty(1, dtype=torch.float32, device=w0.device) if reg_scal is not None else None
    ws = torch.empty(_lib.lib().ddnerf_dp_loss_workspace_bytes(n), dtype=torch.uint8, device=w0.device)
    _lib.check(_lib.lib().ddnerf_dp_loss_forward(_ptr(t1), _ptr(t0), _ptr(w1), _ptr(w0), _ptr(mus0), _ptr(sig0), _ptr(left0),
                                                 _ptr(part0), n, nc, nf, int(blender), _ptr(loss), _ptr(_f32c(reg_scal, "reg_scal")),
                                                 _ptr(total), _ptr(ws), _stream()),
               "ddnerf_dp_loss_forward")
    return loss if reg_scal is None else (loss, total)


# ---- backward entry points ---------------------------------------------------------------------------------
def composite_backward(raw, t_vals, rays, noise, white_bkgd, blender, g_rgb_map, g_weights):
    raw, t_vals, rays, noise = _f32c(raw, "raw"), _f32c(t_vals, "t_vals"), _f32c(rays, "rays"), _f32c(noise, "noise")
    g_rgb_map, g_weights = _f32c(g_rgb_map, "g_rgb_map"), _f32c(g_weights, "g_weights")
    n, S, ldr = raw.shape
    g_raw = torch.empty_like(raw)
    flags = (1 if white_bkgd else 0) | (2 if blender else 0)
    _lib.check(_lib.lib().ddnerf_composite_backward(_ptr(raw), ldr, _ptr(t_vals), _ptr(rays), _ptr(noise), n, S, flags,
                                                    _ptr(g_rgb_map), _ptr(g_weights), _ptr(g_raw), _stream()),
               "ddnerf_composite_backward")
    return g_raw


def dd_head_backward_(raw6, dist_reg, g_mus, g_sigmas, g_scal, g_raw6):
    """adds the DD-head gradient into g_raw6[..., 4:6] in place"""
    raw6 = _f32c(raw6, "raw6")
    n, nc = raw6.shape[0], raw6.shape[1]
    _lib.check(_lib.lib().ddnerf_dd_head_backward(_ptr(raw6), n, nc, float(dist_reg), _ptr(_f32c(g_mus, "g_mus")),
                                                  _ptr(_f32c(g_sigmas, "g_sigmas")), _ptr(_f32c(g_scal, "g_scal")),
                                                  _ptr(g_raw6), _stream()), "ddnerf_dd_head_backward")
    return g_raw6


def dp_loss_backward(t1, t0, w1, w0, mus0, sig0, left0, part0, blender, g_loss):
    t1, t0, w1, w0, mus0, sig0, left0, part0 = (_f32c(t, "dp_loss arg") for t in (t1, t0, w1, w0, mus0, sig0, left0, part0))
    n, nc = w0.shape
    nf = w1.shape[1]
    g_loss = _f32c(g_loss.reshape(1), "g_loss")
    gw, gm, gs = torch.empty_like(w0), torch.empty_like(w0), torch.empty_like(w0)
    ws = torch.empty(_lib.lib().ddnerf_dp_loss_workspace_bytes(n), dtype=torch.uint8, device=w0.device)
    _lib.check(_lib.lib().ddnerf_dp_loss_backward(_ptr(t1), _ptr(t0), _ptr(w1), _ptr(w0), _ptr(mus0), _ptr(sig0), _ptr(left0),
                                                  _ptr(part0), n, nc, nf, int(blender), _ptr(g_loss), _ptr(gw), _ptr(gm),
                                                  _ptr(gs), _ptr(ws), _stream()), "ddnerf_dp_loss_backward")
    return gw, gm, gs


ACT_ROWS = 2560
ROW_FEAT, ROW_DIR, ROW_X = 2048, 2304, 2432

# Test hook: how the training kernels' record / sign-word buffers are allocated.  The kernels write every element a weight
# gradient may read and mask the rest, so the product allocates uninitialised memory; the tests swap in an allocator that fills
# the buffers with NaN patterns first (tests/test_hip_backward.py) -- deterministic, unlike hoping that the caching allocator hands
# poisoned blocks back.
RECORD_ALLOC = torch.empty


def _record(shape, dtype, device):
    return RECORD_ALLOC(shape, dtype=dtype, device=device)


def mlp_f32_pack_t(params_flat, depth_head):
    params_flat = _f32c(params_flat, "params")
    packed = torch.empty(_lib.lib().ddnerf_mlp_f32_packed_t_floats(int(depth_head)), dtype=torch.float32,
                         device=params_flat.device)
    _lib.check(_lib.lib().ddnerf_mlp_f32_pack_t(_ptr(params_flat), int(depth_head), _ptr(packed), _stream()),
               "ddnerf_mlp_f32_pack_t")
    return packed


_REC_SUFFIX = {False: "", None: "", True: "_rec", "hilo": "_rec", "pairs": "_recp", "values": "_recf"}


def mlp_f32_forward_train(feat, packed, depth_head, rec=False, dirs=None, S=0):
    """forward + recorded activations: returns raw [M,4|6], acts [2560, ld] (fp32, transposed: row = feature).  rec=True / "hilo": a
    record of blocked hi/lo words instead -- x3_unsplit reads it back -- for the packed-operand weight-gradient kernel; rec="pairs": a
    record of bf16 row pairs ([1280, ld] words, x3_unpair reads it back) for the one-MFMA weight-gradient kernel; rec="values": the blocked
    layout of the hi/lo words holding the fp32 values themselves (x3_unblock reads it back), for ddnerf_mlp_x3_wgrad_blocked -- and a THIRD
    result, the sign record (uint8 [ddnerf_mlp_f32_sign_bytes(ld)]) mlp_f32_backward_data(rec="values") takes its ReLU masks from; with
    `dirs` [M / S, 32] (encode_rays) that build takes the view-direction columns from the per-ray table instead of columns 96..127 of `feat`"""
    fn = "ddnerf_mlp_f32_forward_train" + _REC_SUFFIX[rec]
    feat = _f32c(feat, "feat")
    M = feat.shape[0]
    ld = (M + 127) // 128 * 128
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)
    acts = _record((ACT_ROWS // 2 if rec == "pairs" else ACT_ROWS, ld), torch.float32, feat.device)
    signs = _record((_lib.lib().ddnerf_mlp_f32_sign_bytes(ld),), torch.uint8, feat.device) if rec == "values" else None
    if dirs is not None and rec != "values":
        raise _lib.DDNerfHipError("mlp_f32_forward_train: per-ray view directions need rec='values'")
    extra = (_ptr(signs), _ptr(_f32c(dirs, "dirs")) if dirs is not None else None, int(S)) if rec == "values" else ()

    def launch():
        _lib.check(getattr(_lib.lib(), fn)(_ptr(feat), _ptr(packed), int(depth_head), _ptr(raw), _ptr(acts), *extra, M, ld, _stream()), fn)

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return (raw, acts, signs) if rec == "values" else (raw, acts)


def mlp_f32_backward_data(g_raw, packed_t, acts, depth_head, rec=False, signs=None):
    """rec: `acts` is a record (mlp_f32_forward_train(rec=...)) and so is the result, in the same format; rec="values" also takes the
    forward's sign record"""
    fn = "ddnerf_mlp_f32_backward_data" + _REC_SUFFIX[rec]
    g_raw = _f32c(g_raw, "g_raw")
    M = g_raw.shape[0]
    ld = acts.shape[1]
    deltas = _record(tuple(acts.shape), acts.dtype, acts.device)
    extra = ()
    if rec == "values":
        if signs is None or signs.dtype != torch.uint8 or signs.numel() != _lib.lib().ddnerf_mlp_f32_sign_bytes(ld) or signs.device != acts.device:
            raise _lib.DDNerfHipError("mlp_f32_backward_data(rec='values'): signs must be the sign record mlp_f32_forward_train returned")
        extra = (_ptr(signs),)
    _lib.check(getattr(_lib.lib(), fn)(_ptr(g_raw), _ptr(packed_t), _ptr(acts), *extra, int(depth_head), _ptr(deltas), M, ld, _stream()), fn)
    return deltas


def mlp_x3_pack_t(params_flat, depth_head):
    params_flat = _f32c(params_flat, "params")
    packed = torch.empty(_lib.lib().ddnerf_mlp_x3_packed_t_bytes(int(depth_head)), dtype=torch.uint8, device=params_flat.device)
    _lib.check(_lib.lib().ddnerf_mlp_x3_pack_t(_ptr(params_flat), int(depth_head), _ptr(packed), _stream()), "ddnerf_mlp_x3_pack_t")
    return packed


def mlp_x3_forward_train(feat, packed, depth_head, dirs=None, S=0):
    """x3 forward that records activations: -> raw [M,4|6], acts = a record of bf16 ROW PAIRS, [1280, ld] 32-bit words in a
    float32-typed tensor (x3_unpair gives the [2560, ld] matrix of the bf16-rounded values), bits [160, ld] uint16 (sign words);
    with `dirs` [M / S, 32] (encode_rays) the view-direction columns come from the per-ray table instead of columns 96..127 of `feat`"""
    feat = _f32c(feat, "feat")
    M = feat.shape[0]
    ld = (M + 127) // 128 * 128
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)
    acts = _record((ACT_ROWS // 2, ld), torch.float32, feat.device)
    bits = _record((ACT_ROWS // 32 * 2, ld), torch.int16, feat.device)

    def launch():
        if dirs is not None:
            _lib.check(_lib.lib().ddnerf_mlp_x3_forward_train_rays(_ptr(feat), _ptr(_f32c(dirs, "dirs")), int(S), _ptr(packed), int(depth_head), _ptr(raw),
                                                                   _ptr(acts), _ptr(bits), M, ld, _stream()), "ddnerf_mlp_x3_forward_train_rays")
            return
        _lib.check(_lib.lib().ddnerf_mlp_x3_forward_train(_ptr(feat), _ptr(packed), int(depth_head), _ptr(raw), _ptr(acts),
                                                          _ptr(bits), M, ld, _stream()), "ddnerf_mlp_x3_forward_train")

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw, acts, bits


def mlp_x3_backward_data(g_raw, packed_t, bits, depth_head):
    """-> deltas: a record of bf16 row pairs ([1280, ld] words), like mlp_x3_forward_train's acts"""
    g_raw = _f32c(g_raw, "g_raw")
    M = g_raw.shape[0]
    ld = bits.shape[1]
    deltas = _record((ACT_ROWS // 2, ld), torch.float32, g_raw.device)
    _lib.check(_lib.lib().ddnerf_mlp_x3_backward_data(_ptr(g_raw), _ptr(packed_t), _ptr(bits), int(depth_head), _ptr(deltas),
                                                      M, ld, _stream()), "ddnerf_mlp_x3_backward_data")
    return deltas


# ---- the x3 training tier with EXACT records (DDNERF_X3_WGRAD=exact; csrc/mlp_x3e_*.hip): the same chains, every layer's output / delta
# recorded as blocked hi/lo words ([2560, ld]: the fp32 value's exact split) for the three-MFMA weight gradients (mode "x3p")
def mlp_x3e_pack(params_flat, depth_head):
    params_flat = _f32c(params_flat, "params")
    packed = torch.empty(_lib.lib().ddnerf_mlp_x3e_packed_bytes(int(depth_head)), dtype=torch.uint8, device=params_flat.device)
    _lib.check(_lib.lib().ddnerf_mlp_x3e_pack(_ptr(params_flat), int(depth_head), _ptr(packed), _stream()), "ddnerf_mlp_x3e_pack")
    return packed


def mlp_x3e_pack_t(params_flat, depth_head):
    params_flat = _f32c(params_flat, "params")
    packed = torch.empty(_lib.lib().ddnerf_mlp_x3e_packed_t_bytes(int(depth_head)), dtype=torch.uint8, device=params_flat.device)
    _lib.check(_lib.lib().ddnerf_mlp_x3e_pack_t(_ptr(params_flat), int(depth_head), _ptr(packed), _stream()), "ddnerf_mlp_x3e_pack_t")
    return packed


def mlp_x3e_forward_train(feat, packed, depth_head):
    """-> raw [M,4|6], acts = a [2560, ld] record of blocked hi/lo words (x3_unsplit gives the fp32 matrix), bits [160, ld] sign words"""
    feat = _f32c(feat, "feat")
    M = feat.shape[0]
    ld = (M + 127) // 128 * 128
    raw = torch.empty((M, 6 if depth_head else 4), dtype=torch.float32, device=feat.device)
    acts = _record((ACT_ROWS, ld), torch.float32, feat.device)
    bits = _record((ACT_ROWS // 32 * 2, ld), torch.int16, feat.device)

    def launch():
        _lib.check(_lib.lib().ddnerf_mlp_x3e_forward_train(_ptr(feat), _ptr(packed), int(depth_head), _ptr(raw), _ptr(acts),
                                                           _ptr(bits), M, ld, _stream()), "ddnerf_mlp_x3e_forward_train")

    if MLP_LAUNCH_HOOK is not None:
        MLP_LAUNCH_HOOK(M, launch)
    else:
        launch()
    return raw, acts, bits


def mlp_x3e_backward_data(g_raw, packed_t, bits, depth_head):
    """-> deltas: a [2560, ld] record of blocked hi/lo words, like mlp_x3e_forward_train's acts"""
    g_raw = _f32c(g_raw, "g_raw")
    M = g_raw.shape[0]
    ld = bits.shape[1]
    deltas = _record((ACT_ROWS, ld), torch.float32, g_raw.device)
    _lib.check(_lib.lib().ddnerf_mlp_x3e_backward_data(_ptr(g_raw), _ptr(packed_t), _ptr(bits), int(depth_head), _ptr(deltas),
                                                       M, ld, _stream()), "ddnerf_mlp_x3e_backward_data")
    return deltas


# ---- ray generation -------------------------------------------------------------------
def ray_bundle(H, W, focal, cam2world, device="cuda"):
    """general_utils/nerf_helpers.py:67-125 -> origins [H,W,3], directions [H,W,3], radii [H,W,1]"""
    import ctypes

    pose = torch.as_tensor(cam2world, dtype=torch.float32).cpu().contiguous()[:3, :4].contiguous()
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=device)
    o, d, r = e(H, W, 3), e(H, W, 3), e(H, W, 1)
    with torch.cuda.device(o.device):
        _lib.check(_lib.lib().ddnerf_ray_bundle(int(H), int(W), float(focal), ctypes.c_void_p(pose.data_ptr()), _ptr(o),
                                                _ptr(d), _ptr(r), _stream()), "ddnerf_ray_bundle")
    return o, d, r


def ndc_rays(H, W, focal, origins, directions, near=1.0):
    """data_utils/dataset_helpers.py:3-42 -> NDC origins, directions [H,W,3], radii [H,W] (the reference's shape)"""
    origins, directions = _f32c(origins, "origins"), _f32c(directions, "directions")
    o, d = torch.empty_like(origins), torch.empty_like(directions)
    r = torch.empty((H, W), dtype=torch.float32, device=origins.device)
    _lib.check(_lib.lib().ddnerf_ndc_rays(int(H), int(W), float(focal), float(near), _ptr(origins), _ptr(directions), _ptr(o),
                                          _ptr(d), _ptr(r), _stream()), "ddnerf_ndc_rays")
    return o, d, r


def ndc_depth_to_regular(ndc_depth, origins, directions):
    """data_utils/dataset_helpers.py:45-49 (switch_t_ndc_to_regular): NDC depth map [H,W] + the regular bundle [H,W,3] x 2 of
    the same view -> camera-space depth [H,W]"""
    ndc_depth, origins, directions = _f32c(ndc_depth, "ndc_depth"), _f32c(origins, "origins"), _f32c(directions, "directions")
    assert origins.shape == directions.shape == tuple(ndc_depth.shape) + (3,)
    out = torch.empty_like(ndc_depth)
    _lib.check(_lib.lib().ddnerf_ndc_depth_to_regular(ndc_depth.numel(), _ptr(ndc_depth), _ptr(origins), _ptr(directions), _ptr(out),
                                                      _stream()), "ddnerf_ndc_depth_to_regular")
    return out


# weight-gradient arithmetic: "x3" = bf16 matrix cores with exact hi/lo operand splits (3 MFMAs per product, ~2^-16
# relative product error, HBM-bound); "f32" = the fp32 matrix cores (exact fp32 products, MFMA-bound, 3x slower)
# "x3p" = the same three-MFMA product on operands that already hold hi/lo words (what the x3 training kernels record)
# DDNERF_WGRAD (the fp32 tier's weight gradients): "x3" (default) = bf16 matrix cores on exact hi/lo splits, fp32-class; since round 5 the
# recording kernels write blocked records of the fp32 VALUES and the weight-gradient kernel splits them per fragment ("x3b") | "x3words" = the
# same gradients bit for bit from records of hi/lo WORDS split by the recording kernels (rounds 3-4; kept for the A/B: +0.3 ms per step)
# | "f32" = fp32 matrices and the fp32-MFMA kernel | "pairs" (an opt-in speed mode: records of bf16 row pairs, one MFMA per product,
# "x3h": NOT fp32-class).  Anything else is an error at import, not a KeyError in the first backward pass.
WGRAD_MODE = os.environ.get("DDNERF_WGRAD", "x3")
if WGRAD_MODE not in ("x3", "x3words", "f32", "pairs"):
    raise _lib.DDNerfHipError("DDNERF_WGRAD=%r: expected x3, x3words, f32 or pairs" % WGRAD_MODE)
# "x3h" = ONE MFMA per product on records of bf16 row pairs (the x3 training tier: its forward / backward-data chains stay fp32-class,
# the weight gradients contract bf16-rounded activations and deltas with fp32 accumulation -- half the record bytes, a third of the MFMAs)
_WGRAD_FN = {"x3": "ddnerf_mlp_x3_wgrad", "f32": "ddnerf_mlp_f32_wgrad", "x3p": "ddnerf_mlp_x3_wgrad_packed", "x3h": "ddnerf_mlp_x3_wgrad_pairs",
             "x3b": "ddnerf_mlp_x3_wgrad_blocked"}
_RECORD_MODES = ("x3p", "x3h", "x3b")


WGRAD_PAIRED = os.environ.get("DDNERF_WGRAD_PAIRED", "1") != "0"
WGRAD_PLAN = os.environ.get("DDNERF_WGRAD_PLAN", "1") != "0"     # (0: the jobs dealt to the two lanes round-robin in layer order, as before round 5)
_SIDE = {}


def _side_stream(device, i):
    key = (str(device), i)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def x3_split(x):
    """fp32 [2560, ld] ([feature][sample]) -> the x3 training tier's record of the same matrix: blocked hi/lo words
    (include/ddnerf_hip.h, ddnerf_mlp_x3_wgrad_packed), in a float32-typed tensor of the same shape (bit patterns, not values)"""
    x = _f32c(x, "x")
    assert x.shape[0] == ACT_ROWS and x.shape[1] % 16 == 0
    w = torch.empty_like(x)
    _lib.check(_lib.lib().ddnerf_mlp_x3_split(_ptr(x), x.shape[0], x.shape[1], 0, _ptr(w), _stream()), "ddnerf_mlp_x3_split")
    return w


def x3_split_pairs(x):
    """fp32 [2560, ld] ([feature][sample]) -> a record of bf16 row pairs ([1280, ld] words in a float32-typed tensor): word
    ((m >> 4) * 1280 + (row >> 1)) * 16 + (m & 15) = bf16(x[row even][m]) | bf16(x[row odd][m]) << 16"""
    x = _f32c(x, "x")
    assert x.shape[0] == ACT_ROWS and x.shape[1] % 16 == 0
    w = torch.empty((ACT_ROWS // 2, x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ddnerf_mlp_x3_split_pairs(_ptr(x), x.shape[0], x.shape[1], 0, _ptr(w), _stream()), "ddnerf_mlp_x3_split_pairs")
    return w


def x3_unpair(rec):
    """a record of bf16 row pairs ([1280, ld] words) -> fp32 [2560, ld]: the bf16 values it holds"""
    ld = rec.shape[1]
    i = rec.view(torch.int32).view(ld // 16, ACT_ROWS // 2, 16).permute(1, 0, 2).reshape(ACT_ROWS // 2, ld)
    out = torch.empty((ACT_ROWS, ld), dtype=torch.float32, device=rec.device)
    out[0::2] = (i << 16).view(torch.float32)
    out[1::2] = (i & -65536).view(torch.float32)
    return out


def x3_block(x):
    """fp32 [2560, ld] ([feature][sample]) -> the blocked record of the same VALUES (ddnerf_mlp_x3_wgrad_blocked's operands, what
    mlp_f32_forward_train(rec="values") writes): element (row, m) at word ((m >> 4) * 2560 + row) * 16 + (m & 15)"""
    assert x.shape[0] == ACT_ROWS and x.shape[1] % 16 == 0
    return x.view(ACT_ROWS, x.shape[1] // 16, 16).permute(1, 0, 2).contiguous().view(ACT_ROWS, x.shape[1])


def x3_unblock(rec):
    """the inverse of x3_block"""
    ld = rec.shape[1]
    return rec.view(ld // 16, ACT_ROWS, 16).permute(1, 0, 2).reshape(ACT_ROWS, ld)


def x3_unsplit(rec):
    """a record of the x3 training tier -> fp32 [2560, ld] values hi + lo (differs from the recorded value by <= 2^-17 relative)"""
    ld = rec.shape[1]
    i = rec.view(torch.int32).view(ld // 16, ACT_ROWS, 16).permute(1, 0, 2).reshape(ACT_ROWS, ld)
    return (i & -65536).view(torch.float32) + (i << 16).view(torch.float32)


def mlp_f32_wgrad_job(deltas, drow0, n_out, acts, arow0, n_in, n_in_used, M, dst, dst_ld, dst_col0, dst_bias, workspace,
                      mode=None, max_wg=0):
    """dst[r*dst_ld + dst_col0 + c] = sum_s deltas[drow0+r][s] * acts[arow0+c][s]; dst_bias[r] = sum_s deltas[drow0+r][s]"""
    mode = mode or {"pairs": "x3h", "x3words": "x3p"}.get(WGRAD_MODE, WGRAD_MODE)    # (a record-format name is not a kernel name)
    fn = getattr(_lib.lib(), _WGRAD_FN[mode])
    extra = (int(max_wg),) if mode in _RECORD_MODES else ()
    _lib.check(fn(_ptr(deltas), drow0, n_out, _ptr(acts), arow0, n_in, n_in_used, M, deltas.shape[1], _ptr(dst), dst_ld,
                  dst_col0, _ptr(dst_bias), _ptr(workspace), *extra, _stream()), _WGRAD_FN[mode])


def mlp_f32_weight_grads(net, acts, deltas, M, mode=None):
    """all parameter gradients of one network -> (flat gradient buffer in registration order, per-parameter views).
    mode "x3p": acts / deltas are records of blocked hi/lo words (the fp32 tier's record build); "x3h": records of bf16 row pairs
    (the x3 training kernels')."""
    params = list(net.parameters())
    flat_g = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=acts.device)
    off, gv, views = 0, {}, []
    for (name, p) in net.named_parameters():
        v = flat_g[off:off + p.numel()].view(p.shape)
        gv[name] = v
        views.append(v)
        off += p.numel()
    nws = _lib.lib().ddnerf_mlp_f32_wgrad_workspace_floats(M)
    main = torch.cuda.current_stream()
    records = mode in _RECORD_MODES
    pair = records and WGRAD_PAIRED
    if pair:
        # two half-width jobs side by side (each fills half the CUs: one workgroup per CU): half the partial slabs per job, and
        # one job's prologue / slab epilogue / reduction under the other's main loop
        lanes = [(_side_stream(acts.device, i), torch.empty(nws, dtype=torch.float32, device=acts.device)) for i in range(2)]
        ready = torch.cuda.Event()
        ready.record(main)
        for st, _ in lanes:
            st.wait_event(ready)
    else:
        lanes = [(main, torch.empty(nws, dtype=torch.float32, device=acts.device))]
    turn = [0]
    # Two lanes: which job runs on which, and in what order, is PLANNED (round 5) instead of dealt round-robin in layer order.  The jobs'
    # times at BASELINE size (paired, fine pass): the skip layer 395 us, the 256 x 256 jobs 305 - 315, layer 0 (96 columns) 258, the two small
    # ones (heads 6 x 128, the dir layer's 32 view columns) 110 - 115 at a fraction of the chip's bandwidth.  Longest-first assignment balances the
    # lanes within 30 us (round-robin: 142), and the small jobs sit at the FRONT of one lane and at the END of the other, beside a big job
    # of the other lane each, instead of both at the end beside each other.
    planned = pair and records and WGRAD_PLAN
    pending, lane_of = {}, {}
    plan = (("dir32", "l5", "l1", "l3", "l7", "l0"), ("t1", "l2", "l4", "l6", "feat", "t2"))

    def on_lane(name, fn):
        """fn(stream, workspace) launches job `name`: now, on the next lane in turn -- or, planned, when its lane's turn comes"""
        if planned:
            pending[name] = fn
        else:
            st, ws = lanes[turn[0] % len(lanes)]
            turn[0] += 1
            lane_of[name] = st
            with torch.cuda.stream(st):
                fn(st, ws)

    def job(name, *a):
        on_lane(name, lambda st, ws: mlp_f32_wgrad_job(deltas, a[0], a[1], acts, a[2], a[3], a[4], M, a[5], a[6], a[7], a[8], ws, mode=mode,
                                                       max_wg=128 if pair else 0))

    for l in range(8):
        w, b = gv["layers_xyz.%d.weight" % l], gv["layers_xyz.%d.bias" % l]
        if l == 0:
            job("l0", 0, 256, ROW_X, 96, 96, w, 96, 0, b)
        elif l == 5 and records:  # cat(xyz, h4) in one pass over the deltas
            skip = _WGRAD_FN[mode] + "_skip"
            on_lane("l5", lambda st, ws, w=w, b=b: _lib.check(getattr(_lib.lib(), skip)(_ptr(deltas), 1280, _ptr(acts), ROW_X, 1024, M, deltas.shape[1], _ptr(w),
                                                                                    _ptr(b), _ptr(ws), 128 if pair else 0, _stream()), skip))
        elif l == 5:
            job("l5a", 1280, 256, ROW_X, 96, 96, w, 352, 0, b)
            job("l5b", 1280, 256, 1024, 256, 256, w, 352, 96, None)
        else:
            job("l%d" % l, 256 * l, 256, 256 * (l - 1), 256, 256, w, 256, 0, b)
    job("feat", ROW_FEAT, 256, 256 * 7, 256, 256, gv["fc_feat.weight"], 256, 0, gv["fc_feat.bias"])
    wd = gv["layers_dir.0.weight"]
    if records:
        # Jobs that contract over the same activations share ONE pass over them (rows ROW_DIR .. ROW_X + 5 of `deltas` are
        # adjacent: d(dir hidden) 128 rows, then d(raw): 0..2 rgb, 3 alpha, 4..5 mu / sigma):
        #   [d(dir hidden) ; d(raw)] x fc_feat out  -> the dir layer's hidden columns and fc_alpha   (saves 288 - 32 rows x M)
        #   d(raw) x dir hidden                     -> fc_rgb and fc_mu_sigma                        (saves 160 rows x M)
        # the sub-blocks are copied out of two small scratch matrices (rows of no interest are products nobody reads).
        t1 = torch.empty((160, 256), dtype=torch.float32, device=acts.device)
        b1 = torch.empty(160, dtype=torch.float32, device=acts.device)
        t2 = torch.empty((6, 128), dtype=torch.float32, device=acts.device)
        b2 = torch.empty(6, dtype=torch.float32, device=acts.device)
        job("t1", ROW_DIR, 160, ROW_FEAT, 256, 256, t1, 256, 0, b1)
        job("t2", ROW_X, 6, ROW_DIR, 128, 128, t2, 128, 0, b2)
        job("dir32", ROW_DIR, 128, ROW_X + 96, 32, 27, wd, 283, 256, None)
        if planned:
            assert sorted(pending) == sorted(plan[0] + plan[1]), sorted(pending)
            for i in range(len(plan[0])):          # (enqueued alternately: neither stream waits for the host)
                for li in (0, 1):
                    st, ws = lanes[li]
                    lane_of[plan[li][i]] = st
                    with torch.cuda.stream(st):
                        pending[plan[li][i]](st, ws)
        st1, st2 = lane_of["t1"], lane_of["t2"]
        with torch.cuda.stream(st1):
            wd[:, :256].copy_(t1[:128])
            gv["layers_dir.0.bias"].copy_(b1[:128])
            gv["fc_alpha.weight"].copy_(t1[131:132])
            gv["fc_alpha.bias"].copy_(b1[131:132])
        with torch.cuda.stream(st2):
            gv["fc_rgb.weight"].copy_(t2[:3])
            gv["fc_rgb.bias"].copy_(b2[:3])
            if net.depth_head:
                gv["fc_mu_sigma.weight"].copy_(t2[4:6])
                gv["fc_mu_sigma.bias"].copy_(b2[4:6])
    else:
        job("dir", ROW_DIR, 128, ROW_FEAT, 256, 256, wd, 283, 0, gv["layers_dir.0.bias"])
        job("dir32", ROW_DIR, 128, ROW_X + 96, 32, 27, wd, 283, 256, None)
        # heads: d(raw) rows 0..2 rgb, 3 alpha, 4..5 mu/sigma (deltas rows ROW_X..)
        job("rgb", ROW_X, 3, ROW_DIR, 128, 128, gv["fc_rgb.weight"], 128, 0, gv["fc_rgb.bias"])
        job("alpha", ROW_X + 3, 1, ROW_FEAT, 256, 256, gv["fc_alpha.weight"], 256, 0, gv["fc_alpha.bias"])
        if net.depth_head:
            job("musigma", ROW_X + 4, 2, ROW_DIR, 128, 128, gv["fc_mu_sigma.weight"], 128, 0, gv["fc_mu_sigma.bias"])
    if pair:
        events = []
        for st, _ in lanes:
            done = torch.cuda.Event()
            done.record(st)
            events.append(done)
        again = any(entry[1] is net for entry in _DEFERRED)
        if again:   # a second backward node of the SAME network (one shared MLP): autograd ADDS its gradients to the first node's on the
            join_deferred(net)   # caller's stream right behind this call -- both must be complete
        if DEFER_JOIN and not again:
            # the caller promised that nothing reads these gradients before join_deferred(): its stream goes on (the other network's
            # backward chain) while the lanes finish -- the tail of a network's jobs (heads, reductions, copies) fills a fraction of the
            # chip.  Everything the lanes still read or write stays referenced until the join (the allocator must not hand it out).
            _DEFERRED.append((events, net, (acts, deltas, flat_g, lanes, locals().get("t1"), locals().get("b1"), locals().get("t2"), locals().get("b2"))))
        else:
            for done in events:  # the caller's stream continues behind both lanes (which also orders the frees of acts / deltas / ws)
                main.wait_event(done)
    return flat_g, views


# mlp_f32_weight_grads leaves its two lanes unjoined while this is set; join_deferred() joins them (train_step.TrainStepper: around
# loss.backward(), when no gradient reducer wants the buffers earlier)
DEFER_JOIN = False
_DEFERRED = []


def join_deferred(net=None):
    """the current stream waits for the weight-gradient lanes left unjoined under DEFER_JOIN -- all of them, or those of `net` --; their
    operands are released"""
    if _DEFERRED:
        main = torch.cuda.current_stream()
        keep = []
        for entry in _DEFERRED:
            if net is None or entry[1] is net:
                for e in entry[0]:
                    main.wait_event(e)
            else:
                keep.append(entry)
        _DEFERRED[:] = keep
