"""Differentiable front ends of the HIP kernels (torch.autograd.Function pairs a forward kernel with its
hand-written backward kernel; PyTorch owns the graph, the optimiser and the RNG -- SURVEY.md 7).

Which gradients exist is dictated by the reference's graph: nothing flows through the
samplers or the encoder inputs; the MLP needs weight gradients only (no gradient w.r.t. its 123 input
features); compositing needs d(raw) from d(rgb_map) and d(weights); the DD head from d(mus), d(sigmas) and
the two regularisers; the dp loss w.r.t. (w0, mus0, sig0).  Outputs nothing in the reference's losses reads
(disp, acc, depth, corrected disparity, the Phi tails) are marked non-differentiable."""
from __future__ import annotations

import torch

from torch.optim.optimizer import register_optimizer_step_post_hook

from . import ops

# ---- MLP -------------------------------------------------------------------------------------------------
# Fused optimisers (torch.optim.Adam(fused=True): one kernel per step, what TrainStepper uses on the GPU) update the parameters
# WITHOUT bumping their version counters, so the counters alone would leave a stale weight image behind the first step.  Every
# optimiser step in the process therefore advances an epoch that is part of the cache tag (an optimiser of another model costs
# one spare repack, 26 us).
_OPTIMIZER_EPOCH = [0]


def _optimizer_stepped(*_args, **_kwargs):
    _OPTIMIZER_EPOCH[0] += 1


register_optimizer_step_post_hook(_optimizer_stepped)


def _cached_pack(net, kind, builder):
    """kernel-format weights, repacked only when a parameter changed (optimizer step, load_state_dict).
    The cache lives ON the module: a global table keyed by id(net) would hand a new model the packed weights of a
    garbage-collected one that happened to get the same id and buffer address.  The parameters alias the flat buffer
    through `.data`, so in-place updates bump THEIR version counters, not the buffer's: the tag sums those (and carries the
    optimiser epoch above)."""
    flat = net.flat_params()
    cache = net.__dict__.setdefault("_packed_cache", {})
    tag = (flat.data_ptr(), net.param_version(), _OPTIMIZER_EPOCH[0])
    hit = cache.get(kind)
    if hit is not None and hit[0] == tag:
        return hit[1]
    packed = builder(flat, net.depth_head)
    cache[kind] = (tag, packed)
    return packed


_PACK = {"fp32": ops.mlp_f32_pack, "x3": ops.mlp_x3_pack, "bf16": ops.mlp_bf16_pack, "fp16": ops.mlp_f16_pack}
_FORWARD = {"fp32": ops.mlp_f32_forward, "x3": ops.mlp_x3_forward, "bf16": ops.mlp_bf16_forward, "fp16": ops.mlp_f16_forward}


def _packed_weights(net):
    return _cached_pack(net, net.mlp_dtype, _PACK[net.mlp_dtype])


def _forward_kernel(feat, net):
    return _FORWARD[net.mlp_dtype](feat, _packed_weights(net), net.depth_head)


def x3_wgrad_exact():
    """DDNERF_X3_WGRAD=exact: the x3 training tier records exact hi/lo words and its weight gradients run three MFMAs per product
    (fp32-class parameter gradients: 3e-4 of the norm against fp64 autograd) instead of the default's bf16 row pairs with one MFMA per
    product (2.2e-3 of the norm; 7.2 ms against 9.9 ms per step at BASELINE size).  Read per call, so a test can switch it."""
    import os

    return os.environ.get("DDNERF_X3_WGRAD", "pairs") == "exact"


class _MLPFunction(torch.autograd.Function):
    """training forward + hand-written backward of one network.  `mlp_dtype` "fp32" runs the exact fp32-MFMA kernels (forward and
    backward-data exact; weight gradients as bf16 hi+lo splits, three MFMAs per product: fp32-class).  "x3" runs the split-precision
    bf16-MFMA kernels: forward and backward-data at fp32-class accuracy (three MFMAs per product on exact hi/lo splits), and weight
    gradients that by DEFAULT contract bf16-ROUNDED activation / delta records (one MFMA per product: 2.2e-3 of the gradient's norm
    from the fp32 tier's -- NOT the fp32 tier's accuracy class; 2.7x its speed), or with DDNERF_X3_WGRAD=exact the exact hi/lo-word
    records (three MFMAs per product: the fp32 tier's class again, 2.0x its speed).  The fp32 tier's weight gradients follow
    ops.WGRAD_MODE."""

    @staticmethod
    def forward(ctx, feat, net, dirs, S, *params):
        if net.mlp_dtype in ("bf16", "fp16"):
            raise NotImplementedError("training runs on the fp32 / x3 MLP kernels; the plain bf16 / fp16 kernels are inference-only")
        ctx.x3 = net.mlp_dtype == "x3"
        ctx.x3e = ctx.x3 and x3_wgrad_exact()
        ctx.rec = False
        if dirs is not None and ctx.x3e:
            raise NotImplementedError("per-ray view directions in training: not with DDNERF_X3_WGRAD=exact (mlp_rays_trainable)")
        if ctx.x3e:
            raw, acts, bits = ops.mlp_x3e_forward_train(feat, _cached_pack(net, "x3e", ops.mlp_x3e_pack), net.depth_head)
            ctx.save_for_backward(feat, acts, bits)
        elif ctx.x3:
            raw, acts, bits = ops.mlp_x3_forward_train(feat, _packed_weights(net), net.depth_head, dirs=dirs, S=S)
            ctx.save_for_backward(feat, acts, bits)
        else:
            # with the default (bf16x3) weight gradients the exact-fp32 kernels write blocked RECORDS for the record-operand kernel
            # DDNERF_WGRAD: "x3" (default) records of the fp32 values, split into hi/lo bf16 by the weight-gradient kernel, three MFMAs per
            # product: fp32-class; "x3words": the same gradients from records of hi/lo words split by the recording kernels (round 4's
            # form); "pairs": bf16 row-pair records, one MFMA per product (an opt-in speed mode: bf16-rounded operands); "f32": fp32
            # matrices and the fp32-MFMA kernel
            ctx.rec = {"x3": "values", "x3words": "hilo", "pairs": "pairs"}.get(ops.WGRAD_MODE, False)
            if dirs is not None and ctx.rec != "values":
                raise NotImplementedError("per-ray view directions in training: the fp32 tier's values-record kernels only (mlp_rays_trainable)")
            if ctx.rec == "values":
                raw, acts, signs = ops.mlp_f32_forward_train(feat, _packed_weights(net), net.depth_head, rec=ctx.rec, dirs=dirs, S=S)
                ctx.save_for_backward(feat, acts, signs)
            else:
                raw, acts = ops.mlp_f32_forward_train(feat, _packed_weights(net), net.depth_head, rec=ctx.rec)
                ctx.save_for_backward(feat, acts)
        net._fwd_calls = getattr(net, "_fwd_calls", 0) + 1
        ctx.net = net
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        net = ctx.net
        g_raw = g_raw.contiguous()
        if ctx.x3e:
            feat, acts, bits = ctx.saved_tensors
            deltas = ops.mlp_x3e_backward_data(g_raw, _cached_pack(net, "x3e_t", ops.mlp_x3e_pack_t), bits, net.depth_head)
        elif ctx.x3:
            feat, acts, bits = ctx.saved_tensors
            packed_t = _cached_pack(net, "x3_t", ops.mlp_x3_pack_t)
            deltas = ops.mlp_x3_backward_data(g_raw, packed_t, bits, net.depth_head)
        else:
            feat, acts, *signs = ctx.saved_tensors
            packed_t = _cached_pack(net, "fp32_t", ops.mlp_f32_pack_t)
            deltas = ops.mlp_f32_backward_data(g_raw, packed_t, acts, net.depth_head, rec=ctx.rec, signs=signs[0] if signs else None)
        # (the x3 tier's records of bf16 row pairs and the fp32 tier's records of hi/lo words go to the record-operand weight-gradient
        # kernel; DDNERF_WGRAD=f32 keeps fp32 matrices on the fp32 tier)
        flat_g, views = ops.mlp_f32_weight_grads(net, acts, deltas, g_raw.shape[0], mode="x3p" if (ctx.x3e or ctx.rec == "hilo") else ("x3h" if (ctx.x3 or ctx.rec == "pairs") else ("x3b" if ctx.rec == "values" else None)))
        net.last_flat_grad = flat_g  # the data-parallel bucket (ddnerf_amd.dist) reduces this buffer
        reducer = getattr(net, "grad_reducer", None)
        if reducer is not None:
            reducer.on_flat_grad_ready(net, flat_g)
        return (None, None, None, None) + tuple(views)


def mlp_rays_trainable(net):
    """training with the view-direction columns from a per-ray table: the fp32 tier's default (values-record) kernels and the x3 tier's
    default (row-pair) kernels take them"""
    return (net.mlp_dtype == "fp32" and ops.WGRAD_MODE == "x3") or (net.mlp_dtype == "x3" and not x3_wgrad_exact())


def needs_grad(net):
    return torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters())


def encode_mlp_bf16(table, t_vals, net):
    """models/models.py:117-142 in one launch (bf16 tier, inference): ray table [n,32] + fenceposts [n,S+1] -> raw [n*S,4|6]"""
    return ops.encode_mlp_bf16_forward(table, t_vals, _packed_weights(net), net.depth_head, kind=net.mlp_dtype)


def mlp_rays(feat, dirs, S, net):
    """the fp32 / x3 kernels with the view-direction columns from the per-ray table `dirs` [n,32] (ops.encode_rays): inference, or -- the
    tiers' default training kernels, mlp_rays_trainable -- a training forward"""
    if needs_grad(net):
        return _MLPFunction.apply(feat, net, dirs, int(S), *net.parameters())
    fwd = {"fp32": ops.mlp_f32_forward_rays, "x3": ops.mlp_x3_forward_rays}[net.mlp_dtype]
    return fwd(feat, dirs, S, _packed_weights(net), net.depth_head)


def mlp(feat, net):
    """feat [M,128] (fp32, or k-ordered bf16 / fp16 for the bf16 / fp16 kernels) -> raw [M,4|6]"""
    if torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters()):
        return _MLPFunction.apply(feat, net, None, 0, *net.parameters())
    return _forward_kernel(feat, net)


# ---- DD head ---------------------------------------------------------------------------------------------
class _DDHeadFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw6, smooth, dist_reg):
        d = ops.dd_head(raw6, smooth, dist_reg)
        ctx.save_for_backward(raw6)
        ctx.dist_reg = float(dist_reg)
        ctx.set_materialize_grads(False)      # (an output nobody differentiates arrives as None, not as a zero-filled tensor: the kernel takes NULL)
        outs = (d["mus"], d["sigmas"], d["left"], d["part"], d["ssig"], d["sleft"], d["spart"], d["scal"])
        ctx.mark_non_differentiable(*outs[2:7])
        return outs

    @staticmethod
    def backward(ctx, g_mus, g_sigmas, _gl, _gp, _gss, _gsl, _gsp, g_scal):
        (raw6,) = ctx.saved_tensors
        if g_mus is None and g_sigmas is None and g_scal is None:
            return None, None, None
        g_raw6 = torch.zeros_like(raw6)
        ops.dd_head_backward_(raw6, ctx.dist_reg, g_mus, g_sigmas, g_scal, g_raw6)
        return g_raw6, None, None


def dd_head(raw6, smooth, dist_reg):
    if torch.is_grad_enabled() and raw6.requires_grad:
        o = _DDHeadFunction.apply(raw6, float(smooth), float(dist_reg))
        return dict(zip(("mus", "sigmas", "left", "part", "ssig", "sleft", "spart", "scal"), o))
    return ops.dd_head(raw6, smooth, dist_reg)


# ---- compositing -----------------------------------------------------------------------------------------
class _CompositeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, t_vals, rays, noise, mus, white_bkgd, blender):
        c = ops.composite_forward(raw, t_vals, rays, noise, mus, white_bkgd, blender)
        ctx.save_for_backward(raw, t_vals, rays, noise)
        ctx.flags = (bool(white_bkgd), bool(blender))
        ctx.set_materialize_grads(False)      # (the fine pass's weights carry no gradient: None instead of a zero-filled [n,S] tensor)
        ctx.has_cdisp = c["cdisp"] is not None
        nd = [c["disp"], c["acc"], c["depth"]] + ([c["cdisp"]] if ctx.has_cdisp else [])
        ctx.mark_non_differentiable(*nd)
        if ctx.has_cdisp:
            return c["rgb_map"], c["weights"], c["disp"], c["acc"], c["depth"], c["cdisp"]
        return c["rgb_map"], c["weights"], c["disp"], c["acc"], c["depth"]

    @staticmethod
    def backward(ctx, g_rgb_map, g_weights, *_unused):
        raw, t_vals, rays, noise = ctx.saved_tensors
        if g_rgb_map is None and g_weights is None:
            return None, None, None, None, None, None, None
        if g_rgb_map is None:
            g_rgb_map = torch.zeros((raw.shape[0], 3), dtype=torch.float32, device=raw.device)
        g_raw = ops.composite_backward(raw, t_vals, rays, noise, ctx.flags[0], ctx.flags[1], g_rgb_map, g_weights)
        return g_raw, None, None, None, None, None, None


def composite(raw, t_vals, rays, noise, mus, white_bkgd, blender):
    """-> dict(rgb_map, disp, acc, weights, depth, cdisp)"""
    if torch.is_grad_enabled() and raw.requires_grad:
        o = _CompositeFunction.apply(raw, t_vals, rays, noise, None if mus is None else mus.detach(), white_bkgd, blender)
        d = dict(rgb_map=o[0], weights=o[1], disp=o[2], acc=o[3], depth=o[4], cdisp=o[5] if len(o) > 5 else None)
        return d
    return ops.composite_forward(raw, t_vals, rays, noise, mus, white_bkgd, blender)


# ---- dp loss ---------------------------------------------------------------------------------------------
class _DPLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w0, mus0, sig0, t1, t0, w1, left0, part0, blender):
        loss = ops.dp_loss_forward(t1, t0, w1, w0, mus0, sig0, left0, part0, blender)
        ctx.save_for_backward(w0, mus0, sig0, t1, t0, w1, left0, part0)
        ctx.blender = bool(blender)
        return loss

    @staticmethod
    def backward(ctx, g):
        w0, mus0, sig0, t1, t0, w1, left0, part0 = ctx.saved_tensors
        gw, gm, gs = ops.dp_loss_backward(t1, t0, w1, w0, mus0, sig0, left0, part0, ctx.blender, g)
        return gw, gm, gs, None, None, None, None, None, None


def dp_loss(t1, t0, w1, w0, mus0, sig0, left0, part0, blender):
    """models/dd_utils.py:6-78; differentiable w.r.t. (w0, mus0, sig0) like the reference's call site"""
    if torch.is_grad_enabled() and (w0.requires_grad or mus0.requires_grad or sig0.requires_grad):
        return _DPLossFunction.apply(w0, mus0, sig0, t1, t0, w1, left0, part0, blender)
    return ops.dp_loss_forward(t1, t0, w1, w0, mus0, sig0, left0, part0, blender)


# ---- the training loss -------------------------------------------------------------------------------------
class _TrainLossFunction(torch.autograd.Function):
    """train_model.py:156-172: c0 mse(rgb_coarse, target) + c1 mse(rgb_fine, target) + c_dp mean(dp_loss) as ONE launch, its gradient as one
    (ops.train_loss_forward / _backward; the torch op chain it replaces was ~26 small launches per step)"""

    @staticmethod
    def forward(ctx, rgb0, rgb1, target, dp, c0, c1, c_dp):
        out = ops.train_loss_forward(rgb0, rgb1, target, dp, c0, c1, c_dp)
        ctx.save_for_backward(rgb0, rgb1, target)
        ctx.coef = (float(c0), float(c1), float(c_dp))
        ctx.n_dp = 0 if dp is None else dp.numel()
        ctx.dp_shape = None if dp is None else dp.shape
        parts = out[1:]
        ctx.mark_non_differentiable(parts)
        return out[0], parts

    @staticmethod
    def backward(ctx, g, _g_parts):
        rgb0, rgb1, target = ctx.saved_tensors
        g0, g1, gd = ops.train_loss_backward(rgb0, rgb1, target, ctx.n_dp, *ctx.coef, g.reshape(1))
        return g0.view_as(rgb0), None if g1 is None else g1.view_as(rgb1), None, None if gd is None else gd.view(ctx.dp_shape), None, None, None


def train_loss(rgb0, rgb1, target, dp, c0, c1, c_dp):
    """-> (loss [0-dim, differentiable w.r.t. rgb0, rgb1, dp], parts [3] = (mse0, mse1, mean dp), detached)"""
    return _TrainLossFunction.apply(rgb0, rgb1, target, dp, float(c0), float(c1), float(c_dp))
