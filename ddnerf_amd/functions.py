"""Differentiable front ends of the HIP kernels (torch.autograd.Function pairs a forward kernel with its
hand-written backward kernel; PyTorch owns the graph, the optimiser and the RNG -- SURVEY.md 7).

Which gradients exist is dictated by the reference's graph: nothing flows through the
samplers or the encoder inputs; the MLP needs weight gradients only (no gradient w.r.t. its 123 input
features); compositing needs d(raw) from d(rgb_map) and d(weights); the DD head from d(mus), d(sigmas) and
the two regularisers."""
from __future__ import annotations

import torch

from . import ops


# ---- MLP -------------------------------------------------------------------------------------------------
_pack_cache = {}


def _packed_weights(net):
    flat = net.flat_params()
    key = (id(net), net.mlp_dtype)
    tag = (flat.data_ptr(), flat._version, net.mlp_dtype)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    if net.mlp_dtype == "fp32":
        packed = ops.mlp_f32_pack(flat, net.depth_head)
    else:
        packed = ops.mlp_bf16_pack(flat, net.depth_head)
    _pack_cache[key] = (tag, packed)
    return packed


class _MLPFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, net, *params):
        packed = _packed_weights(net)
        if net.mlp_dtype == "fp32":
            raw = ops.mlp_f32_forward(feat, packed, net.depth_head)
        else:
            raw = ops.mlp_bf16_forward(feat, packed, net.depth_head)
        ctx.net = net
        ctx.save_for_backward(feat)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        raise NotImplementedError("MLP backward kernel (K2b) not built yet")


def mlp(feat, net):
    """feat [M,128] (fp32, or bf16 for the bf16 kernel) -> raw [M,4|6]"""
    if torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters()):
        return _MLPFunction.apply(feat, net, *net.parameters())
    packed = _packed_weights(net)
    if net.mlp_dtype == "fp32":
        return ops.mlp_f32_forward(feat, packed, net.depth_head)
    return ops.mlp_bf16_forward(feat, packed, net.depth_head)


# ---- DD head ---------------------------------------------------------------------------------------------
def dd_head(raw6, smooth, dist_reg):
    return ops.dd_head(raw6, smooth, dist_reg)


# ---- compositing -----------------------------------------------------------------------------------------
def composite(raw, t_vals, rays, noise, mus, white_bkgd, blender):
    return ops.composite_forward(raw, t_vals, rays, noise, mus, white_bkgd, blender)


# ---- dp loss ---------------------------------------------------------------------------------------------
def dp_loss(t1, t0, w1, w0, mus0, sig0, left0, part0, blender):
    return ops.dp_loss_forward(t1, t0, w1, w0, mus0, sig0, left0, part0, blender)
