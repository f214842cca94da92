"""Plain-PyTorch fp32 restatements of the floating-point kernels, used ONLY by the tests as the autograd
reference for the hand-written backward kernels (the forward values are pinned by the oracle / golden vectors)."""
import torch


def composite(raw, t_vals, rd_norm, noise, white, blender):
    """-> rgb_map, weights (as returned: incl. the detached blender epsilon)"""
    delta = (t_vals[:, 1:] - t_vals[:, :-1]) * rd_norm[:, None]
    rgb = torch.sigmoid(raw[..., :3]) * 1.002 - 0.001
    dens = raw[..., 3] + (noise if noise is not None else 0.0)
    sig = torch.nn.functional.softplus(dens - 1)
    alpha = 1 - torch.exp(-sig * delta)
    T = torch.cumprod(1 - alpha + 1e-10, -1)
    T = torch.cat([torch.ones_like(T[:, :1]), T[:, :-1]], -1)
    w = alpha * T
    rgb_map = (w[..., None] * rgb).sum(-2)
    acc = w.sum(-1)
    if blender:
        eps = torch.zeros_like(w)
        eps[:, -1] = 1e-10
        w = w + eps
        acc = w.sum(-1)
    if white:
        rgb_map = rgb_map + (1 - acc[:, None])
    return rgb_map, w


def dd_head(raw6, dist_reg):
    rm, rs = raw6[..., 4], raw6[..., 5]
    mus, sig = torch.sigmoid(rm), torch.sigmoid(rs) + 0.001
    n = raw6.shape[0]
    ml, sl = (rm ** 2).sum() / n, (rs ** 2).sum() / n
    return mus, sig, torch.stack([ml, sl, dist_reg * ml, dist_reg * sl])


def mlp(x, sd, depth_head, pre=None):
    """`pre` (a list) collects the pre-activations of the nine ReLU layers"""
    lin = lambda h, n: h @ sd[n + ".weight"].t() + sd[n + ".bias"]
    xyz, dirs = x[:, :96], x[:, 96:123]
    h = xyz
    for i in range(8):
        z = lin(torch.cat([xyz, h], -1) if i == 5 else h, "layers_xyz.%d" % i)
        if pre is not None:
            pre.append(z)
        h = torch.relu(z)
    feat = lin(h, "fc_feat")
    alpha = lin(feat, "fc_alpha")
    z = lin(torch.cat([feat, dirs], -1), "layers_dir.0")
    if pre is not None:
        pre.append(z)
    hd = torch.relu(z)
    outs = [lin(hd, "fc_rgb"), alpha]
    if depth_head:
        outs.append(lin(hd, "fc_mu_sigma"))
    return torch.cat(outs, -1)


def dp_loss(t1, t0, w1, w0, mus0, sig0, left0, part0, blender):
    """models/dd_utils.py:6-78 in plain torch (any dtype), incl. the un-filtered left_tails gather."""
    import math
    if blender:
        rows = w1.sum(1) > 1e-10
        if rows.sum() == 0:
            return w0.sum() * 0
        w0, w1, mus0, sig0, part0, t1, t0 = w0[rows], w1[rows], mus0[rows], sig0[rows], part0[rows], t1[rows], t0[rows]
    eps = 1e-12
    p0 = (w0 + eps) / (w0 + eps).sum(-1, keepdim=True)
    p1 = (w1 + eps) / (w1 + eps).sum(-1, keepdim=True)
    mr = t0[:, :-1] + mus0 * (t0[:, 1:] - t0[:, :-1])
    sr = sig0 * (t0[:, 1:] - t0[:, :-1])
    cdf = torch.minimum(torch.ones((), dtype=p0.dtype, device=p0.device), torch.cumsum(p0[:, :-1], -1))
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf, torch.ones_like(cdf[:, :1])], -1)
    mask = t1[:, None, :] > t0[:, :, None]
    est, idx = torch.max(torch.where(mask, cdf[:, :, None], cdf[:, :1, None]), -2)
    g = lambda x: torch.gather(x, index=idx, dim=-1)
    x = (t1 - g(mr)) / g(sr)
    phi = 0.5 * (1 + torch.erf(x / math.sqrt(2.0)))
    est = est + ((phi - g(left0[: w0.shape[0]])) / g(part0)) * g(p0)
    est = torch.where(est > 1, torch.ones_like(est), est)
    d = est[:, 1:] - est[:, :-1]
    d = torch.where(d < 0, torch.zeros_like(d), d)
    q = (d + eps) / (d + eps).sum(-1, keepdim=True)
    return torch.nn.functional.kl_div(q.log(), p1, reduction="mean")
