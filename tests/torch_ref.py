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


def mlp(x, sd, depth_head):
    lin = lambda h, n: h @ sd[n + ".weight"].t() + sd[n + ".bias"]
    xyz, dirs = x[:, :96], x[:, 96:123]
    h = xyz
    for i in range(8):
        h = torch.relu(lin(torch.cat([xyz, h], -1) if i == 5 else h, "layers_xyz.%d" % i))
    feat = lin(h, "fc_feat")
    alpha = lin(feat, "fc_alpha")
    hd = torch.relu(lin(torch.cat([feat, dirs], -1), "layers_dir.0"))
    outs = [lin(hd, "fc_rgb"), alpha]
    if depth_head:
        outs.append(lin(hd, "fc_mu_sigma"))
    return torch.cat(outs, -1)
