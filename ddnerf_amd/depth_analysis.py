"""The `depth_analysis_validation=True` outputs of `run_iter` (reference models/models.py:108-112, 307-319): per-ray density
histograms over 1000 cells between dataset.near and dataset.far, drawn by validation_utils/visualization.py.  Debug
visualisation of a handful of rays, off the hot path: plain torch, results on the CPU like the reference's
(general_utils/math_utils.py:210-277 builds them in CPU tensors).  Vectorised over the sampling intervals (every fine cell
belongs to at most one interval, so the reference's loop adds one non-zero term per cell), with the reference's quirks kept:
an interval that contains no cell centre turns the whole uniform row into NaN (0 * p / 0), and zero cells of the Gaussian
histogram are filled with the mean of their two neighbours."""
from __future__ import annotations

import torch

N_CELLS = 1000


def _norm_cdf(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5))  # general_utils/math_utils.py:193-199


def uniform_incell_pdf(t_vals, weights, near, far):
    """math_utils.py:210-233 -> [n, 1000] (CPU)"""
    t, w = t_vals.detach().cpu(), weights.detach().cpu()
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    bins = torch.linspace(near, far, N_CELLS).reshape(1, 1, -1)
    inside = (bins >= t[:, :-1, None]) & (bins < t[:, 1:, None])      # [n, S, 1000]: cell centres of every interval
    count = inside.sum(-1, keepdim=True)
    return (inside * pdf[:, :, None] / count).sum(1)                    # empty interval: 0 * p / 0 = NaN for the whole row


def gaussian_incell_pdf(t_vals, weights, mus, sigmas, part_inside, near, far):
    """math_utils.py:236-277 -> [n, 1000] (CPU)"""
    t, w = t_vals.detach().cpu(), weights.detach().cpu()
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    width = t[:, 1:] - t[:, :-1]
    mu = (t[:, :-1] + mus.detach().cpu() * width)[:, :, None]          # section space -> ray space
    sg = (sigmas.detach().cpu() * width)[:, :, None]
    edges = torch.linspace(near, far, N_CELLS + 1)
    x0, x1 = edges[:-1].reshape(1, 1, -1), edges[1:].reshape(1, 1, -1)
    inside = (x0 >= t[:, :-1, None]) & (x1 <= t[:, 1:, None])
    mass = (_norm_cdf((x1 - mu) / sg) - _norm_cdf((x0 - mu) / sg)) * (1 / part_inside.detach().cpu()[:, :, None])
    est = (inside * mass * pdf[:, :, None]).sum(1)
    # zero cells take the mean of their neighbours, read before any of them is written (one indexed assignment upstream)
    left = torch.cat((est[:, :1], est[:, :-1]), 1)
    right = torch.cat((est[:, 1:], est[:, -1:]), 1)
    return torch.where(est == 0, (right + left) / 2, est)
