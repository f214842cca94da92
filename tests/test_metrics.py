"""SSIM of the eval harness (ddnerf_amd/metrics.py) against a direct numpy evaluation of scikit-image's published
defaults (7x7 uniform window through scipy.ndimage.uniform_filter, sample covariance, cropped mean)."""
import numpy as np
import torch
from scipy.ndimage import uniform_filter

from ddnerf_amd import metrics


def _ssim_numpy(x, y, data_range, win=7):
    x, y = x.astype(np.float64), y.astype(np.float64)
    n = win * win
    cov_norm = n / (n - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    vx = cov_norm * (uniform_filter(x * x, size=win) - ux * ux)
    vy = cov_norm * (uniform_filter(y * y, size=win) - uy * uy)
    vxy = cov_norm * (uniform_filter(x * y, size=win) - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return s[pad:-pad, pad:-pad].mean()


def test_ssim_matches_the_published_algorithm():
    rng = np.random.default_rng(0)
    a = rng.random((37, 52, 3)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape).astype(np.float32), 0, 1)
    ga = (a * np.array([0.299, 0.587, 0.114], np.float32)).sum(-1)
    gb = (b * np.array([0.299, 0.587, 0.114], np.float32)).sum(-1)
    v1, v2 = metrics.calc_ssim(torch.from_numpy(b), torch.from_numpy(a))
    assert abs(v1 - _ssim_numpy(ga, gb, 2.0)) < 1e-6
    assert abs(v2 - _ssim_numpy(ga, gb, float(gb.max() - gb.min()))) < 1e-6
    same1, same2 = metrics.calc_ssim(torch.from_numpy(a), torch.from_numpy(a))
    assert abs(same1 - 1.0) < 1e-12 and abs(same2 - 1.0) < 1e-12
    assert v2 < v1 < 1.0      # a larger assumed range hides more of the difference
