"""Image metrics of the evaluation harness.  The reference computes SSIM with scikit-image on the
grey-scale images (validation_utils/validation.py:7-17: cv2 RGB->grey, `structural_similarity` with its defaults);
scikit-image / cv2 are not available here, so this is that published algorithm in torch (runs on the device the images
are on): 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, mean over the window-valid interior.
LPIPS needs AlexNet weights that cannot be fetched offline and stays unavailable."""
from __future__ import annotations

import torch


def rgb_to_gray(img: torch.Tensor) -> torch.Tensor:
    """[H,W,3] -> [H,W] with the ITU-R 601 weights cv2.COLOR_RGB2GRAY uses"""
    w = torch.tensor([0.299, 0.587, 0.114], dtype=img.dtype, device=img.device)
    return (img[..., :3] * w).sum(-1)


def ssim_gray(x: torch.Tensor, y: torch.Tensor, data_range: float, win: int = 7) -> float:
    """skimage.metrics.structural_similarity(x, y, data_range=..., win_size=7, gaussian_weights=False,
    use_sample_covariance=True) for 2-D float images"""
    if x.shape != y.shape or x.dim() != 2:
        raise ValueError("ssim_gray wants two [H,W] images of the same shape")
    if min(x.shape) < win:
        raise ValueError("win_size exceeds image extent")
    x, y = x.double()[None, None], y.double()[None, None]
    n = win * win
    box = lambda t: torch.nn.functional.avg_pool2d(t, win, stride=1)   # window-valid positions only (= skimage's crop)
    ux, uy = box(x), box(y)
    cov_norm = n / (n - 1.0)
    vx = cov_norm * (box(x * x) - ux * ux)
    vy = cov_norm * (box(y * y) - uy * uy)
    vxy = cov_norm * (box(x * y) - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return float(s.mean())


def calc_ssim(image: torch.Tensor, target: torch.Tensor):
    """validation_utils/validation.py:7-17 -> (ssim_v1, ssim_v2): v1 = the legacy `compare_ssim` call without a
    data_range (scikit-image then assumes the float range [-1, 1], i.e. 2.0); v2 = data_range = max - min of the image"""
    gi, gt = rgb_to_gray(image.float()), rgb_to_gray(target.float())
    v1 = ssim_gray(gt, gi, 2.0)
    v2 = ssim_gray(gt, gi, float(gi.max() - gi.min()))
    return v1, v2
