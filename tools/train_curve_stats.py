#!/usr/bin/env python3
"""How far the HIP path's training curves are from the reference's (tests/golden/train*_*.npz) per training tier: max / mean |d PSNR| over
the recorded iterations, the difference of the means over the last six records, max relative loss difference.  GPU box:
python3 tools/train_curve_stats.py [fixture ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_baseline_size as T  # noqa: E402


class MP:
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


def main():
    names = sys.argv[1:] or ["train1500_dd_blender", "train300_dd_blender", "train300_mip_blender"]
    for name in names:
        for tier in ("fp32", "x3", "x3-exact"):
            loss, mse, g = T._training_curve(name, tier, MP())
            d = np.abs(T._psnr(mse) - T._psnr(g["mse"]))
            last = abs(T._psnr(mse[-6:, 1]).mean() - T._psnr(g["mse"][-6:, 1]).mean())
            rl = np.abs(loss - g["loss"]) / np.abs(g["loss"])
            print("%-22s %-8s max|dPSNR| %.3f dB (record %d of %d)  mean %.3f  last-six-mean diff %.3f  max rel loss diff %.3f  ref PSNR %.1f -> %.1f dB"
                  % (name, tier, d.max(), int(d.argmax()) // 2, len(d), d.mean(), last, rl.max(), T._psnr(g["mse"][0, 1]), T._psnr(g["mse"][-1, 1])), flush=True)


if __name__ == "__main__":
    main()
