#!/usr/bin/env python3
"""Runs the two-group bf16 kernel of a library back to back (for rocprofv3 counter passes): python3 tools/g2_run.py <lib.so> [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import g2_clock  # noqa: E402

L, launch, keep = g2_clock.setup(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for _ in range(n):
    launch()
torch.cuda.synchronize()
print("done", n)
