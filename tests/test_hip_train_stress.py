"""The fp32 tier's values-record training kernels move their ReLU masks through the scalar data cache (scalar stores in the forward, scalar
loads in the backward: the only such path in the library).  A short version of tools/stress_train.py: forward + backward pairs that reuse
the same record buffers with fresh inputs, from one ragged tile to the coarse pass, every delta and sign record compared bit for bit with
the hi/lo-word path / the recorded activations (profiles/r05_stress_train.log holds the long run)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_values_record_kernels_under_buffer_reuse():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import stress_train

    assert stress_train.run(24, hammer=True, sizes=(1, 127, 128, 129, 200, 4096, 33333, 262144), max_random=20000, log=lambda m: None) == 0
