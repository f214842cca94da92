"""Parameter containers of the two MLP backbones, with the reference's constructor signature, sub-module
names, parameter shapes and registration order (models/base_architectures.py:3-37, 64-99) so that
state_dicts / checkpoints / optimizers are interchangeable.  `forward` runs the fused HIP kernel.

All parameters of a network are views into ONE flat fp32 buffer laid out in registration order -- the
"flat parameter buffer" of include/ddnerf_hip.h -- so the kernel-side repack reads one contiguous tensor and
the data-parallel gradient all-reduce moves one bucket per network."""
from __future__ import annotations

import torch

from . import functions as F


class _FusedMLP(torch.nn.Module):
    depth_head = False

    def __init__(self, num_layers=8, hidden_size=256, skip_connect_every=4, max_ipe_deg=16, num_encoding_fn_dir=4,
                 include_input_xyz=False, include_input_dir=False, use_viewdirs=True):
        super().__init__()
        # like the reference, num_layers / skip_connect_every are accepted and ignored (8 layers, skip at 5)
        self.dim_xyz = (3 if include_input_xyz else 0) + 2 * 3 * max_ipe_deg
        self.dim_dir = (3 if include_input_dir else 0) + 2 * 3 * num_encoding_fn_dir
        if (hidden_size, self.dim_xyz, self.dim_dir, bool(use_viewdirs)) != (256, 96, 27, True):
            raise ValueError("the fused HIP MLP is built for hidden 256, 96 IPE + 27 view-dir features "
                             "(got hidden=%s dim_xyz=%s dim_dir=%s)" % (hidden_size, self.dim_xyz, self.dim_dir))
        self.use_viewdirs = use_viewdirs
        self.layers_xyz = torch.nn.ModuleList()
        self.layers_xyz.append(torch.nn.Linear(self.dim_xyz, hidden_size))
        for i in range(1, 8):
            self.layers_xyz.append(torch.nn.Linear(self.dim_xyz + hidden_size if i == 5 else hidden_size, hidden_size))
        self.fc_feat = torch.nn.Linear(hidden_size, hidden_size)
        self.fc_alpha = torch.nn.Linear(hidden_size, 1)
        self.layers_dir = torch.nn.ModuleList([torch.nn.Linear(hidden_size + self.dim_dir, 128)])
        self.fc_rgb = torch.nn.Linear(128, 3)
        if self.depth_head:
            self.fc_mu_sigma = torch.nn.Linear(128, 2)
        self.mlp_dtype = "fp32"  # "fp32": exact-fp32 MFMA kernel; "x3": bf16 MFMA on exact hi/lo splits; "bf16" / "fp16": plain 16-bit MFMA kernels
        self._flat = None
        self._flatten()

    # -- flat parameter buffer -------------------------------------------------------------------------
    def _flatten(self):
        params = list(self.parameters())
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1).float())
                p.data = flat[off:off + n].view(p.shape)
                off += n
        self._flat = flat

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._flatten()
        return self

    def flat_params(self) -> torch.Tensor:
        """The flat fp32 buffer all parameters alias (re-made if someone re-pointed a .data)."""
        first = self.layers_xyz[0].weight
        last = self.fc_mu_sigma.bias if self.depth_head else self.fc_rgb.bias  # registration order: the last parameter
        if (self._flat is None or first.data_ptr() != self._flat.data_ptr()
                or last.data_ptr() + last.numel() * 4 != self._flat.data_ptr() + self._flat.numel() * 4):
            self._flatten()
        return self._flat

    def param_version(self) -> int:
        """changes whenever any parameter is written in place (Adam step, load_state_dict, manual edits)"""
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        return sum(p._version for p in plist)

    def invalidate_packed(self):
        """drop the kernel-format weight images (call after writing the flat buffer directly, e.g. a broadcast)"""
        self.__dict__.pop("_packed_cache", None)

    def forward(self, x):
        """x: [M,123] (`embedded`, models/models.py:133) or the 128-column padded feature rows -> [M,4|6]"""
        if x.shape[-1] == 123:
            x = torch.nn.functional.pad(x, (0, 5))
        return F.mlp(x, self)


class MipNeRFModel(_FusedMLP):
    """models/base_architectures.py:3-61: outputs (rgb3, alpha)"""
    depth_head = False


class DepthMipNeRFModel(_FusedMLP):
    """models/base_architectures.py:64-126: outputs (rgb3, alpha, raw mu, raw sigma)"""
    depth_head = True
