"""ctypes binding of libddnerf_hip.so (the C ABI in include/ddnerf_hip.h).

There is no CPU fallback: if the library is missing or a kernel launch fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libddnerf_hip.so")

c_fp = C.c_void_p
_SIGS = {
    "ddnerf_abi_version": (C.c_int, []),
    "ddnerf_error_string": (C.c_char_p, [C.c_int]),
    "ddnerf_build_info": (C.c_char_p, []),
    "ddnerf_pack_rays": (C.c_int, [c_fp, c_fp, c_fp, C.c_float, C.c_float, c_fp, C.c_int, c_fp]),
    "ddnerf_sample_first_cycle": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp]),
    "ddnerf_encode": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp]),
    "ddnerf_encode_first_cycle": (C.c_int, [c_fp, c_fp, c_fp, C.c_float, C.c_float, c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int,
                                            C.c_int, c_fp]),
    "ddnerf_encode_rays": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp]),
    "ddnerf_encode_first_cycle_rays": (C.c_int, [c_fp, c_fp, c_fp, C.c_float, C.c_float, c_fp, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp]),
    "ddnerf_mlp_x3_forward_rays": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_f32_forward_rays": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_f32_packed_floats": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_f32_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_f32_forward": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_bf16_packed_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_bf16_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_bf16_forward": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_bf16g1_packed_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_bf16g1_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_bf16g1_forward": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_bf16g2_packed_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_bf16g2_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_bf16g2_forward": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_train_loss_forward": (C.c_int, [c_fp, c_fp, c_fp, C.c_long, c_fp, C.c_int, C.c_float, C.c_float, C.c_float, c_fp, c_fp]),
    "ddnerf_train_loss_backward": (C.c_int, [c_fp, c_fp, c_fp, C.c_long, C.c_int, C.c_float, C.c_float, C.c_float, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "ddnerf_ray_table_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_ray_table": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, c_fp]),
    "ddnerf_pack_rays_first_cycle_table": (C.c_int, [c_fp, c_fp, c_fp, C.c_float, C.c_float, c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, c_fp]),
    "ddnerf_encode_mlp_bf16_scratch_bytes": (C.c_size_t, []),
    "ddnerf_encode_mlp_bf16_forward": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_int, C.c_int, c_fp, c_fp]),
    "ddnerf_encode_mlp_f16_scratch_bytes": (C.c_size_t, []),
    "ddnerf_encode_mlp_f16_forward": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_int, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_f16_packed_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_f16_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_f16_forward": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_f16g1_packed_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_f16g1_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_f16g1_forward": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_f16g2_packed_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_f16g2_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_f16g2_forward": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_x3_packed_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_x3_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_x3_forward": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_long, c_fp]),
    "ddnerf_mlp_x3_forward_train": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_x3_forward_train_rays": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_x3_packed_t_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_x3_pack_t": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_x3_backward_data": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_x3e_packed_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_x3e_pack": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_x3e_forward_train": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_x3e_packed_t_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_x3e_pack_t": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_x3e_backward_data": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_dd_records_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ddnerf_dd_records": (C.c_int, [c_fp] * 4 + [C.c_int, C.c_int] + [c_fp] * 5 + [c_fp]),
    "ddnerf_dd_head_workspace_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "ddnerf_dd_head": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_float, C.c_float] + [c_fp] * 9 + [c_fp]),
    "ddnerf_composite_forward": (C.c_int, [c_fp, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int] + [c_fp] * 7 + [c_fp]),
    "ddnerf_sample_pdf": (C.c_int, [c_fp] * 5 + [C.c_int] * 4 + [c_fp]),
    "ddnerf_sample_pdf_mu_sigma": (C.c_int, [c_fp] * 8 + [C.c_float, C.c_float, c_fp, c_fp] + [C.c_int] * 4 + [c_fp]),
    "ddnerf_dp_loss_workspace_bytes": (C.c_size_t, [C.c_int]),
    "ddnerf_dp_loss_forward": (C.c_int, [c_fp] * 8 + [C.c_int] * 4 + [c_fp, c_fp, c_fp, c_fp, c_fp]),
    "ddnerf_composite_backward": (C.c_int, [c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]),
    "ddnerf_dd_head_backward": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_float, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "ddnerf_dp_loss_backward": (C.c_int, [c_fp] * 8 + [C.c_int] * 4 + [c_fp] * 5 + [c_fp]),
    "ddnerf_mlp_act_rows": (C.c_size_t, []),
    "ddnerf_mlp_f32_packed_t_floats": (C.c_size_t, [C.c_int]),
    "ddnerf_mlp_f32_pack_t": (C.c_int, [c_fp, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_f32_forward_train": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_f32_backward_data": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_f32_forward_train_rec": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_f32_backward_data_rec": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_f32_forward_train_recp": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_f32_backward_data_recp": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_f32_sign_bytes": (C.c_size_t, [C.c_long]),
    "ddnerf_mlp_f32_forward_train_recf": (C.c_int, [c_fp, c_fp, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_long, C.c_long, c_fp]),
    "ddnerf_mlp_f32_backward_data_recf": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_long, C.c_long, c_fp]),
    "ddnerf_ray_bundle": (C.c_int, [C.c_int, C.c_int, C.c_float, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "ddnerf_ndc_rays": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "ddnerf_ndc_depth_to_regular": (C.c_int, [C.c_long, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "ddnerf_pack_rays_first_cycle": (C.c_int, [c_fp, c_fp, c_fp, C.c_float, C.c_float, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp]),
    "ddnerf_dd_coarse_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ddnerf_dd_coarse_forward": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float] + [c_fp] * 20),
    "ddnerf_dd_coarse_sample_forward": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float] + [c_fp] * 19
                                        + [c_fp, c_fp, C.c_float, C.c_float, c_fp, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, c_fp]),
    "ddnerf_debug_philox_normal": (C.c_int, [c_fp, C.c_long, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, c_fp]),
    "ddnerf_composite_forward_keep": (C.c_int, [c_fp, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int] + [c_fp] * 8),
    "ddnerf_composite_forward_keep_rng": (C.c_int, [c_fp, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int] + [c_fp] * 7
                                          + [C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, c_fp]),
    "ddnerf_dp_loss_forward_kept": (C.c_int, [c_fp] * 8 + [C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "ddnerf_mlp_f32_wgrad_workspace_floats": (C.c_size_t, [C.c_long]),
    "ddnerf_mlp_f32_wgrad": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, c_fp,
                                       C.c_int, C.c_int, c_fp, c_fp, c_fp]),
    "ddnerf_mlp_x3_wgrad": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, c_fp,
                                       C.c_int, C.c_int, c_fp, c_fp, c_fp]),
    "ddnerf_mlp_x3_wgrad_packed": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, c_fp,
                                              C.c_int, C.c_int, c_fp, c_fp, C.c_int, c_fp]),
    "ddnerf_mlp_x3_wgrad_packed_skip": (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_long, C.c_long, c_fp, c_fp, c_fp, C.c_int, c_fp]),
    "ddnerf_mlp_x3_wgrad_pairs": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, c_fp,
                                            C.c_int, C.c_int, c_fp, c_fp, C.c_int, c_fp]),
    "ddnerf_mlp_x3_wgrad_pairs_skip": (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_long, C.c_long, c_fp, c_fp, c_fp, C.c_int, c_fp]),
    "ddnerf_mlp_x3_wgrad_blocked": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, c_fp,
                                              C.c_int, C.c_int, c_fp, c_fp, C.c_int, c_fp]),
    "ddnerf_mlp_x3_wgrad_blocked_skip": (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_long, C.c_long, c_fp, c_fp, c_fp, C.c_int, c_fp]),
    "ddnerf_mlp_x3_split_pairs": (C.c_int, [c_fp, C.c_int, C.c_long, C.c_int, c_fp, c_fp]),
    "ddnerf_mlp_x3_split": (C.c_int, [c_fp, C.c_int, C.c_long, C.c_int, c_fp, c_fp]),
}

_lib = None


class DDNerfHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise DDNerfHipError(
                "libddnerf_hip.so not found at %s -- build it with `python -m ddnerf_amd.build` "
                "(there is no CPU fallback)" % SO_PATH)
        l = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, what: str):
    if code != 0:
        msg = lib().ddnerf_error_string(code)
        raise DDNerfHipError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))
