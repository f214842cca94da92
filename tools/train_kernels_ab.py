#!/usr/bin/env python3
"""Same-box A/B of builds of the fp32 tier's values-record training kernels (mlp_f32_train_recf.hip): launch times of
ddnerf_mlp_f32_forward_train_recf and ddnerf_mlp_f32_backward_data_recf at M = 524,288 (the fine pass), interleaved.

    python tools/train_kernels_ab.py build NAME [-DNAME ...]    (CPU: tools/lib/f32t_NAME.so)
    python tools/train_kernels_ab.py run NAME [NAME ...]        (GPU box; 'product' = the shipped library)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddnerf_amd", "csrc")


def so(name):
    return os.path.join(CSRC, "libddnerf_hip.so") if name == "product" else os.path.join(ROOT, "tools", "lib", "f32t_%s.so" % name)


def build(name, defs):
    os.makedirs(os.path.join(ROOT, "tools", "lib"), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function",
                           "-I" + os.path.join(ROOT, "include")] + defs
                          + ["-shared", os.path.join(CSRC, "mlp_f32.hip"), os.path.join(CSRC, "mlp_f32_train.hip"), os.path.join(CSRC, "mlp_f32_train_recf.hip"), "-o", so(name)])
    print(so(name))


def run(names):
    import torch
    M = 524288
    P, L = ctypes.c_void_p, ctypes.c_long
    torch.manual_seed(0)
    feat = torch.randn(M, 128, device="cuda")
    nout = [256] * 9 + [1, 128, 3, 2]
    nin = [96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128]
    params = torch.randn(sum(o * i + o for o, i in zip(nout, nin)), device="cuda") * 0.05
    acts = torch.empty(2560, M, device="cuda")
    deltas = torch.empty(2560, M, device="cuda")
    graw = torch.randn(M, 6, device="cuda")
    raw = torch.empty(M, 6, device="cuda")
    libs = {nm: ctypes.CDLL(so(nm)) for nm in names}
    fw, bw, ref = {}, {}, None
    for rep in range(4):
        for nm, lib in libs.items():
            for f in ("ddnerf_mlp_f32_packed_floats", "ddnerf_mlp_f32_packed_t_floats", "ddnerf_mlp_f32_sign_bytes"):
                getattr(lib, f).restype = ctypes.c_size_t
            packed = torch.empty(lib.ddnerf_mlp_f32_packed_floats(1), device="cuda")
            packed_t = torch.empty(lib.ddnerf_mlp_f32_packed_t_floats(1), device="cuda")
            signs = torch.empty(lib.ddnerf_mlp_f32_sign_bytes(L(M)), dtype=torch.uint8, device="cuda")
            assert lib.ddnerf_mlp_f32_pack(P(params.data_ptr()), 1, P(packed.data_ptr()), None) == 0
            assert lib.ddnerf_mlp_f32_pack_t(P(params.data_ptr()), 1, P(packed_t.data_ptr()), None) == 0
            go = lambda: lib.ddnerf_mlp_f32_forward_train_recf(P(feat.data_ptr()), P(packed.data_ptr()), 1, P(raw.data_ptr()), P(acts.data_ptr()),
                                                              P(signs.data_ptr()), None, 0, L(M), L(M), None)
            gob = lambda: lib.ddnerf_mlp_f32_backward_data_recf(P(graw.data_ptr()), P(packed_t.data_ptr()), P(acts.data_ptr()), P(signs.data_ptr()), 1,
                                                               P(deltas.data_ptr()), L(M), L(M), None)
            for fn, res in ((go, fw), (gob, bw)):
                for _ in range(3):
                    assert fn() == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(nm, []).append(e0.elapsed_time(e1) / 10)
            if rep == 0:   # every build must produce the same deltas (rows 0 .. 2437 of the record)
                d = deltas.view(-1, 2560, 16)[:, :2438].clone()
                if ref is None:
                    ref = d
                print("%-10s deltas equal to the first build's: %s" % (nm, bool(torch.equal(ref, d))), flush=True)
    for nm in names:
        print("%-10s forward_train_recf %.4f ms  backward_data_recf %.4f ms per launch (median of 4 x 10)   [%s | %s]"
              % (nm, sorted(fw[nm])[2], sorted(bw[nm])[2], " ".join("%.4f" % v for v in fw[nm]), " ".join("%.4f" % v for v in bw[nm])))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        run(sys.argv[2:])
