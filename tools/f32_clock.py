#!/usr/bin/env python3
"""Where the fp32 MLP kernel's non-MFMA cycles go: clock stamps per weight slice.

(The per-slice stamp build, -DF32_STAMP, keeps a 4-byte index in LDS: since the forward parks its xyz columns there it uses exactly the
CU's 160 KB and that build no longer links -- its numbers in profiles/r04_f32_fwd_experiments.log are of the kernels before that step.
The tile-level build, -DF32_STAMP_TILE, and the ingredient builds, -DF32_EXP_*, are unaffected.)

    python tools/f32_clock.py build [-DNAME ...]   (CPU: tools/lib/f32_stamp.so = mlp_f32.hip + api-free, -DF32_STAMP and any extra defines)
    python tools/f32_clock.py run                  (GPU box: one fine-pass launch, per-slice table from 64 workgroups)

Stamps (mlp_f32_common.h, F32_STAMP): 0 slice entry, 1 after chunk 0, 2 after chunk 1, 3 after chunk NQ/2-1, 4 after chunk NQ-3,
5 after chunk NQ-2, 6 behind the barrier and the carry reads, 7 after the last chunk; a chunk = 4 MFMAs = 256 cycles when the pipe is fed."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddnerf_amd", "csrc")
SO = os.path.join(ROOT, "tools", "lib", os.environ.get("F32_SO", "f32_stamp.so"))
K = [96] * 8 + [256] * 32 + [352] * 8 + [256] * 24 + [288] * 5 + [128]


def build(defs):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function"]
                          + defs + ["-shared", os.path.join(CSRC, "mlp_f32.hip"), os.path.join(CSRC, "mlp_f32_train.hip"), os.path.join(CSRC, "mlp_f32_train_rec.hip"), "-o", SO])
    print(SO)


def setup(lib):
    import torch
    M = 524288
    torch.manual_seed(0)
    feat = torch.randn(M, 128, device="cuda")
    n = lib.ddnerf_mlp_f32_packed_floats
    n.restype = ctypes.c_size_t
    nout = [256] * 9 + [1, 128, 3, 2]
    nin = [96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128]
    params = torch.randn(sum(o * i + o for o, i in zip(nout, nin)), device="cuda") * 0.05
    packed = torch.empty(n(1), device="cuda")
    raw = torch.empty(M, 6, device="cuda")
    P = ctypes.c_void_p
    assert lib.ddnerf_mlp_f32_pack(P(params.data_ptr()), 1, P(packed.data_ptr()), None) == 0
    return lambda: lib.ddnerf_mlp_f32_forward(P(feat.data_ptr()), P(packed.data_ptr()), 1, P(raw.data_ptr()), ctypes.c_long(M), None), raw


def time_train(names):
    """Interleaved launch times of the record-writing training forward (ddnerf_mlp_f32_forward_train_rec) of several builds."""
    import torch
    M = 524288
    res = {}
    libs = {}
    for nm in names:
        path = os.path.join(CSRC, "libddnerf_hip.so") if nm == "product" else os.path.join(ROOT, "tools", "lib", "f32_%s.so" % nm)
        lib = ctypes.CDLL(path)
        go, raw = setup(lib)
        libs[nm] = lib
    P = ctypes.c_void_p
    torch.manual_seed(0)
    feat = torch.randn(M, 128, device="cuda")
    nout = [256] * 9 + [1, 128, 3, 2]
    nin = [96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128]
    params = torch.randn(sum(o * i + o for o, i in zip(nout, nin)), device="cuda") * 0.05
    acts = torch.empty(2560, M, device="cuda")
    deltas = torch.empty(2560, M, device="cuda")
    graw = torch.randn(M, 6, device="cuda")
    raw = torch.empty(M, 6, device="cuda")
    resb = {}
    for rep in range(4):
        for nm, lib in libs.items():
            n = lib.ddnerf_mlp_f32_packed_floats
            n.restype = ctypes.c_size_t
            packed = torch.empty(n(1), device="cuda")
            assert lib.ddnerf_mlp_f32_pack(P(params.data_ptr()), 1, P(packed.data_ptr()), None) == 0
            go = lambda: lib.ddnerf_mlp_f32_forward_train_rec(P(feat.data_ptr()), P(packed.data_ptr()), 1, P(raw.data_ptr()), P(acts.data_ptr()),
                                                             ctypes.c_long(M), ctypes.c_long(M), None)
            for _ in range(3):
                assert go() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                go()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(nm, []).append(e0.elapsed_time(e1) / 10)
            nt = lib.ddnerf_mlp_f32_packed_t_floats
            nt.restype = ctypes.c_size_t
            packed_t = torch.empty(nt(1), device="cuda")
            assert lib.ddnerf_mlp_f32_pack_t(P(params.data_ptr()), 1, P(packed_t.data_ptr()), None) == 0
            gob = lambda: lib.ddnerf_mlp_f32_backward_data_rec(P(graw.data_ptr()), P(packed_t.data_ptr()), P(acts.data_ptr()), 1, P(deltas.data_ptr()),
                                                               ctypes.c_long(M), ctypes.c_long(M), None)
            for _ in range(3):
                assert gob() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gob()
            e1.record()
            torch.cuda.synchronize()
            resb.setdefault(nm, []).append(e0.elapsed_time(e1) / 10)
    for nm in names:
        print("%-12s backward_data_rec %.4f ms per launch (median of 4 x 10)  [%s]" % (nm, sorted(resb[nm])[2], " ".join("%.4f" % v for v in resb[nm])))
        print("%-12s forward_train_rec %.4f ms per launch (median of 4 x 10)  [%s]" % (nm, sorted(res[nm])[2], " ".join("%.4f" % v for v in res[nm])))


def timeit(names):
    """Interleaved launch times of several builds (tools/lib/f32_NAME.so; 'product' = the shipped library's kernel)."""
    import torch
    libs = {}
    for nm in names:
        path = os.path.join(CSRC, "libddnerf_hip.so") if nm == "product" else os.path.join(ROOT, "tools", "lib", "f32_%s.so" % nm)
        libs[nm] = setup(ctypes.CDLL(path))
    ref = None
    for nm, (go, raw) in libs.items():
        go()
        torch.cuda.synchronize()
        if ref is None:
            ref = raw.clone()
        print("%-10s output equal to the first build's: %s" % (nm, bool(torch.equal(ref, raw))))
    res = {nm: [] for nm in names}
    for rep in range(6):
        for nm, (go, raw) in libs.items():
            for _ in range(3):
                go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                go()
            e1.record()
            torch.cuda.synchronize()
            res[nm].append(e0.elapsed_time(e1) / 20)
    for nm in names:
        ms = sorted(res[nm])[len(res[nm]) // 2]
        print("%-10s %.4f ms per launch (median of 6 x 20), frac of 157.3 TFLOP/s %.4f   [%s]" % (
            nm, ms, 1220608 * 524288 / (ms * 1e-3) / 157.3e12, " ".join("%.4f" % v for v in res[nm])))


def run():
    import numpy as np
    import torch
    lib = ctypes.CDLL(SO)
    M = 524288
    torch.manual_seed(0)
    feat = torch.randn(M, 128, device="cuda")
    n = lib.ddnerf_mlp_f32_packed_floats
    n.restype = ctypes.c_size_t
    nparam = 0
    nout = [256] * 9 + [1, 128, 3, 2]
    nin = [96, 256, 256, 256, 256, 352, 256, 256, 256, 256, 283, 128, 128]
    nparam = sum(o * i + o for o, i in zip(nout, nin))
    params = torch.randn(nparam, device="cuda") * 0.05
    packed = torch.empty(n(1), device="cuda")
    raw = torch.empty(M, 6, device="cuda")
    P = ctypes.c_void_p
    assert lib.ddnerf_mlp_f32_pack(P(params.data_ptr()), 1, P(packed.data_ptr()), None) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(4):
        if it == 3:
            e0.record()
        assert lib.ddnerf_mlp_f32_forward(P(feat.data_ptr()), P(packed.data_ptr()), 1, P(raw.data_ptr()), ctypes.c_long(M), None) == 0
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = np.zeros(64 * 96 * 8, dtype=np.uint64)
    assert lib.ddnerf_debug_f32_stamps(P(st.ctypes.data)) == 0
    st = st.reshape(64, 96, 8)[:, :78].astype(np.int64)
    print("launch %.3f ms (stamp build), %.1f TFLOP/s-equivalent frac %.4f" % (ms, 0, 1220608 * M / (ms * 1e-3) / 157.3e12))
    d = np.diff(st, axis=2)                      # 7 intervals inside each slice
    gap = st[:, 1:, 0] - st[:, :-1, 7]           # last chunk's stamp -> next slice's entry (flush of the stamps included)
    total = st[:, 77, 7] - st[:, 0, 0]
    mf = sum(4 * k // 8 for k in K) * 64
    print("workgroup: %.0f cycles entry of slice 0 .. end of slice 77 (median of 64), MFMAs need %d: busy %.4f" % (np.median(total), mf, mf / np.median(total)))
    names = ["chunk0", "chunk1", "..half", "..NQ-3", "NQ-2", "barrier+carry", "last chunk"]
    ideal = lambda k: [256, 256, (k // 16 - 2) * 256, (k // 8 - 2 - k // 16) * 256, 256, 0, 256]
    print("median cycles over the expected count, by layer width (all slices of that width, 64 workgroups):")
    for k in (96, 256, 352, 288, 128):
        idx = [i for i, kk in enumerate(K) if kk == k]
        med = np.median(d[:, idx, :].reshape(-1, 7), axis=0)
        g = np.median(gap[:, [i for i in idx if i < 77]]) if k != 128 else 0
        exc = med - np.array(ideal(k))
        print("  K=%3d (%2d slices): " % (k, len(idx)) + ", ".join("%s %+d" % (n, e) for n, e in zip(names, exc)) + ", to next entry %d" % g)
    exc_all = 0.0
    for i, k in enumerate(K):
        exc_all += np.median(d[:, i, :].sum(axis=1)) - k // 8 * 256
    print("sum over slices of (slice cycles - MFMA cycles): %.0f; sum of gaps between slices %.0f" % (exc_all, np.median(gap, axis=0).sum()))
    print("K=256 layers, by slice position inside the layer (median over the 7 layers x 64 workgroups), interval - expected:")
    starts = [8, 16, 24, 32, 48, 56, 64]
    for b in range(8):
        idx = [s0 + b for s0 in starts]
        med = np.median(d[:, idx, :].reshape(-1, 7), axis=0) - np.array(ideal(256))
        g = np.median(gap[:, idx])
        print("  slice %d: " % b + ", ".join("%s %+d" % (n, e) for n, e in zip(names, med)) + ", to next entry %d" % g)
    first = [i for i in range(78) if i in (0, 8, 16, 24, 32, 40, 48, 56, 64, 72, 77)]
    print("per-slice excess (median), first slice of each layer marked *:")
    print(" ".join(("*" if i in first else "") + "%d" % (np.median(d[:, i, :].sum(axis=1)) - K[i] // 8 * 256 + (np.median(gap[:, i]) if i < 77 else 0)) for i in range(78)))


def tiles():
    """Tile-level stamps (-DF32_STAMP_TILE build): workgroup duration on the shader clock, the clock itself, the share of the launch between workgroups."""
    import numpy as np
    import torch
    lib = ctypes.CDLL(SO)
    go, raw = setup(lib)
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = np.zeros(4096 * 6, dtype=np.uint64)
    assert lib.ddnerf_debug_f32_tile_stamps(ctypes.c_void_p(st.ctypes.data)) == 0
    st = st.reshape(4096, 6).astype(np.int64)
    st = st[st[:, 2] > 0]                        # (persistent kernel: one row per workgroup = per CU)
    per_wg = 4096 // len(st)
    dur = st[:, 2] - st[:, 0]
    pro = st[:, 1] - st[:, 0]
    real = (st[:, 4] - st[:, 3]) / 100e6
    clock = dur / real / 1e6
    mf = sum(4 * k // 8 for k in K) * 64 * per_wg
    span = (st[:, 4].max() - st[:, 3].min()) / 100e6 * 1e3
    print("launch %.4f ms by events; first workgroup start .. last workgroup end %.4f ms on the 100 MHz clock" % (ms, span))
    print("workgroup: %.0f cycles (median; p10 %.0f, p90 %.0f), of which prologue (features + slice 0 staged) %.0f; MFMAs need %d = %.4f of it" % (
        np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), np.median(pro), mf, mf / np.median(dur)))
    print("in-kernel clock %.0f MHz (median); workgroups per CU x median duration = %.4f ms = %.4f of the launch" % (
        np.median(clock), 16 // per_wg * np.median(real) * 1e3, 16 // per_wg * np.median(real) * 1e3 / ms))
    return
    # consecutive workgroups on one CU: sort by start time, greedy chain by end->start proximity is overkill: report the start-time waves
    order = np.argsort(st[:, 3])
    starts = (st[order, 3] - st[:, 3].min()) / 100e6 * 1e6
    print("start times (us) of workgroups 0, 255, 256, 511, 512, 4095 in start order: %s" % " ".join("%.1f" % starts[i] for i in (0, 255, 256, 511, 512, 4095)))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "tiles":
        tiles()
    elif sys.argv[1] == "time_train":
        time_train(sys.argv[2:])
    elif sys.argv[1] == "time":
        timeit(sys.argv[2:])
    else:
        run()
