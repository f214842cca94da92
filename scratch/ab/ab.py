"""interleaved A/B timing of MLP-forward variants in ONE process (devices and DVFS states differ between runs).
usage: ab.py fp32|bf16|x3 lib1.so[:symbol_infix] lib2.so ...   (each lib exports the ddnerf_mlp_* C ABI and packs with its own
pack kernel; `lib.so:bf16v1` calls ddnerf_mlp_bf16v1_* instead of ddnerf_mlp_bf16_*)"""
import ctypes as C, sys, os, statistics, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddnerf_amd import synthetic
kind, libs = sys.argv[1], sys.argv[2:]
M = 524288
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
KO = [(p & ~15) | ((p & 3) + 4 * (((p >> 2) & 1) * 2 + ((p >> 3) & 1))) for p in range(128)]
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
fb = feat[:, KO].to(torch.bfloat16).contiguous()
raw = torch.empty(M, 4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
V = C.c_void_p
runs = []
for so in libs:
    so, _, infix = so.partition(":")
    L = C.CDLL(so)
    if kind == "bf16":
        infix = infix or "bf16"
        pb = getattr(L, "ddnerf_mlp_%s_packed_bytes" % infix); pb.restype = C.c_size_t
        packed = torch.empty(pb(0), dtype=torch.uint8, device="cuda")
        pk = getattr(L, "ddnerf_mlp_%s_pack" % infix); pk.argtypes = [V, C.c_int, V, V]
        assert pk(flat.data_ptr(), 0, packed.data_ptr(), st) == 0
        f = getattr(L, "ddnerf_mlp_%s_forward" % infix); f.argtypes = [V, V, C.c_int, V, C.c_long, V]
        runs.append((so, lambda f=f, packed=packed: f(fb.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)))
    elif kind == "x3":
        infix = infix or "x3"
        pb = getattr(L, "ddnerf_mlp_%s_packed_bytes" % infix); pb.restype = C.c_size_t
        packed = torch.empty(pb(0), dtype=torch.uint8, device="cuda")
        pk = getattr(L, "ddnerf_mlp_%s_pack" % infix); pk.argtypes = [V, C.c_int, V, V]
        assert pk(flat.data_ptr(), 0, packed.data_ptr(), st) == 0
        f = getattr(L, "ddnerf_mlp_%s_forward" % infix); f.argtypes = [V, V, C.c_int, V, C.c_long, V]
        runs.append((so, lambda f=f, packed=packed: f(feat.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)))
    else:
        L.ddnerf_mlp_f32_packed_floats.restype = C.c_size_t
        packed = torch.empty(L.ddnerf_mlp_f32_packed_floats(0), dtype=torch.float32, device="cuda")
        L.ddnerf_mlp_f32_pack.argtypes = [V, C.c_int, V, V]
        assert L.ddnerf_mlp_f32_pack(flat.data_ptr(), 0, packed.data_ptr(), st) == 0
        f = L.ddnerf_mlp_f32_forward; f.argtypes = [V, V, C.c_int, V, C.c_long, V]
        runs.append((so, lambda f=f, packed=packed: f(feat.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)))
outs = []
for so, fn in runs:
    assert fn() == 0
    torch.cuda.synchronize(); outs.append(raw.clone())
for o in outs[1:]:
    print("max |diff| vs first variant: %.3g" % float((o - outs[0]).abs().max()))
times = {so: [] for so, _ in runs}
reps = 20 if kind == "bf16" else (10 if kind == "x3" else 4)
for _ in range(3):
    for so, fn in runs: fn()
for rnd in range(12):
    for so, fn in runs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        times[so].append(e0.elapsed_time(e1) / reps)
peak = 2500 if kind == "bf16" else (2500 / 3 if kind == "x3" else 157.3)
for so, ts in times.items():
    med = statistics.median(ts)
    print("%-40s median %.4f ms  min %.4f  frac(median) %.4f" % (os.path.basename(so), med, min(ts), 1220608 * M / med / 1e9 / peak))
