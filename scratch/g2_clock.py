"""in-kernel clock / cycles per 512-sample tile of the two-group bf16 kernel (diagnostic build scratch/ab/lib/g2_stamp.so)"""
import ctypes as C, sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import synthetic
so = sys.argv[1]
M = 524288
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
fb = (torch.rand(M, 128, device="cuda") * 2 - 1).to(torch.bfloat16).contiguous()
raw = torch.empty(M, 4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
V = C.c_void_p
L = C.CDLL(so)
L.ddnerf_mlp_bf16g2_packed_bytes.restype = C.c_size_t
packed = torch.empty(L.ddnerf_mlp_bf16g2_packed_bytes(0), dtype=torch.uint8, device="cuda")
L.ddnerf_mlp_bf16g2_pack.argtypes = [V, C.c_int, V, V]
assert L.ddnerf_mlp_bf16g2_pack(flat.data_ptr(), 0, packed.data_ptr(), st) == 0
f = L.ddnerf_mlp_bf16g2_forward; f.argtypes = [V, V, C.c_int, V, C.c_long, V]
stamps = torch.zeros(256 * 6 + 256 * 192, dtype=torch.int64, device="cuda")
L.ddnerf_debug_set_stamps_g2.argtypes = [V]
assert L.ddnerf_debug_set_stamps_g2(stamps.data_ptr()) == 0
t0 = time.time(); n = 0
while time.time() - t0 < 2.5:
    for _ in range(50): f(fb.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)
    torch.cuda.synchronize(); n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f(fb.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
allst = stamps.cpu().numpy()
s = allst[:256 * 6].reshape(256, 6).astype(np.float64)
clk = (s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 100.0
cyc = (s[:, 2] - s[:, 0]) / s[:, 4]
print("%s: launch %.4f ms (%.3f of peak); clock median %.0f MHz; %.0f cycles per 512-sample tile (ideal 154240 -> pipe %.1f %% busy); prologue %.1f us, loop %.1f us"
      % (os.path.basename(so), ms, 1220608 * M / ms / 1e9 / 2500, np.median(clk), np.median(cyc), 100 * 154240 / np.median(cyc),
         np.median(s[:, 1] - s[:, 5]) / 100, np.median(s[:, 3] - s[:, 1]) / 100))

import importlib.util
spec = importlib.util.spec_from_file_location("gen", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ddnerf_amd", "csrc", "gen_bf16_g2.py"))
sys.argv = ["gen"]
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
allp = allst[256 * 6:].reshape(256, 192).astype(np.float64)
ps = allp[:, :gen.NPER + 1]
d = np.diff(ps, axis=1)
med = np.median(d, axis=0)
tot_ideal = tot = 0
passes = {}
for p_, (pi, ci) in enumerate(gen.PERIODS):
    ideal = sum(gen.K[l] // 32 for l, b in gen.CHUNKS[ci]) * 64
    passes.setdefault(gen.PASSES[pi], []).append((med[p_], ideal))
print("pass     periods: cycles lost ...")
for (l, g), lst in passes.items():
    print("L%d g%d  %6.0f lost %5.0f | %s" % (l, g, sum(x for x, _ in lst), sum(x - y for x, y in lst), "  ".join("%5.0f(%+5.0f)" % (x, x - y) for x, y in lst)))
    tot += sum(x for x, _ in lst); tot_ideal += sum(y for _, y in lst)
print("sum of periods %.0f (ideal %d); tile boundary (per-tile cycles - periods) %.0f" % (tot, tot_ideal, np.median(cyc) - tot))

# per-block stamps of the window
lo, hi = gen.STAMP_BLOCKS
j = gen.NPER + 1
for p_ in range(lo, hi):
    nb = len(gen.CHUNKS[gen.PERIODS[p_][1]])
    t = np.concatenate([ps[:, p_:p_ + 1], allp[:, j:j + nb], ps[:, p_ + 1:p_ + 2]], axis=1)
    dd = np.median(np.diff(t, axis=1), axis=0)
    print("period %2d %s: blocks %s | barrier+tail %4.0f" % (p_, gen.PASSES[gen.PERIODS[p_][0]], " ".join("%5.0f" % x for x in dd[:-1]), dd[-1]))
    j += nb

ks = [tuple(int(v) for v in x.split(":")) for x in os.environ.get("G2_STAMP_KSTEPS", "").split(",") if x]
for i, (per, b) in enumerate(ks):
    t = allp[:, 130 + 12 * i:130 + 12 * i + 11]
    print("period %d block %d: cycles between k-step stamps: %s" % (per, b, " ".join("%5.0f" % x for x in np.median(np.diff(t, axis=1), axis=0))))
