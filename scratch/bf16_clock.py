"""In-kernel clock of the bf16 MLP kernel (diagnostic build scratch/ab/lib/bf16_stamp.so, -DBF16_STAMP): >= 2 s of
back-to-back launches on random data, then the stamps of the last launch: clock = d(s_memtime) / d(s_memrealtime) x 100 MHz,
median over workgroups, and cycles per 256-sample tile (ideal: 4820 MFMAs x 16 cycles = 77,120)."""
import ctypes as C, sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import synthetic
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "lib", "bf16_stamp.so")
M = 524288
sd = synthetic.make_state_dict(False, 12, 20.0)
names = [n for n, _, _ in synthetic.layer_table(False)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
fb = (torch.rand(M, 128, device="cuda") * 2 - 1).to(torch.bfloat16).contiguous()
raw = torch.empty(M, 4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
V = C.c_void_p
L = C.CDLL(so)
L.ddnerf_mlp_bf16_packed_bytes.restype = C.c_size_t
packed = torch.empty(L.ddnerf_mlp_bf16_packed_bytes(0), dtype=torch.uint8, device="cuda")
L.ddnerf_mlp_bf16_pack.argtypes = [V, C.c_int, V, V]
assert L.ddnerf_mlp_bf16_pack(flat.data_ptr(), 0, packed.data_ptr(), st) == 0
f = L.ddnerf_mlp_bf16_forward; f.argtypes = [V, V, C.c_int, V, C.c_long, V]
stamps = torch.zeros(256 * 6, dtype=torch.int64, device="cuda")
L.ddnerf_debug_set_stamps.argtypes = [V]
assert L.ddnerf_debug_set_stamps(stamps.data_ptr()) == 0
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.time() - t0 < 2.5:
    for _ in range(50): f(fb.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)
    torch.cuda.synchronize(); n += 50
e0.record()
for _ in range(50): f(fb.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), M, st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
s = stamps.cpu().numpy().reshape(256, 6).astype(np.float64)
clk = (s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 100.0
cyc = (s[:, 2] - s[:, 0]) / s[:, 4]
print("launch %.4f ms (%.3f of the bf16 MFMA peak) after %d warm launches; in-kernel clock median %.0f MHz (min %.0f, max %.0f); "
      "%.0f shader cycles per tile (median; ideal 77120 -> matrix pipe %.1f %% busy)"
      % (ms, 1220608 * M / ms / 1e9 / 2500, n, np.median(clk), clk.min(), clk.max(), np.median(cyc), 100 * 77120 / np.median(cyc)))
