"""Is the bf16 fine-MLP kernel slower inside a render step than back to back -- and if so, fewer MHz or more cycles?
The diagnostic stamp build (ddnerf_amd/csrc/libddnerf_diag.so) takes the place of ddnerf_mlp_bf16_forward in the product library's
function table (same weight image: same source); after each render step the stamps of the LAST launch (the fine pass) are read."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ddnerf_amd import _lib, build as hip_build, ops, synthetic

args = bench.parse(["--mlp", "bf16"])
dev = torch.device("cuda", 0)
model, cfg, sd_c, sd_f = bench.build_model(args, dev)
ro, rd, rad, tgt = (torch.from_numpy(x).to(dev) for x in synthetic.make_rays("blender", 4096, 1))
D = C.CDLL(hip_build.DIAG_SO)
V = C.c_void_p
D.ddnerf_mlp_bf16_forward.argtypes = [V, V, C.c_int, V, C.c_long, V]
D.ddnerf_debug_set_stamps.argtypes = [V]
stamps = torch.zeros(256 * 6, dtype=torch.int64, device=dev)
assert D.ddnerf_debug_set_stamps(stamps.data_ptr()) == 0
L = _lib.lib()
orig = L.ddnerf_mlp_bf16_forward
def step():
    with torch.no_grad():
        return model.run_iter(ro, rd, rad, mode="validation", rgb_target=tgt)
def read():
    s = stamps.cpu().numpy().reshape(256, 6).astype(np.float64)
    s = s[s[:, 4] > 0]
    clk = np.median((s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 100.0)
    cyc = np.median((s[:, 2] - s[:, 0]) / s[:, 4])
    dur = np.median((s[:, 3] - s[:, 1]) / 100.0)   # us of the tile loop (s_memrealtime ticks at 100 MHz)
    span = (s[:, 3].max() - s[:, 5].min()) / 100.0  # first workgroup ENTRY .. last workgroup end
    pro = np.median((s[:, 1] - s[:, 5]) / 100.0)    # prologue: entry .. tile loop
    skew = (s[:, 5].max() - s[:, 5].min()) / 100.0  # first .. last workgroup entry
    return clk, cyc, dur, span, pro, skew
for _ in range(5): step()
torch.cuda.synchronize()
L.ddnerf_mlp_bf16_forward = D.ddnerf_mlp_bf16_forward
res = []
ev = []
def hook(M, launch):
    if M != 524288: return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o = launch(); e1.record(); ev.append((e0, e1)); return o
ops.MLP_LAUNCH_HOOK = hook
for _ in range(40):
    step(); torch.cuda.synchronize(); res.append(read())
# ... and as bench.py runs them: the steps enqueued back to back, no host sync in between
ev2 = ev[:]
del ev[:]
cont = []
for rep in range(6):
    for _ in range(20): step()
    torch.cuda.synchronize(); cont.append(read())
print("CONTINUOUS STEPS (no sync between steps): events median %.1f us;  clock %.0f MHz, %.0f cycles/tile, tile loop %.1f us, first entry .. last end %.1f us, prologue %.1f us, entry skew %.1f us"
      % ((1e3 * np.median([a.elapsed_time(b) for a, b in ev[20:]]),) + tuple(np.median(np.array(cont[1:]), 0))))
ev[:] = ev2
ops.MLP_LAUNCH_HOOK = None
print("HIP events around the fine launch in the step: median %.1f us" % (1e3 * np.median([a.elapsed_time(b) for a, b in ev[5:]])))
r = np.array(res[5:])
print("IN STEP       : clock %.0f MHz, %.0f cycles/tile, tile loop %.1f us (median WG), first entry .. last end %.1f us, prologue %.1f us, entry skew %.1f us" % tuple(np.median(r, 0)))
# back to back, same process, same buffers: the fine pass's features / weights
feat = ops.encode(ops.pack_rays(ro, rd, rad, 2.0, 6.0), torch.sort(torch.rand(4096, 129, device=dev) * 4 + 2, dim=1)[0].contiguous(), bf16=True)
packed = torch.empty(L.ddnerf_mlp_bf16_packed_bytes(0), dtype=torch.uint8, device=dev)
L.ddnerf_mlp_bf16_pack(model.fine.flat_params().data_ptr(), 0, packed.data_ptr(), torch.cuda.current_stream().cuda_stream)
raw = torch.empty(524288, 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(50): D.ddnerf_mlp_bf16_forward(feat.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), 524288, st)
    torch.cuda.synchronize()
print("BACK TO BACK  : clock %.0f MHz, %.0f cycles/tile, tile loop %.1f us (median WG), first entry .. last end %.1f us, prologue %.1f us, entry skew %.1f us" % read())
# in between: each launch preceded by ~60 us of other work (the encode kernel), timed launch by launch
res = []
rays = ops.pack_rays(ro, rd, rad, 2.0, 6.0)
tv = torch.sort(torch.rand(4096, 129, device=dev) * 4 + 2, dim=1)[0].contiguous()
for _ in range(40):
    f2 = ops.encode(rays, tv, bf16=True)
    D.ddnerf_mlp_bf16_forward(f2.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), 524288, st)
    torch.cuda.synchronize(); res.append(read())
print("ENCODE + MLP  : clock %.0f MHz, %.0f cycles/tile, tile loop %.1f us (median WG), first entry .. last end %.1f us, prologue %.1f us, entry skew %.1f us" % tuple(np.median(np.array(res[5:]), 0)))
res = []
for _ in range(40):
    D.ddnerf_mlp_bf16_forward(feat.data_ptr(), packed.data_ptr(), 0, raw.data_ptr(), 524288, st)
    torch.cuda.synchronize(); time.sleep(0.002); res.append(read())
print("ISOLATED (idle gaps): clock %.0f MHz, %.0f cycles/tile, tile loop %.1f us (median WG), first entry .. last end %.1f us, prologue %.1f us, entry skew %.1f us" % tuple(np.median(np.array(res[5:]), 0)))
