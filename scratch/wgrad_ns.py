"""timing of ddnerf_mlp_x3_wgrad_packed in variant libraries (kernel + reduce), fine and coarse sizes"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnerf_amd import ops
V = C.c_void_p
st = torch.cuda.current_stream().cuda_stream
for M in (4096 * 128, 4096 * 64):
    pa = torch.randint(0, 2 ** 31 - 1, (2560, M), dtype=torch.int32, device="cuda") & 0x3fff3fff
    pd = torch.randint(0, 2 ** 31 - 1, (2560, M), dtype=torch.int32, device="cuda") & 0x3fff3fff
    ws = torch.empty(ops._lib.lib().ddnerf_mlp_f32_wgrad_workspace_floats(M), dtype=torch.float32, device="cuda")
    for so in sys.argv[1:]:
        L = C.CDLL(so)
        f = L.ddnerf_mlp_x3_wgrad_packed
        f.argtypes = [V, C.c_int, C.c_int, V, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, V, C.c_int, C.c_int, V, V, V]
        for drow0, n_out, arow0, n_in in ((512, 256, 256, 256), (0, 256, 2432, 96), (2304, 128, 2048, 256)):
            w = torch.zeros(n_out, n_in, device="cuda"); b = torch.zeros(n_out, device="cuda")
            run = lambda: f(pd.data_ptr(), drow0, n_out, pa.data_ptr(), arow0, n_in, n_in, M, M, w.data_ptr(), n_in, 0, b.data_ptr(), ws.data_ptr(), st)
            assert run() == 0; torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
            t = sorted(ts)[2]
            print("M=%d %dx%d %-12s %.3f ms %.2f TB/s" % (M, n_out, n_in, os.path.basename(so), t, (n_out + n_in) * M * 4 / 1e9 / t), flush=True)
