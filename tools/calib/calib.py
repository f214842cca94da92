import ctypes as C, os, torch
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "calib.so"))
M = 524288
feat = torch.rand(M, 128, device="cuda"); out = torch.zeros(M * 2, device="cuda")
st = torch.cuda.current_stream().cuda_stream
V = C.c_void_p
L.run_rows.argtypes = [V, V, C.c_long, V]; L.run_stream.argtypes = [V, V, C.c_long, V]; L.run_bf16rows.argtypes = [V, V, C.c_long, V]
fb = torch.rand(M, 128, device="cuda").to(torch.bfloat16)
for _ in range(3):
    assert L.run_bf16rows(fb.data_ptr(), out.data_ptr(), M, st) == 0
    assert L.run_rows(feat.data_ptr(), out.data_ptr(), M, st) == 0
    assert L.run_stream(feat.data_ptr(), out.data_ptr(), M * 32, st) == 0
torch.cuda.synchronize()
print("bytes per launch", M * 512)
