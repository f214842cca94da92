import ctypes as C, sys, torch, glob, os
sys.path.insert(0,'/root/repo')
from ddnerf_amd import ops
M=524288
feat=(torch.rand(M,128,device='cuda')*2-1).to(torch.bfloat16)
raw=torch.empty(M,4,device='cuda')
for so in sorted(glob.glob('/root/repo/scratch/exp/libexp_*.so')):
    L=C.CDLL(so)
    L.ddnerf_mlp_bf16_packed_bytes.restype=C.c_size_t
    nb=L.ddnerf_mlp_bf16_packed_bytes(0)
    packed=(torch.randn(nb//2,device='cuda')*0.05).to(torch.bfloat16).view(torch.uint8)
    f=L.ddnerf_mlp_bf16_forward; f.restype=C.c_int
    f.argtypes=[C.c_void_p,C.c_void_p,C.c_int,C.c_void_p,C.c_long,C.c_void_p]
    st=torch.cuda.current_stream().cuda_stream
    for _ in range(3): f(feat.data_ptr(),packed.data_ptr(),0,raw.data_ptr(),M,st)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f(feat.data_ptr(),packed.data_ptr(),0,raw.data_ptr(),M,st)
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/20
    print('%-55s %.4f ms  frac %.3f'%(os.path.basename(so),t,1220608*M/t/1e9/2500))
