import numpy as np, torch, sys
sys.path.insert(0,'.')
from ddnerf_amd import ops
for tag in ("blender_drop","blender_full","llff"):
    g=dict(np.load('tests/golden/dploss_%s.npz'%tag))
    args=[torch.from_numpy(g[k]).cuda() for k in ("t1","t0","w1","w0","mus","sig","left","part")]
    gw,gm,gs=ops.dp_loss_backward(*args,bool(g["is_blender"]),torch.ones((),device="cuda"))
    np.savez('gpurun_out/dbg_dpl_%s.npz'%tag,gw=gw.cpu().numpy(),gm=gm.cpu().numpy(),gs=gs.cpu().numpy())
