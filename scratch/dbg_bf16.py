import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from ddnerf_amd import ops, synthetic
depth = True
for sharpen in (1.0, 20.0):
    sd = synthetic.make_state_dict(depth, 4, sharpen)
    names = [n for n, _, _ in synthetic.layer_table(depth)]
    flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
    M = 1500
    torch.manual_seed(0)
    feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
    bf = lambda x: x.to(torch.bfloat16).double()
    W = {k: (bf(torch.from_numpy(v).cuda()) if k.endswith("weight") else torch.from_numpy(v).cuda().double()) for k, v in sd.items()}
    x = bf(feat); xyz, dirs = x[:, :96], x[:, 96:123]; h = xyz
    for i in range(8):
        inp = torch.cat([xyz, h], 1) if i == 5 else h
        h = bf(torch.relu(inp @ W["layers_xyz.%d.weight" % i].T + W["layers_xyz.%d.bias" % i]).float())
    ft = bf((h @ W["fc_feat.weight"].T + W["fc_feat.bias"]).float())
    alpha = ft @ W["fc_alpha.weight"].T + W["fc_alpha.bias"]
    hd = bf(torch.relu(torch.cat([ft, dirs], 1) @ W["layers_dir.0.weight"].T + W["layers_dir.0.bias"]).float())
    ref = torch.cat([hd @ W["fc_rgb.weight"].T + W["fc_rgb.bias"], alpha, hd @ W["fc_mu_sigma.weight"].T + W["fc_mu_sigma.bias"]], 1)
    fb = feat[:, ops.K_ORDER].to(torch.bfloat16).contiguous()
    raw = ops.mlp_bf16_forward(fb, ops.mlp_bf16_pack(flat, depth), depth)
    print("sharpen", sharpen, "bf16 kernel vs bf16 emulation: max err per col", (raw.double() - ref).abs().max(0).values.cpu().numpy(), "scale", ref.abs().max(0).values.cpu().numpy())
