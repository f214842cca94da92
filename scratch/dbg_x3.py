import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ddnerf_amd import ops, synthetic
import torch_ref
depth = True
sd = synthetic.make_state_dict(depth, 3, 1.0)
names = [n for n, _, _ in synthetic.layer_table(depth)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
M = 300
feat = torch.zeros(M, 128, device="cuda"); feat[:, :123] = torch.rand(M, 123, device="cuda") * 2 - 1
raw = ops.mlp_x3_forward(feat, ops.mlp_x3_pack(flat, depth), depth)
ref = ops.mlp_f32_forward(feat, ops.mlp_f32_pack(flat, depth), depth)
print("x3", raw[:3].cpu().numpy()); print("f32", ref[:3].cpu().numpy())
print("max err per col", (raw - ref).abs().max(0).values.cpu().numpy())
print("rows bad", ((raw - ref).abs().max(1).values > 1e-4).nonzero().flatten()[:40].cpu().numpy())
# (a) bf16-exact weights: isolates the activation-lo path
flat_b = flat.to(torch.bfloat16).float()
raw = ops.mlp_x3_forward(feat, ops.mlp_x3_pack(flat_b, depth), depth)
ref = ops.mlp_f32_forward(feat, ops.mlp_f32_pack(flat_b, depth), depth)
print("(a) bf16-exact weights: max err", float((raw - ref).abs().max()))
# (b) bf16-exact weights AND only first layer nonlinear chain short: zero features lo
feat_b = feat.to(torch.bfloat16).float()
raw = ops.mlp_x3_forward(feat_b, ops.mlp_x3_pack(flat, depth), depth)
ref = ops.mlp_f32_forward(feat_b, ops.mlp_f32_pack(flat, depth), depth)
print("(b) bf16-exact features, full weights: max err", float((raw - ref).abs().max()))
raw = ops.mlp_x3_forward(feat_b, ops.mlp_x3_pack(flat_b, depth), depth)
ref = ops.mlp_f32_forward(feat_b, ops.mlp_f32_pack(flat_b, depth), depth)
print("(c) both bf16-exact: max err", float((raw - ref).abs().max()))
