import torch, numpy as np, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from ddnerf_amd import ops, synthetic
depth, M = True, 200
sd = synthetic.make_state_dict(depth, 9, 3.0)
names = [n for n, _, _ in synthetic.layer_table(depth)]
flat = torch.from_numpy(np.concatenate([np.concatenate([sd[k + ".weight"].ravel(), sd[k + ".bias"].ravel()]) for k in names])).cuda()
g = torch.Generator().manual_seed(5)
feat = torch.zeros(M, 128); feat[:, :123] = torch.rand(M, 123, generator=g) * 2 - 1; feat = feat.cuda()
G = torch.randn(M, 6, generator=g).cuda()
pk, pt = ops.mlp_f32_pack(flat, depth), ops.mlp_f32_pack_t(flat, depth)
raw_w, a_w = ops.mlp_f32_forward_train(feat, pk, depth, rec="hilo")
raw_v, a_v = ops.mlp_f32_forward_train(feat, pk, depth, rec="values")
print("raw equal", torch.equal(raw_w, raw_v))
A_w, A_v = ops.x3_unsplit(a_w), ops.x3_unblock(a_v)
d = (A_w - A_v).abs()
print("acts max diff", float(d[:, :M].max()), "rows bad", torch.nonzero(d[:, :M].max(1).values > 1e-5 * float(A_v[:, :M].abs().max())).flatten()[:20].tolist())
d_w = ops.mlp_f32_backward_data(G, pt, a_w, depth, rec="hilo")
d_v = ops.mlp_f32_backward_data(G, pt, a_v, depth, rec="values")
D_w, D_v = ops.x3_unsplit(d_w)[:2438], ops.x3_unblock(d_v)[:2438]
d = (D_w - D_v).abs()
bad = torch.nonzero(d[:, :M].max(1).values > 1e-5 * float(D_w[:, :M].abs().max())).flatten()
print("deltas max diff", float(d[:, :M].max()), "n rows bad", bad.numel(), bad[:40].tolist())
print("pad cols deltas values: max", float(D_v[:, M:].abs().max()), "words", float(D_w[:, M:].abs().max()))
from ddnerf_amd import base_architectures as BA
net = BA.DepthMipNeRFModel(hidden_size=256, include_input_dir=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.cuda()
for poison in (False, True):
    if poison:
        def poisoned(shape, dtype, device):
            t = torch.empty(shape, dtype=dtype, device=device); t.view(torch.int16).fill_(-1); return t
        ops.RECORD_ALLOC = poisoned
        raw_w, a_w = ops.mlp_f32_forward_train(feat, pk, depth, rec="hilo")
        raw_v, a_v = ops.mlp_f32_forward_train(feat, pk, depth, rec="values")
        d_w = ops.mlp_f32_backward_data(G, pt, a_w, depth, rec="hilo")
        d_v = ops.mlp_f32_backward_data(G, pt, a_v, depth, rec="values")
    fw, vw = ops.mlp_f32_weight_grads(net, a_w, d_w, M, mode="x3p")
    fv, vv = ops.mlp_f32_weight_grads(net, a_v, d_v, M, mode="x3b")
    torch.cuda.synchronize()
    for (name, p), x, y in zip(net.named_parameters(), vw, vv):
        e = float((x - y).abs().max()); s = float(x.abs().max())
        print("poison", poison, name, "equal" if torch.equal(x, y) else "max diff %.3g of %.3g" % (e, s), "nan" if torch.isnan(y).any() else "")
