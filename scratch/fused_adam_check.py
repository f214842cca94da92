import torch
torch.manual_seed(0)
def run(fused, views):
    flat = torch.arange(1, 1001, dtype=torch.float32, device="cuda") / 1000
    if views:
        ps = [torch.nn.Parameter(flat[:600].view(20, 30)), torch.nn.Parameter(flat[600:].view(400))]
    else:
        ps = [torch.nn.Parameter(flat[:600].clone().view(20, 30)), torch.nn.Parameter(flat[600:].clone())]
    o = torch.optim.Adam(ps, lr=1e-2, **({"fused": True} if fused else {}))
    g = torch.Generator(device="cuda").manual_seed(1)
    for it in range(3):
        for gp in o.param_groups: gp["lr"] = 1e-2 * (it + 1)
        fg = torch.randn(1000, device="cuda", generator=g)
        ps[0].grad = fg[:600].view(20, 30); ps[1].grad = fg[600:]
        o.step(); o.zero_grad()
    return torch.cat([p.detach().reshape(-1) for p in ps])
for views in (False, True):
    a, b = run(False, views), run(True, views)
    print("views", views, "max diff fused vs foreach", float((a - b).abs().max()))
flat = torch.zeros(100, device="cuda")
p = torch.nn.Parameter(torch.empty(0, device="cuda")); p.data = flat[:100].view(10, 10)
for fused in (False, True):
    o = torch.optim.Adam([p], lr=1e-2, **({"fused": True} if fused else {}))
    v0 = p._version
    p.grad = torch.ones(10, 10, device="cuda"); o.step()
    print("fused", fused, "version", v0, "->", p._version)
