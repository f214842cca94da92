"""world_size-2 gloo tests (CPU, no kernels) of the data-parallel gradient buckets: every rank ends with the
world-average gradient, for the aliased-flat-buffer fast path, the early (in-backward) launch, the gathered
fallback, pre-existing .grad tensors (no early launch: autograd would accumulate into the buffer being reduced) and
gradient accumulation (loud failure with early launches, correct sums without); parameters are broadcast from rank 0."""
import os

import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ddnerf_amd import base_architectures as BA
        from ddnerf_amd import dist as ddp

        torch.manual_seed(100 + rank)
        fine = BA.MipNeRFModel(hidden_size=256, include_input_dir=True)
        coarse = BA.DepthMipNeRFModel(hidden_size=256, include_input_dir=True)
        ddp.broadcast_parameters([coarse, fine])
        w_sum = float(fine.flat_params().double().sum()) + float(coarse.flat_params().double().sum())
        buckets = ddp.GradBuckets([fine, coarse, fine], early=(mode != "accumulate_deferred"))   # duplicates collapse
        assert len(buckets.nets) == 2 and buckets.world == world
        scale = 1.0

        def views_of(net, flat):
            off, out = 0, []
            for p in net.parameters():
                out.append(flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            return out

        for net in (fine, coarse):
            n = sum(p.numel() for p in net.parameters())
            flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
            net.last_flat_grad = flat
            if mode == "aliased":          # .grad = views of the flat buffer, reduced in finish()
                for p, v in zip(net.parameters(), views_of(net, flat)):
                    p.grad = v
            elif mode == "early":          # what the fused backward does: bucket ready while every .grad is None -> launch
                net._fwd_calls = 1
                buckets.on_flat_grad_ready(net, flat)
                for p, v in zip(net.parameters(), views_of(net, flat)):   # autograd steals the views
                    p.grad = v
            elif mode == "preexisting":    # .grad tensors exist (zero_grad(set_to_none=False)): autograd would ACCUMULATE
                for p in net.parameters():
                    p.grad = torch.zeros_like(p)
                net._fwd_calls = 1
                buckets.on_flat_grad_ready(net, flat)        # must not launch: the accumulation reads `flat`
                for p, v in zip(net.parameters(), views_of(net, flat)):
                    p.grad += v
            elif mode in ("accumulate", "accumulate_deferred"):   # two backward passes before finish()
                net._fwd_calls = 1
                buckets.on_flat_grad_ready(net, flat)
                for p, v in zip(net.parameters(), views_of(net, flat)):
                    p.grad = v
                net._fwd_calls = 2
                buckets.on_flat_grad_ready(net, flat)
                for p in net.parameters():
                    p.grad += p.grad.clone() * 0 + 1.0        # a second, un-reduced contribution (1.0 everywhere)
            else:  # gathered: independent .grad tensors, no flat buffer
                for p, v in zip(net.parameters(), views_of(net, flat)):
                    p.grad = v.clone()
        if mode == "early":
            assert buckets.early_launches == 2
        if mode in ("preexisting", "accumulate_deferred"):
            assert buckets.early_launches == 0
        if mode == "accumulate":            # reduced + un-reduced gradients in one buffer: must fail loudly, not average wrongly
            try:
                buckets.finish()
                ok = False
            except RuntimeError:
                ok = True
            q.put((rank, ok, w_sum))
            return
        buckets.finish()
        ok = True
        extra = 1.0 if mode == "accumulate_deferred" else 0.0
        for net in (fine, coarse):
            off = 0
            for p in net.parameters():
                exp = torch.arange(off, off + p.numel(), dtype=torch.float32).view(p.shape) * (sum(range(1, world + 1)) / world) + extra
                ok &= bool(torch.allclose(p.grad, exp, rtol=1e-6, atol=0))
                off += p.numel()
            ok &= net._fwd_calls == 0
        q.put((rank, ok, w_sum))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("mode", ["aliased", "early", "gathered", "preexisting", "accumulate", "accumulate_deferred"])
def test_grad_buckets_world2_gloo(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + hash(mode)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert abs(res[0][2] - res[1][2]) == 0.0      # identical parameters after the broadcast


def test_schedules_match_reference_formulas():
    from ddnerf_amd import schedules
    from ddnerf_amd.cfgnode import CfgNode

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # general_utils/nerf_helpers.py:211-245 with train_model.py:101-107's constants
    assert abs(schedules.lr_at(0, 200001) - 0.0005 * 0.01) < 1e-12
    assert abs(schedules.lr_at(2500, 200001) - 0.0005 * (5e-6 / 0.0005) ** (2500 / 200001)) < 1e-12
    assert abs(schedules.lr_at(200001, 200001) - 5e-6) < 1e-15
    assert schedules.mse2psnr(0) == 50.0 and abs(schedules.mse2psnr(0.01) - 20.0) < 1e-12
    cfg = CfgNode.load(os.path.join(root, "configs", "config_blender.yml"))
    s = schedules.SmoothingSchedule(cfg)
    assert cfg.train_params.dist_reg_coeficient == min(max(1 / 32, 0.01), 0.12)      # train_model.py:124-125
    s.apply(cfg, 75000)
    assert abs(cfg.train_params.gaussian_smooth_factor - (1.7 - 0.6 * 0.5)) < 1e-12
    s.apply(cfg, 20000)
    assert cfg.train_params.pdf_padding is False
    s.apply(cfg, 160000)
    assert cfg.train_params.gaussian_smooth_factor == 1.1
