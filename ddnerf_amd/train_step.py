"""One training iteration of the reference's loop (train_model.py:132-177) on the HIP path, optionally data
parallel.  Loss assembly, the two Adam optimisers and the schedules are the reference's; what changes is that
nothing forces a host sync per iteration (the reference calls .item() twice per step for logging)."""
from __future__ import annotations

import os

import torch

from . import dist as ddp
from . import functions as F
from . import schedules

# the loss of train_model.py:156-172 as one launch + one for its gradient (ddnerf_train_loss_*); DDNERF_FUSED_LOSS=0: the torch op chain
FUSED_LOSS = os.environ.get("DDNERF_FUSED_LOSS", "1") != "0"
# the weight-gradient lanes of both networks joined once, behind loss.backward() (DDNERF_DEFER_WGRAD_JOIN=0: at the end of each network's node)
DEFER_WGRAD_JOIN = os.environ.get("DDNERF_DEFER_WGRAD_JOIN", "1") != "0"


class TrainStepper:
    def __init__(self, model, cfg, dist=False, optimizers=None, single_rank_collectives=False):
        self.model, self.cfg = model, cfg
        self.dd = cfg.nerf.type == "DDNerfModel"
        self.own_optims = optimizers is None   # (then optims[k] is known to be network k's: coarse, fine)
        if optimizers is None:  # train_model.py:84-98: one optimiser per network, lr set per step
            opt = getattr(torch.optim, cfg.optimizer.type)
            # (same update rule; on the GPU torch's fused implementation is ONE kernel per optimiser instead of six multi-tensor
            # launches: 0.29 -> 0.08 ms of a 10 ms step)
            on_gpu = next(model.coarse.parameters()).is_cuda
            kw = {"fused": True} if (on_gpu and cfg.optimizer.type in ("Adam", "AdamW") and os.environ.get("DDNERF_FUSED_ADAM", "1") != "0") else {}
            optimizers = [opt(model.coarse.parameters(), lr=cfg.optimizer.lr, **kw)]
            if cfg.nerf.type != "GeneralMipNerfModel":
                optimizers.append(opt(model.fine.parameters(), lr=cfg.optimizer.lr, **kw))
        self.optims = optimizers
        self.smooth = schedules.SmoothingSchedule(cfg)
        self.buckets = ddp.GradBuckets([model.fine, model.coarse], single_rank_collectives=single_rank_collectives) if dist else None
        if dist:
            ddp.broadcast_parameters([model.coarse, model.fine])
        self.iter = 0

    def step(self, ray_origins, ray_directions, ray_rad, target):
        cfg, model, i = self.cfg, self.model, self.iter
        self.smooth.apply(model.cfg, i)
        model.train()
        lr = schedules.lr_at(i, cfg.experiment.train_iters, cfg.get("scheduler", None) if hasattr(cfg, "get") else None)
        for o in self.optims:
            for gp in o.param_groups:
                gp["lr"] = lr
        out = model.run_iter(ray_origins, ray_directions, ray_rad, mode="train", rgb_target=target)
        lc = cfg.train_params.loss_coeficients
        if FUSED_LOSS and len(out) == 2 and out[0]["rgb"].is_cuda:
            # the whole of train_model.py:156-172 in one launch (and one for its gradient)
            dp_vec = out[1]["dp_loss"] if self.dd else None
            loss, parts = F.train_loss(out[0]["rgb"], out[1]["rgb"], target, dp_vec, lc[0], lc[1], cfg.train_params.dp_coeficient if self.dd else 0.0)
            losses = [parts[0], parts[1]] + ([parts[2]] if self.dd else [])
        else:
            losses = [torch.nn.functional.mse_loss(out[j]["rgb"], target) for j in range(len(out))]
            loss = lc[0] * losses[0]                                                               # :159-161 (no `0 +` launch in front)
            for j in range(1, len(out)):
                loss = loss + lc[j] * losses[j]
            if self.dd:
                dp = out[1]["dp_loss"].mean()                                                      # :163-167
                loss = loss + cfg.train_params.dp_coeficient * dp
                losses.append(dp)
        # (single GPU: the weight-gradient lanes of a network are joined behind the whole backward pass, not at the end of that network's
        # node: the other network's backward chain runs beside their tail.  With gradient buckets the reducer wants each buffer at once.)
        from . import ops
        # (one shared MLP, GeneralMipNerfModel: its two backward nodes' gradients are ADDED on this stream as soon as the second returns)
        ops.DEFER_JOIN = self.buckets is None and DEFER_WGRAD_JOIN and len(self.optims) > 1
        try:
            loss.backward()
        except BaseException:
            ops.DEFER_JOIN = False
            ops.join_deferred()      # (nothing stays referenced behind a failed step)
            raise
        ops.DEFER_JOIN = False
        if not self.own_optims:
            ops.join_deferred()
        if self.buckets is not None:
            self.buckets.finish()
        if self.own_optims:
            # (the fine network's backward ran first: its lanes are done first, and its optimiser step runs beside the coarse lanes' tail)
            nets = [model.coarse] + ([model.fine] if len(self.optims) > 1 else [])
            for net, o in reversed(list(zip(nets, self.optims))):
                ops.join_deferred(net)
                o.step()
                o.zero_grad()
            ops.join_deferred()
        else:
            for o in self.optims:
                o.step()
                o.zero_grad()
        self.iter += 1
        return loss.detach(), [l.detach() for l in losses], out
