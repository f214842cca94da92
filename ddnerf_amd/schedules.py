"""Host-side schedules of the training loop (train_model.py:101-107, 120-142; general_utils/nerf_helpers.py:12-16,
211-245) -- scalar Python, no device work."""
import math


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
    """log-linear interpolation lr_init -> lr_final with an eased warm-up (general_utils/nerf_helpers.py:211-245)"""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
    else:
        delay_rate = 1.0
    t = min(max(step / max_steps, 0), 1)
    return delay_rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def lr_at(step, train_iters, scheduler=None):
    """the schedule train_model.py hard-wires (:101-107): 5e-4 -> 5e-6, 2500-step x0.01 warm-up; cfg.optimizer.lr and
    the shipped cfg.scheduler keys are ignored by the reference.  Extension: optional `scheduler.lr_init / lr_final /
    lr_delay_steps / lr_delay_mult` keys override the hard-wired values (short procedural runs would otherwise spend
    their whole life inside the warm-up); absent keys keep the reference's numbers."""
    get = (lambda k, d: scheduler.get(k, d)) if scheduler is not None and hasattr(scheduler, "get") else (lambda k, d: d)
    return learning_rate_decay(step, get("lr_init", 0.0005), get("lr_final", 5e-6), train_iters,
                               lr_delay_steps=get("lr_delay_steps", 2500), lr_delay_mult=get("lr_delay_mult", 0.01))


def mse2psnr(mse):
    if mse == 0:
        mse = 1e-5
    return -10.0 * math.log10(mse)


class SmoothingSchedule:
    """gaussian_smooth_factor: linear initial -> final over `finnish_smooth` iterations; pdf_padding switched off at
    max_pdf_pad_iters (train_model.py:120-142).  Writes into the live cfg the model reads."""

    def __init__(self, cfg):
        tp = cfg.train_params
        self.initial = tp.gaussian_smooth_factor
        self.d = (tp.gaussian_smooth_factor - tp.final_smooth) / tp.finnish_smooth
        if tp.set_automatic_dist_reg_coeficient:
            tp.dist_reg_coeficient = min(max(1 / cfg.nerf.train.num_coarse, 0.01), 0.12)

    def apply(self, cfg, i):
        tp = cfg.train_params
        tp.gaussian_smooth_factor = self.initial - self.d * i if i < tp.finnish_smooth else tp.final_smooth
        if i == tp.max_pdf_pad_iters:
            tp.pdf_padding = False
