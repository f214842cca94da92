"""Data parallelism over rays: one process per GPU, model replicated, every rank works on its own
ray shard; the ONLY collective is an all-reduce (sum, / world) of each network's flat fp32 gradient buffer over
RCCL (`backend="nccl"` on ROCm; gloo in the CPU tests).  The reference has no distributed code at all.

The fused MLP backward writes all parameter gradients of a network into one flat buffer in registration order
(`net.last_flat_grad`; the `.grad` tensors are views of it), so a network is ONE bucket: 612,740 floats (fine) /
612,998 floats (coarse DD), 2.45 MB each.  The fine network's backward finishes first (it is last in the forward),
so its all-reduce is launched from inside the backward pass and overlaps the coarse network's backward."""
from __future__ import annotations

import torch
import torch.distributed as td


class GradBuckets:
    """early=True launches a network's all-reduce from inside its backward pass (overlap with the rest of the backward).
    That is only sound when autograd will STEAL the flat buffer's views as the .grad tensors (every .grad is None) and the
    network is back-propagated once per finish(): otherwise the accumulation `p.grad += view` would read the buffer while
    the collective rewrites it, or add un-reduced gradients to reduced ones.  Both cases fall back to / demand the plain
    reduce in finish().  single_rank_collectives=True issues the collectives even in a one-rank group (test hook: it runs
    the RCCL code path on a one-GPU box; results are unchanged)."""

    def __init__(self, nets, group=None, early=True, single_rank_collectives=False):
        seen, self.nets = set(), []
        for n in nets:  # GeneralMipNerfModel: fine is coarse -> one bucket
            if id(n) not in seen:
                seen.add(id(n))
                self.nets.append(n)
        self.group = group
        self.world = td.get_world_size(group) if td.is_initialized() else 1
        self.early = early
        self.collect = self.world > 1 or (single_rank_collectives and td.is_initialized())
        self.early_launches = 0  # statistics for the tests
        self._pending = {}
        for n in self.nets:
            n.grad_reducer = self
            n._fwd_calls = 0
            n._bwd_calls = 0

    # called by functions._MLPFunction.backward right after the flat gradient of `net` is complete
    def on_flat_grad_ready(self, net, flat):
        net._bwd_calls = getattr(net, "_bwd_calls", 0) + 1
        steal = all(p.grad is None for p in net.parameters())  # autograd will adopt the views of `flat` as .grad
        if self.collect and self.early and net._fwd_calls == 1 and net._bwd_calls == 1 and steal:
            self._pending[id(net)] = (flat, td.all_reduce(flat, op=td.ReduceOp.SUM, group=self.group, async_op=True))
            self.early_launches += 1

    def _bucket_of(self, net):
        """(flat tensor, aliased) -- the flat buffer the .grad tensors are views of, or a gathered copy"""
        params = [p for p in net.parameters() if p.grad is not None]
        flat = getattr(net, "last_flat_grad", None)
        if flat is not None and len(params) == len(list(net.parameters())):
            off, ok = 0, True
            for p in params:
                ok &= p.grad.data_ptr() == flat.data_ptr() + 4 * off and p.grad.is_contiguous()
                off += p.numel()
            if ok and off == flat.numel():
                return flat, True
        return torch.cat([p.grad.reshape(-1) for p in params]), False

    def finish(self):
        """Complete the gradient all-reduce of every network; afterwards every .grad holds the world average."""
        if not self.collect:
            for n in self.nets:
                n._fwd_calls = n._bwd_calls = 0
            self._pending.clear()
            return
        for net in self.nets:
            pend = self._pending.pop(id(net), None)
            if pend is not None:
                pend[1].wait()
                if net._bwd_calls > 1:  # un-reduced gradients were added to the reduced ones: nothing can repair that
                    raise RuntimeError("gradients were accumulated after an in-backward all-reduce; "
                                       "use GradBuckets(..., early=False) when a network is back-propagated more than once per step")
            flat, aliased = self._bucket_of(net)
            if pend is None:
                td.all_reduce(flat, op=td.ReduceOp.SUM, group=self.group)
            elif not (aliased and pend[0].data_ptr() == flat.data_ptr()):
                raise RuntimeError("the early-reduced gradient buffer is no longer the network's .grad storage")
            flat.div_(self.world)
            if not aliased:
                off = 0
                for p in net.parameters():
                    if p.grad is not None:
                        p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
                        off += p.numel()
            net._fwd_calls = net._bwd_calls = 0


def broadcast_parameters(nets, src=0, group=None):
    """make every rank start from rank `src`'s weights (one broadcast per network: the flat parameter buffer)"""
    seen = set()
    for n in nets:
        if id(n) in seen:
            continue
        seen.add(id(n))
        td.broadcast(n.flat_params(), src=src, group=group)
        n.invalidate_packed()  # the flat buffer was written behind the parameters' version counters
