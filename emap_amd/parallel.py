"""Data-parallel training step over rays: one process per GPU, ``torch.distributed`` ("nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

EMAP itself is single-GPU (SURVEY.md par. 2a); this is the sharding the hot path admits (par. 8e):
rays are independent through sampling, MLP and compositing, so every rank renders its slice of ONE
global batch with replicated weights and the only exchange is

  1. a 2-float all-reduce of the eikonal mask counts *before* backward (the loss's masked means
     ``sum(m*e)/(sum(m)+1e-5)`` run over all rays of the batch, reference udf_renderer_blending.py:618-625,
     so the denominators must be global for the result to equal the single-GPU step), and
  2. ONE all-reduce(sum) of a flat fp32 bucket holding every parameter gradient (about 1.85 MB) after
     backward.

The step arithmetic restates reference src/runner/runner_udf.py:90-168 (mask of ones, MSE * edge_weight
+ igr_ns_weight * ge_ns + igr_weight * ge, zero_grad / backward / step).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank's contiguous slice [rank*N/world, (rank+1)*N/world) of a global per-ray tensor."""
    n = t.shape[0]
    assert n % world == 0, "global ray batch must divide evenly over ranks"
    per = n // world
    return t[rank * per:(rank + 1) * per].contiguous()


class GradBucket:
    """Flat fp32 bucket over the gradients of `params`; one all-reduce(sum) for all of them."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        self.numel = sum(p.numel() for p in self.params)
        self.flat: Optional[torch.Tensor] = None

    def all_reduce(self, group=None):
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.requires_grad:
                g = self.flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            off += n


def training_step(render_fn: Callable[[], Dict[str, torch.Tensor]], true_edge: torch.Tensor, params, optimizer,
                  edge_weight: float = 1.0, igr_weight: float = 0.1, igr_ns_weight: float = 0.0, group=None,
                  bucket: Optional[GradBucket] = None, n_rays_global: Optional[int] = None):
    """One optimizer step on this rank's ray shard; equals the single-GPU step on the global batch.

    render_fn() -> render dict for this rank's rays (must contain "edge", "gradient_error",
    "gradient_error_near_surface" and "eikonal_sums" = [sum(relax*err), sum(relax), sum(near*err), sum(near)]).
    Returns (loss_global, edge_loss_global) as detached 0-d tensors (identical on every rank).
    """
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    out = render_fn()
    edge = out["edge"]
    n_local = edge.shape[0]
    n_glob = n_rays_global if n_rays_global is not None else n_local * world
    sums = out["eikonal_sums"].detach()
    counts = torch.stack([sums[1], sums[3]]).to(torch.float32)  # local denominators (no gradient)
    counts_glob = counts.clone()
    if world > 1:
        dist.all_reduce(counts_glob, op=dist.ReduceOp.SUM, group=group)
    # local numerators with gradient: ge_local * (c_local + 1e-5)
    e_rel = out["gradient_error"] * (counts[0] + 1e-5)
    e_ns = out["gradient_error_near_surface"] * (counts[1] + 1e-5)
    mse_sum = ((edge - true_edge) ** 2).sum()
    n_elem_glob = n_glob * (edge.numel() // n_local)
    loss_local = mse_sum / n_elem_glob * edge_weight \
        + e_ns / (counts_glob[1] + 1e-5) * igr_ns_weight + e_rel / (counts_glob[0] + 1e-5) * igr_weight
    optimizer.zero_grad()
    loss_local.backward()
    if bucket is None:
        bucket = GradBucket(params)
    bucket.all_reduce(group)
    optimizer.step()
    stats = torch.stack([loss_local.detach(), (mse_sum / n_elem_glob * edge_weight).detach()])
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats[0], stats[1]
