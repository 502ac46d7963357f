"""Frame-sharded data parallelism for the fitting loop: one process per GPU, torch.distributed over
RCCL/xGMI (backend "nccl" on ROCm), gloo on CPU for tests.

The reference has no multi-GPU path; what shards here is its per-frame structure:
every loss term of smal_fitter.py:107-175 is a sum over frames given the shared shape parameters, and
the temporal term (smal_fitter.py:177-190) couples only adjacent frames.  Per iteration each rank

  1. all-gathers a 2x108-float record (masked pose + translation of its first and last frame) so that
     the temporal pairs that straddle a shard boundary see their neighbour (the pair (i, i+1) is owned,
     for the loss value, by the rank that owns frame i),
  2. evaluates its own frames (HIP engine),
  3. all-reduces (sum) the 26-float gradient of the shared betas / limb scales,
  4. applies Adam locally; the shared parameters evolve identically on every rank because they see the
     same reduced gradient from the same state.

Messages are a few hundred bytes, i.e. pure latency on xGMI; there is no bulk exchange to overlap.
Shards must start on window boundaries so that the per-window normalisers match the unsharded run.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(num_frames, rank, world_size, window=None):
    """contiguous [lo, hi) of frames for `rank`; equal shards, optionally aligned to `window`"""
    if num_frames % world_size != 0:
        raise ValueError("num_frames (%d) must be divisible by the number of ranks (%d)" % (num_frames, world_size))
    per = num_frames // world_size
    if window is not None and world_size > 1 and per % window != 0:
        raise ValueError("frames per rank (%d) must be a multiple of WINDOW_SIZE (%d)" % (per, window))
    return rank * per, (rank + 1) * per


class ShardedFitter:
    """Wraps a local fitter (FusedFitter protocol: evaluate / apply_adam / shared_grad / boundary_records /
    halo_prev / halo_next / trainable / begin_stage / losses) for rank `rank` of `world_size`."""

    def __init__(self, local_fitter, rank, world_size, group=None):
        self.fitter = local_fitter
        self.rank, self.world, self.group = rank, world_size, group
        self._gather = None

    def begin_stage(self, stage_id):
        self.fitter.begin_stage(stage_id)

    def exchange_halos(self):
        f = self.fitter
        rec = f.boundary_records().reshape(-1)          # (2*108,)
        if self._gather is None or self._gather.numel() != self.world * 216:
            self._gather = torch.empty(self.world * 216, device=rec.device, dtype=rec.dtype)
        dist.all_gather_into_tensor(self._gather, rec, group=self.group)
        g = self._gather.view(self.world, 2, 108)
        f.halo_prev = g[self.rank - 1, 1].contiguous() if self.rank > 0 else None
        f.halo_next = g[self.rank + 1, 0].contiguous() if self.rank + 1 < self.world else None

    def step(self, weights, w_temp, lr, stage_id):
        f = self.fitter
        names = f.trainable(stage_id)
        if self.world > 1 and float(w_temp) > 0.0:
            self.exchange_halos()
        f.evaluate(weights, w_temp, stage_id, want=names)
        if self.world > 1 and ("betas" in names or "log_beta_scales" in names):
            dist.all_reduce(f.shared_grad(), op=dist.ReduceOp.SUM, group=self.group)
        f.apply_adam(names, lr)
        return f.losses

    def global_losses(self):
        """sum of the per-rank loss terms (reporting only)"""
        l = self.fitter.losses.clone()
        if self.world > 1:
            dist.all_reduce(l, op=dist.ReduceOp.SUM, group=self.group)
        return l
