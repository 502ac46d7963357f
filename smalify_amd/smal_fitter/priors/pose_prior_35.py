"""Drop-in for reference smal_fitter/priors/pose_prior_35.py :: Prior (the class; the offline prior
builder in that file needs psbody/chumpy and is out of scope).

    Prior(prior_path, device)(x) -> ((x.reshape(-1,105) - mean) @ pic * mask)**2      (N,105)

The pickle is read without chumpy; the mask is built exactly like the reference (only the three
global-rotation entries are masked, pose_prior_35.py:78-92)."""
from __future__ import annotations

import torch

from ... import engine as eng, model_io, runtime


class _PriorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x):
        x = x.contiguous().float()
        ctx.owner = owner
        ctx.save_for_backward(x)
        return eng.pose_prior(owner._engine(x.shape[0]), x)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return None, eng.pose_prior_backward(ctx.owner._engine(x.shape[0]), x, dout.contiguous().float())


class Prior(object):
    def __init__(self, prior_path, device, prior_data=None):
        prec, mean, mask = prior_data if prior_data is not None else model_io.load_pose_prior(prior_path)
        self._data = (prec, mean, mask)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.precs = torch.from_numpy(prec).to(dev)
        self.mean = torch.from_numpy(mean).to(dev)
        self.use_ind_tch = torch.from_numpy(mask).to(dev)

    def _engine(self, frames):
        e = runtime.get_engine(None, frames, 16)
        if getattr(e, "_pose_prior_owner", None) is not self:
            e.set_pose_prior(*self._data)
            e._pose_prior = self._data
            e._pose_prior_owner = self
        return e

    def __call__(self, x):
        return _PriorFn.apply(self, x.reshape(-1, 105))        # (-1, 105) for any input layout, like the reference (:117)
