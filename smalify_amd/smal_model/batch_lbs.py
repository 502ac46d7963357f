"""Drop-in for reference smal_model/batch_lbs.py (the functions the fitting path uses).

batch_rodrigues runs the HIP kernel (smalfit_rodrigues) with an analytic adjoint.
batch_global_rigid_transformation runs smalfit_global_rigid_transformation / _backward (inside the fitting path the chain
and its adjoint are fused into SMAL.__call__: lbs_head_kernel / chain_bwd_kernel)."""
from __future__ import annotations

import numpy as np
import torch

from .. import engine as eng


class _Rodrigues(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta):
        theta = theta.contiguous().float()
        ctx.save_for_backward(theta)
        return eng.rodrigues(theta)

    @staticmethod
    def backward(ctx, dR):
        (theta,) = ctx.saved_tensors
        return eng.rodrigues_backward(theta, dR.contiguous().float())


def batch_rodrigues(theta, opts=None):
    """theta (N,3) axis-angle -> (N,3,3)   (reference batch_lbs.py:33-52, incl. the +1e-8 inside the norm)"""
    return _Rodrigues.apply(theta)


class _GlobalRigid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Rs, Js, logscale, parent):
        ls = None if logscale is None else logscale.contiguous().float()
        Rs, Js = Rs.contiguous().float(), Js.contiguous().float()
        new_J, A = eng.global_rigid_transformation(Rs, Js, parent, ls)
        ctx.parent, ctx.has_ls = parent, ls is not None
        ctx.save_for_backward(Rs, Js, ls if ls is not None else Rs.new_zeros(1))
        return new_J, A

    @staticmethod
    def backward(ctx, d_new_J, d_A):
        Rs, Js, ls = ctx.saved_tensors
        ls = ls if ctx.has_ls else None
        dn = torch.zeros_like(Js) if d_new_J is None else d_new_J.contiguous().float()
        dA = Rs.new_zeros(Rs.shape[0], 35, 4, 4) if d_A is None else d_A.contiguous().float()
        dRs, dJs, dls = eng.global_rigid_transformation_backward(Rs, Js, ctx.parent, ls, dn, dA)
        return dRs, dJs, dls, None


def batch_global_rigid_transformation(Rs, Js, parent, rotate_base=False, betas_logscale=None, opts=None):
    """Rs (N,35,3,3), Js (N,35,3), parent (35,) -> new_J (N,35,3), A (N,35,4,4)   (reference batch_lbs.py:75-170).
    rotate_base=True fails in the reference too (`torch.repeat` does not exist, batch_lbs.py:93) and is rejected."""
    if rotate_base:
        raise NotImplementedError("rotate_base=True is broken in the reference (batch_lbs.py:91-94) and not supported")
    return _GlobalRigid.apply(Rs, Js, betas_logscale, np.asarray(parent))
