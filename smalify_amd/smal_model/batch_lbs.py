"""Drop-in for reference smal_model/batch_lbs.py (the functions the fitting path uses).

batch_rodrigues runs the HIP kernel (smalfit_rodrigues) with an analytic adjoint.
batch_global_rigid_transformation as a free-standing differentiable function is not exposed yet: the
kinematic chain lives inside SMAL.__call__ (pose_kernel / chain_bwd_kernel); calling it raises."""
from __future__ import annotations

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


def batch_global_rigid_transformation(Rs, Js, parent, rotate_base=False, betas_logscale=None, opts=None):
    raise NotImplementedError(
        "smalify_amd fuses the kinematic chain into SMAL.__call__ (HIP pose_kernel); the free-standing "
        "batch_global_rigid_transformation of reference batch_lbs.py:75-170 is not exposed")
