"""Drop-in for reference smal_model/smal_torch.py :: SMAL — same constructor and call signature, HIP inside.

    SMAL(device, shape_family_id=-1, dtype=torch.float)(beta, theta, trans=None, del_v=None,
         betas_logscale=None, get_skin=True, v_template=None) -> verts, joints, Rs, v_shaped | joints

Blend shapes, Rodrigues, the 34-step kinematic chain, skinning and joint regression are HIP kernels
(smalfit_lbs_forward_ex / smalfit_lbs_backward_ex); gradients flow to beta, theta (axis-angles or rotation matrices),
betas_logscale, del_v and v_template through `verts` and `joints` (Rs and v_shaped are returned detached)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import config, engine as eng, model_io, runtime


class _LBS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, beta, theta, logscale, v_offset):
        e = runtime.get_engine(owner.device_model, theta.shape[0], owner.engine_image_size)
        beta, theta = beta.contiguous().float(), theta.contiguous().float()
        ls = None if logscale is None else logscale.contiguous().float()
        off = None if v_offset is None else v_offset.contiguous().float()
        verts, joints, Rs, vs = e.lbs_forward(beta, theta, ls, v_offset=off)
        ctx.owner, ctx.has_ls, ctx.has_off = owner, ls is not None, off is not None
        ctx.save_for_backward(beta, theta, ls if ls is not None else beta.new_zeros(1), off if off is not None else beta.new_zeros(1))
        ctx.mark_non_differentiable(Rs, vs)
        return verts, joints, Rs, vs

    @staticmethod
    def backward(ctx, dverts, djoints, _dRs, _dvs):
        beta, theta, ls, off = ctx.saved_tensors
        ls = ls if ctx.has_ls else None
        off = off if ctx.has_off else None
        e = runtime.get_engine(ctx.owner.device_model, theta.shape[0], ctx.owner.engine_image_size)
        dv = None if dverts is None else dverts.contiguous().float()
        dj = None if djoints is None else djoints.contiguous().float()
        out = e.lbs_backward(beta, theta, ls, dv, dj, v_offset=off)
        return None, out[0], out[1], out[2], (out[3] if off is not None else None)


class SMAL(nn.Module):
    def __init__(self, device, shape_family_id=-1, dtype=torch.float, model_data=None, engine_image_size=16):
        super().__init__()
        if dtype not in (torch.float, torch.float32):
            raise ValueError("smalify_amd SMAL computes in float32 like the reference's default")
        md = model_data if model_data is not None else model_io.load_smal_model(
            config.SMAL_FILE, config.SMAL_DATA_FILE, config.SMAL_SYM_FILE, shape_family_id)
        self.model_data = md
        self.device_model = eng.DeviceModel(md)
        self.engine_image_size = engine_image_size
        runtime.set_current_model(self.device_model)
        dev = torch.device("cuda", torch.cuda.current_device())
        # attributes the reference exposes (smal_torch.py:36-96)
        self.f = md.faces
        self.faces = torch.from_numpy(md.faces.astype(np.int64)).to(dev)
        self.size = [md.v_template.shape[0], 3]
        self.num_betas = md.shapedirs.shape[0]
        self.v_template = torch.from_numpy(md.v_template).to(dev)
        self.shapedirs = torch.from_numpy(md.shapedirs).to(dev)
        self.posedirs = torch.from_numpy(md.posedirs).to(dev)
        self.J_regressor = torch.from_numpy(md.J_regressor).to(dev)
        self.weights = torch.from_numpy(md.weights).to(dev)
        self.parents = md.parents
        self.left_inds, self.right_inds, self.center_inds = md.left_inds, md.right_inds, md.center_inds

    def __call__(self, beta, theta, trans=None, del_v=None, betas_logscale=None, get_skin=True, v_template=None):
        # per-call template / offset (smal_torch.py:107-122): v_shaped = v_template + del_v + shape blend.  The engine adds
        # one per-frame offset to its own template; gradients flow to del_v and to a v_template that requires them
        N = beta.shape[0]
        offset = None
        if v_template is not None:
            offset = (v_template - self.v_template).expand(N, self.size[0], 3)
        if del_v is not None:
            offset = del_v.expand(N, self.size[0], 3) if offset is None else offset + del_v
        if theta.dim() == 4:                               # rotation matrices (:132-133)
            theta = theta.reshape(N, 35, 3, 3)
        else:
            theta = theta.reshape(N, 35, 3)
        verts, joints, Rs, v_shaped = _LBS.apply(self, beta, theta, betas_logscale, offset)
        if trans is not None:
            verts = verts + trans[:, None, :]          # reference adds trans after skinning (smal_torch.py:165-168)
            # joints are regressed from the translated vertices in the reference (smal_torch.py:171-184)
            joints = joints + self._joint_translation_factor()[None, :, None] * trans[:, None, :]
        if get_skin:
            return verts, joints, Rs, v_shaped
        return joints

    def _joint_translation_factor(self):
        """column sums of the joint regressor (1 for the 6 landmark vertices): d joints / d trans"""
        if not hasattr(self, "_jsum"):
            s = self.J_regressor.sum(0)
            self._jsum = torch.cat([s, torch.ones(6, device=s.device)])
        return self._jsum
