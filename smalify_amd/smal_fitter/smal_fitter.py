"""Drop-in for reference smal_fitter/smal_fitter.py :: SMALFitter(nn.Module).

Same constructor, parameters, `forward(batch_range, weights, stage_id) -> (loss, objs)`,
`get_temporal(w_temp)`, `load_checkpoint`, so the reference's own driver loop (optimize_to_joints.py:90-140,
including its `torch.optim.Adam(model.parameters(), ...)`) runs unmodified on top of the HIP engine.

forward() evaluates the window with ONE fused call (smalfit_fit_eval: LBS, projection, soft silhouette,
all loss terms and the gradient of their sum); the returned loss carries a custom autograd node that
hands those gradients to the five parameter tensors.  The entries of `objs` are detached scalars (the
reference only prints them, optimize_to_joints.py:123)."""
from __future__ import annotations

import os
import pickle as pkl

import numpy as np
import torch
import torch.nn as nn

from .. import config, engine as eng, model_io, runtime
from ..smal_model.smal_torch import SMAL
from .p3d_renderer import Renderer
from .priors.pose_prior_35 import Prior

_TERMS = ("joint", "pose", "splay", "betas", "sil_reproj")


class _WindowLoss(torch.autograd.Function):
    """loss of one window = sum of the weighted terms; gradients come from the same fused HIP evaluation"""

    @staticmethod
    def forward(ctx, fitter, batch_range, weights, betas, log_beta_scales, global_rotation, joint_rotations, trans):
        idx = torch.as_tensor(batch_range, device=global_rotation.device, dtype=torch.long)
        grot = global_rotation.detach()[idx].contiguous()
        jrot = joint_rotations.detach()[idx].contiguous()
        tr = trans.detach()[idx].contiguous()
        b = betas.detach().contiguous()
        per_frame_ls = log_beta_scales.dim() == 2
        ls = (log_beta_scales.detach()[idx] if per_frame_ls else log_beta_scales.detach()).contiguous()
        M = len(batch_range)
        e = fitter._engine(M)
        tj = fitter.target_joints[idx].contiguous()
        vis_all = fitter.target_visibility          # the reference driver swaps in a CPU float tensor after stage 0
        vis = vis_all.to(idx.device)[idx].float().contiguous()
        sil = fitter.sil_imgs[idx].reshape(M, fitter.image_size, fitter.image_size).contiguous()
        losses, grads = e.fit_eval(betas=b, log_beta_scales=ls, global_rotation=grot, joint_rotations=jrot, trans=tr,
                                   target_joints=tj, target_visibility=vis, target_sil=sil, weights=weights, w_temp=0.0,
                                   window=M, temporal=False, global_mask=fitter.global_mask.reshape(3).contiguous(),
                                   rotation_mask=fitter.rotation_mask.contiguous())
        ctx.saved = (grads, idx, per_frame_ls, global_rotation.shape, joint_rotations.shape, trans.shape,
                     log_beta_scales.shape)
        ctx.mark_non_differentiable(losses)
        return losses[:5].sum() + losses[8], losses          # the window's terms (temporal is separate); [8] = joint limits

    @staticmethod
    def backward(ctx, gtotal, _glosses):
        grads, idx, per_frame_ls, s_g, s_j, s_t, s_ls = ctx.saved
        dev = grads["trans"].device

        def scatter(g, shape):
            out = torch.zeros(shape, device=dev)
            out[idx] = g
            return out * gtotal

        g_ls = scatter(grads["log_beta_scales"], s_ls) if per_frame_ls else grads["log_beta_scales"] * gtotal
        return (None, None, None, grads["betas"] * gtotal, g_ls, scatter(grads["global_rotation"], s_g),
                scatter(grads["joint_rotations"], s_j), scatter(grads["trans"], s_t))


class _Temporal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fitter, w_temp, global_rotation, joint_rotations, trans):
        e = fitter._engine(global_rotation.shape[0])
        losses, gg, gj, gt = eng.temporal(e, w_temp, global_rotation.detach().contiguous(),
                                          joint_rotations.detach().contiguous(), trans.detach().contiguous(),
                                          fitter.global_mask.reshape(3).contiguous(), fitter.rotation_mask.contiguous())
        ctx.saved = (gg, gj, gt)
        return losses[0], losses[1], losses[2]

    @staticmethod
    def backward(ctx, dj, dg, dt):
        gg, gj, gt = ctx.saved
        # the three terms touch disjoint parameters: joint -> joint_rotations, global -> global_rotation, trans -> trans
        return None, None, gg * dg, gj * dj, gt * dt


class SMALFitter(nn.Module):
    def __init__(self, device, data_batch, batch_size, shape_family, use_unity_prior, model_data=None,
                 pose_prior_data=None, shape_prior_data=None, enable_joint_limits=False):
        """model_data / pose_prior_data / shape_prior_data let tests inject the synthetic stand-ins; by default
        everything is read from the paths in smalify_amd.config exactly like the reference (smal_fitter.py:40-74).
        enable_joint_limits: switch on the w_limit term the reference has commented out (smal_fitter.py:76-79,146-151)."""
        super().__init__()
        self.enable_joint_limits = bool(enable_joint_limits)
        self.rgb_imgs, self.sil_imgs, self.target_joints, self.target_visibility = data_batch
        dev = torch.device("cuda", torch.cuda.current_device())
        self.target_visibility = self.target_visibility.long().to(dev)
        assert self.rgb_imgs.max() <= 1.0 and self.rgb_imgs.min() >= 0.0, "RGB Image range is incorrect"
        self.device = dev
        self.num_images = self.rgb_imgs.shape[0]
        self.image_size = self.rgb_imgs.shape[2]
        self.use_unity_prior = use_unity_prior
        self.batch_size = batch_size
        self.n_betas = config.N_BETAS
        self.sil_imgs = self.sil_imgs.float().to(dev)
        self.target_joints = self.target_joints.float().to(dev)

        if shape_prior_data is not None:
            prec, mean = shape_prior_data
        elif use_unity_prior:
            prec, mean = model_io.unity_shape_prior(config.UNITY_SHAPE_PRIOR)
        else:
            prec, mean = model_io.family_shape_prior(model_io.load_pickle(config.SMAL_DATA_FILE), shape_family)
        self._shape_prior = (prec, mean)
        self.betas_prec = torch.from_numpy(prec).to(dev)
        self.mean_betas = torch.from_numpy(mean).to(dev)
        if use_unity_prior:
            self.betas = nn.Parameter(self.mean_betas[:20].clone())
            self.log_beta_scales = nn.Parameter(self.mean_betas[20:].clone())
        else:
            self.betas = nn.Parameter(self.mean_betas.clone())
            self.log_beta_scales = nn.Parameter(torch.zeros(self.num_images, 6, device=dev), requires_grad=False)

        self.pose_prior = Prior(config.WALKING_PRIOR_FILE, dev, prior_data=pose_prior_data)
        init = torch.from_numpy(model_io.initial_global_rotation()).float().to(dev)
        self.global_rotation = nn.Parameter(init.unsqueeze(0).repeat(self.num_images, 1))
        self.trans = nn.Parameter(torch.zeros(self.num_images, 3, device=dev))
        self.joint_rotations = nn.Parameter(torch.zeros(self.num_images, config.N_POSE, 3, device=dev))
        self.global_mask = torch.ones(1, 3, device=dev)
        self.rotation_mask = torch.ones(config.N_POSE, 3, device=dev)
        self.smal_model = SMAL(dev, shape_family_id=shape_family, model_data=model_data,
                               engine_image_size=self.image_size)
        self.renderer = Renderer(self.image_size, dev, model=self.smal_model.device_model)

    def _engine(self, frames):
        e = runtime.get_engine(self.smal_model.device_model, max(frames, self.batch_size), self.image_size)
        if getattr(e, "_fitter_priors", None) is not self:
            e.set_pose_prior(*self.pose_prior._data)
            e.set_shape_prior(*self._shape_prior)
            if self.enable_joint_limits:                            # reference smal_fitter.py:76-79, commented out there
                e.set_joint_limits(*model_io.joint_limit_table(), owner=self)
            else:                                                   # a shared engine may carry another fitter's table
                e.clear_joint_limits()
            e._pose_prior, e._shape_prior, e._fitter_priors = self.pose_prior._data, self._shape_prior, self
        return e

    def forward(self, batch_range, weights, stage_id):
        weights = [float(w) for w in weights]
        if not self.enable_joint_limits:        # like the reference: the weight table says 100, the term does not exist
            weights[4] = 0.0
        total, losses = _WindowLoss.apply(self, list(batch_range), weights, self.betas,
                                          self.log_beta_scales, self.global_rotation, self.joint_rotations, self.trans)
        w_j2d, w_reproj, w_betas, w_pose, w_limit, w_splay = [float(w) for w in weights]
        active = dict(joint=w_j2d > 0, pose=w_pose > 0, splay=w_splay > 0, betas=w_betas > 0, sil_reproj=w_reproj > 0)
        objs = {}
        for i, k in enumerate(_TERMS):
            if active[k]:
                objs[k] = losses[i]
            if k == "joint" and w_limit > 0 and self.enable_joint_limits:   # the reference's (disabled) term order: joint, limit, pose, ...
                objs["limit"] = losses[8]
        return total, objs

    def get_temporal(self, w_temp):
        return _Temporal.apply(self, float(w_temp), self.global_rotation, self.joint_rotations, self.trans)

    def load_checkpoint(self, checkpoint_path, epoch):
        """reference smal_fitter.py:192-207"""
        beta_list, scale_list = [], []
        with torch.no_grad():
            for frame_id in range(self.num_images):
                with open(os.path.join(checkpoint_path, "{0:04}".format(frame_id), "{0}.pkl".format(epoch)), "rb") as f:
                    p = pkl.load(f)
                self.global_rotation[frame_id] = torch.from_numpy(np.asarray(p["global_rotation"])).float().to(self.device)
                self.joint_rotations[frame_id] = torch.from_numpy(np.asarray(p["joint_rotations"])).float().to(
                    self.device).view(config.N_POSE, 3)
                self.trans[frame_id] = torch.from_numpy(np.asarray(p["trans"])).float().to(self.device)
                beta_list.append(np.asarray(p["betas"])[:self.n_betas])
                scale_list.append(np.asarray(p["log_betascale"]))
        self.betas = nn.Parameter(torch.from_numpy(np.mean(beta_list, axis=0)).float().to(self.device))
        self.log_beta_scales = nn.Parameter(torch.from_numpy(np.mean(scale_list, axis=0)).float().to(self.device))

    def frame_parameters(self, batch_range):
        """per-frame parameter dicts with the reference's keys (smal_fitter.py:213-219,268)"""
        with torch.no_grad():
            ls = self.log_beta_scales
            out = []
            for i in batch_range:
                out.append({"global_rotation": (self.global_rotation[i] * self.global_mask[0]).cpu().numpy(),
                            "joint_rotations": (self.joint_rotations[i] * self.rotation_mask).cpu().numpy(),
                            "betas": self.betas.detach().cpu().numpy(),
                            "log_betascale": (ls if ls.dim() == 1 else ls[i]).detach().cpu().numpy(),
                            "trans": self.trans[i].cpu().numpy()})
            return out

    @staticmethod
    def _draw_joints(images, landmarks, visible=None):
        """Marks keypoints on (B,3,S,S) images in [0,1]: landmarks (B,25,2) as (row, col).  Stands in for the reference's
        cv2 marker drawer (draw_smal_joints.py:22-46): plus-shaped marks, one hue per limb group; invisible joints are
        parked along the top edge like there."""
        imgs = np.transpose(np.asarray(images.detach().cpu().numpy(), np.float32), (0, 2, 3, 1)).copy()
        lm = np.asarray(landmarks.detach().cpu().numpy())
        vis = np.ones(lm.shape[:2], bool) if visible is None else np.asarray(visible.detach().cpu().numpy()) > 0
        S = imgs.shape[1]
        for b in range(imgs.shape[0]):
            parked = 0
            for j in range(lm.shape[1]):
                h = (j // 3) / 9.0
                col = np.clip(np.abs((h * 6.0 + np.array([0.0, 4.0, 2.0])) % 6.0 - 3.0) - 1.0, 0.0, 1.0)
                r, c = (int(round(lm[b, j, 0])), int(round(lm[b, j, 1]))) if vis[b, j] else (0, parked * 10)
                parked += 0 if vis[b, j] else 1
                for dr, dc in [(0, k) for k in range(-4, 5)] + [(k, 0) for k in range(-4, 5)]:
                    rr, cc = r + dr, c + dc
                    if 0 <= rr < S and 0 <= cc < S:
                        imgs[b, rr, cc] = col
        return torch.from_numpy(np.transpose(imgs, (0, 3, 1, 2)))

    def generate_visualization(self, image_exporter):
        """reference smal_fitter.py:209-272: per frame a five-panel collage (target + keypoints | colour render +
        projected keypoints | overlay | 1 - |silhouette error| | the mesh seen from behind), the parameter dict and the
        posed mesh go to image_exporter.export.  The colour renders come from smalfit_render_color; the keypoint marks
        are drawn in numpy instead of cv2."""
        rot_y180 = torch.tensor([[-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]], device=self.device)
        for j in range(0, self.num_images, self.batch_size):
            batch_range = list(range(j, min(self.num_images, j + self.batch_size)))
            with torch.no_grad():
                theta = torch.cat([(self.global_rotation[batch_range] * self.global_mask).unsqueeze(1),
                                   self.joint_rotations[batch_range] * self.rotation_mask], dim=1)
                ls = self.log_beta_scales
                ls = ls.expand(len(batch_range), 6) if ls.dim() == 1 else ls[batch_range]
                verts, joints, _, _ = self.smal_model(self.betas.expand(len(batch_range), self.n_betas).contiguous(),
                                                      theta.contiguous(), betas_logscale=ls.contiguous())
                trans = self.trans[batch_range].unsqueeze(1)
                verts = verts + trans
                canonical = (joints + trans)[:, config.CANONICAL_MODEL_JOINTS]
                sil, proj, rendered = self.renderer(verts, canonical, None, render_texture=True)
                centre = verts.mean(dim=1, keepdim=True)
                _, rev_proj, rev_rendered = self.renderer(((verts - centre) @ rot_y180.T).contiguous(),
                                                          ((canonical - centre) @ rot_y180.T).contiguous(), None,
                                                          render_texture=True)
                rgb = self.rgb_imgs[batch_range].to(self.device).float()
                vis = self.target_visibility[batch_range]
                overlay = rendered * 0.8 + rgb * 0.2
                sil_err = (1.0 - (self.sil_imgs[batch_range].reshape(sil.shape) - sil).abs()).expand(-1, 3, -1, -1).cpu()
                collage = torch.cat([self._draw_joints(rgb, self.target_joints[batch_range], vis),
                                     self._draw_joints(rendered, proj, vis), self._draw_joints(overlay, proj, vis),
                                     sil_err, self._draw_joints(rev_rendered, rev_proj, vis)], dim=3)
            for batch_id, (global_id, params) in enumerate(zip(batch_range, self.frame_parameters(batch_range))):
                collage_np = (np.transpose(collage[batch_id].numpy(), (1, 2, 0)) * 255.0).astype(np.uint8)
                image_exporter.export(collage_np, batch_id, global_id, params, verts, self.smal_model.f)
