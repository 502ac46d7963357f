"""Drop-in for reference smal_fitter/p3d_renderer.py :: Renderer without PyTorch3D.

    Renderer(image_size, device)(vertices, points, faces, render_texture=False) -> sil (N,1,S,S), proj (N,P,2)

Fixed camera (look_at_view_transform(2.7,0,0) + OpenGL perspective fov 60), soft silhouette with
sigma = 1e-4, blur = log(1/1e-4 - 1)*sigma, 100 faces per pixel, keypoints returned as (row, col):
all HIP (smalfit_render_forward / _backward, smalfit_project_points_backward).  render_texture=True adds the
colour image of the reference's second renderer (hard rasterisation + HardPhongShader, mesh colour
config.MESH_COLOR, white background; smalfit_render_color) -- visualisation only, it carries no gradient."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import runtime


class _Silhouette(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, verts):
        verts = verts.contiguous().float()
        e = owner._engine(verts.shape[0])
        sil, _ = e.render_forward(verts, None)
        ctx.owner = owner
        ctx.save_for_backward(verts, sil)
        return sil

    @staticmethod
    def backward(ctx, dsil):
        verts, sil = ctx.saved_tensors
        e = ctx.owner._engine(verts.shape[0])
        return None, e.render_backward(verts, sil, dsil.contiguous().float())


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, points):
        points = points.contiguous().float()
        ctx.owner = owner
        ctx.save_for_backward(points)
        return owner._project(points)

    @staticmethod
    def backward(ctx, dproj):
        (points,) = ctx.saved_tensors
        e = ctx.owner._engine(points.shape[0])
        return None, e.project_points_backward(points, dproj.contiguous().float())


class Renderer(nn.Module):
    def __init__(self, image_size, device, model=None):
        super().__init__()
        self.image_size = int(image_size)
        self.device_model = model

    def _engine(self, frames):
        return runtime.get_engine(self.device_model, frames, self.image_size)

    def _project(self, points):
        from .. import engine as eng
        e = self._engine(points.shape[0])
        proj = torch.empty(points.shape[0], points.shape[1], 2, device=points.device)
        # verts argument is unused when sil is NULL
        eng.check(e.lib.smalfit_render_forward(e.handle, eng._stream(), int(points.shape[0]), eng._ptr(points),
                                               eng._ptr(points), int(points.shape[1]), None, eng._ptr(proj)),
                  "smalfit_render_forward")
        return proj

    def _check_faces(self, faces, e):
        """the rasteriser always draws the topology registered with the engine: refuse anything else (checked by content
        once per faces tensor, then by identity)"""
        key = (faces.data_ptr(), tuple(faces.shape), faces._version)
        if getattr(self, "_faces_ok", None) == key:
            return
        f = faces.reshape(-1, faces.shape[-2], 3)[0] if faces.dim() > 2 else faces
        ref = torch.as_tensor(np.asarray(e.model.data.faces), device=f.device)
        if tuple(f.shape) != tuple(ref.shape) or not bool((f.to(ref.dtype) == ref).all()):
            raise ValueError("faces do not match the SMAL topology the rasteriser was built for")
        self._faces_ok = key

    def forward(self, vertices, points, faces, render_texture=False):
        e = self._engine(vertices.shape[0])
        if faces is not None:
            self._check_faces(faces, e)
        sil = _Silhouette.apply(self, vertices)
        proj = _Project.apply(self, points)
        if render_texture:
            from .. import config
            with torch.no_grad():
                color = e.render_color(vertices.detach().contiguous().float(), [c / 255.0 for c in config.MESH_COLOR])
            return sil.unsqueeze(1), proj, color
        return sil.unsqueeze(1), proj
