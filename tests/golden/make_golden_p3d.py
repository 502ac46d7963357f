#!/usr/bin/env python3
"""GUARDED generator of tests/golden/reference_golden_p3d.npz: the one fixture this repository cannot produce itself.

The soft-silhouette arithmetic of the path lives in pytorch3d==0.2.5 (reference requirements.txt:60; call sites
smal_fitter/p3d_renderer.py:22-39,65-68), which is neither vendored in the reference nor installable in the build container
(no network), so oracle/smal_oracle.py's renderer is "parity unpinned".  Anyone who HAS pytorch3d 0.2.5 turns that into a pin
with one command:

    python tests/golden/make_golden_p3d.py            (needs `import pytorch3d` to succeed; CPU is enough)

It renders the synthetic SMAL-topology model (smalify_amd/synthetic.py) through the REFERENCE'S OWN `Renderer` class when the
reference checkout is importable ($SMALIFY_REFERENCE or /root/reference), else through the same pytorch3d objects built here
from the reference's constants (camera look_at_view_transform(2.7, 0, 0) + OpenGLPerspectiveCameras, BlendParams(sigma = gamma =
1e-4), blur_radius = log(1/1e-4 - 1) sigma, faces_per_pixel = 100, SoftSilhouetteShader), and stores inputs and outputs:

    general view, 32 x 32 and 64 x 64:  silhouette, projected keypoints (row, col), d(sum w * sil)/d(verts) for a random w
    head-on view (the reference's initial pose), 32 x 32: every covered pixel has far more than 100 candidate faces -- the
        K = 100 nearest-in-depth truncation, its tie handling and the pz >= 0 / kEpsilon culls are what this case pins

    vertex-nearest anchor (tests/raster_anchors.py::vertex_nearest_gradient): one triangle, one pixel whose nearest feature is a
        VERTEX -- d sil / d verts there tells whether this pytorch3d clamps the edge parameter t in PointLineDistanceBackward
        (last row).  The two conventions are different closed forms; tests/test_p3d_fixture.py reports which one the
        fixture follows, i.e. whether smalfit_engine_set_option(SMALFIT_OPT_UNCLAMPED_EDGE_T, 1) is the setting that reproduces it.

tests/test_p3d_fixture.py consumes the file when it exists (oracle on the CPU, HIP kernels on the GPU) and is reported as
skipped otherwise.  Without pytorch3d this script exits with a message and writes nothing.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)
OUT = os.path.join(HERE, "reference_golden_p3d.npz")


def build_renderer(image_size, torch):
    ref = os.environ.get("SMALIFY_REFERENCE", "/root/reference")
    if os.path.isdir(os.path.join(ref, "smal_fitter")):
        sys.path.insert(0, ref)
        sys.path.insert(0, os.path.join(ref, "smal_fitter"))
        try:
            from p3d_renderer import Renderer                      # the reference's own class
            return Renderer(image_size, "cpu"), "reference smal_fitter/p3d_renderer.py::Renderer"
        except Exception as exc:                                   # e.g. its other imports (cv2 ...) are missing
            print("reference Renderer not importable (%s): building the pytorch3d objects directly" % exc)
    from pytorch3d.renderer import (BlendParams, MeshRasterizer, MeshRenderer, OpenGLPerspectiveCameras, RasterizationSettings,
                                    SoftSilhouetteShader, look_at_view_transform)
    from pytorch3d.structures import Meshes
    rot, tra = look_at_view_transform(2.7, 0, 0)
    cams = OpenGLPerspectiveCameras(R=rot, T=tra)
    blend = BlendParams(sigma=1e-4, gamma=1e-4)
    settings = RasterizationSettings(image_size=image_size, blur_radius=float(np.log(1.0 / 1e-4 - 1.0) * blend.sigma), faces_per_pixel=100)
    soft = MeshRenderer(rasterizer=MeshRasterizer(cameras=cams, raster_settings=settings), shader=SoftSilhouetteShader(blend_params=blend))

    class Direct(torch.nn.Module):
        def forward(self, vertices, points, faces):
            sil = soft(Meshes(verts=vertices, faces=faces))[..., -1].unsqueeze(1)
            size = torch.ones(vertices.shape[0], 2) * image_size
            return sil, cams.transform_points_screen(points, size)[:, :, [1, 0]]

    return Direct(), "pytorch3d objects built from the reference's constants (p3d_renderer.py:22-39,65-68)"


def main():
    try:
        import pytorch3d
    except Exception as exc:
        raise SystemExit("pytorch3d is not importable here (%s): nothing written.  Run this where pytorch3d==0.2.5 is installed." % exc)
    import torch
    from oracle import smal_oracle as so
    from smalify_amd import model_io, synthetic
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    om = so.OracleModel(md)
    faces = torch.from_numpy(np.asarray(md.faces).astype(np.int64))
    rs = np.random.RandomState(2025)
    out = {"pytorch3d_version": np.array(str(getattr(pytorch3d, "__version__", "unknown")))}

    def pose(kind, M):
        init = model_io.initial_global_rotation()
        if kind == "head_on":                                      # reference initial state (smal_fitter.py:81-89)
            grot, jrot, trans = np.tile(init, (M, 1)), np.zeros((M, 34, 3)), np.zeros((M, 3))
        else:
            grot = init[None] + 0.25 * rs.randn(M, 3)
            jrot = 0.2 * rs.randn(M, 34, 3)
            trans = np.array([0.03, -0.02, 1.45])[None] + 0.03 * rs.randn(M, 3)
        theta = np.concatenate([grot[:, None], jrot], 1)
        with torch.no_grad():
            v, j, _, _ = so.smal_forward(om, torch.zeros(M, 20).double(), torch.from_numpy(theta).double(), torch.zeros(M, 6).double())
        t = torch.from_numpy(trans).double()[:, None]
        return (v + t).float(), (j + t)[:, so.CANONICAL].float()

    for tag, kind, S, M in (("general32", "general", 32, 2), ("general64", "general", 64, 1), ("headon32", "head_on", 32, 1)):
        renderer, how = build_renderer(S, torch)
        verts, pts = pose(kind, M)
        verts.requires_grad_(True)
        res = renderer(verts, pts, faces[None].expand(M, -1, -1))
        sil, proj = res[0], res[1]
        w = torch.from_numpy(rs.randn(M, 1, S, S).astype(np.float32))
        (sil * w).sum().backward()
        out.update({tag + "_verts": verts.detach().numpy(), tag + "_points": pts.numpy(), tag + "_sil": sil.detach().numpy()[:, 0],
                    tag + "_proj": proj.detach().numpy(), tag + "_w": w.numpy()[:, 0], tag + "_dverts": verts.grad.numpy(),
                    tag + "_image_size": np.array(S)})
        out["produced_with"] = np.array(how)
        print(tag, "coverage %.3f" % float((sil > 0.5).float().mean()), how)
    # the vertex-nearest anchor: d sil[row, col] / d verts of ONE triangle (answers SURVEY App. B's open question for this pytorch3d)
    from tests import raster_anchors as ra
    av, af, aS, (arow, acol), _ = ra.vertex_nearest_gradient()
    renderer, how = build_renderer(aS, torch)
    averts = torch.from_numpy(av[None].astype(np.float32)).requires_grad_(True)
    ares = renderer(averts, torch.zeros(1, 1, 3), torch.from_numpy(af.astype(np.int64))[None])
    ares[0][0, 0, arow, acol].backward()
    out.update({"anchor_vertex_sil": ares[0].detach().numpy()[0, 0, arow, acol], "anchor_vertex_dverts": averts.grad.numpy()[0]})
    print("vertex-nearest anchor: sil %.6f, d sil / d verts" % float(out["anchor_vertex_sil"]), out["anchor_vertex_dverts"].tolist())
    out["faces"] = np.asarray(md.faces).astype(np.int32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
