#!/usr/bin/env python3
"""Workload statistics of the soft-silhouette rasteriser for YOUR model, in one command (needs an MI355X: the posed vertices
come from the engine's own LBS kernels; everything after that is numpy on the host):

    python -m smalify_amd.tools.work_stats --smal my_smpl_00781_4_all.pkl --data my_smpl_data_00781_4_all.pkl --sym symIdx.pkl
    python -m smalify_amd.tools.work_stats --synthetic                       (the stand-in model of smalify_amd/synthetic.py)

What decides the time of every rasteriser kernel is a property of the MESH in the image, not of the
engine: how many pixels the faces' blur-expanded boxes hold (= (face, pixel) pairs the sweep visits per launch), how they are
spread over box sizes (which candidate-list format a face gets: list <= 256 px, masks <= 1024 px, box walk beyond), how many
candidates a pixel sees and how many pixels see more than K = 100 (the exact-selection regime).  Every number this repository
quotes was measured on a procedural stand-in mesh (tools/make_synth_mesh.py: near-uniform triangles); a real SMAL template has
dense heads / paws and long thin leg faces.  This prints the same table for the model it is given, for the reference's initial
state (smal_fitter.py:81-89: the animal a third of the image wide, head-on) and for a pose as a converged fit has it (a smooth
random ground-truth pose, BASELINE.md section 4, moved towards the camera until the animal fills the crop the way the
reference's loaders deliver it, utils.py:5-36) -- next to the stand-in's table under profiles/r6_work_stats_standin.txt.

Reads what smal_torch.py:36-96 reads from the pickle (f, v_template, shapedirs, posedirs, J_regressor, weights, kintree_table),
the family means of the data pickle and symIdx (config.py:32-49)."""
import argparse
import math
import os
import sys

import numpy as np

K = 100                                            # faces_per_pixel, p3d_renderer.py:31
SIGMA = 1e-4
BLUR = math.log(1.0 / 1e-4 - 1.0) * SIGMA          # p3d_renderer.py:26-30
CAM_DIST, CAM_SCALE = 2.7, 1.0 / math.tan(math.radians(30.0))
BOX_BINS = (16, 64, 128, 256, 512, 1024)
CAND_BINS = (1, 10, 25, 50, 100, 200, 400)


def project(verts):
    """world -> (x_ndc, y_ndc, z_view): look_at_view_transform(2.7, 0, 0) + OpenGL perspective, fov 60"""
    zv = CAM_DIST - verts[..., 2]
    return -verts[..., 0] * CAM_SCALE / zv, verts[..., 1] * CAM_SCALE / zv, zv


def face_boxes(x, y, faces, S):
    """the pixel box of every face exactly as face_bbox_kernel forms it: the pixels whose centres lie in the sqrt(blur)-expanded
    bounding box.  -> c0, c1, r0, r1 (inclusive; empty: c1 < c0)"""
    r = math.sqrt(BLUR)
    fx, fy = x[faces], y[faces]
    xlo, xhi, ylo, yhi = fx.min(1) - r, fx.max(1) + r, fy.min(1) - r, fy.max(1) + r
    c0 = np.ceil(((1.0 - xhi) * S - 1.0) * 0.5 - 1.0 / 64).clip(0, S - 1)
    c1 = np.floor(((1.0 - xlo) * S - 1.0) * 0.5 + 1.0 / 64).clip(-1, S - 1)
    r0 = np.ceil(((1.0 - yhi) * S - 1.0) * 0.5 - 1.0 / 64).clip(0, S - 1)
    r1 = np.floor(((1.0 - ylo) * S - 1.0) * 0.5 + 1.0 / 64).clip(-1, S - 1)
    on = (xhi >= -1) & (xlo <= 1) & (yhi >= -1) & (ylo <= 1) & (c1 >= c0) & (r1 >= r0)
    c1 = np.where(on, c1, c0 - 1)
    return c0.astype(np.int64), c1.astype(np.int64), r0.astype(np.int64), r1.astype(np.int64)


def seg_dist2(px, py, ax, ay, bx, by):
    ex, ey = bx - ax, by - ay
    l2 = ex * ex + ey * ey
    t = np.clip(((px - ax) * ex + (py - ay) * ey) / np.where(l2 > 1e-8, l2, 1.0), 0.0, 1.0)
    t = np.where(l2 > 1e-8, t, 1.0)
    qx, qy = ax + t * ex - px, ay + t * ey - py
    return qx * qx + qy * qy


def frame_stats(verts, faces, S):
    """one frame: box sizes, and per pixel the number of candidate faces (the naive rasteriser's inclusion test)"""
    x, y, z = project(verts.astype(np.float64))
    c0, c1, r0, r1 = face_boxes(x, y, faces, S)
    w, h = np.maximum(c1 - c0 + 1, 0), np.maximum(r1 - r0 + 1, 0)
    npx = np.where(w > 0, w * h, 0)
    total = int(npx.sum())
    face = np.repeat(np.arange(len(faces)), npx)
    local = np.arange(total) - np.repeat(np.cumsum(npx) - npx, npx)
    wf = w[face]
    rr, cc = r0[face] + local // wf, c0[face] + local % wf
    px, py = 1.0 - (2.0 * cc + 1.0) / S, 1.0 - (2.0 * rr + 1.0) / S
    f = faces[face]
    ax, ay, az, bx, by, bz, cx, cy, cz = x[f[:, 0]], y[f[:, 0]], z[f[:, 0]], x[f[:, 1]], y[f[:, 1]], z[f[:, 1]], x[f[:, 2]], y[f[:, 2]], z[f[:, 2]]
    area = (cx - ax) * (by - ay) - (cy - ay) * (bx - ax)
    den = area + 1e-8
    w0 = ((px - bx) * (cy - by) - (py - by) * (cx - bx)) / den
    w1 = ((px - cx) * (ay - cy) - (py - cy) * (ax - cx)) / den
    w2 = ((px - ax) * (by - ay) - (py - ay) * (bx - ax)) / den
    inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
    dist = np.minimum(np.minimum(seg_dist2(px, py, ax, ay, bx, by), seg_dist2(px, py, ax, ay, cx, cy)), seg_dist2(px, py, bx, by, cx, cy))
    cand = (np.abs(area) > 1e-8) & (np.maximum(np.maximum(az, bz), cz) >= 0) & (w0 * az + w1 * bz + w2 * cz >= 0) & (inside | (dist < BLUR))
    per_pixel = np.bincount((rr * S + cc)[cand], minlength=S * S)
    per_face = np.bincount(face[cand], minlength=len(faces))
    return npx, per_pixel, per_face, int(cand.sum())


def hist_line(values, weights, bins, unit):
    edges = (0,) + tuple(bins) + (1 << 62,)
    n, wsum = max(len(values), 1), max(float(weights.sum()), 1.0)
    parts = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (values > lo) & (values <= hi)
        label = ("<= %d" % hi) if lo == 0 else (("> %d" % lo) if hi > (1 << 60) else "%d-%d" % (lo, hi))
        parts.append("%s %s: %.1f %% (%.1f %%)" % (label, unit, 100.0 * m.sum() / n, 100.0 * weights[m].sum() / wsum))
    return "; ".join(parts)


def report(name, verts, faces, S, out):
    boxes, ppx, pfc, ncand = [], [], [], 0
    for v in verts:
        npx, per_pixel, per_face, nc = frame_stats(v, faces, S)
        boxes.append(npx); ppx.append(per_pixel); pfc.append(per_face); ncand += nc
    npx, per_pixel, per_face = np.concatenate(boxes), np.concatenate(ppx), np.concatenate(pfc)
    live = npx > 0
    N, F = len(verts), len(faces)
    covered = per_pixel > 0
    out("%s  (%d frames of %d x %d, %d faces)" % (name, N, S, S, F))
    out("  (face, pixel) pairs the sweep visits per frame: %.2f M  (= box pixels; %.1f per face on screen, %.1f %% of the faces on screen); of these candidates: %.3f"
        % (npx.sum() / N / 1e6, npx[live].mean() if live.any() else 0.0, 100.0 * live.mean(), ncand / max(float(npx.sum()), 1.0)))
    out("  box size, share of the faces (share of the pairs): " + hist_line(npx[live], npx[live].astype(np.float64), BOX_BINS, "px"))
    lst = live & (npx <= 256) & (per_face <= 128)
    msk = live & ~lst & (npx <= 1024)
    out("  candidate-list format a face gets on its own: list %.1f %%, masks %.1f %% (box > 256 px: %.1f %%, more than 128 candidates: %.1f %%), whole box %.1f %%   [largest box %d px]"
        % (100.0 * lst.sum() / max(live.sum(), 1), 100.0 * msk.sum() / max(live.sum(), 1), 100.0 * (live & (npx > 256) & (npx <= 1024)).sum() / max(live.sum(), 1),
           100.0 * (live & (npx <= 256) & (per_face > 128)).sum() / max(live.sum(), 1), 100.0 * (live & (npx > 1024)).sum() / max(live.sum(), 1), int(npx.max())))
    out("  silhouette coverage: %.1f %% of the pixels see a candidate; candidates per covered pixel: mean %.1f, median %.0f, max %d"
        % (100.0 * covered.mean(), per_pixel[covered].mean() if covered.any() else 0.0, np.median(per_pixel[covered]) if covered.any() else 0.0, int(per_pixel.max())))
    out("  candidates per covered pixel, share of the pixels (share of the candidates): " +
        hist_line(per_pixel[covered], per_pixel[covered].astype(np.float64), CAND_BINS, ""))
    out("  K-overflow: %.1f %% of the covered pixels see more than K = %d candidates (they need the depth bounds / the exact selection); they hold %.1f %% of the candidates"
        % (100.0 * (per_pixel > K).sum() / max(covered.sum(), 1), K, 100.0 * per_pixel[per_pixel > K].sum() / max(float(per_pixel.sum()), 1.0)))


def posed_vertices(md, params, S):
    import torch
    from smalify_amd import engine as eng
    N = params["trans"].shape[0]
    e = eng.Engine(eng.DeviceModel(md), N, S)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device="cuda").contiguous()  # noqa: E731
    verts = torch.empty(N, md.num_verts, 3, device="cuda")
    e.fit_eval(betas=t(params["betas"]), log_beta_scales=t(params["log_beta_scales"]), global_rotation=t(params["global_rotation"]),
               joint_rotations=t(params["joint_rotations"]), trans=t(params["trans"]), target_joints=None, target_visibility=None,
               target_sil=None, weights=(0, 0, 0, 0, 0, 0), w_temp=0.0, window=N, want=(), verts_out=verts)
    assert e.status() == 0
    return verts.cpu().numpy()


def fill_crop(md, params, S, fill=0.8):
    """move the pose along the view axis until the silhouette's larger extent is `fill` of the image: what crop_to_silhouette delivers
    (utils.py:5-36 pads the silhouette's bounding square by a margin).  Three fixed-point steps on the mean extent of the frames."""
    p = {k: np.array(v, copy=True) for k, v in params.items()}
    for _ in range(4):
        v = posed_vertices(md, p, S)
        x, y, z = project(v.astype(np.float64))
        ext = np.maximum(x.max(1) - x.min(1), y.max(1) - y.min(1)).mean() / 2.0          # fraction of the image
        zbar = float(z.mean())
        p["trans"][:, 2] += zbar * (1.0 - ext / fill)                                    # extent ~ 1 / z_view
    return p


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--smal", help="SMAL model pickle (config.SMAL_FILE)")
    ap.add_argument("--data", help="SMAL data pickle with cluster_means (config.SMAL_DATA_FILE); needed with --family >= 0")
    ap.add_argument("--sym", help="symIdx pickle (config.SMAL_SYM_FILE)")
    ap.add_argument("--family", type=int, default=1, help="shape family (config.SHAPE_FAMILY; -1: plain template)")
    ap.add_argument("--synthetic", action="store_true", help="the procedural stand-in model instead of pickles")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--out", help="also write the table to this file")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from smalify_amd import model_io, synthetic
    if args.synthetic:
        md, what = synthetic.synthetic_model(seed=0, shape_family_id=args.family), "procedural stand-in (smalify_amd/synthetic.py, tools/make_synth_mesh.py)"
    else:
        if not (args.smal and args.sym):
            ap.error("--smal and --sym are required (or --synthetic)")
        md, what = model_io.load_smal_model(args.smal, args.data, args.sym, args.family), os.path.basename(args.smal)
    lines = []

    def out(s):
        print(s, flush=True)
        lines.append(s)
    faces = np.asarray(md.faces).astype(np.int64)
    N, S = args.frames, args.image_size
    out("rasteriser workload of model %s: V = %d, F = %d, shape family %d, K = %d, blur radius %.2f px" % (what, md.num_verts, md.num_faces, args.family, K, math.sqrt(BLUR) * S / 2))
    gt = synthetic.ground_truth_params(N, seed=1234)
    init = dict(gt, global_rotation=np.tile(model_io.initial_global_rotation(), (N, 1)).astype(np.float32),
                joint_rotations=np.zeros_like(gt["joint_rotations"]), trans=np.zeros_like(gt["trans"]))
    report("initial state of a fit (smal_fitter.py:81-89)", posed_vertices(md, init, S), faces, S, out)
    report("a converged pose as the benchmark's headline scene has it (ground-truth draw of BASELINE.md section 4)", posed_vertices(md, gt, S), faces, S, out)
    report("the same pose filling the crop (what crop_to_silhouette delivers, utils.py:5-36)", posed_vertices(md, fill_crop(md, gt, S), S), faces, S, out)
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
