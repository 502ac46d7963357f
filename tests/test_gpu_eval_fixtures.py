"""The HIP evaluation (per-term losses + every analytic gradient of one epoch objective, optimize_to_joints.py:113-137,
smal_fitter.py:107-190) against the float64 ORACLE at BASELINE.json's own sizes (-m gpu):

  crop8    8 frames, 256 x 256, the CROP-FILLING scene -- what BADJA / StanfordExtra input looks like after the reference's
           crop_to_silhouette (utils.py:5-36, data_loader.py:48,117) -- at four states incl. the HIP fit's own states after
           stage 1 and at the end, where the rasteriser's large-face paths run (faces without a byte candidate list, pixels
           outside the sweep's LDS window, box-walking backward)
  config3  the headline workload itself: 64 frames, 256 x 256, WINDOW 8 (BASELINE config 3), at three states

Fixtures: tests/golden/oracle_eval_<case>.npz (tests/golden/make_oracle_eval.py; tests/eval_cases.py defines the problems,
tests/test_oracle_golden.py pins the files to today's oracle).  Bounds: north_star's 1e-4 relative on every loss term and on
the total; 5e-4 relative L2 on every gradient tensor, or twice the deviation of the oracle's own float32 evaluation of the
same state where that is larger (printed next to every number: float32 arithmetic alone is worth ~1e-6 on terms and 3e-6 ..
1e-3 on gradients here, the latter at the end of a fit, where the gradient is what is left after the terms cancel).
The tables are printed past pytest's capture, so the driver's log of the GPU run shows them.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TERM_TOL = 1e-4
GRAD_TOL = 5e-4
# Regression ratchet (round 6): the bars above are north_star's and 20-30 x what is measured at most states, so an order of magnitude
# could be lost unnoticed.  tests/golden/hip_eval_measured.json records what THESE kernels measure per (case, state, quantity) --
# written by this test under SMALFIT_WRITE_EVAL_MEASURED=<path> on a GPU box, committed with the kernels -- and every deviation must
# also stay within RATCHET x its recorded value (or the floor: below it a deviation is float32 noise whose draw changes with any
# reordering of a sum).
RATCHET = 3.0
RATCHET_FLOOR_TERM, RATCHET_FLOOR_GRAD = 1e-6, 1e-5
MEASURED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hip_eval_measured.json")
YARD = 2.0        # ... or YARD x the float32 ORACLE's own deviation from its float64 self at that state, whichever is larger: near
                  # the end of a fit the gradient is what is left after the terms cancel, and float32 arithmetic alone (the oracle's
                  # included) is worth 1e-3 relative there (crop8 / hip_final: 0.8e-3 .. 1.1e-3 for the float32 oracle)


def _rel(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _along(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.dot(a - b, b) / max(np.dot(b, b), 1e-300))


def _setup(case):
    from tests import eval_cases as ec
    fx, tg = ec.load_fixture(case), ec.load_targets(case)
    if fx is None or tg is None:
        pytest.skip("tests/golden/oracle_eval_%s.npz missing: run tests/golden/make_oracle_eval.py" % case)
    from smalify_amd import engine as eng, fitter as fit, synthetic
    c = ec.CASES[case]
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    e = eng.Engine(eng.DeviceModel(md), c["frames"], c["image_size"])
    e.set_pose_prior(*synthetic.synthetic_pose_prior())
    sp = synthetic.synthetic_shape_prior()
    e.set_shape_prior(*sp)

    def fitter_at(params):
        f = fit.FusedFitter(e, tg["tj"], tg["vis"], tg["tsil"].astype(np.float32), c["window"], True, params["betas"], params["log_beta_scales"])
        for k in ("global_rotation", "joint_rotations", "trans"):
            f.p[k].copy_(torch.as_tensor(np.asarray(params[k], np.float32)).cuda().reshape(f.p[k].shape))
        return f
    return ec, fx, e, fitter_at


@pytest.mark.parametrize("case", ["crop8", "config3"])
def test_losses_and_gradients_match_the_float64_oracle(case, capsys):
    ec, fx, e, fitter_at = _setup(case)
    assert set(fx["states"]) >= {"initial", "near_gt"}
    lines, bad = [], []
    recorded = json.load(open(MEASURED)) if os.path.exists(MEASURED) else {}
    measured = {}

    def ratchet(key, err, floor, line):
        measured[key] = max(measured.get(key, 0.0), err)
        if key in recorded and err > max(RATCHET * recorded[key], floor):
            bad.append(line + "   [ratchet: %.1f x the recorded %.2e]" % (err / max(recorded[key], 1e-300), recorded[key]))
    for name, st in fx["states"].items():
        weights, w_temp, _ = ec.stage_weights(st["stage"])
        for cache in ("cold", "warm"):            # first evaluation of a forgotten cache, then the same state on its cached bounds
            if cache == "cold":
                e.reset_raster_cache()
                f = fitter_at(st["params"])
            f.evaluate(weights, w_temp, st["stage"])
            hip = f.losses.cpu().numpy().astype(np.float64)[:8]
            ref = st["terms"]
            scale = abs(ref.sum())
            for i, t in enumerate(ec.TERMS):
                if ref[i] == 0.0 and hip[i] == 0.0:
                    continue
                # a term is held to 1e-4 of ITSELF, or of 1e-3 x the objective when it is a negligible part of it
                err = abs(hip[i] - ref[i]) / max(abs(ref[i]), 1e-3 * scale)
                y = abs(st["terms_f32"][i] - ref[i]) / max(abs(ref[i]), 1e-3 * scale) if "terms_f32" in st else float("nan")
                lines.append("%-8s %-10s %-4s %-11s hip %.8g  f64 %.8g  rel %.2e  (f32 oracle %.2e)" % (case, name, cache, t, hip[i], ref[i], err, y))
                if err > max(TERM_TOL, YARD * (y if y == y else 0.0)):
                    bad.append(lines[-1])
                ratchet("%s/%s/%s" % (case, name, t), err, RATCHET_FLOOR_TERM, lines[-1])
            tot = abs(hip.sum() - ref.sum()) / scale
            lines.append("%-8s %-10s %-4s %-11s hip %.8g  f64 %.8g  rel %.2e" % (case, name, cache, "TOTAL", hip.sum(), ref.sum(), tot))
            if tot > TERM_TOL:
                bad.append(lines[-1])
            for k, g in st["grads"].items():
                err = _rel(f.g[k].cpu().numpy(), g)
                y = _rel(st["grads_f32"][k], g) if "grads_f32" in st and k in st["grads_f32"] else float("nan")
                # the part of the deviation that lies ALONG the gradient (a common scale error, signed), HIP / float32 oracle
                along = _along(f.g[k].cpu().numpy(), g)
                along32 = _along(st["grads_f32"][k], g) if "grads_f32" in st and k in st["grads_f32"] else float("nan")
                lines.append("%-8s %-10s %-4s d/d%-16s rel-L2 %.2e  (f32 oracle %.2e)   along the gradient %+.1e  (f32 oracle %+.1e)" %
                             (case, name, cache, k, err, y, along, along32))
                if err > max(GRAD_TOL, YARD * (y if y == y else 0.0)):
                    bad.append(lines[-1])
                ratchet("%s/%s/d%s" % (case, name, k), err, RATCHET_FLOOR_GRAD, lines[-1])
        assert e.status() == 0
    with capsys.disabled():
        print("\n[eval fixtures: HIP vs float64 oracle]\n" + "\n".join(lines))
    if os.environ.get("SMALFIT_WRITE_EVAL_MEASURED"):
        path = os.environ["SMALFIT_WRITE_EVAL_MEASURED"]
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc.update(measured)
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    assert recorded, "tests/golden/hip_eval_measured.json is missing: run this test with SMALFIT_WRITE_EVAL_MEASURED=<path> on a GPU box and commit the file"
    assert not bad, "\n".join(bad)


def test_crop_states_run_the_large_face_paths(capsys):
    """the crop8 states the fixture holds really are the regime it is there for: thousands of faces whose pixel boxes exceed
    256 pixels (no byte candidate list: kernels_raster.inc `listed`, box-walking backward) and boxes far larger than the
    sweep's 32 x 32 LDS window.  Boxes are recomputed on the host from the vertices the evaluation hands out
    (FusedFitter.evaluate(verts_out=...): the general argument path)."""
    ec, fx, e, fitter_at = _setup("crop8")
    name = "hip_final" if "hip_final" in fx["states"] else "near_gt"
    st = fx["states"][name]
    weights, w_temp, _ = ec.stage_weights(st["stage"])
    f = fitter_at(st["params"])
    N, S = ec.CASES["crop8"]["frames"], ec.CASES["crop8"]["image_size"]
    from smalify_amd import synthetic
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    verts = torch.full((N, md.num_verts, 3), float("nan"), device="cuda")
    sil = torch.full((N, S, S), float("nan"), device="cuda")
    f.evaluate(weights, w_temp, st["stage"], verts_out=verts, sil_out=sil)
    assert bool(torch.isfinite(verts).all()) and bool(torch.isfinite(sil).all()), "evaluate() left verts_out / sil_out unwritten"
    faces = np.asarray(md.faces).astype(np.int64)
    big = wide = 0
    for i in range(N):
        px = ec.face_box_pixels(verts[i].cpu().numpy(), faces, S)
        big += int((px > 256).sum())
        wide += int((px > 1024).sum())
    rows = (sil > 0.5).any(2).sum(1).float().mean().item()
    with capsys.disabled():
        print("\n[crop8 %s] faces with boxes > 256 px: %d of %d, > 1024 px: %d; silhouette spans %.0f of %d rows"
              % (name, big, N * len(faces), wide, rows, S))
    assert big >= 0.02 * N * len(faces), big
    assert wide > 0
    assert rows > 0.5 * S


def test_config4_shards_add_up_to_the_oracle_at_full_size(capsys):
    """BASELINE config 4 at its own size, oracle-backed: the 64-frame 256 x 256 sequence split 8 frames per rank (WINDOW 8) -- here the
    eight shards of `config3`'s fixture evaluated one after the other on one GPU, each a FusedFitter that is told where its frames sit in
    the sequence (frame_offset / total_frames) and holds its neighbours' boundary frames as halos, exactly what a rank of the sharded loop
    evaluates (smalify_amd/distributed.py; the collective only moves these records).  The ranks' loss terms must add up to the float64
    oracle's terms of the WHOLE sequence, the per-frame gradients concatenated and the partial shape gradients summed over the ranks must be
    its gradients -- at the HIP fit's own state after stage 1, with stage 2's weights (silhouette and temporal terms on)."""
    ec, fx, _, _ = _setup("config3")
    from smalify_amd import distributed, engine as eng, fitter as fit, synthetic
    tg = ec.load_targets("config3")
    c = ec.CASES["config3"]
    N, world = c["frames"], 8
    st = fx["states"]["hip_stage1"]
    weights, w_temp, _ = ec.stage_weights(st["stage"])
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    e = eng.Engine(eng.DeviceModel(md), N // world, c["image_size"])
    e.set_pose_prior(*synthetic.synthetic_pose_prior())
    e.set_shape_prior(*synthetic.synthetic_shape_prior())
    P = st["params"]
    fitters = []
    for r in range(world):
        lo, hi = distributed.shard_range(N, r, world, c["window"])
        f = fit.FusedFitter(e, tg["tj"][lo:hi], tg["vis"][lo:hi], tg["tsil"][lo:hi].astype(np.float32), c["window"], True, P["betas"],
                            P["log_beta_scales"], frame_offset=lo, total_frames=N)
        for k in ("global_rotation", "joint_rotations", "trans"):
            f.p[k].copy_(torch.as_tensor(np.asarray(P[k][lo:hi], np.float32)).cuda().reshape(f.p[k].shape))
        fitters.append((lo, hi, f))
    recs = [f.boundary_records().clone() for _, _, f in fitters]           # (2, 108): first and last frame of every shard
    terms = np.zeros(8)
    grads = {k: np.zeros_like(np.asarray(st["grads"][k], np.float64)) for k in st["grads"]}
    for r, (lo, hi, f) in enumerate(fitters):
        f.halo_prev = recs[r - 1][1].contiguous() if r > 0 else None
        f.halo_next = recs[r + 1][0].contiguous() if r + 1 < world else None
        e.reset_raster_cache()
        f.evaluate(weights, w_temp, st["stage"])
        terms += f.losses.cpu().numpy().astype(np.float64)[:8]
        for k in grads:
            g = f.g[k].cpu().numpy().astype(np.float64)
            if k in ("betas", "log_beta_scales"):
                grads[k] += g.reshape(grads[k].shape)                  # partial shape gradients: summed over the ranks
            else:
                grads[k][lo:hi] = g.reshape(grads[k][lo:hi].shape)
        assert e.status() == 0
    ref = st["terms"]
    lines, bad = [], []
    for i, t in enumerate(ec.TERMS):
        err = abs(terms[i] - ref[i]) / max(abs(ref[i]), 1e-3 * abs(ref.sum()))
        lines.append("config4 (8 shards of 8)  %-11s sum over ranks %.8g  f64 oracle %.8g  rel %.2e" % (t, terms[i], ref[i], err))
        if err > TERM_TOL:
            bad.append(lines[-1])
    for k, g in st["grads"].items():
        err, y = _rel(grads[k], g), _rel(st["grads_f32"][k], g)
        lines.append("config4 (8 shards of 8)  d/d%-16s rel-L2 %.2e  (f32 oracle %.2e)" % (k, err, y))
        if err > max(GRAD_TOL, YARD * y):
            bad.append(lines[-1])
    with capsys.disabled():
        print("\n" + "\n".join(lines))
    assert not bad, "\n".join(bad)
