"""Oracle EVALUATION fixtures at BASELINE.json's own sizes: per-term losses and every gradient of one epoch objective
(optimize_to_joints.py:113-137 without the Adam step) in float64, at a handful of named states, for

  crop8    8 frames, 256 x 256, WINDOW 8, the CROP-FILLING scene (bench.build_problem(..., "crop")'s construction: the
           ground-truth animal 1.2 units nearer the camera, so that it fills the crop as the reference's loaders deliver it,
           utils.py:5-36 crop_to_silhouette, data_loader.py:48,117) -- the regime the large-face paths of the rasteriser run in
  config3  64 frames, 256 x 256, WINDOW 8, the headline scene (BASELINE config 3: exactly bench.py's ground-truth draw)

Targets are made by the ORACLE in float64 (projected canonical joints + 1 px noise, Bernoulli(0.85) visibility, hard
silhouette = soft silhouette > 0.5) and travel with the fixture (packed bits), so the GPU test fits and evaluates the
same bytes.  States:

  initial      the reference's initial state (smal_fitter.py:48-61,81-89), evaluated with stage 1's weights
  hip_stage1   the HIP fit's own state at the end of stage 1 (150 + 400 iterations from `initial`; dumped ON THE GPU by
               tools/dump_fit_states.py into tests/golden/hip_states_<case>.npz -- an INPUT of the fixture), stage 2's weights
  hip_final    (crop8 only) the HIP fit's state after the whole 1950-iteration schedule, stage 3's weights
  near_gt      the ground truth + a seeded perturbation of 0.01 (rotations) / 0.005 (shape, translation), stage 3's weights

Shared by tests/golden/make_oracle_eval.py (writes tests/golden/oracle_eval_<case>.npz), tests/test_oracle_golden.py (the
fixture belongs to today's problem) and tests/test_gpu_eval_fixtures.py (the HIP evaluation against it).
Everything here is CPU/oracle-side test infrastructure; the product never imports it.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np
import torch

from oracle import smal_oracle as so
from smalify_amd import config as cfg
from smalify_amd import model_io, synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TERMS = ("joint", "pose", "splay", "betas", "sil_reproj", "temp_joint", "temp_global", "temp_trans")
PARAMS = ("betas", "log_beta_scales", "global_rotation", "joint_rotations", "trans")
CASES = {
    "crop8": dict(frames=8, image_size=256, window=8, dz=1.2, states=("initial", "hip_stage1", "hip_final", "near_gt")),
    "config3": dict(frames=64, image_size=256, window=8, dz=0.0, states=("initial", "hip_stage1", "near_gt")),
    # targets only (bench.py's second scene, `value_crop`): 64 frames of the crop-filling scene, rendered by the oracle
    "crop64": dict(frames=64, image_size=256, window=8, dz=1.2, states=()),
}
STATE_STAGE = {"initial": 1, "hip_stage1": 2, "hip_final": 3, "near_gt": 3}      # whose weight column a state is evaluated with
HIP_STATE_AFTER = {"hip_stage1": 2, "hip_final": 4}                              # number of completed stages


def fixture_path(case):
    return os.path.join(GOLDEN_DIR, "oracle_eval_%s.npz" % case)


def hip_states_path(case):
    return os.path.join(GOLDEN_DIR, "hip_states_%s.npz" % case)


def targets_path(case):
    return os.path.join(GOLDEN_DIR, "eval_targets_%s.npz" % case)


def ground_truth(case):
    c = CASES[case]
    sp = synthetic.synthetic_shape_prior()
    gt = synthetic.ground_truth_params(c["frames"], seed=1234, mean_betas=sp[1][:20], mean_logscale=sp[1][20:26])
    gt["trans"][:, 2] += np.float32(c["dz"])
    return gt


def make_targets(case):
    """-> dict(tj (N,25,2) f32, vis (N,25) f32, tsil (N,S,S) u8): the oracle's float64 rendering of the ground truth"""
    c = CASES[case]
    N, S = c["frames"], c["image_size"]
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    gt = ground_truth(case)
    om = so.OracleModel(md)
    with torch.no_grad():
        theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
        vo, jo, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(gt["betas"], (N, 1))).double(), torch.from_numpy(theta).double(),
                                       torch.from_numpy(np.tile(gt["log_beta_scales"], (N, 1))).double())
        t = torch.from_numpy(gt["trans"]).double()[:, None]
        noise, vis = synthetic.keypoint_noise_and_visibility(N)
        tj = (so.project_points((jo + t)[:, so.CANONICAL], S).numpy() + noise).astype(np.float32)
        tsil = (so.soft_silhouette(vo + t, om.faces, S) > 0.5).numpy().astype(np.uint8)
    return dict(tj=tj, vis=vis.astype(np.float32), tsil=tsil)


def save_targets(case, tg):
    np.savez_compressed(targets_path(case), tj=tg["tj"], vis=tg["vis"], tsil_bits=np.packbits(tg["tsil"].reshape(-1)),
                        shape=np.array(tg["tsil"].shape))


def load_targets(case):
    p = targets_path(case)
    if not os.path.exists(p):
        return None
    z = np.load(p, allow_pickle=False)
    shape = tuple(int(x) for x in z["shape"])
    return dict(tj=z["tj"], vis=z["vis"], tsil=np.unpackbits(z["tsil_bits"])[:int(np.prod(shape))].reshape(shape))


def initial_params(case):
    """SMALFitter.__init__ (smal_fitter.py:48-61,81-89)"""
    sp = synthetic.synthetic_shape_prior()
    N = CASES[case]["frames"]
    return dict(betas=sp[1][:20].astype(np.float32).copy(), log_beta_scales=sp[1][20:26].astype(np.float32).copy(),
                global_rotation=np.tile(model_io.initial_global_rotation(), (N, 1)).astype(np.float32),
                joint_rotations=np.zeros((N, 34, 3), np.float32), trans=np.zeros((N, 3), np.float32))


def near_gt_params(case):
    gt = ground_truth(case)
    rs = np.random.RandomState(97)
    out = {}
    for k in PARAMS:
        amp = 0.01 if k in ("global_rotation", "joint_rotations") else 0.005
        out[k] = (gt[k] + amp * rs.randn(*gt[k].shape)).astype(np.float32)
    return out


def states(case):
    """-> {state name: parameter dict (float32 arrays)}; the hip_* states only when their dump exists"""
    out = {"initial": initial_params(case), "near_gt": near_gt_params(case)}
    p = hip_states_path(case)
    if os.path.exists(p):
        z = np.load(p, allow_pickle=False)
        for name in CASES[case]["states"]:
            if name.startswith("hip_") and (name + "_betas") in z.files:
                out[name] = {k: z["%s_%s" % (name, k)].astype(np.float32) for k in PARAMS}
    return {k: out[k] for k in CASES[case]["states"] if k in out}


def fingerprint(tg, st):
    h = hashlib.sha256()
    for k in sorted(tg):
        h.update(np.ascontiguousarray(tg[k]).tobytes())
    for name in sorted(st):
        for k in PARAMS:
            h.update(np.ascontiguousarray(st[name][k], np.float32).tobytes())
    return h.hexdigest()


def problem(case, tg, dtype=torch.float64):
    c = CASES[case]
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    pp = synthetic.synthetic_pose_prior()
    sp = synthetic.synthetic_shape_prior()
    om = so.OracleModel(md, dtype=dtype)
    return so.FitProblem(om, c["image_size"], tg["tj"], tg["vis"], tg["tsil"].astype(np.float32), pp[0], pp[1], pp[2], sp[0], sp[1],
                         c["window"], True, dtype=dtype)


def stage_weights(stage):
    w = np.array(cfg.OPT_WEIGHTS).T[stage]
    return w[:6].copy(), float(w[6]), float(w[8])


def oracle_eval(prob, params, stage, dtype=torch.float64):
    """-> (terms (8,), {param: gradient}) of one epoch objective at `params` with stage `stage`'s weights"""
    weights, w_temp, _ = stage_weights(stage)
    names = so.trainable_names(stage)
    p = {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in params.items()}
    _, sums, grads = so.loss_and_grads(prob, p, weights, w_temp, names)
    return np.array([float(sums.get(k, 0.0)) for k in TERMS]), {k: v.double().numpy() for k, v in grads.items()}


def load_fixture(case):
    p = fixture_path(case)
    if not os.path.exists(p):
        return None
    z = np.load(p, allow_pickle=False)
    out = {"fingerprint": str(z["fingerprint"]), "states": {}}
    for name in CASES[case]["states"]:
        if name + "_terms" not in z.files:
            continue
        out["states"][name] = dict(stage=int(z[name + "_stage"]), terms=z[name + "_terms"],
                                   params={k: z["%s_p_%s" % (name, k)] for k in PARAMS},
                                   grads={k: z["%s_g_%s" % (name, k)] for k in PARAMS if "%s_g_%s" % (name, k) in z.files})
        if name + "_terms_f32" in z.files:       # the oracle's own float32 evaluation of the same state: the yardstick
            out["states"][name]["terms_f32"] = z[name + "_terms_f32"]
            out["states"][name]["grads_f32"] = {k: z["%s_g32_%s" % (name, k)] for k in PARAMS if "%s_g32_%s" % (name, k) in z.files}
    return out


def face_box_pixels(verts, faces, S, blur=so.BLUR_RADIUS):
    """numpy: per face the number of pixel centres inside its sqrt(blur)-expanded bounding box (pytorch3d's
    CheckPointOutsideBoundingBox), for ONE frame's world-space vertices (V, 3) -- the box the rasteriser's face sweep walks"""
    xn, yn, _ = so.world_to_ndc(torch.from_numpy(np.asarray(verts, np.float64)))
    x, y = xn.numpy(), yn.numpy()
    fx, fy = x[faces], y[faces]
    r = np.sqrt(blur)
    xlo, xhi, ylo, yhi = fx.min(1) - r, fx.max(1) + r, fy.min(1) - r, fy.max(1) + r
    # pixel centre of column c: x_p = 1 - (2c + 1) / S, inside when xlo <= x_p <= xhi
    c_lo = np.ceil((1.0 - xhi) * S / 2.0 - 0.5)
    c_hi = np.floor((1.0 - xlo) * S / 2.0 - 0.5)
    r_lo = np.ceil((1.0 - yhi) * S / 2.0 - 0.5)
    r_hi = np.floor((1.0 - ylo) * S / 2.0 - 0.5)
    w = np.clip(np.minimum(c_hi, S - 1) - np.maximum(c_lo, 0) + 1, 0, None)
    h = np.clip(np.minimum(r_hi, S - 1) - np.maximum(r_lo, 0) + 1, 0, None)
    return (w * h).astype(np.int64)
