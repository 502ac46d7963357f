#!/usr/bin/env python3
"""BASELINE.md section 4, second CPU leg (BUILD CONTAINER ONLY -- the reference never travels to the GPU box): times the
IMPORTED REFERENCE (/root/reference, its own SMALFitter.forward + get_temporal + autograd backward + torch.optim.Adam,
reference smal_fitter/smal_fitter.py:107-190, smal_fitter/optimize_to_joints.py:113-137) for the NON-RENDER part of an
iteration next to the oracle port (oracle/smal_oracle.py) on the same inputs, to anchor the port's speed -- the CPU
baseline bench.py reports on the GPU box -- to the real code.

PyTorch3D is not installable here, so the reference's Renderer is replaced by the stand-in of tests/golden/make_golden.py
(zero silhouette, closed-form keypoint projection) and the silhouette weight is 0 in both legs: what is timed is LBS +
keypoint projection + priors + temporal term + backward + Adam over the whole 64-frame batch (windows of 8), the
reference's stage-0 type (global rotation / translation only, torso keypoints) and stage-1 type (all parameters) iterations.

usage: python tools/time_reference_cpu.py [iterations (default 20)] [threads (default all cores)]
writes profiles/r3_reference_cpu_anchor.json
"""
import json
import os
import pickle
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
REF = "/root/reference"
N, S, WINDOW = 64, 256, 8


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    import warnings
    warnings.filterwarnings("ignore")
    from tests.golden import make_golden as mg
    from smalify_amd import synthetic
    from oracle import smal_oracle as so
    mg.install_stubs()
    tmp = tempfile.mkdtemp(prefix="smal_anchor_")
    dd, data, sym = synthetic.synthetic_smal_dicts(seed=0)
    paths = {}
    for name, obj in (("smal", dd), ("data", data), ("sym", sym)):
        paths[name] = os.path.join(tmp, name + ".pkl")
        with open(paths[name], "wb") as f:
            pickle.dump(obj, f, protocol=2)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "smal_fitter"))
    import config as rconfig
    rconfig.SMAL_FILE, rconfig.SMAL_DATA_FILE, rconfig.SMAL_SYM_FILE = paths["smal"], paths["data"], paths["sym"]
    rconfig.WALKING_PRIOR_FILE = os.path.join(REF, rconfig.WALKING_PRIOR_FILE)
    rconfig.UNITY_SHAPE_PRIOR = os.path.join(REF, rconfig.UNITY_SHAPE_PRIOR)
    import smal_fitter as rfit                                    # the reference's module

    # the benchmark's inputs: ground-truth draw -> projected keypoints + noise, visibility
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    om64 = so.OracleModel(md)
    sp_syn = synthetic.synthetic_shape_prior()
    gt = synthetic.ground_truth_params(N, seed=1234, mean_betas=sp_syn[1][:20], mean_logscale=sp_syn[1][20:26])
    with torch.no_grad():
        theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
        _, jo, _, _ = so.smal_forward(om64, torch.from_numpy(np.tile(gt["betas"], (N, 1))).double(), torch.from_numpy(theta).double(),
                                      torch.from_numpy(np.tile(gt["log_beta_scales"], (N, 1))).double())
        noise, vis = synthetic.keypoint_noise_and_visibility(N)
        tj = (so.project_points((jo + torch.from_numpy(gt["trans"]).double()[:, None])[:, so.CANONICAL], S).numpy() + noise).astype(np.float32)
    rgb, sil = torch.zeros(N, 3, S, S), torch.zeros(N, 1, S, S)
    W = np.array(rconfig.OPT_WEIGHTS).T

    def weights_of(stage):
        w = W[stage][:6].copy()
        w[1] = 0.0                                                # silhouette off: the renderer is a stand-in
        return w

    # ---- leg 1: the reference's own loop body -----------------------------------------------------------------------
    def reference_leg(stage):
        f = rfit.SMALFitter("cpu", (rgb.clone(), sil.clone(), torch.from_numpy(tj), torch.from_numpy(vis.astype(np.float32))), WINDOW, 1, True)
        f.renderer = mg.StandInRenderer(S)
        opt = torch.optim.Adam(f.parameters(), lr=float(W[stage][8]), betas=(0.5, 0.999))
        if stage == 0:                                            # optimize_to_joints.py:98-104
            f.joint_rotations.requires_grad = False
            f.betas.requires_grad = False
            f.log_beta_scales.requires_grad = False
            tv = f.target_visibility.clone()
            f.target_visibility *= 0
            f.target_visibility[:, rconfig.TORSO_JOINTS] = tv[:, rconfig.TORSO_JOINTS]
        w, w_temp = weights_of(stage), float(W[stage][6])

        def epoch():
            acc = 0
            opt.zero_grad()
            for j in range(0, N, WINDOW):
                loss, _ = f(list(range(j, min(N, j + WINDOW))), w, stage)
                acc = acc + loss.mean()
            jl, gl, tl = f.get_temporal(w_temp)
            acc = acc + jl + gl + tl
            acc.backward()
            opt.step()
            return float(acc)

        epoch()
        epoch()                                                   # untimed: allocator, thread pool
        t0 = time.perf_counter()
        for _ in range(iters):
            last = epoch()
        return (time.perf_counter() - t0) / iters, last

    # ---- leg 2: the oracle port (what bench.py times on the GPU box), float32, same inputs ---------------------------------
    from smalify_amd import model_io
    pose_prior = model_io.load_pose_prior(rconfig.WALKING_PRIOR_FILE)
    shape_prior = model_io.unity_shape_prior(rconfig.UNITY_SHAPE_PRIOR)

    def port_leg(stage):
        om = so.OracleModel(md, dtype=torch.float32)
        prob = so.FitProblem(om, S, tj, vis, np.zeros((N, S, S), np.float32), pose_prior[0], pose_prior[1], pose_prior[2],
                             shape_prior[0], shape_prior[1], WINDOW, True, dtype=torch.float32)
        params = dict(betas=torch.from_numpy(shape_prior[1][:20].copy()), log_beta_scales=torch.from_numpy(shape_prior[1][20:26].copy()),
                      global_rotation=torch.from_numpy(np.tile(model_io.initial_global_rotation(), (N, 1))).float(),
                      trans=torch.zeros(N, 3), joint_rotations=torch.zeros(N, 34, 3))
        names = so.trainable_names(stage)
        vis0 = so.stage0_visibility(prob.vis) if stage == 0 else None
        opt = so.Adam(so.PARAM_ORDER, lr=float(W[stage][8]))
        w, w_temp = weights_of(stage), float(W[stage][6])

        def epoch():
            total, _, grads = so.loss_and_grads(prob, params, w, w_temp, names, visibility=vis0)
            opt.step(params, grads)
            return float(total)

        epoch()
        epoch()
        t0 = time.perf_counter()
        for _ in range(iters):
            last = epoch()
        return (time.perf_counter() - t0) / iters, last

    out = {"what": "non-render part of one iteration (64 frames, WINDOW_SIZE 8, keypoints + priors + temporal + backward + Adam, "
                   "silhouette weight 0) on the build container's CPU: imported reference vs the oracle port",
           "frames": N, "window": WINDOW, "timed_iterations": iters, "threads": threads, "cpu_count": os.cpu_count(),
           "torch": torch.__version__}
    for stage, label in ((0, "stage0_type"), (1, "stage1_type_no_silhouette")):
        t_ref, l_ref = reference_leg(stage)
        t_port, l_port = port_leg(stage)
        out[label] = {"reference_s_per_iteration": t_ref, "port_s_per_iteration": t_port, "reference_over_port": t_ref / t_port,
                      "reference_iterations_per_s": 1.0 / t_ref, "port_iterations_per_s": 1.0 / t_port,
                      "loss_after_timed_iterations": {"reference": l_ref, "port": l_port}}
        print(label, out[label], flush=True)
    dst = os.path.join(ROOT, "profiles", "r3_reference_cpu_anchor.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
