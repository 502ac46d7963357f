"""Full-schedule parity problems (ScheduleCase).  CONFIG2 = BASELINE.json config 2's shape: 8 frames, 256 x 256, WINDOW_SIZE 8, shape family 1 with the
unity-style prior, the reference's FULL 150/400/600/800 schedule (reference config.py:63-72) from the reference's own
initial state (smal_fitter.py:48-61,81-89).  Shared by

  tests/golden/make_oracle_config2.py   writes tests/golden/oracle_config2_{f64,f32}.npz (hours of CPU, build container)
  tests/test_oracle_golden.py           the fixtures belong to today's problem and today's oracle (CPU)
  tests/test_gpu_config2.py             the HIP loop against them (GPU)
  bench.py::final_loss_parity           `final_loss_vs_ref` of the bench line

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

SCHEDULE = tuple(int(w[7]) for w in np.array(cfg.OPT_WEIGHTS).T)          # (150, 400, 600, 800)
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TERMS = ("joint", "pose", "splay", "betas", "sil_reproj", "temp_joint", "temp_global", "temp_trans")
PARAMS = ("betas", "log_beta_scales", "global_rotation", "joint_rotations", "trans")


class ScheduleCase:
    """One full-schedule parity problem.  Instances expose what this module exposed as globals when it held config 2 only
    (FRAMES, IMAGE_SIZE, WINDOW, SCHEDULE, TERMS, PARAMS, targets(), initial_params(), fingerprint(), problem(),
    oracle_schedule(), fixture_path(), load_fixture()), so tests and generators take a case where they took the module."""
    SCHEDULE, TERMS, PARAMS = SCHEDULE, TERMS, PARAMS

    def __init__(self, name, frames, image_size, window, dz=0.0):
        self.name, self.FRAMES, self.IMAGE_SIZE, self.WINDOW, self.dz = name, frames, image_size, window, float(dz)

    def fixture_path(self, tag):
        return os.path.join(GOLDEN_DIR, "oracle_%s_%s.npz" % (self.name, tag))

    def targets(self):
        """the bench's ground-truth draw (BASELINE.md section 4 / smalify_amd.synthetic.ground_truth_params) for FRAMES frames
        (translated by dz along the view axis: dz = 1.2 is bench.py's crop-filling scene), targets made by the ORACLE in
        float64: projected canonical joints + 1 px noise, Bernoulli(0.85) visibility, hard silhouette = soft silhouette > 0.5"""
        N, S = self.FRAMES, self.IMAGE_SIZE
        md = synthetic.synthetic_model(seed=0, shape_family_id=1)
        sp = synthetic.synthetic_shape_prior()
        gt = synthetic.ground_truth_params(N, seed=1234, mean_betas=sp[1][:20], mean_logscale=sp[1][20:26])
        if self.dz:
            gt["trans"][:, 2] += np.float32(self.dz)
        om = so.OracleModel(md)
        with torch.no_grad():
            theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
            vo, jo, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(gt["betas"], (N, 1))).double(), torch.from_numpy(theta).double(),
                                           torch.from_numpy(np.tile(gt["log_beta_scales"], (N, 1))).double())
            t = torch.from_numpy(gt["trans"]).double()[:, None]
            noise, vis = synthetic.keypoint_noise_and_visibility(N)
            tj = (so.project_points((jo + t)[:, so.CANONICAL], S).numpy() + noise).astype(np.float32)
            tsil = (so.soft_silhouette(vo + t, om.faces, S) > 0.5).numpy().astype(np.uint8)
        return md, dict(tj=tj, vis=vis.astype(np.float32), tsil=tsil)

    def initial_params(self):
        """SMALFitter.__init__ (smal_fitter.py:48-61,81-89): betas / limb scales at the prior mean, the reference's initial
        global rotation, everything else zero"""
        sp = synthetic.synthetic_shape_prior()
        N = self.FRAMES
        return dict(betas=sp[1][:20].astype(np.float32).copy(), log_beta_scales=sp[1][20:26].astype(np.float32).copy(),
                    global_rotation=np.tile(model_io.initial_global_rotation(), (N, 1)).astype(np.float32),
                    joint_rotations=np.zeros((N, 34, 3), np.float32), trans=np.zeros((N, 3), np.float32))

    @staticmethod
    def fingerprint(tg, start):
        h = hashlib.sha256()
        for k in sorted(tg):
            h.update(np.ascontiguousarray(tg[k]).tobytes())
        for k in sorted(start):
            h.update(np.ascontiguousarray(start[k]).tobytes())
        return h.hexdigest()

    def problem(self, md, tg, dtype):
        pp = synthetic.synthetic_pose_prior()
        sp = synthetic.synthetic_shape_prior()
        om = so.OracleModel(md, dtype=dtype)
        return so.FitProblem(om, self.IMAGE_SIZE, tg["tj"], tg["vis"], tg["tsil"].astype(np.float32), pp[0], pp[1], pp[2], sp[0], sp[1],
                             self.WINDOW, True, dtype=dtype)

    @staticmethod
    def oracle_schedule(prob, start, dtype, schedule=SCHEDULE, checkpoint=None, state=None):
        """the oracle's stage loop (optimize_to_joints.py:90-137) in `dtype`.  Returns per-iteration per-term losses
        (sum(schedule), 8), the parameters at the START of every stage and at the end of the run.  `state` (a dict this function
        made earlier, handed to `checkpoint(state)` after every iteration) continues an interrupted run."""
        W = np.array(cfg.OPT_WEIGHTS).T
        if state is None:
            state = dict(stage=0, it=0, params={k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in start.items()},
                         opt=None, trace=[], stage_start={})
        while state["stage"] < 4:
            stage = state["stage"]
            w = W[stage]
            weights, w_temp, lr = w[:6].copy(), float(w[6]), float(w[8])
            names = so.trainable_names(stage)
            vis = so.stage0_visibility(prob.vis) if stage == 0 else None
            if state["it"] == 0:                                   # optimize_to_joints.py:96: a new Adam per stage
                state["opt"] = so.Adam(so.PARAM_ORDER, lr=lr)
                state["stage_start"][stage] = {k: v.double().numpy().copy() for k, v in state["params"].items()}
            while state["it"] < schedule[stage]:
                total, sums, grads = so.loss_and_grads(prob, state["params"], weights, w_temp, names, visibility=vis)
                state["trace"].append([float(sums.get(k, 0.0)) for k in TERMS])
                state["opt"].step(state["params"], grads)
                state["it"] += 1
                if checkpoint is not None:
                    checkpoint(state)
            state["stage"], state["it"] = stage + 1, 0
        return np.array(state["trace"]), state["stage_start"], {k: v.double().numpy() for k, v in state["params"].items()}

    def load_fixture(self, tag):
        p = self.fixture_path(tag)
        if not os.path.exists(p):
            return None
        z = np.load(p, allow_pickle=False)
        out = {"trace": z["trace"], "schedule": tuple(int(x) for x in z["schedule"]), "fingerprint": str(z["fingerprint"]),
               "dtype": str(z["dtype"]), "complete": bool(z["complete"]),
               "final": {k: z["final_" + k] for k in PARAMS if "final_" + k in z.files},
               "stage_start": {s: {k: z["stage%d_%s" % (s, k)] for k in PARAMS} for s in range(4) if "stage%d_betas" % s in z.files}}
        if "tj" in z.files:
            out["targets"] = dict(tj=z["tj"], vis=z["vis"], tsil=np.unpackbits(z["tsil_bits"])[:self.FRAMES * self.IMAGE_SIZE ** 2]
                                  .reshape(self.FRAMES, self.IMAGE_SIZE, self.IMAGE_SIZE))
        return out


# BASELINE config 2's shape: 8 frames, 256 x 256, WINDOW_SIZE 8, the headline scene's draw
CONFIG2 = ScheduleCase("config2", 8, 256, 8)
# BASELINE config 1 as worded -- ONE image, shape family 1, ALL stages -- at 256 x 256 on the crop-filling scene (a StanfordExtra
# image is cropped to the animal: data_loader.py:117), WINDOW_SIZE 10 as in the reference's config.py:25
CONFIG1 = ScheduleCase("config1", 1, 256, 10, dz=1.2)
CASES = {"config2": CONFIG2, "config1": CONFIG1}

# the module itself still reads as config 2 (tests / bench.py written against it)
FRAMES, IMAGE_SIZE, WINDOW = CONFIG2.FRAMES, CONFIG2.IMAGE_SIZE, CONFIG2.WINDOW
fixture_path, targets, initial_params, fingerprint = CONFIG2.fixture_path, CONFIG2.targets, CONFIG2.initial_params, CONFIG2.fingerprint
problem, oracle_schedule, load_fixture = CONFIG2.problem, CONFIG2.oracle_schedule, CONFIG2.load_fixture
