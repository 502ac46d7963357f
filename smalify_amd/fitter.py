"""Fused fitting loop: the reference's optimize_to_joints stage/epoch loop with every tensor resident
in HBM and every arithmetic step a HIP kernel (smalfit_fit_eval + smalfit_adam_step).

Reference semantics reproduced (file:line into /root/reference):
  * parameters and their initial values          smal_fitter/smal_fitter.py:48-97
  * 4-stage schedule from OPT_WEIGHTS            config.py:63-72, smal_fitter/optimize_to_joints.py:90-94
  * new Adam(lr, betas=(0.5, 0.999)) per stage   optimize_to_joints.py:96
  * stage 0 trains global_rotation + trans only, torso keypoints only   optimize_to_joints.py:98-104
  * per epoch: sum over windows + temporal, one backward, one step      optimize_to_joints.py:113-137
  * per-frame checkpoint dict                    smal_fitter.py:213-219, optimize_to_joints.py:46-48

Differences by design: all windows of an epoch are evaluated in one launch sequence (the per-window
normalisers are applied per frame, the sum over windows is identical); nothing synchronises with the
host inside the loop (the reference reads 4 device scalars per epoch for tqdm).
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from . import config
from . import engine as eng
from . import model_io
from . import smal_topology as topo

PARAM_NAMES = ("betas", "log_beta_scales", "joint_rotations", "global_rotation", "trans")


class FusedFitter:
    """Owns the fit parameters of N frames in one flat HBM buffer laid out as
    [betas(20) | log_beta_scales(6 or N*6) | joint_rotations(N*102) | global_rotation(N*3) | trans(N*3)]
    so that the stage-0 trainable set (global_rotation, trans) is one contiguous Adam segment."""

    def __init__(self, engine: eng.Engine, target_joints, target_visibility, target_sil, window_size,
                 use_unity_prior=True, mean_betas=None, mean_log_scales=None, allow_limb_scaling=True,
                 rank=0, world_size=1, group=None, sil_storage="auto", frame_offset=0, total_frames=None):
        """frame_offset / total_frames: these N frames are frames [frame_offset, frame_offset + N) of a sequence of
        total_frames (a shard of a sequence fitted by several ranks, smalify_amd/distributed.py): the reference's per-window
        normalisers (smal_fitter.py:144,157,173) follow the SEQUENCE's windows (optimize_to_joints.py:119-120), so a shard
        may start anywhere, down to one frame of a window per rank."""
        self.e = engine
        dev = engine.device
        self.N = int(target_joints.shape[0])
        self.S = engine.image_size
        self.window = int(window_size)
        self.frame_offset = int(frame_offset)
        self.total_frames = self.frame_offset + self.N if total_frames is None else int(total_frames)
        if self.frame_offset < 0 or self.total_frames < self.frame_offset + self.N:
            raise ValueError("frames [%d, %d) do not fit a sequence of %d" % (self.frame_offset, self.frame_offset + self.N, self.total_frames))
        self.unity = bool(use_unity_prior)
        self.allow_limb_scaling = allow_limb_scaling
        self.rank, self.world_size, self.group = rank, world_size, group
        f32 = dict(device=dev, dtype=torch.float32)

        def to_dev(x):
            if not isinstance(x, torch.Tensor):
                x = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
            return x.to(**f32).contiguous()

        self.target_joints = to_dev(target_joints)
        self.visibility_full = to_dev(target_visibility)
        self.target_sil = to_dev(target_sil).reshape(self.N, self.S, self.S).contiguous()
        # Device-resident targets as bytes when that loses nothing: the masks the reference's loaders read are 8-bit
        # images / 255 (data_loader.py:43), and a binary mask stays binary through the crop.  b / 255 in the kernels is the
        # correctly rounded quotient, i.e. exactly the float32 value such a mask holds -> bit-identical results, a quarter
        # of the target traffic.  Masks with other values (bilinear resampling of the crop) stay float32.
        if sil_storage not in ("auto", "f32", "u8"):
            raise ValueError("sil_storage must be 'auto', 'f32' or 'u8'")
        if sil_storage != "f32":
            q = torch.round(self.target_sil * 255.0)
            exact = bool(((q / 255.0) == self.target_sil).all()) and bool(((q >= 0) & (q <= 255)).all())
            if exact:
                self.target_sil_f32 = self.target_sil if sil_storage == "u8" else None
                self.target_sil = q.to(torch.uint8).contiguous()
            elif sil_storage == "u8":
                raise ValueError("target silhouettes are not multiples of 1/255: they cannot be stored as bytes exactly")
        vis0 = torch.zeros_like(self.visibility_full)
        vis0[:, config.TORSO_JOINTS] = self.visibility_full[:, config.TORSO_JOINTS]
        self.visibility_stage0 = vis0.contiguous()

        N = self.N
        self.ls_shared = self.unity            # smal_fitter.py:61 vs :71-72
        n_ls = 6 if self.ls_shared else N * 6
        self.offsets = {}
        off = 0
        for name, count in (("betas", 20), ("log_beta_scales", n_ls), ("joint_rotations", N * 102),
                            ("global_rotation", N * 3), ("trans", N * 3)):
            self.offsets[name] = (off, count)
            off += count
        self.flat = torch.zeros(off, **f32)
        self.grad = torch.zeros(off, **f32)
        self.exp_avg = torch.zeros(off, **f32)
        self.exp_avg_sq = torch.zeros(off, **f32)
        shapes = dict(betas=(20,), log_beta_scales=(6,) if self.ls_shared else (N, 6),
                      joint_rotations=(N, 34, 3), global_rotation=(N, 3), trans=(N, 3))
        self.p = {k: self.flat[o:o + c].view(shapes[k]) for k, (o, c) in self.offsets.items()}
        self.g = {k: self.grad[o:o + c].view(shapes[k]) for k, (o, c) in self.offsets.items()}
        if mean_betas is not None:
            self.p["betas"].copy_(torch.as_tensor(np.asarray(mean_betas)[:20], **f32))
        if mean_log_scales is not None and self.ls_shared:
            self.p["log_beta_scales"].copy_(torch.as_tensor(np.asarray(mean_log_scales), **f32))
        self.p["global_rotation"].copy_(torch.as_tensor(model_io.initial_global_rotation(), **f32)[None].expand(N, 3))
        self.global_mask = torch.ones(3, **f32)
        self.rotation_mask = torch.ones(34, 3, **f32)
        self.losses = torch.zeros(eng.NUM_LOSS_TERMS, **f32)
        self.step_count = 0
        self.halo_prev = self.halo_next = None
        self._plan = None
        self.use_joint_limits = False          # opt-in (enable_joint_limits): the reference's term is commented out
        # the two argument-block templates (stage 0 / later stages differ in the visibility tensor) are marshalled here, with the
        # rest of the construction, not in front of the first launch of the first stage
        for stage_id in (0, 1):
            self._fit_args((0, 0, 0, 0, 0, 0), 0.0, stage_id, PARAM_NAMES)

    def enable_joint_limits(self, min_values=None, max_values=None):
        """switch on the joint-limit hinge the reference has commented out (smal_fitter.py:76-79,146-151): from then on
        the w_limit column of the weight table (100 in stages 1-3, config.py:68) takes effect in THIS fitter's evaluations
        (other fitters on the same engine keep passing w_limit = 0, like the reference)"""
        if min_values is None:
            min_values, max_values = model_io.joint_limit_table()
        self._joint_limits = (np.array(min_values, np.float32), np.array(max_values, np.float32))
        self.use_joint_limits = True
        self.assert_joint_limits()
        self._plan = None                      # argument blocks built so far carry w_limit = 0

    def assert_joint_limits(self):
        """The limit table is state of the (possibly shared) engine: another fitter may have replaced or cleared it since this
        one opted in (SMALFitter._engine clears it for fitters without limits).  Called before every evaluation of a fitter
        that uses limits: puts this fitter's table back when the engine's is not this fitter's -- the w_limit term can never
        silently drop out of the objective."""
        if self.use_joint_limits and self.e.joint_limits_owner is not self:
            self.e.set_joint_limits(self._joint_limits[0], self._joint_limits[1], owner=self)

    # ---- stage control (optimize_to_joints.py:96-110) --------------------------------------------------
    def trainable(self, stage_id):
        if stage_id == 0:
            return ("global_rotation", "trans")
        names = ["betas", "joint_rotations", "global_rotation", "trans"]
        if self.allow_limb_scaling:
            names.insert(1, "log_beta_scales")
        return tuple(names)

    def begin_stage(self, stage_id):
        """fresh optimiser state, as `torch.optim.Adam(model.parameters(), ...)` per stage.  Nothing is filled: the
        first Adam step of a stage takes the moments as zero (smalfit_adam_args.step == 0)."""
        self.step_count = 0
        self.stage_id = stage_id
        # (the argument blocks of _stage_plan depend on weights / stage / trainable set / halo buffers only, not on the
        # optimiser's step count: they survive a stage change, see prepare_schedule)

    def _segments(self, names):
        """contiguous [start, end) runs of the flat buffer covering the trainable tensors"""
        segs = []
        for k in PARAM_NAMES:
            if k not in names:
                continue
            o, c = self.offsets[k]
            if segs and segs[-1][1] == o:
                segs[-1][1] = o + c
            else:
                segs.append([o, o + c])
        return segs

    # ---- one epoch (optimize_to_joints.py:113-137) ---------------------------------------------------------
    def _fit_args(self, weights, w_temp, stage_id, want, **outs):
        vis = self.visibility_stage0 if stage_id == 0 else self.visibility_full
        weights = [float(w) for w in weights]
        if not self.use_joint_limits:          # the engine's limit table may belong to another fitter: w_limit is per call
            weights[4] = 0.0
        if outs:                               # extra outputs (silhouette image, projections, vertices): the general path
            return self.e.build_fit_args(
                betas=self.p["betas"], log_beta_scales=self.p["log_beta_scales"],
                global_rotation=self.p["global_rotation"], joint_rotations=self.p["joint_rotations"],
                trans=self.p["trans"], target_joints=self.target_joints, target_visibility=vis,
                target_sil=self.target_sil, weights=weights, w_temp=w_temp, window=self.window,
                temporal=True, global_mask=self.global_mask, rotation_mask=self.rotation_mask,
                halo_prev=self.halo_prev, halo_next=self.halo_next,
                losses=self.losses, grads=self.g, want=want, frame_offset=self.frame_offset, total_frames=self.total_frames, **outs)
        # Every pointer of the block is known once the fitter exists: it is marshalled ONCE per set of tensors (~100 us of ctypes work:
        # 35 checked pointers) and a stage's block is a copy of that template with the handful of per-stage fields set -- a stage
        # change costs the host ~10 us instead of ~100 (it sits in front of the first launch of every stage).
        pk = self._pointer_key(stage_id)
        bases = self.__dict__.setdefault("_arg_templates", {})
        base = bases.get(pk)
        if base is None:
            if len(bases) > 8:
                bases.clear()
            a, _, _, keep = self.e.build_fit_args(
                betas=self.p["betas"], log_beta_scales=self.p["log_beta_scales"],
                global_rotation=self.p["global_rotation"], joint_rotations=self.p["joint_rotations"],
                trans=self.p["trans"], target_joints=self.target_joints, target_visibility=vis,
                target_sil=self.target_sil, weights=(0, 0, 0, 0, 0, 0), w_temp=0.0, window=self.window,
                temporal=True, global_mask=self.global_mask, rotation_mask=self.rotation_mask,
                halo_prev=self.halo_prev, halo_next=self.halo_next,
                losses=self.losses, grads=self.g, want=PARAM_NAMES, frame_offset=self.frame_offset, total_frames=self.total_frames)
            base = bases[pk] = (bytes(a), keep)
        a = eng.FitArgs.from_buffer_copy(base[0])
        a.w_j2d, a.w_sil, a.w_betas, a.w_pose, a.w_limit, a.w_splay = weights
        a.w_temp = float(w_temp)
        for k in PARAM_NAMES:
            if k not in want:
                setattr(a, "g_" + k, None)
        return a, self.losses, self.g, base[1]

    def evaluate(self, weights, w_temp, stage_id, want=None, **outs):
        want = self.trainable(stage_id) if want is None else want
        self.assert_joint_limits()
        a, _, _, _keep = self._fit_args(weights, w_temp, stage_id, want, **outs)
        eng.check(self.e.lib.smalfit_fit_eval(self.e.handle, eng._stream(), eng.C.byref(a)), "smalfit_fit_eval")
        return self.losses

    def _adam_args(self, names, lr):
        return eng.make_adam_args(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self._segments(names), lr,
                                  step=self.step_count)

    def apply_adam(self, names, lr, advance=True):
        """Adam on the trainable tensors `names` (one launch); advance=False: a second group of tensors within the
        same iteration"""
        a = self._adam_args(names, lr)
        if not advance:
            a.step -= 1
        eng.adam_segments(a)
        self.step_count = a.step

    def _pointer_key(self, stage_id):
        """addresses of every tensor an argument block holds a raw pointer to and that a caller may rebind (masks, targets,
        visibility, halos): part of the plan key, so a rebound tensor gets a fresh block instead of a stale pointer"""
        vis = self.visibility_stage0 if stage_id == 0 else self.visibility_full
        return tuple(None if t is None else t.data_ptr() for t in
                     (self.global_mask, self.rotation_mask, self.target_joints, vis, self.target_sil, self.halo_prev, self.halo_next))

    def _stage_plan(self, weights, w_temp, lr, stage_id, names):
        """argument blocks of the stage's iterations, rebuilt only when something they depend on changes.  The halo buffers
        (like the masks, targets and visibility tensors: _pointer_key) enter the key by ADDRESS (the block holds raw device pointers): ShardedFitter hands out views of one persistent
        gather buffer, so the key is stable across iterations; a caller that allocates fresh halo tensors every step gets a
        fresh block every step -- correct, but it pays the ~100 us of marshalling each time (reuse the buffers instead)."""
        key = (tuple(float(w) for w in weights), float(w_temp), float(lr), stage_id, tuple(names), self.use_joint_limits) + self._pointer_key(stage_id)
        plans = self._plan if isinstance(getattr(self, "_plan", None), dict) else {}
        if key not in plans:
            if len(plans) > 8:
                plans.clear()
            fa, _, _, keep = self._fit_args(weights, w_temp, stage_id, names)
            plans[key] = (fa, self._adam_args(names, lr), keep)
        self._plan = plans
        fa, aa, _ = plans[key]
        aa.step = self.step_count
        return fa, aa

    def prepare_schedule(self, opt_weights=None):
        """builds the argument blocks of every stage of the schedule up front (config.OPT_WEIGHTS by default), so that a stage
        change inside the loop is a dictionary lookup and one library call instead of ~100 us of ctypes marshalling while
        the GPU idles.  Optional: run_iterations builds what it does not find."""
        W = np.array(config.OPT_WEIGHTS if opt_weights is None else opt_weights).T
        for stage_id, w in enumerate(W):
            self._stage_plan(w[:6], float(w[6]), float(w[8]), stage_id, self.trainable(stage_id))

    def run_iterations(self, weights, w_temp, lr, stage_id, iterations):
        """`iterations` epochs of the reference loop in ONE library call (smalfit_fit_run): evaluation + analytic
        backward + Adam, all enqueued from C"""
        self.assert_joint_limits()
        fa, aa = self._stage_plan(weights, w_temp, lr, stage_id, self.trainable(stage_id))
        self.e.fit_run(fa, aa, iterations)
        self.step_count = aa.step
        return self.losses

    def step(self, weights, w_temp, lr, stage_id):
        return self.run_iterations(weights, w_temp, lr, stage_id, 1)

    # ---- frame-sharded protocol (smalify_amd/distributed.py) ----------------------------------------------------
    def num_shared(self):
        return 20 + (6 if self.ls_shared else 0)

    def local_step(self, weights, w_temp, lr, stage_id, record):
        """evaluation + Adam on the per-frame parameters + this rank's record (partial shared gradient | boundary
        frames after the step) written into `record` (num_shared() + 216 floats)"""
        self.assert_joint_limits()
        names = self.trainable(stage_id)
        local = tuple(k for k in names if k not in ("betas", "log_beta_scales") or (k == "log_beta_scales" and not self.ls_shared))
        key = (tuple(float(w) for w in weights), float(w_temp), float(lr), stage_id, tuple(names), "sharded", self.use_joint_limits) + self._pointer_key(stage_id)
        plans = self._plan if isinstance(getattr(self, "_plan", None), dict) else {}
        if key not in plans:
            if len(plans) > 8:
                plans.clear()
            fa, _, _, keep = self._fit_args(weights, w_temp, stage_id, names)
            plans[key] = (fa, self._adam_args(local, lr), keep)
        self._plan = plans
        fa, aa, _ = plans[key]
        aa.step = self.step_count
        eng.shard_local_step(self.e, fa, aa, self.num_shared(), self.grad, record)

    def shard_run(self, weights, w_temp, lr, stage_id, iterations, rank, world_size, record, gathered, allgather, allgather_ctx):
        """`iterations` whole sharded iterations in ONE library call (smalfit_shard_run): evaluation + per-frame Adam + record,
        the caller's all-gather (`allgather`: address of a smalfit_allgather_fn, `allgather_ctx`: its context) and the
        rank-ordered reduction + shared Adam, all enqueued from C.  halo_prev / halo_next must be views of `gathered`."""
        self.assert_joint_limits()
        names = self.trainable(stage_id)
        local = tuple(k for k in names if k not in ("betas", "log_beta_scales") or (k == "log_beta_scales" and not self.ls_shared))
        ntrain = 0
        if "betas" in names:
            ntrain = 20 + (6 if (self.ls_shared and "log_beta_scales" in names) else 0)
        key = (tuple(float(w) for w in weights), float(w_temp), float(lr), stage_id, tuple(names), "shard_run", self.use_joint_limits,
               rank, world_size, record.data_ptr(), gathered.data_ptr(), int(allgather), int(allgather_ctx or 0)) + self._pointer_key(stage_id)
        plans = self._plan if isinstance(getattr(self, "_plan", None), dict) else {}
        if key not in plans:
            if len(plans) > 8:
                plans.clear()
            fa, _, _, keep = self._fit_args(weights, w_temp, stage_id, names)
            sa = eng.ShardArgs()
            sa.world_size, sa.rank, sa.num_shared, sa.num_trainable_shared = int(world_size), int(rank), self.num_shared(), ntrain
            sa.shared_grad, sa.record, sa.gathered = eng._ptr(self.grad), eng._ptr(record), eng._ptr(gathered)
            sa.allgather, sa.allgather_ctx = int(allgather), allgather_ctx
            plans[key] = (fa, self._adam_args(local, lr), eng.make_adam_args(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, [], lr), sa,
                          keep + [record, gathered])
        self._plan = plans
        fa, al, ash, sa, _ = plans[key]
        al.step = ash.step = self.step_count
        eng.shard_run(self.e, fa, al, ash, sa, iterations)
        self.step_count = al.step
        return self.losses

    def shared_step(self, gathered, world_size, lr, stage_id):
        """sum of the ranks' partial shared gradients + Adam on the shared parameters the stage trains; closes the
        iteration (advances the step count)"""
        names = self.trainable(stage_id)
        ntrain = 0
        if "betas" in names:
            ntrain = 20 + (6 if (self.ls_shared and "log_beta_scales" in names) else 0)
        aa = eng.make_adam_args(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, [], lr, step=self.step_count)
        eng.shard_reduce_step(world_size, gathered.shape[-1], gathered, self.num_shared(), ntrain, aa)
        self.step_count += 1

    def shared_grad(self):
        """view of the gradient of the parameters every frame shares (betas, and the limb scales when shared)"""
        return self.grad[: 20 + (6 if self.ls_shared else 0)]

    def boundary_records(self, out=None):
        """(2,108) masked theta(105)|trans(3) of the first and the last local frame (temporal halo exchange): one
        gather from the flat parameter buffer and one multiply with the masks; `out` (216 floats) avoids allocations"""
        if getattr(self, "_brec_index", None) is None:
            idx, msk = [], []
            for n in (0, self.N - 1):
                o_g, o_j, o_t = self.offsets["global_rotation"][0], self.offsets["joint_rotations"][0], self.offsets["trans"][0]
                idx += [o_g + n * 3 + k for k in range(3)] + [o_j + n * 102 + k for k in range(102)] + [o_t + n * 3 + k for k in range(3)]
            self._brec_index = torch.tensor(idx, device=self.flat.device, dtype=torch.long)
        key = (id(self.global_mask), self.global_mask._version, id(self.rotation_mask), self.rotation_mask._version)
        if getattr(self, "_brec_mask_key", None) != key:
            m = torch.cat([self.global_mask.reshape(-1), self.rotation_mask.reshape(-1), torch.ones(3, device=self.flat.device)])
            self._brec_mask, self._brec_mask_key = torch.cat([m, m]), key
        rec = torch.index_select(self.flat, 0, self._brec_index)
        if out is None:
            return (rec * self._brec_mask).view(2, 108)
        torch.mul(rec, self._brec_mask, out=out)
        return out.view(2, 108)

    def target_sil_float(self):
        """the target silhouettes as float32 in [0, 1], whichever way they are stored on the device (bytes hold 255 t)"""
        if self.target_sil.dtype == torch.uint8:
            return self.target_sil.to(torch.float32) / 255.0
        return self.target_sil

    def snapshot(self):
        """-> (verts (N,V,3) translated, silhouettes (N,S,S), projected keypoints (N,25,2)) at the current parameters.
        Forward only, into scratch buffers: the fit's loss vector and gradients are left alone."""
        dev = self.flat.device
        V = self.e.model.num_verts
        verts = torch.empty(self.N, V, 3, device=dev)
        sil = torch.empty(self.N, self.S, self.S, device=dev)
        proj = torch.empty(self.N, 25, 2, device=dev)
        self.e.fit_eval(betas=self.p["betas"], log_beta_scales=self.p["log_beta_scales"],
                        global_rotation=self.p["global_rotation"], joint_rotations=self.p["joint_rotations"],
                        trans=self.p["trans"], target_joints=None, target_visibility=None, target_sil=None,
                        weights=(0, 0, 0, 0, 0, 0), w_temp=0.0, window=self.window, temporal=False,
                        global_mask=self.global_mask, rotation_mask=self.rotation_mask,
                        losses=torch.empty(eng.NUM_LOSS_TERMS, device=dev), grads={}, want=(), sil_out=sil, proj_out=proj, verts_out=verts)
        return verts, sil, proj

    def run_schedule(self, opt_weights=None, iters_scale=1.0, on_visualize=None, vis_frequency=None):
        """The reference's full stage loop. Returns per-stage final loss vectors (host)."""
        W = np.array(config.OPT_WEIGHTS if opt_weights is None else opt_weights).T
        vis_frequency = config.VIS_FREQUENCY if vis_frequency is None else vis_frequency
        history = []
        for stage_id, w in enumerate(W):
            weights, w_temp, epochs, lr = w[:6], float(w[6]), max(1, int(round(int(w[7]) * iters_scale))), float(w[8])
            self.begin_stage(stage_id)
            epoch_id = 0
            while epoch_id < epochs:
                # up to the next visualisation epoch (or the end of the stage) in one library call
                if on_visualize is None:
                    n = epochs - epoch_id
                else:
                    n = min(epochs - epoch_id, 1 if epoch_id % vis_frequency == 0 else vis_frequency - epoch_id % vis_frequency)
                self.run_iterations(weights, w_temp, lr, stage_id, n)
                if on_visualize is not None and epoch_id % vis_frequency == 0:
                    on_visualize(self, stage_id, epoch_id)
                epoch_id += n
            history.append(self.losses.cpu().numpy().copy())
        return history

    # ---- checkpoints (smal_fitter.py:192-219, optimize_to_joints.py:43-53) -------------------------------------
    def frame_parameters(self):
        """list of per-frame dicts with the reference's keys, dtypes and masking"""
        gr = (self.p["global_rotation"] * self.global_mask).cpu().numpy()
        jr = (self.p["joint_rotations"] * self.rotation_mask).cpu().numpy()
        tr = self.p["trans"].cpu().numpy()
        betas = self.p["betas"].cpu().numpy()
        ls = self.p["log_beta_scales"].cpu().numpy()
        out = []
        for i in range(self.N):
            out.append({"global_rotation": gr[i].astype(np.float32), "joint_rotations": jr[i].astype(np.float32),
                        "betas": betas.astype(np.float32),
                        "log_betascale": (ls if self.ls_shared else ls[i]).astype(np.float32),
                        "trans": tr[i].astype(np.float32)})
        return out

    def export_checkpoints(self, output_dirs, stage_id, epoch_name):
        """writes <dir>/st{stage}_ep{epoch}.pkl per frame, layout of optimize_to_joints.py:46-48"""
        for d, params in zip(output_dirs, self.frame_parameters()):
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "st{0}_ep{1}.pkl".format(stage_id, epoch_name)), "wb") as f:
                pickle.dump(params, f)

    def load_checkpoint(self, checkpoint_path, epoch):
        """reference SMALFitter.load_checkpoint (smal_fitter.py:192-207): per-frame params, betas and scales
        averaged over frames"""
        beta_list, scale_list = [], []
        for frame_id in range(self.N):
            with open(os.path.join(checkpoint_path, "{0:04}".format(frame_id), "{0}.pkl".format(epoch)), "rb") as f:
                d = pickle.load(f)
            dev = self.flat.device
            self.p["global_rotation"][frame_id] = torch.from_numpy(np.asarray(d["global_rotation"], np.float32)).to(dev)
            self.p["joint_rotations"][frame_id] = torch.from_numpy(np.asarray(d["joint_rotations"], np.float32)).to(dev).view(34, 3)
            self.p["trans"][frame_id] = torch.from_numpy(np.asarray(d["trans"], np.float32)).to(dev)
            beta_list.append(np.asarray(d["betas"])[:topo.NUM_BETAS])
            scale_list.append(np.asarray(d["log_betascale"]))
        self.p["betas"].copy_(torch.from_numpy(np.mean(beta_list, axis=0).astype(np.float32)))
        mean_scale = torch.from_numpy(np.mean(scale_list, axis=0).astype(np.float32)).to(self.flat.device)
        if self.ls_shared:
            self.p["log_beta_scales"].copy_(mean_scale)
        else:
            self.p["log_beta_scales"].copy_(mean_scale[None].expand(self.N, 6))
