"""Drop-in for reference fitter_3d/trainer.py: SMAL3DFitter, SMALParamGroup, Stage, StageManager.

What stays identical: constructor arguments, parameter names and shapes, the five optimisation schemes
(trainer.py:113-119), per-parameter learning rates, one fresh Adam per Stage (trainer.py:194, default betas), the loss
weights and their defaults (trainer.py:31), the .npz written by Stage.save_npz (trainer.py:264-279).

What is different underneath: one Stage.step is ONE C-ABI call on the current HIP stream (smalfit_fit3d_step: SMAL
forward, target-point sampling, all four loss terms, the gradient back through the SMAL model, Adam on the scheme's
parameters -- 14 kernel launches) with no autograd graph and no host synchronisation; the loss history stays on
the device until it is plotted or printed.  Stage.evaluate / Stage.step_unfused compose the same iteration from the
component entry points (smalfit_lbs_forward, smalfit_mesh_targets_sample, smalfit_mesh_objective_eval,
smalfit_lbs_backward, smalfit_adam_step) and expose the gradients.  The target points are drawn by a counter-based
generator keyed by (seed, global iteration), not by torch's global generator, so runs are reproducible.

Two reference quirks kept on purpose: log_beta_scales sits in the parameter groups but has requires_grad=False, so no
scheme ever changes it (trainer.py:64-65); deform_verts is trainable by the "deform" scheme as it is on the reference's
CPU path (on its CUDA path `nn.Parameter(...).to(device)` yields a non-leaf tensor the optimiser rejects, trainer.py:91).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, config, engine as eng, model_io, runtime
from ..smal_model.smal_torch import SMAL

default_weights = dict(w_chamfer=1.0, w_edge=1.0, w_normal=0.01, w_laplacian=0.1)     # trainer.py:31
_WEIGHT_ORDER = ("w_chamfer", "w_edge", "w_normal", "w_laplacian")
N_SAMPLE_POINTS = 3000                                                                 # trainer.py:209
_PARAM_ORDER = ("betas", "log_beta_scales", "global_rot", "joint_rot", "trans", "deform_verts")


class SMAL3DFitter(nn.Module):
    def __init__(self, batch_size=1, device="cuda", shape_family=-1, model_data=None, smal_data=None):
        """model_data / smal_data let tests inject synthetic stand-ins for the SMAL pickles; by default both are read
        from the paths in smalify_amd.config like the reference (trainer.py:47-58)."""
        super().__init__()
        if not torch.cuda.is_available():
            raise eng.SmalfitError("no HIP device available: smalify_amd has no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.batch_size = int(batch_size)
        self.n_betas = config.N_BETAS
        self.shape_family_list = np.array(shape_family)
        if smal_data is None:
            smal_data = model_io.load_pickle(config.SMAL_DATA_FILE)
        prec, mean = model_io.family_shape_prior(smal_data, shape_family, config.N_BETAS)
        self.betas_prec = torch.from_numpy(np.ascontiguousarray(prec, np.float32)).to(dev)
        self.mean_betas = torch.from_numpy(np.ascontiguousarray(mean, np.float32)).to(dev)
        N = self.batch_size
        self.betas = nn.Parameter(self.mean_betas.unsqueeze(0).repeat(N, 1))
        self.log_beta_scales = nn.Parameter(torch.zeros(N, 6, device=dev), requires_grad=False)
        self.global_rot = nn.Parameter(torch.zeros(N, 3, device=dev))          # eul_to_axis([0, 0, 0]) = 0
        self.trans = nn.Parameter(torch.zeros(N, 3, device=dev))
        self.joint_rot = nn.Parameter(torch.zeros(N, config.N_POSE, 3, device=dev))
        self.global_mask = torch.ones(1, 3, device=dev)                        # unused by forward, as in the reference
        self.rotation_mask = torch.ones(config.N_POSE, 3, device=dev)
        self.smal_model = SMAL(dev, shape_family_id=shape_family, model_data=model_data)
        self.faces = self.smal_model.faces.unsqueeze(0).repeat(N, 1, 1)
        self.deform_verts = nn.Parameter(torch.zeros(N, *self.smal_model.v_template.shape, device=dev))
        self._theta = torch.zeros(N, 35, 3, device=dev)

    # ---- C-ABI plumbing ---------------------------------------------------------------------------------------
    def _engine(self):
        return runtime.get_engine(self.smal_model.device_model, self.batch_size, self.smal_model.engine_image_size)

    def _pack_theta(self):
        self._theta[:, 0].copy_(self.global_rot.detach())
        self._theta[:, 1:].copy_(self.joint_rot.detach())
        return self._theta

    def lbs_verts(self):
        """SMAL(betas, [global_rot | joint_rot], betas_logscale) -> (N,V,3), trainer.py:95-100"""
        e = self._engine()
        return e.lbs_forward(self.betas.detach().contiguous(), self._pack_theta(),
                             self.log_beta_scales.detach().contiguous(), want_Rs=False, want_v_shaped=False)[0]

    def forward(self):
        """verts + trans + deform_verts (trainer.py:94-108), composed on the device by the objective's first kernel;
        detached -- gradients come from Stage.step"""
        o = _objective_for(self, N_SAMPLE_POINTS).eval(self.lbs_verts(), self.trans.detach().contiguous(),
                                                        self.deform_verts.detach().contiguous(), None, (0.0, 0.0, 0.0, 0.0))
        return o["verts"]


class SMALParamGroup:
    """Same parameter map and per-parameter learning rates as trainer.py:111-153."""
    param_map = {
        "init": ["global_rot", "trans"],
        "default": ["global_rot", "joint_rot", "trans", "betas", "log_beta_scales"],
        "shape": ["global_rot", "trans", "betas", "log_beta_scales"],
        "pose": ["global_rot", "trans", "joint_rot"],
        "deform": ["deform_verts"],
    }

    def __init__(self, model, group="smbld", lrs=None):
        self.model = model
        self.group = group
        assert group in self.param_map, f"Group {group} not in list of available params: {list(self.param_map.keys())}"
        self.lrs = dict(lrs) if lrs is not None else {}

    def names(self):
        return list(self.param_map[self.group])

    def __iter__(self):
        out = []
        for param_name in self.param_map[self.group]:
            d = {"params": [getattr(self.model, param_name)], "name": param_name}
            if param_name in self.lrs:
                d["lr"] = self.lrs[param_name]
            out.append(d)
        return iter(out)


class Stage:
    """One stage of optimisation (trainer.py:157-262)."""

    def __init__(self, nits: int, scheme: str, smal_3d_fitter: SMAL3DFitter, target_meshes, mesh_names=[],
                 name="optimise", loss_weights=None, lr=1e-3, out_dir="static_fits_output", custom_lrs=None,
                 device="cuda", seed=0, iteration_offset=0):
        self.n_it = int(nits)
        self.name = name
        self.out_dir = out_dir
        self.target_meshes = target_meshes
        self.mesh_names = mesh_names
        self.smal_3d_fitter = smal_3d_fitter
        self.device = smal_3d_fitter.device
        self.loss_weights = default_weights.copy()
        if loss_weights is not None:
            for k, v in loss_weights.items():
                if k not in self.loss_weights:
                    raise KeyError("unknown loss weight %r (have %s)" % (k, sorted(self.loss_weights)))
                self.loss_weights[k] = float(v)
        if custom_lrs is not None:
            for attr in custom_lrs:
                assert hasattr(smal_3d_fitter, attr), f"attr '{attr}' not in SMAL."
        self.param_group = SMALParamGroup(smal_3d_fitter, scheme, custom_lrs)
        self.scheduler = None
        self.lr = float(lr)
        if len(target_meshes) != smal_3d_fitter.batch_size:
            raise ValueError("%d target meshes for a fitter of batch size %d" % (len(target_meshes), smal_3d_fitter.batch_size))
        # fresh Adam state per stage (trainer.py:194); frozen parameters (requires_grad=False) get none
        self._adam = {}
        for g in self.param_group:
            p = g["params"][0]
            if p.requires_grad:
                self._adam[g["name"]] = dict(lr=float(g.get("lr", self.lr)), m=torch.zeros_like(p), v=torch.zeros_like(p))
        fit = smal_3d_fitter
        V = int(fit.smal_model.v_template.shape[0])
        self.n_verts = V
        self.faces = fit.faces.detach()
        self.src_verts = fit().detach()
        self._objective = _objective_for(fit, N_SAMPLE_POINTS)
        self._buffers = {}
        self._points = None
        self._last_points = None
        self.seed = int(seed)
        self.iteration_offset = int(iteration_offset)      # global iteration of this stage's first step (StageManager)
        self._loss_history = torch.zeros(max(self.n_it, 1), device=self.device)
        self._done = 0
        self._t = 0              # Adam step count of this stage
        self._args = None        # smalfit_fit3d_args, built on the first step
        self.consider_loss = lambda loss_name: self.loss_weights[f"w_{loss_name}"] > 0

    # ---- evaluation ----------------------------------------------------------------------------------------------
    @property
    def losses_to_plot(self):
        """total loss per completed iteration (host floats; reading it synchronises)"""
        return self._loss_history[:self._done].cpu().tolist()

    @property
    def last_points(self):
        """the target points of the most recent evaluation or step, (N, 3000, 3)"""
        return self._last_points if self._last_points is not None else self._points

    @property
    def last_terms(self):
        """device tensor (5,): chamfer, edge, normal, laplacian (unweighted) and the weighted total"""
        return self._buffers.get("losses")

    def _weights(self):
        return [self.loss_weights[k] for k in _WEIGHT_ORDER]

    def evaluate(self, iteration, points=None):
        """loss terms and gradients at the current parameters -> (total (device scalar), grads by parameter name)"""
        fit = self.smal_3d_fitter
        e = fit._engine()
        betas = fit.betas.detach().contiguous()
        ls = fit.log_beta_scales.detach().contiguous()
        theta = fit._pack_theta()
        lbs = e.lbs_forward(betas, theta, ls, want_Rs=False, want_v_shaped=False)[0]
        if points is None and self.loss_weights["w_chamfer"] > 0:
            points = self.target_meshes.sample(N_SAMPLE_POINTS, self.seed, self.iteration_offset + iteration, out=self._sample_buffer())
        self._last_points = points            # the sampler's buffer is never rebound: smalfit_fit3d_step writes through its raw pointer
        o = self._objective.eval(lbs, fit.trans.detach().contiguous(), fit.deform_verts.detach().contiguous(), points,
                                 self._weights(), out=self._buffers)
        self._buffers = o
        dbeta, dtheta, dls = e.lbs_backward(betas, theta, ls, o["dverts"], None)
        grads = {"betas": dbeta, "global_rot": dtheta[:, 0].contiguous(), "joint_rot": dtheta[:, 1:].contiguous(),
                 "log_beta_scales": dls, "trans": o["dtrans"], "deform_verts": o["dverts"]}
        return o["losses"][4], grads

    def forward(self, src_mesh=None):
        """total loss at the current parameters (the reference takes the offset source mesh; the engine composes it)"""
        return self.evaluate(self._done)[0]

    def step(self, epoch):
        """one iteration (trainer.py:229-241): loss, gradients and Adam on the parameters of the scheme in ONE C-ABI
        call (smalfit_fit3d_step).  Returns the total loss before the update (device scalar)."""
        fit = self.smal_3d_fitter
        e = fit._engine()
        a = self._step_args()
        for name in _PARAM_ORDER:
            setattr(a, name, getattr(fit, name).data_ptr())
        a.weights = (C.c_float * 4)(*self._weights())
        a.iteration = (self.iteration_offset + int(epoch)) & 0xFFFFFFFF
        self._t += 1
        a.adam_t = self._t
        dev_targets = getattr(self.target_meshes, "_dev", self.target_meshes)
        eng.check(e.lib.smalfit_fit3d_step(e.handle, self._objective.handle, dev_targets.handle, eng._stream(), C.byref(a)),
                  "smalfit_fit3d_step")
        self._last_points = self._points
        return self._buffers["losses"][4]

    def step_unfused(self, epoch):
        """the same iteration as five separate C-ABI calls + one smalfit_adam_step per parameter; leaves the gradients
        in `.grad` of the trained parameters (development / inspection path, ~3x the host time of step())"""
        loss, grads = self.evaluate(epoch)
        fit = self.smal_3d_fitter
        self._t += 1
        for name, st in self._adam.items():
            p = getattr(fit, name)
            g = grads[name]
            p.grad = g
            eng.adam_step(p.data, g, st["m"], st["v"], st["lr"], self._t, beta1=0.9, beta2=0.999, eps=1e-8)
        return loss

    def _sample_buffer(self):
        """the one (N, 3000, 3) tensor the sampler writes into, allocated once and kept for the life of the stage (the cached
        argument block of step() holds its device pointer)"""
        if self._points is None:
            self._points = torch.empty(self.smal_3d_fitter.batch_size, N_SAMPLE_POINTS, 3, device=self.device)
        return self._points

    def _step_args(self):
        if self._args is None:
            fit = self.smal_3d_fitter
            N = fit.batch_size
            a = _lib.Fit3dArgs()
            a.num_meshes, a.num_betas, a.num_points = N, int(fit.betas.shape[1]), N_SAMPLE_POINTS
            for name in _PARAM_ORDER:
                if name == "log_beta_scales":          # read by the forward, never trained (trainer.py:64-65)
                    continue
                st = self._adam.get(name)
                setattr(a, "lr_" + name, st["lr"] if st else 0.0)
                setattr(a, "m_" + name, st["m"].data_ptr() if st else None)
                setattr(a, "v_" + name, st["v"].data_ptr() if st else None)
            a.beta1, a.beta2, a.eps = 0.9, 0.999, 1e-8
            a.points = None
            a.seed = self.seed & (2 ** 64 - 1)
            if "losses" not in self._buffers:
                self._buffers["losses"] = torch.zeros(5, device=self.device)
            a.points_out = self._sample_buffer().data_ptr()
            a.losses = self._buffers["losses"].data_ptr()
            a.verts_out = None
            self._args = a
        return self._args

    def run(self, plot=False, progress=True, report_every=50):
        """Run the entire Stage (trainer.py:257-270).  The description line is refreshed every `report_every`
        iterations: each refresh reads the loss back and so synchronises with the device."""
        it = range(self.n_it)
        bar = None
        if progress:
            try:
                from tqdm import tqdm
                bar = tqdm(it)
                it = bar
            except ImportError:
                bar = None
        for i in it:
            loss = self.step(i)
            self._loss_history[i].copy_(loss)
            self._done = i + 1
            if bar is not None and (i % report_every == 0 or i == self.n_it - 1):
                bar.set_description(f"STAGE = {self.name}, TOT_LOSS = {float(loss):.6f}")
        if plot:
            self.plot()

    def plot(self):
        from .utils import plot_meshes
        verts = self.smal_3d_fitter()
        figtitle = f"{self.name}, its = {self.n_it}"
        plot_meshes(self.target_meshes, verts.cpu().numpy(), self.faces[0].cpu().numpy(), self.mesh_names, title=self.name,
                    figtitle=figtitle, out_dir=os.path.join(self.out_dir, "meshes"))

    def save_npz(self, labels=None):
        """same keys as trainer.py:264-279"""
        out = {}
        for param in ["global_rot", "joint_rot", "betas", "log_beta_scales", "trans", "deform_verts"]:
            out[param] = getattr(self.smal_3d_fitter, param).cpu().detach().numpy()
        out["verts"] = self.smal_3d_fitter().cpu().detach().numpy()
        out["faces"] = self.faces.cpu().detach().numpy()
        out["labels"] = labels
        os.makedirs(self.out_dir, exist_ok=True)
        np.savez(os.path.join(self.out_dir, f"{self.name}.npz"), **out)


def _objective_for(fitter, max_points):
    """one smalfit_mesh_objective per fitter (topology tables + work buffers), shared by its stages"""
    obj = getattr(fitter, "_mesh_objective", None)
    if obj is None or obj.max_points < max_points or obj.max_meshes < fitter.batch_size:
        obj = eng.MeshObjective(int(fitter.smal_model.v_template.shape[0]), fitter.smal_model.f, fitter.batch_size, max_points)
        fitter._mesh_objective = obj
    return obj


class StageManager:
    """Container for multiple stages of optimisation (trainer.py:282-323)."""

    def __init__(self, out_dir="static_fits_output", labels=None):
        self.stages = []
        self.out_dir = out_dir
        self.labels = labels

    def run(self, plot=True, progress=True):
        for stage in self.stages:
            stage.run(plot=plot, progress=progress)
            stage.save_npz(labels=self.labels)
        self.plot_losses()

    def plot_losses(self, out_src="losses"):
        """semilog plot of the total loss over all stages (trainer.py:299-319)"""
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot as plt
        fig, ax = plt.subplots()
        it_start = 0
        for stage in self.stages:
            hist = stage.losses_to_plot
            ax.semilogy(np.arange(it_start, it_start + len(hist)), hist, label=stage.name)
            it_start += stage.n_it
        ax.set_xlabel("Epoch")
        ax.set_ylabel("Total loss")
        ax.legend()
        os.makedirs(self.out_dir, exist_ok=True)
        plt.tight_layout()
        fig.savefig(os.path.join(self.out_dir, out_src + ".png"))
        plt.close(fig)

    def add_stage(self, stage):
        # every stage continues the sampler's iteration counter where the previous one stopped
        stage.iteration_offset = sum(s.n_it for s in self.stages)
        self.stages.append(stage)
