#!/usr/bin/env python3
"""Generate tests/golden/reference_golden_fit3d.npz by running the REFERENCE'S OWN fitter_3d code (imported from
/root/reference) as far as it runs without PyTorch3D.  Build container only:

    python tests/golden/make_golden_fit3d.py

What the reference's fitter_3d/trainer.py can do here with pytorch3d replaced by import stubs:
  * SMAL3DFitter.__init__ / forward   -> initial parameters, vertices = SMAL(...) + trans + deform_verts     (pinned)
  * SMALParamGroup + torch.optim.Adam -> which parameters a scheme trains, per-parameter learning rates, default betas,
                                         log_beta_scales frozen by requires_grad=False                        (pinned)
What it cannot do: Stage.forward (the four PyTorch3D losses and the sampler) -- that part of the path stays
"parity unpinned" (oracle/mesh3d_oracle.py).  The optimiser semantics are therefore exercised with a stand-in loss that
needs nothing from PyTorch3D: mean squared distance of the vertices to a fixed target.

Data only is stored (inputs + the reference's outputs), never reference source.  The SMAL model is the synthetic stand-in
written to a temp dir in the pickle layout the reference loads, exactly as tests/golden/make_golden.py does.
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (import stubs shared with the main fixture generator)
from smalify_amd import synthetic  # noqa: E402


def main():
    import warnings
    warnings.filterwarnings("ignore")
    mg.install_stubs()
    dummy = lambda *a, **k: None  # noqa: E731
    p3d = sys.modules["pytorch3d"]
    p3d.ops = mg._stub("pytorch3d.ops", sample_points_from_meshes=dummy)
    p3d.loss = mg._stub("pytorch3d.loss", chamfer_distance=dummy, mesh_edge_loss=dummy, mesh_laplacian_smoothing=dummy,
                        mesh_normal_consistency=dummy)
    sys.modules["pytorch3d.io"].load_obj = dummy
    import matplotlib
    matplotlib.use("Agg")

    tmp = tempfile.mkdtemp(prefix="smal_golden_fit3d_")
    dd, data, sym = synthetic.synthetic_smal_dicts(seed=0)
    paths = {}
    for name, obj in (("smal.pkl", dd), ("smal_data.pkl", data), ("sym.pkl", sym)):
        paths[name] = os.path.join(tmp, name)
        with open(paths[name], "wb") as f:
            pickle.dump(obj, f, protocol=2)
    sys.path.insert(0, REF)          # smal_fitter / smal_model / fitter_3d resolve as (namespace) packages from the root
    import config as rconfig
    rconfig.SMAL_FILE, rconfig.SMAL_DATA_FILE, rconfig.SMAL_SYM_FILE = paths["smal.pkl"], paths["smal_data.pkl"], paths["sym.pkl"]
    from fitter_3d.trainer import SMAL3DFitter, SMALParamGroup, default_weights

    torch.manual_seed(0)
    rs = np.random.RandomState(31)
    out = {}
    N = 2
    fit = SMAL3DFitter(batch_size=N, device="cpu", shape_family=-1)
    out["default_weights"] = np.array([default_weights[k] for k in ("w_chamfer", "w_edge", "w_normal", "w_laplacian")])
    for k in ("betas", "log_beta_scales", "global_rot", "joint_rot", "trans", "deform_verts"):
        out["init_" + k] = getattr(fit, k).detach().numpy().copy()
    out["init_verts"] = fit().detach().numpy().copy()
    out["requires_grad"] = np.array([int(getattr(fit, k).requires_grad) for k in
                                     ("betas", "log_beta_scales", "global_rot", "joint_rot", "trans", "deform_verts")])

    # forward at perturbed parameters (incl. vertex offsets and limb scales)
    with torch.no_grad():
        fit.betas += torch.from_numpy(0.3 * rs.randn(N, 20).astype(np.float32))
        fit.log_beta_scales += torch.from_numpy(0.1 * rs.randn(N, 6).astype(np.float32))
        fit.global_rot += torch.from_numpy(0.2 * rs.randn(N, 3).astype(np.float32))
        fit.joint_rot += torch.from_numpy(0.15 * rs.randn(N, 34, 3).astype(np.float32))
        fit.trans += torch.from_numpy(0.1 * rs.randn(N, 3).astype(np.float32))
        fit.deform_verts += torch.from_numpy(0.01 * rs.randn(*fit.deform_verts.shape).astype(np.float32))
    for k in ("betas", "log_beta_scales", "global_rot", "joint_rot", "trans", "deform_verts"):
        out["p_" + k] = getattr(fit, k).detach().numpy().copy()
    out["p_verts"] = fit().detach().numpy().copy()

    # optimiser semantics of a Stage (trainer.py:192-194): SMALParamGroup + Adam(lr), stand-in loss
    target = out["p_verts"] * 1.05 + 0.02
    tt = torch.from_numpy(target)
    for scheme, lr, custom in (("default", 0.01, {"joint_rot": 0.005}), ("init", 0.05, None), ("deform", 2e-4, None)):
        f2 = SMAL3DFitter(batch_size=N, device="cpu", shape_family=-1)
        with torch.no_grad():
            for k in ("betas", "log_beta_scales", "global_rot", "joint_rot", "trans", "deform_verts"):
                getattr(f2, k).copy_(torch.from_numpy(out["p_" + k]))
        opt = torch.optim.Adam(SMALParamGroup(f2, scheme, custom), lr=lr)
        hist = []
        for _ in range(6):
            opt.zero_grad()
            loss = ((f2() - tt) ** 2).sum(-1).mean()
            loss.backward()
            opt.step()
            hist.append(loss.item())
        out["adam_%s_loss" % scheme] = np.array(hist)
        for k in ("betas", "log_beta_scales", "global_rot", "joint_rot", "trans", "deform_verts"):
            out["adam_%s_%s" % (scheme, k)] = getattr(f2, k).detach().numpy().copy()
    out["adam_target"] = target.astype(np.float32)
    out["param_map_default"] = np.array(SMALParamGroup.param_map["default"])
    out["param_map_init"] = np.array(SMALParamGroup.param_map["init"])
    out["param_map_shape"] = np.array(SMALParamGroup.param_map["shape"])
    out["param_map_pose"] = np.array(SMALParamGroup.param_map["pose"])
    out["param_map_deform"] = np.array(SMALParamGroup.param_map["deform"])

    # keep the fixture small: vertices on a stride, everything float32
    vsel = np.arange(0, out["p_verts"].shape[1], 7)
    out["vsel"] = vsel
    for k in ("init_verts", "p_verts"):
        out[k] = out[k][:, vsel].astype(np.float32)
    path = os.path.join(HERE, "reference_golden_fit3d.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")
    print("init betas[0,:4]", out["init_betas"][0, :4], "requires_grad", out["requires_grad"])
    for s in ("default", "init", "deform"):
        print(s, out["adam_%s_loss" % s])


if __name__ == "__main__":
    main()
