#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE'S OWN CODE (imported from /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

What is produced is data (inputs + the reference's outputs), never reference source.  The reference
modules are imported unmodified; third-party packages that are not installable here are replaced
by import stubs:
  cv2, chumpy, torchvision.utils, nibabel.eulerangles, pytorch3d.*, imageio, trimesh, pycocotools
pytorch3d is a stub, so the reference's Renderer cannot produce silhouettes: fixtures that go
through SMALFitter.forward use w_sil = 0 and a stand-in renderer whose keypoint projection is the
closed form of SURVEY Appendix A.2 (documented as "parity unpinned" in oracle/smal_oracle.py).

The SMAL model is the synthetic stand-in (smalify_amd/synthetic.py) written to a temp dir in the
pickle layout the reference loads; the pose / shape priors are the reference's real data files.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)

from smalify_amd import synthetic, smal_topology as topo  # noqa: E402


# ------------------------------------------------------------------------------------------------
# import stubs
# ------------------------------------------------------------------------------------------------
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Ch:
    def __setstate__(self, s):
        self.__dict__.update(s)

    @property
    def r(self):
        return self.x


def _euler2angle_axis(z=0, y=0, x=0):
    from scipy.spatial.transform import Rotation
    m = (Rotation.from_euler("x", x) * Rotation.from_euler("y", y) * Rotation.from_euler("z", z))   # nibabel: Rx Ry Rz
    rv = m.as_rotvec()
    ang = np.linalg.norm(rv)
    if ang == 0.0:                      # nibabel's quat2angle_axis returns the x axis for the identity
        return 0.0, np.array([1.0, 0.0, 0.0])
    return ang, rv / ang


def install_stubs():
    _stub("cv2", MARKER_TRIANGLE_DOWN=5, MARKER_STAR=2, MARKER_CROSS=0)
    ch = _stub("chumpy", Ch=_Ch)
    chch = _stub("chumpy.ch", Ch=_Ch)
    ch.ch = chch
    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", make_grid=lambda *a, **k: None)
    nb = _stub("nibabel")
    nb.eulerangles = _stub("nibabel.eulerangles", euler2angle_axis=_euler2angle_axis)
    dummy = lambda *a, **k: None  # noqa: E731
    names = ("OpenGLPerspectiveCameras look_at_view_transform look_at_rotation RasterizationSettings "
             "MeshRenderer MeshRasterizer BlendParams PointLights HardPhongShader SoftSilhouetteShader "
             "Materials Textures").split()

    class _Blend:
        def __init__(self, sigma=1e-4, gamma=1e-4):
            self.sigma, self.gamma = sigma, gamma

    p3d = _stub("pytorch3d")
    p3d.structures = _stub("pytorch3d.structures", Meshes=dummy)
    rend = {n: dummy for n in names}
    rend["look_at_view_transform"] = lambda *a, **k: (None, None)
    rend["BlendParams"] = _Blend
    p3d.renderer = _stub("pytorch3d.renderer", **rend)
    p3d.io = _stub("pytorch3d.io", load_objs_as_meshes=dummy)
    for n in ("imageio", "trimesh", "pycocotools", "pycocotools.mask"):
        _stub(n)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mp = _stub("matplotlib")
        mp.pyplot = _stub("matplotlib.pyplot")


class StandInRenderer(torch.nn.Module):
    """Replaces reference Renderer (pytorch3d unavailable): zero silhouette + closed-form projection."""

    def __init__(self, image_size):
        super().__init__()
        self.image_size = image_size

    def forward(self, vertices, points, faces, render_texture=False):
        s = 1.0 / np.tan(np.radians(60.0) / 2.0)
        zv = 2.7 - points[..., 2]
        xn = s * (-points[..., 0]) / zv
        yn = s * points[..., 1] / zv
        half = (self.image_size - 1.0) / 2.0
        proj = torch.stack([half * (1.0 - yn), half * (1.0 - xn)], -1)
        sil = torch.zeros(vertices.shape[0], 1, self.image_size, self.image_size)
        return sil, proj


def main():
    import warnings
    warnings.filterwarnings("ignore")
    install_stubs()
    tmp = tempfile.mkdtemp(prefix="smal_golden_")
    dd, data, sym = synthetic.synthetic_smal_dicts(seed=0)
    smal_file = os.path.join(tmp, "smal.pkl")
    data_file = os.path.join(tmp, "smal_data.pkl")
    sym_file = os.path.join(tmp, "sym.pkl")
    for path, obj in ((smal_file, dd), (data_file, data), (sym_file, sym)):
        with open(path, "wb") as f:
            pickle.dump(obj, f, protocol=2)

    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "smal_fitter"))
    import pdb
    pdb.set_trace = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("reference hit pdb.set_trace"))
    import config as rconfig
    rconfig.SMAL_FILE, rconfig.SMAL_DATA_FILE, rconfig.SMAL_SYM_FILE = smal_file, data_file, sym_file
    rconfig.WALKING_PRIOR_FILE = os.path.join(REF, rconfig.WALKING_PRIOR_FILE)
    rconfig.UNITY_SHAPE_PRIOR = os.path.join(REF, rconfig.UNITY_SHAPE_PRIOR)

    from smal_model.smal_torch import SMAL
    from smal_model import batch_lbs as rlbs
    import smal_fitter as rfit
    from priors.pose_prior_35 import Prior

    torch.manual_seed(0)
    rs = np.random.RandomState(2024)
    out = {}
    dev = "cpu"

    # ---- model checksums (detect drift of the synthetic generator) ------------------------------
    smal = SMAL(dev, shape_family_id=1)
    out["chk_v_template"] = np.array([smal.v_template.double().sum().item(), smal.v_template.double().abs().sum().item()])
    out["chk_posedirs"] = np.array([smal.posedirs.double().sum().item(), smal.posedirs.double().abs().sum().item()])
    out["chk_shapedirs"] = np.array([smal.shapedirs.double().sum().item(), smal.shapedirs.double().abs().sum().item()])
    out["chk_weights"] = np.array([(smal.weights.double() * torch.arange(35).double()).sum().item()])
    out["chk_jreg"] = np.array([(smal.J_regressor.double() * torch.arange(35).double()).sum().item()])
    out["parents"] = np.asarray(smal.parents).astype(np.int32)

    # ---- G1 rodrigues ---------------------------------------------------------------------------
    th = rs.randn(12, 3).astype(np.float32)
    th[0] = 0.0
    th[1] = [1e-6, -2e-6, 3e-6]
    th[2] = [np.pi * 0.999, 0, 0]
    th[3] = [-1.20919958] * 3
    th[4] *= 3.0
    out["g1_theta"] = th
    out["g1_R"] = rlbs.batch_rodrigues(torch.from_numpy(th)).numpy()

    # ---- G2 global rigid transformation ----------------------------------------------------------
    n = 3
    th2 = (0.4 * rs.randn(n * 35, 3)).astype(np.float32)
    Rs = rlbs.batch_rodrigues(torch.from_numpy(th2)).reshape(n, 35, 3, 3)
    Js = rs.randn(n, 35, 3).astype(np.float32) * 0.3
    ls = (0.3 * rs.randn(n, 6)).astype(np.float32)
    parents = smal.parents
    nj0, a0 = rlbs.batch_global_rigid_transformation(Rs, torch.from_numpy(Js), parents)
    nj1, a1 = rlbs.batch_global_rigid_transformation(Rs, torch.from_numpy(Js), parents,
                                                     betas_logscale=torch.from_numpy(ls))
    out.update(g2_theta=th2, g2_Js=Js, g2_ls=ls, g2_newJ_noscale=nj0.numpy(), g2_A_noscale=a0.numpy(),
               g2_newJ_scale=nj1.numpy(), g2_A_scale=a1.numpy())

    # ---- G3 SMAL.__call__ values + grads -----------------------------------------------------------
    beta = torch.tensor(0.5 * rs.randn(n, 20).astype(np.float32), requires_grad=True)
    theta = torch.tensor(0.3 * rs.randn(n, 35, 3).astype(np.float32), requires_grad=True)
    lsc = torch.tensor(0.2 * rs.randn(n, 6).astype(np.float32), requires_grad=True)
    verts, joints, Rs3, v_shaped = smal(beta, theta, betas_logscale=lsc)
    vsel = np.arange(0, topo.NUM_VERTS, 37)
    wv = rs.randn(n, len(vsel), 3).astype(np.float32)
    wj = rs.randn(n, 41, 3).astype(np.float32)
    func = (verts[:, vsel] * torch.from_numpy(wv)).sum() + (joints * torch.from_numpy(wj)).sum()
    func.backward()
    out.update(g3_beta=beta.detach().numpy(), g3_theta=theta.detach().numpy(), g3_ls=lsc.detach().numpy(),
               g3_vsel=vsel, g3_verts=verts.detach().numpy()[:, vsel], g3_joints=joints.detach().numpy(),
               g3_vshaped=v_shaped.detach().numpy()[:, vsel], g3_Rs=Rs3.detach().numpy(),
               g3_wv=wv, g3_wj=wj, g3_func=np.array(func.item()),
               g3_dbeta=beta.grad.numpy(), g3_dtheta=theta.grad.numpy(), g3_dls=lsc.grad.numpy())
    # no-scale / theta = 0 case (stage-1 start: every joint at exactly zero rotation)
    theta0 = torch.zeros(1, 35, 3, requires_grad=True)
    beta0 = torch.zeros(1, 20)
    v0, j0, _, _ = smal(beta0, theta0)
    (j0 * torch.from_numpy(wj[:1])).sum().backward()
    out.update(g3z_joints=j0.detach().numpy(), g3z_dtheta=theta0.grad.numpy(), g3z_verts=v0.detach().numpy()[:, vsel])

    # ---- G4 pose prior ----------------------------------------------------------------------------
    prior = Prior(rconfig.WALKING_PRIOR_FILE, dev)
    x = torch.tensor(0.2 * rs.randn(4, 35, 3).astype(np.float32), requires_grad=True)
    val = prior(x)
    val.mean().backward()
    out.update(g4_x=x.detach().numpy(), g4_val=val.detach().numpy(), g4_dx=x.grad.numpy(),
               pose_prec=prior.precs.numpy(), pose_mean=prior.mean.numpy(), pose_mask=prior.use_ind_tch.numpy(),
               g4_zero_mean=np.array(prior(torch.zeros(1, 35, 3)).mean().item()),
               g4_point1_mean=np.array(prior(torch.full((1, 35, 3), 0.1)).mean().item()))

    # ---- G5..G7 SMALFitter --------------------------------------------------------------------------
    N, S = 4, 64
    rgb = torch.zeros(N, 3, S, S)
    sil = torch.zeros(N, 1, S, S)
    tj = torch.from_numpy((rs.rand(N, 25, 2) * S).astype(np.float32))
    vis = torch.from_numpy((rs.rand(N, 25) < 0.8).astype(np.float32))
    vis[:, [2, 5]] = 1.0
    vis[0, 8] = 0.0

    def make_fitter(window, family=1, unity=True):
        fit = rfit.SMALFitter(dev, (rgb.clone(), sil.clone(), tj.clone(), vis.clone()), window, family, unity)
        fit.renderer = StandInRenderer(S)
        return fit

    fit = make_fitter(N)
    out.update(unity_prec=fit.betas_prec.numpy(), unity_mean=fit.mean_betas.numpy(),
               init_global_rotation=fit.global_rotation.detach().numpy()[0],
               g5_init_betas=fit.betas.detach().numpy(), g5_init_ls=fit.log_beta_scales.detach().numpy())
    zero_b = ((torch.zeros(1, 26) - fit.mean_betas[None]) @ fit.betas_prec) ** 2
    out["g5_zero_betas_mean"] = np.array(zero_b.mean().item())

    def set_params(f, seed):
        r2 = np.random.RandomState(seed)
        with torch.no_grad():
            f.global_rotation += torch.from_numpy(0.1 * r2.randn(N, 3).astype(np.float32))
            f.joint_rotations += torch.from_numpy(0.15 * r2.randn(N, 34, 3).astype(np.float32))
            f.trans += torch.from_numpy((np.array([0.05, -0.03, 0.1]) + 0.02 * r2.randn(N, 3)).astype(np.float32))
            f.betas += torch.from_numpy(0.3 * r2.randn(20).astype(np.float32))
            f.log_beta_scales += torch.from_numpy(0.1 * r2.randn(*f.log_beta_scales.shape).astype(np.float32))

    def snapshot(f, prefix):
        for k in ("global_rotation", "joint_rotations", "trans", "betas", "log_beta_scales"):
            out[prefix + k] = getattr(f, k).detach().numpy().copy()

    def run_case(tag, window, weights, w_temp, stage_id, family=1, unity=True):
        f = make_fitter(window, family, unity)
        set_params(f, 77)
        snapshot(f, tag + "_p_")
        if stage_id == 0:                       # driver logic, optimize_to_joints.py:98-110
            f.joint_rotations.requires_grad = False
            f.betas.requires_grad = False
            f.log_beta_scales.requires_grad = False
            tv = f.target_visibility.clone()
            f.target_visibility *= 0
            f.target_visibility[:, rconfig.TORSO_JOINTS] = tv[:, rconfig.TORSO_JOINTS]
        else:
            f.joint_rotations.requires_grad = True
            f.betas.requires_grad = True
            if rconfig.ALLOW_LIMB_SCALING:
                f.log_beta_scales.requires_grad = True
            f.target_visibility = vis.clone()
        acc = 0
        termsum = {}
        for j in range(0, N, window):
            br = list(range(j, min(N, j + window)))
            loss, objs = f(br, weights, stage_id)
            acc = acc + loss.mean()
            for k, v in objs.items():
                termsum[k] = termsum.get(k, 0.0) + v.item()
        jl, gl, tl = f.get_temporal(w_temp)
        total = acc + jl + gl + tl
        total.backward()
        out[tag + "_total"] = np.array(total.item())
        for k, v in termsum.items():
            out[tag + "_term_" + k] = np.array(v)
        out[tag + "_temporal"] = np.array([jl.item(), gl.item(), tl.item()])
        for k in ("global_rotation", "joint_rotations", "trans", "betas", "log_beta_scales"):
            g = getattr(f, k).grad
            out[tag + "_g_" + k] = (torch.zeros_like(getattr(f, k)) if g is None else g).numpy().copy()

    W = np.array(rconfig.OPT_WEIGHTS).T
    w0 = W[0][:6].copy()
    w1 = W[1][:6].copy()
    w1[1] = 0.0                                   # silhouette off (renderer unavailable)
    run_case("g6_stage0_w4", 4, w0, W[0][6], 0)
    run_case("g6_stage1_w4", 4, w1, W[1][6], 1)
    run_case("g6_stage1_w2", 2, w1, W[1][6], 1)   # two windows: betas prior counted twice
    run_case("g6_stage1_w3", 3, w1, W[1][6], 1)   # ragged last window (3 + 1)
    run_case("g6_family0_w4", 4, w1, W[1][6], 1, family=0, unity=False)   # SMAL cluster prior, per-frame scales
    out.update(g6_target_joints=tj.numpy(), g6_visibility=vis.numpy(), g6_image_size=np.array(S),
               g6_w0=w0, g6_w1=w1, g6_wtemp=np.array([W[0][6], W[1][6]]))
    f0 = make_fitter(4, 0, False)
    out.update(fam0_prec=f0.betas_prec.numpy(), fam0_mean=f0.mean_betas.numpy())

    # ---- G8 short Adam trajectory with the reference's loop semantics ---------------------------------
    f = make_fitter(2)
    hist = []
    for stage_id, its in ((0, 6), (1, 14)):
        weights = (w0, w1)[stage_id]
        w_temp = W[stage_id][6]
        lr = W[stage_id][8]
        opt = torch.optim.Adam(f.parameters(), lr=lr, betas=(0.5, 0.999))
        if stage_id == 0:
            f.joint_rotations.requires_grad = False
            f.betas.requires_grad = False
            f.log_beta_scales.requires_grad = False
            tv = f.target_visibility.clone()
            f.target_visibility *= 0
            f.target_visibility[:, rconfig.TORSO_JOINTS] = tv[:, rconfig.TORSO_JOINTS]
        else:
            f.joint_rotations.requires_grad = True
            f.betas.requires_grad = True
            f.log_beta_scales.requires_grad = True
            f.target_visibility = vis.clone()
        for _ in range(its):
            acc = 0
            opt.zero_grad()
            for j in range(0, N, 2):
                loss, _ = f(list(range(j, min(N, j + 2))), weights, stage_id)
                acc = acc + loss.mean()
            jl, gl, tl = f.get_temporal(w_temp)
            acc = acc + jl + gl + tl
            acc.backward()
            opt.step()
            hist.append(acc.item())
        snapshot(f, "g8_after_stage%d_" % stage_id)
    out["g8_loss_history"] = np.array(hist)
    out["g8_schedule"] = np.array([[0, 6], [1, 14]])

    # ---- G9 checkpoint dict round trip through the reference's load_checkpoint ---------------------------
    f = make_fitter(N)
    set_params(f, 99)
    ck = os.path.join(tmp, "ckpt")
    os.makedirs(ck)
    frames = []
    for i in range(N):
        d = {
            "global_rotation": (f.global_rotation[i] * f.global_mask[0]).detach().numpy(),
            "joint_rotations": (f.joint_rotations[i] * f.rotation_mask).detach().numpy(),
            "betas": f.betas.detach().numpy() + 0.01 * i,
            "log_betascale": f.log_beta_scales.detach().numpy() - 0.02 * i,
            "trans": f.trans[i].detach().numpy(),
        }
        d = {k: v.astype(np.float32) for k, v in d.items()}
        os.makedirs(os.path.join(ck, "%04d" % i))
        with open(os.path.join(ck, "%04d" % i, "st10_ep0.pkl"), "wb") as fh:
            pickle.dump(d, fh)
        frames.append(d)
    g = make_fitter(N)
    with torch.no_grad():
        g.load_checkpoint(ck, "st10_ep0")
    for k in frames[0]:
        out["g9_frames_" + k] = np.stack([fr[k] for fr in frames])
    snapshot(g, "g9_loaded_")

    np.savez_compressed(os.path.join(HERE, "reference_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_golden.npz"), len(out), "arrays")
    for k in ("g4_zero_mean", "g4_point1_mean", "g5_zero_betas_mean", "init_global_rotation", "g5_init_ls"):
        print(k, out[k])


if __name__ == "__main__":
    main()
