"""GPU tests of the drop-in layer: the reference's class / function surface backed by HIP.

The strongest check: the reference's own stage loop semantics (new torch.optim.Adam per stage, stage-0
freezing, torso-only visibility, window accumulation, get_temporal) driven over *our* SMALFitter reproduces
the parameter trajectory recorded from the *reference's* SMALFitter (tests/golden, G8)."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import smal_oracle as so  # noqa: E402
from smalify_amd import config as cfg  # noqa: E402
from smalify_amd import synthetic  # noqa: E402
from tests.parity_cases import rel  # noqa: E402


@pytest.fixture(scope="module")
def md():
    return synthetic.synthetic_model(seed=0, shape_family_id=1)


def _data(golden):
    N, S = golden["g6_target_joints"].shape[0], int(golden["g6_image_size"])
    return (torch.zeros(N, 3, S, S), torch.zeros(N, 1, S, S), torch.from_numpy(golden["g6_target_joints"]),
            torch.from_numpy(golden["g6_visibility"])), N, S


def _make_fitter(golden, md, window):
    from smalify_amd.smal_fitter.smal_fitter import SMALFitter
    data, N, S = _data(golden)
    return SMALFitter("cuda", data, window, 1, True, model_data=md,
                      pose_prior_data=(golden["pose_prec"], golden["pose_mean"], golden["pose_mask"]),
                      shape_prior_data=(golden["unity_prec"], golden["unity_mean"]))


def test_smal_module_matches_reference_golden(golden, md):
    from smalify_amd.smal_model.smal_torch import SMAL
    smal = SMAL("cuda", shape_family_id=1, model_data=md)
    beta = torch.tensor(golden["g3_beta"], device="cuda", requires_grad=True)
    theta = torch.tensor(golden["g3_theta"], device="cuda", requires_grad=True)
    ls = torch.tensor(golden["g3_ls"], device="cuda", requires_grad=True)
    verts, joints, Rs, v_shaped = smal(beta, theta, betas_logscale=ls)
    vsel = torch.from_numpy(golden["g3_vsel"]).cuda()
    assert rel(verts[:, vsel].detach().cpu(), golden["g3_verts"]) < 1e-5
    assert rel(joints.detach().cpu(), golden["g3_joints"]) < 1e-5
    assert rel(Rs.cpu(), golden["g3_Rs"]) < 1e-5
    assert rel(v_shaped[:, vsel].cpu(), golden["g3_vshaped"]) < 1e-5
    func = (verts[:, vsel] * torch.from_numpy(golden["g3_wv"]).cuda()).sum() + \
           (joints * torch.from_numpy(golden["g3_wj"]).cuda()).sum()
    func.backward()
    assert rel(beta.grad.cpu(), golden["g3_dbeta"]) < 2e-4
    assert rel(theta.grad.cpu(), golden["g3_dtheta"]) < 2e-4
    assert rel(ls.grad.cpu(), golden["g3_dls"]) < 2e-4
    assert smal.faces.shape == (7774, 3) and smal.faces.dtype == torch.int64


@pytest.mark.parametrize("tag", ["delv", "vtmpl", "both", "rs"])
def test_smal_call_options_match_the_reference(md, tag):
    """SMAL.__call__(del_v=..., v_template=..., theta as (N,35,3,3) rotation matrices) -- smal_torch.py:99-133 -- against values
    and gradients of the reference's own SMAL (tests/golden/reference_golden_smal_options.npz)"""
    from smalify_amd.smal_model.smal_torch import SMAL
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden_smal_options.npz")))
    smal = SMAL("cuda", shape_family_id=1, model_data=md)
    t = lambda a: torch.tensor(np.asarray(a), device="cuda", requires_grad=True)  # noqa: E731
    beta, ls = t(g["beta"]), t(g["ls"])
    theta = t(g["Rs"]) if tag == "rs" else t(g["theta"])
    kw = {}
    if tag in ("delv", "both"):
        kw["del_v"] = t(g["del_v"])
    if tag in ("vtmpl", "both"):
        kw["v_template"] = t(g["v_template"])
    verts, joints, Rs, v_shaped = smal(beta, theta, betas_logscale=ls, **kw)
    vsel = torch.from_numpy(g["vsel"]).cuda()
    assert rel(verts[:, vsel].detach().cpu(), g[tag + "_verts"]) < 1e-5
    assert rel(joints.detach().cpu(), g[tag + "_joints"]) < 1e-5
    assert rel(v_shaped[:, vsel].cpu(), g[tag + "_vshaped"]) < 1e-5
    func = (verts[:, vsel] * torch.from_numpy(g["wv"]).cuda()).sum() + (joints * torch.from_numpy(g["wj"]).cuda()).sum()
    func.backward()
    assert rel(beta.grad.cpu(), g[tag + "_dbeta"]) < 2e-4 and rel(ls.grad.cpu(), g[tag + "_dls"]) < 2e-4
    assert rel(theta.grad.cpu(), g[tag + ("_dRs" if tag == "rs" else "_dtheta")]) < 2e-4
    if "del_v" in kw:
        assert rel(kw["del_v"].grad[:, vsel].cpu(), g[tag + "_ddel_v"]) < 2e-4
    if "v_template" in kw:
        assert rel(kw["v_template"].grad[vsel].cpu(), g[tag + "_dv_template"]) < 2e-4


@pytest.mark.parametrize("scaled", [False, True])
def test_global_rigid_transformation_gradients_match_the_reference(scaled):
    """drop-in batch_global_rigid_transformation: values and autograd against the reference's own function and autograd
    (tests/golden/reference_golden_smal_options.npz)"""
    from smalify_amd.smal_model.batch_lbs import batch_global_rigid_transformation
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden_smal_options.npz")))
    tag = "chain_scale" if scaled else "chain_noscale"
    t = lambda a: torch.tensor(np.asarray(a), device="cuda", requires_grad=True)  # noqa: E731
    R, J = t(g["chain_Rs"]), t(g["chain_Js"])
    L = t(g["chain_ls"]) if scaled else None
    nj, A = batch_global_rigid_transformation(R, J, g["chain_parents"], betas_logscale=L)
    assert rel(nj.detach().cpu(), g[tag + "_newJ"]) < 1e-5 and rel(A.detach().cpu(), g[tag + "_A"]) < 1e-5
    ((nj * torch.from_numpy(g["chain_wn"]).cuda()).sum() + (A * torch.from_numpy(g["chain_wa"]).cuda()).sum()).backward()
    assert rel(R.grad.cpu(), g[tag + "_dRs"]) < 2e-4 and rel(J.grad.cpu(), g[tag + "_dJs"]) < 2e-4
    if scaled:
        assert rel(L.grad.cpu(), g[tag + "_dls"]) < 2e-4


def test_batch_rodrigues_matches_reference_golden(golden):
    from smalify_amd.smal_model.batch_lbs import batch_rodrigues
    R = batch_rodrigues(torch.from_numpy(golden["g1_theta"]).cuda())
    assert np.abs(R.cpu().numpy() - golden["g1_R"]).max() < 5e-6


def test_prior_matches_reference_golden(golden, md):
    from smalify_amd.smal_model.smal_torch import SMAL
    from smalify_amd.smal_fitter.priors.pose_prior_35 import Prior
    SMAL("cuda", shape_family_id=1, model_data=md)
    prior = Prior(None, "cuda", prior_data=(golden["pose_prec"], golden["pose_mean"], golden["pose_mask"]))
    x = torch.tensor(golden["g4_x"], device="cuda", requires_grad=True)
    val = prior(x)
    val.mean().backward()
    assert rel(val.detach().cpu(), golden["g4_val"]) < 1e-5
    assert rel(x.grad.cpu(), golden["g4_dx"]) < 1e-5
    assert abs(prior(torch.zeros(1, 35, 3, device="cuda")).mean().item() - 0.80639112) < 1e-5   # SURVEY §8c known answer


def test_renderer_module(md):
    """Renderer.forward: shapes, (row, col) keypoints and gradients vs the oracle."""
    from smalify_amd.smal_model.smal_torch import SMAL
    from smalify_amd.smal_fitter.p3d_renderer import Renderer
    from tests import parity_cases as pc
    smal = SMAL("cuda", shape_family_id=1, model_data=md)
    S, M = 64, 2
    p = pc.random_pose(M, 31)
    theta = torch.tensor(np.concatenate([p["global_rotation"][:, None], p["joint_rotations"]], 1), device="cuda")
    with torch.no_grad():
        verts, joints, _, _ = smal(torch.tensor(np.tile(p["betas"], (M, 1)), device="cuda"), theta,
                                   betas_logscale=torch.tensor(np.tile(p["log_beta_scales"], (M, 1)), device="cuda"))
        verts = verts + torch.tensor(p["trans"], device="cuda")[:, None]
        pts = (joints + torch.tensor(p["trans"], device="cuda")[:, None])[:, cfg.CANONICAL_MODEL_JOINTS]
    verts = verts.clone().requires_grad_(True)
    pts = pts.clone().requires_grad_(True)
    renderer = Renderer(S, "cuda", model=smal.device_model)
    sil, proj = renderer(verts, pts, smal.faces.unsqueeze(0).expand(M, -1, -1))
    assert sil.shape == (M, 1, S, S) and proj.shape == (M, 25, 2)
    w = torch.randn(M, 1, S, S, device="cuda")
    wp = torch.randn(M, 25, 2, device="cuda")
    ((sil * w).sum() + (proj * wp).sum()).backward()
    v64 = verts.detach().cpu().double().requires_grad_(True)
    p64 = pts.detach().cpu().double().requires_grad_(True)
    om = so.OracleModel(md)
    sil_o = so.soft_silhouette(v64, om.faces, S)
    proj_o = so.project_points(p64, S)
    ((sil_o * w[:, 0].cpu().double()).sum() + (proj_o * wp.cpu().double()).sum()).backward()
    assert np.abs(sil[:, 0].detach().cpu().numpy() - sil_o.detach().numpy()).max() < 2e-3
    assert np.abs(proj.detach().cpu().numpy() - proj_o.detach().numpy()).max() < 1e-3
    assert rel(verts.grad.cpu(), v64.grad) < 1e-2
    assert rel(pts.grad.cpu(), p64.grad) < 1e-5
    # colour branch (visualisation): hard Phong render against the oracle's restatement.  Pixels on a face boundary
    # can go to a different face in float32; everywhere else the colours agree to rounding.
    _, _, color = renderer(verts.detach(), pts.detach(), None, render_texture=True)
    assert color.shape == (M, 3, S, S)
    col_o = so.hard_phong_render(verts.detach().cpu().double(), om.faces, S, np.array(cfg.MESH_COLOR) / 255.0)
    diff = np.abs(color.cpu().numpy() - col_o).max(axis=1)
    assert (diff > 1e-3).mean() < 5e-3, (diff > 1e-3).mean()
    assert np.median(diff) < 1e-5
    covered = (col_o < 1.0).any(axis=1)
    assert covered.mean() > 0.02 and ((color.cpu().numpy() < 1.0).any(axis=1) == covered).mean() > 0.998


def _reference_style_loop(f, golden, vis_full):
    """the reference's driver semantics (optimize_to_joints.py:90-137) over a SMALFitter-like module"""
    W = np.array(cfg.OPT_WEIGHTS).T
    hist = []
    N = f.num_images
    snaps = {}
    for stage_id, its in golden["g8_schedule"]:
        weights = (golden["g6_w0"] if stage_id == 0 else golden["g6_w1"])
        w_temp, lr = W[stage_id][6], W[stage_id][8]
        opt = torch.optim.Adam(f.parameters(), lr=lr, betas=(0.5, 0.999))
        if stage_id == 0:
            f.joint_rotations.requires_grad = False
            f.betas.requires_grad = False
            f.log_beta_scales.requires_grad = False
            tv = f.target_visibility.clone()
            f.target_visibility *= 0
            f.target_visibility[:, cfg.TORSO_JOINTS] = tv[:, cfg.TORSO_JOINTS]
        else:
            f.joint_rotations.requires_grad = True
            f.betas.requires_grad = True
            f.log_beta_scales.requires_grad = True
            f.target_visibility = vis_full.clone()          # CPU float tensor, like data[-1].clone()
        for _ in range(int(its)):
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
        snaps[int(stage_id)] = {k: getattr(f, k).detach().cpu().numpy().copy()
                                for k in ("global_rotation", "joint_rotations", "trans", "betas", "log_beta_scales")}
    return hist, snaps


def test_reference_driver_loop_over_dropin_fitter_reproduces_reference_trajectory(golden, md):
    f = _make_fitter(golden, md, 2)
    hist, snaps = _reference_style_loop(f, golden, torch.from_numpy(golden["g6_visibility"]))
    assert np.allclose(hist, golden["g8_loss_history"], rtol=5e-4), (hist, golden["g8_loss_history"])
    for stage in (0, 1):
        for k, v in snaps[stage].items():
            r = rel(v, golden["g8_after_stage%d_%s" % (stage, k)])
            assert r < 5e-4, (stage, k, r)


def test_fused_fitter_reproduces_reference_trajectory(golden, md):
    """same 20 iterations through the fused on-device loop (FusedFitter + HIP Adam)"""
    from smalify_amd import engine as eng, fitter as fit
    data, N, S = _data(golden)
    e = eng.Engine(eng.DeviceModel(md), N, S)
    e.set_pose_prior(golden["pose_prec"], golden["pose_mean"], golden["pose_mask"])
    e.set_shape_prior(golden["unity_prec"], golden["unity_mean"])
    f = fit.FusedFitter(e, golden["g6_target_joints"], golden["g6_visibility"], np.zeros((N, S, S), np.float32), 2,
                        True, golden["unity_mean"][:20], golden["unity_mean"][20:26])
    W = np.array(cfg.OPT_WEIGHTS).T
    hist = []
    for stage_id, its in golden["g8_schedule"]:
        weights = golden["g6_w0"] if stage_id == 0 else golden["g6_w1"]
        f.begin_stage(int(stage_id))
        for _ in range(int(its)):
            f.step(weights, float(W[stage_id][6]), float(W[stage_id][8]), int(stage_id))
            hist.append(float(f.losses.sum().item()))
        for k in ("global_rotation", "joint_rotations", "trans", "betas", "log_beta_scales"):
            r = rel(f.p[k].cpu().numpy(), golden["g8_after_stage%d_%s" % (stage_id, k)])
            assert r < 5e-4, (stage_id, k, r)
    assert np.allclose(hist, golden["g8_loss_history"], rtol=5e-4)


def test_checkpoint_layout_roundtrip(golden, md, tmp_path):
    """FusedFitter writes the reference's per-frame dict layout; SMALFitter.load_checkpoint reads it with the
    reference's semantics (betas / scales averaged over frames) — compared with the reference's own result (G9)."""
    from smalify_amd import engine as eng, fitter as fit
    data, N, S = _data(golden)
    e = eng.Engine(eng.DeviceModel(md), N, S)
    f = fit.FusedFitter(e, golden["g6_target_joints"], golden["g6_visibility"], np.zeros((N, S, S), np.float32), N,
                        True, golden["unity_mean"][:20], golden["unity_mean"][20:26])
    dirs = [str(tmp_path / ("%04d" % i)) for i in range(N)]
    f.export_checkpoints(dirs, 10, 0)
    with open(os.path.join(dirs[1], "st10_ep0.pkl"), "rb") as fh:
        d = pickle.load(fh)
    assert sorted(d) == ["betas", "global_rotation", "joint_rotations", "log_betascale", "trans"]
    assert d["global_rotation"].shape == (3,) and d["joint_rotations"].shape == (34, 3) and d["betas"].shape == (20,)
    assert d["log_betascale"].shape == (6,) and d["trans"].shape == (3,)
    assert all(v.dtype == np.float32 for v in d.values())
    # reference-written frames (G9) -> our load_checkpoint == reference's load_checkpoint
    ck = tmp_path / "ref"
    for i in range(N):
        os.makedirs(ck / ("%04d" % i))
        with open(ck / ("%04d" % i) / "st10_ep0.pkl", "wb") as fh:
            pickle.dump({k: golden["g9_frames_" + k][i] for k in d}, fh)
    g = _make_fitter(golden, md, N)
    g.load_checkpoint(str(ck), "st10_ep0")
    f.load_checkpoint(str(ck), "st10_ep0")
    for k in ("global_rotation", "joint_rotations", "trans", "betas", "log_beta_scales"):
        ref = golden["g9_loaded_" + k]
        assert np.allclose(getattr(g, k).detach().cpu().numpy(), ref, atol=1e-6), k
        assert np.allclose(f.p[k].cpu().numpy(), ref, atol=1e-6), k


def test_generate_visualization_writes_the_reference_files(golden, md, tmp_path):
    """SMALFitter.generate_visualization + ImageExporter (smal_fitter.py:209-272, optimize_to_joints.py:25-53): per
    frame a five-panel collage .png (colour renders included), the parameter .pkl and the posed mesh .ply."""
    import struct
    import zlib
    from smalify_amd.smal_fitter.optimize_to_joints import ImageExporter
    f = _make_fitter(golden, md, 2)
    N, S = f.num_images, f.image_size
    names = ["%04d.png" % i for i in range(N)]
    exporter = ImageExporter(str(tmp_path), names)
    exporter.stage_id, exporter.epoch_name = 10, 0
    f.generate_visualization(exporter)
    for i in range(N):
        stem = os.path.join(str(tmp_path), "%04d" % i, "st10_ep0")
        assert os.path.getsize(stem + ".ply") > 80 + md.num_verts * 12
        with open(stem + ".pkl", "rb") as fh:
            assert sorted(pickle.load(fh)) == ["betas", "global_rotation", "joint_rotations", "log_betascale", "trans"]
        blob = open(stem + ".png", "rb").read()
        assert blob[:8] == b"\x89PNG\r\n\x1a\n"
        w, h = struct.unpack(">II", blob[16:24])
        assert (w, h) == (5 * S, S)
        n = struct.unpack(">I", blob[33:37])[0]
        rows = np.frombuffer(zlib.decompress(blob[41:41 + n]), np.uint8).reshape(h, 1 + 3 * w)[:, 1:].reshape(h, w, 3)
        render_panel = rows[:, S:2 * S]
        assert (render_panel != 255).any(axis=2).mean() > 0.01        # the mesh is visible on the white background


def test_fit_sequence_runs_the_schedule_and_writes_the_final_files(golden, md, tmp_path):
    """driver-level entry (optimize_to_joints.py:56-144 counterpart): whole 4-stage schedule, shortened, on the synthetic
    sequence; the final st10_ep0 parameter / mesh files exist for every frame and the fit improved the keypoint loss"""
    from smalify_amd.smal_fitter.optimize_to_joints import fit_sequence
    data, N, S = _data(golden)
    names = ["frame_%02d.png" % i for i in range(N)]
    f = fit_sequence(data, names, md, (golden["pose_prec"], golden["pose_mean"], golden["pose_mask"]),
                     (golden["unity_prec"], golden["unity_mean"]), output_dir=str(tmp_path), window_size=2, iters_scale=0.02)
    assert f.e.status() == 0
    for i in range(N):
        stem = os.path.join(str(tmp_path), "frame_%02d" % i, "st10_ep0")
        assert os.path.exists(stem + ".pkl") and os.path.exists(stem + ".ply") and os.path.exists(stem + ".png")
        with open(stem + ".pkl", "rb") as fh:
            d = pickle.load(fh)
        assert d["joint_rotations"].shape == (34, 3) and d["joint_rotations"].dtype == np.float32
    assert np.isfinite(f.losses.cpu().numpy()).all()
    # fourth panel = 1 - |target silhouette - rendered| in [0, 1] (reference smal_fitter.py:243).  The all-zero masks of this
    # sequence are stored as BYTES on the device (sil_storage "auto"): the panel must still be 1 - rendered, i.e. white
    # where nothing is rendered and dark on the animal -- not a wrapped difference of byte values
    import struct
    import zlib
    assert f.target_sil.dtype == torch.uint8
    _, sil_r, _ = f.snapshot()
    expect = np.clip((1.0 - sil_r.cpu().numpy()) * 255.0, 0, 255)
    for i in range(N):
        blob = open(os.path.join(str(tmp_path), "frame_%02d" % i, "st10_ep0.png"), "rb").read()
        w, h = struct.unpack(">II", blob[16:24])
        n = struct.unpack(">I", blob[33:37])[0]
        rows = np.frombuffer(zlib.decompress(blob[41:41 + n]), np.uint8).reshape(h, 1 + 3 * w)[:, 1:].reshape(h, w, 3)
        panel = rows[:, 3 * S:4 * S, 0].astype(np.float64)
        assert np.abs(panel - expect[i]).max() <= 1.0, np.abs(panel - expect[i]).max()
        assert (panel > 250).mean() > 0.5


def test_joint_limits_stay_with_the_fitter_that_asked_for_them(golden, md):
    """engines are shared between fitters of one model and image size (runtime.get_engine): a fitter that enables the joint-limit
    term must not switch it on for the next default fitter -- the reference has the term commented out while its weight table
    says 100 (smal_fitter.py:146-151, config.py:68)"""
    from smalify_amd.smal_fitter.smal_fitter import SMALFitter
    data, N, S = _data(golden)
    pri = dict(model_data=md, pose_prior_data=(golden["pose_prec"], golden["pose_mean"], golden["pose_mask"]),
               shape_prior_data=(golden["unity_prec"], golden["unity_mean"]))
    w = [float(x) for x in golden["g6_w1"]]
    w[4] = 100.0                                        # the weight table's w_limit
    rs = np.random.RandomState(3)
    jr = torch.from_numpy((0.6 * rs.randn(N, 34, 3)).astype(np.float32)).cuda()      # well outside the limits

    def total(fitter):
        with torch.no_grad():
            fitter.joint_rotations.copy_(jr)
        loss, objs = fitter.forward(list(range(N)), w, 1)
        return float(loss.detach()), objs

    plain0, objs0 = total(SMALFitter("cuda", data, N, 1, True, **pri))
    limited, objs1 = total(SMALFitter("cuda", data, N, 1, True, enable_joint_limits=True, **pri))
    plain1, objs2 = total(SMALFitter("cuda", data, N, 1, True, **pri))               # same engine, after the limited fitter
    assert "limit" in objs1 and float(objs1["limit"]) > 0.0 and limited > plain0
    assert "limit" not in objs0 and "limit" not in objs2
    assert plain1 == plain0


def test_fit_args_of_another_header_are_refused(md):
    """smalfit_fit_args.struct_size (ABI v3): a struct laid out by another version of smalfit.h is an error, not a misread"""
    import ctypes as C
    from smalify_amd import engine as eng, synthetic
    e = eng.Engine(eng.DeviceModel(md), 2, 64)
    e.set_pose_prior(*synthetic.synthetic_pose_prior())
    e.set_shape_prior(*synthetic.synthetic_shape_prior())
    z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    a, _, _, keep = e.build_fit_args(betas=z(20), log_beta_scales=z(6), global_rotation=z(2, 3), joint_rotations=z(2, 34, 3), trans=z(2, 3),
                                     target_joints=z(2, 25, 2), target_visibility=z(2, 25), target_sil=None, weights=(1, 0, 0, 0, 0, 0),
                                     w_temp=0.0, window=2)
    assert a.struct_size == C.sizeof(type(a))
    assert e.lib.smalfit_fit_eval(e.handle, eng._stream(), C.byref(a)) == 0
    a.struct_size -= 8
    assert e.lib.smalfit_fit_eval(e.handle, eng._stream(), C.byref(a)) != 0
    assert b"struct_size" in e.lib.smalfit_last_error()
    assert e.lib.smalfit_version() == eng._lib.ABI_VERSION


def test_loaders_feed_device_resident_targets(md, golden, tmp_path):
    """SURVEY section 8f row 2, GPU side: a dataset in BADJA's on-disk format (PNG frames + half-resolution segmentation PNGs +
    the joint-annotation JSON) and one in StanfordExtra's (RLE masks) go through the cv2 / imageio / pycocotools-free loaders
    and from there into device-resident targets -- bytes where the crop is an exact multiple of 1/255, float32 where the
    reference's bilinear resize of BADJA masks leaves other values -- and a short fit runs on them."""
    from tests.test_data_loader_cpu import _write_badja
    from smalify_amd import engine as eng, fitter as fit
    from smalify_amd.smal_fitter import data_loader as dl
    from smalify_amd.smal_fitter.optimize_to_joints import fit_sequence, write_png
    import json
    root = str(tmp_path / "BADJA")
    _write_badja(root)
    (rgb, sil, joints, vis), names = dl.load_badja_sequence(root, "synth", 64, image_range=range(0, 3))
    pri = ((golden["pose_prec"], golden["pose_mean"], golden["pose_mask"]), (golden["unity_prec"], golden["unity_mean"]))
    f = fit_sequence((rgb, sil, joints, vis), names, md, pri[0], pri[1], output_dir=None, window_size=2, iters_scale=0.02)
    assert f.e.status() == 0 and np.isfinite(f.losses.cpu().numpy()).all()
    # whatever the storage, the kernels see exactly the loader's silhouettes
    assert torch.equal(f.target_sil_float().cpu(), sil.reshape(3, 64, 64))
    exact = bool((torch.round(sil * 255.0) / 255.0 == sil).all())
    assert (f.target_sil.dtype == torch.uint8) == exact
    # StanfordExtra: RLE mask -> nearest-neighbour crop: binary, hence bytes on the device
    sroot = str(tmp_path / "StanfordExtra")
    os.makedirs(os.path.join(sroot, "sample_imgs", "n0-dog"))
    rs = np.random.RandomState(3)
    write_png(os.path.join(sroot, "sample_imgs", "n0-dog", "a.png"), (rs.rand(50, 70, 3) * 255).astype(np.uint8))
    mask = np.zeros((50, 70), np.uint8); mask[10:40, 20:60] = 1
    sj = np.concatenate([rs.rand(24, 2) * [70, 50], (rs.rand(24, 1) < 0.8)], 1)
    with open(os.path.join(sroot, "StanfordExtra_sample.json"), "w") as fh:
        json.dump([{"img_path": "n0-dog/a.png", "img_height": 50, "img_width": 70, "seg": dl.encode_rle(mask), "joints": sj.tolist()}], fh)
    (rgb1, sil1, j1, v1), n1 = dl.load_stanford_sequence(sroot, "n0-dog/a.png", 64)
    f1 = fit_sequence((rgb1, sil1, j1, v1), n1, md, pri[0], pri[1], output_dir=None, window_size=1, iters_scale=0.02)
    assert f1.e.status() == 0 and f1.target_sil.dtype == torch.uint8
    assert torch.equal(f1.target_sil_float().cpu(), sil1.reshape(1, 64, 64))
