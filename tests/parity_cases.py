"""Parity cases HIP-vs-oracle, shared by tests/test_gpu_parity.py (asserting) and tests/gpu_diag.py
(printing).  Every function returns {metric_name: value}; nothing here asserts.

Oracle = oracle/smal_oracle.py in float64 (test infrastructure).  HIP = smalify_amd through the C-ABI.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import smal_oracle as so
from smalify_amd import config as cfg
from smalify_amd import engine as eng
from smalify_amd import model_io, synthetic


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda()


_CACHE = {}


def get_oracle_model(dense=False):
    """(model description, oracle model): CPU only"""
    key = ("oracle_model", dense)
    if key not in _CACHE:
        md = synthetic.synthetic_model(seed=0, shape_family_id=1, dense_weights=dense)
        _CACHE[key] = (md, so.OracleModel(md))
    return _CACHE[key]


def get_model(dense=False):
    key = ("model", dense)
    if key not in _CACHE:
        md, om = get_oracle_model(dense)
        _CACHE[key] = (md, om, eng.DeviceModel(md))
    return _CACHE[key]


def get_engine(max_frames, S, dense=False):
    key = ("engine", max_frames, S, dense)
    if key not in _CACHE:
        _, _, dm = get_model(dense)
        e = eng.Engine(dm, max_frames, S)
        pp = synthetic.synthetic_pose_prior()
        sp = synthetic.synthetic_shape_prior()
        e.set_pose_prior(*pp)
        e.set_shape_prior(*sp)
        _CACHE[key] = (e, pp, sp)
    return _CACHE[key]


def random_pose(M, seed, scale=1.0, z=1.45):
    rs = np.random.RandomState(seed)
    init = model_io.initial_global_rotation()
    return dict(
        betas=(0.4 * rs.randn(20)).astype(np.float32),
        log_beta_scales=(0.15 * rs.randn(6)).astype(np.float32),
        global_rotation=(init[None] + 0.25 * scale * rs.randn(M, 3)).astype(np.float32),
        joint_rotations=(0.2 * scale * rs.randn(M, 34, 3)).astype(np.float32),
        trans=(np.array([0.03, -0.02, z])[None] + 0.03 * rs.randn(M, 3)).astype(np.float32))


# ------------------------------------------------------------------------------------------------
def case_rodrigues():
    rs = np.random.RandomState(3)
    th = rs.randn(64, 3).astype(np.float32)
    th[0] = 0
    th[1] = [1e-6, -2e-6, 3e-6]
    th[2] = [3.1, 0, 0]
    G = rs.randn(64, 3, 3).astype(np.float32)
    R = eng.rodrigues(dev(th)).cpu().numpy()
    dth = eng.rodrigues_backward(dev(th), dev(G)).cpu().numpy()
    t64 = torch.from_numpy(th).double().requires_grad_(True)
    Ro = so.rodrigues(t64)
    (Ro * torch.from_numpy(G).double()).sum().backward()
    return {"rodrigues_fwd_maxabs": float(np.abs(R - Ro.detach().numpy()).max()),
            "rodrigues_bwd_rel": rel(dth, t64.grad.numpy()),
            "rodrigues_bwd_zero_row_abs": float(np.abs(dth[0] - t64.grad.numpy()[0]).max())}


def case_lbs(M=3, dense=False, with_scale=True):
    md, om, _ = get_model(dense)
    e, _, _ = get_engine(8 if M <= 8 else 24, 64, dense)
    rs = np.random.RandomState(5)
    beta = (0.5 * rs.randn(M, 20)).astype(np.float32)
    theta = (0.3 * rs.randn(M, 35, 3)).astype(np.float32)
    theta[0, 3:] = 0.0                       # a frame with exactly-zero joint rotations
    ls = (0.2 * rs.randn(M, 6)).astype(np.float32) if with_scale else None
    wv = rs.randn(M, md.num_verts, 3).astype(np.float32)
    wj = rs.randn(M, 41, 3).astype(np.float32)
    v, j, Rs, vs = e.lbs_forward(dev(beta), dev(theta), dev(ls) if with_scale else None)
    db, dt, dl = e.lbs_backward(dev(beta), dev(theta), dev(ls) if with_scale else None, dev(wv), dev(wj))
    b64 = torch.from_numpy(beta).double().requires_grad_(True)
    t64 = torch.from_numpy(theta).double().requires_grad_(True)
    l64 = torch.from_numpy(ls).double().requires_grad_(True) if with_scale else None
    vo, jo, Ro, vso = so.smal_forward(om, b64, t64, l64)
    ((vo * torch.from_numpy(wv).double()).sum() + (jo * torch.from_numpy(wj).double()).sum()).backward()
    out = {"lbs_verts_rel": rel(v.cpu(), vo.detach()), "lbs_joints_rel": rel(j.cpu(), jo.detach()),
           "lbs_Rs_rel": rel(Rs.cpu(), Ro.detach()), "lbs_vshaped_rel": rel(vs.cpu(), vso.detach()),
           "lbs_dbeta_rel": rel(db.cpu(), b64.grad), "lbs_dtheta_rel": rel(dt.cpu(), t64.grad)}
    if with_scale:
        out["lbs_dlogscale_rel"] = rel(dl.cpu(), l64.grad)
    return out


def _sil_metrics(sil_hip, sil_or):
    d = np.abs(np.asarray(sil_hip, np.float64) - np.asarray(sil_or, np.float64))
    return {"sil_maxabs": float(d.max()), "sil_meanabs": float(d.mean()),
            "sil_frac_gt_1e-4": float((d > 1e-4).mean()), "sil_frac_gt_1e-3": float((d > 1e-3).mean())}


def case_render(M=2, S=64, z=1.45, seed=11, shrink=1.0):
    """shrink < 1 scales the mesh about its centroid: every face then lands on the same few pixels"""
    md, om, _ = get_model()
    e, _, _ = get_engine(8, S)
    p = random_pose(M, seed, z=z)
    theta = np.concatenate([p["global_rotation"][:, None], p["joint_rotations"]], 1)
    with torch.no_grad():
        vo, jo, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(p["betas"], (M, 1))).double(),
                                       torch.from_numpy(theta).double(),
                                       torch.from_numpy(np.tile(p["log_beta_scales"], (M, 1))).double())
    verts = vo + torch.from_numpy(p["trans"]).double()[:, None]
    if shrink != 1.0:
        c = verts.mean(1, keepdim=True)
        verts = c + shrink * (verts - c)
    verts = verts.float()                                                          # float32 inputs for both
    pts = (jo + torch.from_numpy(p["trans"]).double()[:, None])[:, so.CANONICAL].float()
    sil, proj = e.render_forward(verts.cuda().contiguous(), pts.cuda().contiguous())
    status = e.status()
    v64 = verts.double().requires_grad_(True)
    sil_o, stats = so.soft_silhouette(v64, om.faces, S, return_stats=True)
    rs = np.random.RandomState(seed + 1)
    w = rs.randn(M, S, S).astype(np.float32)
    (sil_o * torch.from_numpy(w).double()).sum().backward()
    dverts = e.render_backward(verts.cuda().contiguous(), sil, dev(w)).cpu().numpy()
    proj_o = so.project_points(pts.double(), S).numpy()
    out = _sil_metrics(sil.cpu().numpy(), sil_o.detach().numpy())
    out.update({"render_status": status, "render_oracle_max_faces_per_pixel": stats["max_faces_per_pixel"],
                "render_coverage": float((sil_o > 0.5).double().mean()),
                "render_dverts_rel": rel(dverts, v64.grad.numpy()),
                "render_dverts_norm": float(np.linalg.norm(v64.grad.numpy())),
                "render_proj_maxabs_px": float(np.abs(proj.cpu().numpy() - proj_o).max())})
    return out


def make_problem_cpu(M, S, window, seed=21, z=1.45, with_sil=True):
    """Synthetic fitting problem: targets from a ground-truth pose, evaluation at a perturbed pose (oracle side only)."""
    md, om = get_oracle_model()
    pp = synthetic.synthetic_pose_prior()
    sp = synthetic.synthetic_shape_prior()
    gt = random_pose(M, seed, z=z)
    cur = random_pose(M, seed, z=z)
    rs = np.random.RandomState(seed + 7)
    cur["global_rotation"] += (0.05 * rs.randn(M, 3)).astype(np.float32)
    cur["joint_rotations"] += (0.08 * rs.randn(M, 34, 3)).astype(np.float32)
    cur["trans"] += (0.02 * rs.randn(M, 3)).astype(np.float32)
    cur["betas"] += (0.1 * rs.randn(20)).astype(np.float32)
    with torch.no_grad():
        theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
        vo, jo, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(gt["betas"], (M, 1))).double(),
                                       torch.from_numpy(theta).double(),
                                       torch.from_numpy(np.tile(gt["log_beta_scales"], (M, 1))).double())
        t = torch.from_numpy(gt["trans"]).double()[:, None]
        tj = so.project_points((jo + t)[:, so.CANONICAL], S).numpy() + rs.randn(M, 25, 2)
        if with_sil:
            tsil = (so.soft_silhouette(vo + t, om.faces, S) > 0.5).double().numpy()
        else:
            tsil = np.zeros((M, S, S))
    vis = (rs.rand(M, 25) < 0.85).astype(np.float32)
    vis[:, [2, 5, 8]] = 1.0
    prob = so.FitProblem(om, S, tj, vis, tsil, pp[0], pp[1], pp[2], sp[0], sp[1], window, use_unity_prior=True)
    return prob, cur, dict(tj=tj.astype(np.float32), vis=vis, tsil=tsil.astype(np.float32))


def make_problem(M, S, window, seed=21, z=1.45, with_sil=True):
    """make_problem_cpu + the device engine holding the same priors"""
    e, _, _ = get_engine(max(M, 8), S)
    prob, cur, tg = make_problem_cpu(M, S, window, seed, z, with_sil)
    return e, prob, cur, tg


def case_fit(M=4, S=64, window=2, stage=2, seed=21, trainable=None):
    W = np.array(cfg.OPT_WEIGHTS).T
    weights = W[stage][:6].copy()
    w_temp = float(W[stage][6])
    with_sil = weights[1] > 0
    e, prob, cur, tg = make_problem(M, S, window, seed, with_sil=with_sil)
    if trainable is None:
        trainable = so.trainable_names(stage)
    vis = tg["vis"]
    vis_o = None
    if stage == 0:
        vis_o = so.stage0_visibility(torch.from_numpy(vis).double())
        vis = vis_o.numpy().astype(np.float32)
    params64 = {k: torch.from_numpy(v).double() for k, v in cur.items()}
    total, sums, grads_o = so.loss_and_grads(prob, params64, weights, w_temp, trainable, vis_o)
    d = {k: dev(v) for k, v in cur.items()}
    losses, grads = e.fit_eval(betas=d["betas"], log_beta_scales=d["log_beta_scales"],
                               global_rotation=d["global_rotation"], joint_rotations=d["joint_rotations"],
                               trans=d["trans"], target_joints=dev(tg["tj"]), target_visibility=dev(vis),
                               target_sil=dev(tg["tsil"]), weights=weights, w_temp=w_temp, window=window,
                               want=trainable)
    status = e.status()
    l = losses.cpu().numpy().astype(np.float64)
    names = ("joint", "pose", "splay", "betas", "sil_reproj", "temp_joint", "temp_global", "temp_trans")
    out = {"fit_status": status, "fit_total_rel": abs(l.sum() - float(total)) / abs(float(total)),
           "fit_total_oracle": float(total)}
    for i, nme in enumerate(names):
        ref = sums.get(nme, 0.0)
        out["fit_loss_%s_abs" % nme] = abs(l[i] - ref)
        out["fit_loss_%s_oracle" % nme] = ref
    for k in trainable:
        out["fit_grad_%s_rel" % k] = rel(grads[k].cpu().numpy(), grads_o[k].numpy())
        out["fit_grad_%s_norm" % k] = float(np.linalg.norm(grads_o[k].numpy()))
    return out


def case_fit_golden(golden, tag, window, stage, family1=True):
    """HIP fitter evaluation against the *reference's own* outputs (tests/golden, no silhouette).
    family1=False: SMAL cluster prior (20-dim) with per-frame limb scales (reference smal_fitter.py:62-72)."""
    S = int(golden["g6_image_size"])
    N = golden["g6_target_joints"].shape[0]
    key = ("golden_engine", S, family1)
    if key not in _CACHE:
        if family1:
            dm = get_model()[2]
        else:
            dm = eng.DeviceModel(synthetic.synthetic_model(seed=0, shape_family_id=0))
        e = eng.Engine(dm, 8, S)
        e.set_pose_prior(golden["pose_prec"], golden["pose_mean"], golden["pose_mask"])
        if family1:
            e.set_shape_prior(golden["unity_prec"], golden["unity_mean"])
        else:
            e.set_shape_prior(golden["fam0_prec"], golden["fam0_mean"])
        _CACHE[key] = e
    e = _CACHE[key]
    weights = golden["g6_w0"] if stage == 0 else golden["g6_w1"]
    w_temp = float(golden["g6_wtemp"][stage])
    vis = golden["g6_visibility"].astype(np.float32)
    if stage == 0:
        v0 = np.zeros_like(vis)
        v0[:, cfg.TORSO_JOINTS] = vis[:, cfg.TORSO_JOINTS]
        vis = v0
    p = {k: dev(golden["%s_p_%s" % (tag, k)]) for k in
         ("betas", "log_beta_scales", "global_rotation", "joint_rotations", "trans")}
    trainable = ("global_rotation", "trans") if stage == 0 else so.PARAM_ORDER
    losses, grads = e.fit_eval(betas=p["betas"], log_beta_scales=p["log_beta_scales"],
                               global_rotation=p["global_rotation"], joint_rotations=p["joint_rotations"],
                               trans=p["trans"], target_joints=dev(golden["g6_target_joints"]),
                               target_visibility=dev(vis), target_sil=None, weights=weights, w_temp=w_temp,
                               window=window, want=trainable)
    l = losses.cpu().numpy().astype(np.float64)
    out = {"golden_total_rel": abs(l.sum() - float(golden[tag + "_total"])) / abs(float(golden[tag + "_total"]))}
    for i, nme in enumerate(("joint", "pose", "splay", "betas")):
        key = "%s_term_%s" % (tag, nme)
        if key in golden:
            out["golden_loss_%s_rel" % nme] = abs(l[i] - float(golden[key])) / max(abs(float(golden[key])), 1e-12)
    t = golden[tag + "_temporal"]
    out["golden_temporal_rel"] = rel(l[5:8], t)
    for k in trainable:
        out["golden_grad_%s_rel" % k] = rel(grads[k].cpu().numpy(), golden["%s_g_%s" % (tag, k)])
    return out


def case_adam():
    rs = np.random.RandomState(9)
    p0 = rs.randn(1000).astype(np.float32)
    p = dev(p0)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    po = {"x": torch.from_numpy(p0).double()}
    opt = so.Adam(["x"], lr=5e-3)
    for t in range(1, 6):
        g = rs.randn(1000).astype(np.float32) * (10.0 ** rs.randint(-3, 3))
        eng.adam_step(p, dev(g), m, v, 5e-3, t)
        opt.step(po, {"x": torch.from_numpy(g).double()})
    return {"adam_rel": rel(p.cpu().numpy(), po["x"].numpy())}


def case_cache_consistency(M=4, S=64, window=3, stage=2, z=1.45, steps=6, seed=21):
    """Same evaluations on an engine that keeps the rasteriser's depth-bound cache and on one that resets it each time."""
    W = np.array(cfg.OPT_WEIGHTS).T
    weights, w_temp = W[stage][:6].copy(), float(W[stage][6])
    e_a, prob, cur, tg = make_problem(M, S, window, seed, z=z)
    md, om, dm = get_model()
    e_b = eng.Engine(dm, max(M, 8), S)
    _, pp, sp = get_engine(max(M, 8), S)
    e_b.set_pose_prior(*pp)
    e_b.set_shape_prior(*sp)
    names = so.trainable_names(stage)

    def ev(e, p):
        d = {k: dev(v) for k, v in p.items()}
        losses, grads = e.fit_eval(betas=d["betas"], log_beta_scales=d["log_beta_scales"],
                                   global_rotation=d["global_rotation"], joint_rotations=d["joint_rotations"],
                                   trans=d["trans"], target_joints=dev(tg["tj"]), target_visibility=dev(tg["vis"]),
                                   target_sil=dev(tg["tsil"]), weights=weights, w_temp=w_temp, window=window, want=names)
        return losses.cpu().numpy().astype(np.float64), {k: v.cpu().numpy().astype(np.float64) for k, v in grads.items()}

    rs = np.random.RandomState(seed + 3)
    loss_rel, grad_rel = 0.0, 0.0
    for step in range(steps + 2):
        if step >= 2:                      # steps 0 and 1 evaluate the same pose (fresh, then fully cached)
            cur = {k: v.copy() for k, v in cur.items()}
            cur["global_rotation"] += (0.004 * rs.randn(M, 3)).astype(np.float32)
            cur["joint_rotations"] += (0.004 * rs.randn(M, 34, 3)).astype(np.float32)
            cur["trans"] += (0.003 * rs.randn(M, 3)).astype(np.float32)
        la, ga = ev(e_a, cur)
        e_b.reset_raster_cache()
        lb, gb = ev(e_b, cur)
        loss_rel = max(loss_rel, abs(la.sum() - lb.sum()) / abs(lb.sum()))
        for k in ga:
            grad_rel = max(grad_rel, rel(ga[k], gb[k]))
    return {"loss_rel_max": loss_rel, "grad_rel_max": grad_rel, "status": e_a.status() | e_b.status()}


def case_sil_trajectory(M=4, S=64, window=2, iters=8, seed=21):
    """The full loop with the silhouette on (stage-2 weights, lr 5e-4): FusedFitter on the GPU vs the oracle's loss +
    autograd + Adam on the CPU, same start, same targets.  Returns the per-tensor relative differences after `iters`
    iterations and the loss histories."""
    from smalify_amd import fitter as fit
    W = np.array(cfg.OPT_WEIGHTS).T
    stage = 2
    weights, w_temp, lr = W[stage][:6].copy(), float(W[stage][6]), float(W[stage][8])
    e, prob, cur, tg = make_problem(M, S, window, seed)
    names = so.trainable_names(stage)
    # oracle loop
    params = {k: torch.from_numpy(v).double() for k, v in cur.items()}
    opt = so.Adam(so.PARAM_ORDER, lr=lr)
    hist_o = []
    for _ in range(iters):
        total, sums, grads = so.loss_and_grads(prob, params, weights, w_temp, names)
        opt.step(params, grads)
        hist_o.append(float(total))
    # device loop (betas / limb scales start from the perturbed values, as in the oracle)
    f = fit.FusedFitter(e, tg["tj"], tg["vis"], tg["tsil"], window, True, cur["betas"], cur["log_beta_scales"])
    for k in ("global_rotation", "joint_rotations", "trans"):
        f.p[k].copy_(dev(cur[k]))
    f.begin_stage(stage)
    hist = []
    for _ in range(iters):
        f.step(weights, w_temp, lr, stage)
        hist.append(float(f.losses.double().sum().item()))
    out = {"traj_loss_rel_max": float(np.max(np.abs(np.array(hist) - np.array(hist_o)) / np.abs(np.array(hist_o)))),
           "traj_status": e.status()}
    for k in ("betas", "log_beta_scales", "global_rotation", "joint_rotations", "trans"):
        out["traj_%s_rel" % k] = rel(f.p[k].cpu().numpy().reshape(params[k].shape), params[k].numpy())
    return out


def case_fit_limits(M=3, S=64, window=2, seed=23, w_limit=40.0):
    """the joint-limit hinge term (reference smal_fitter.py:146-151 with the table of priors/joint_limits_prior.py): one
    stage-1-type evaluation without silhouette, w_limit switched on, HIP vs oracle"""
    W = np.array(cfg.OPT_WEIGHTS).T
    weights = W[1][:6].copy()
    weights[1] = 0.0                      # no silhouette: this case is about the limit term
    weights[4] = w_limit
    w_temp = float(W[1][6])
    e, prob, cur, tg = make_problem(M, S, window, seed, with_sil=False)
    lo, hi = model_io.joint_limit_table()
    e.set_joint_limits(lo, hi)
    prob.limits = (torch.from_numpy(lo).double(), torch.from_numpy(hi).double())
    names = so.PARAM_ORDER
    params64 = {k: torch.from_numpy(v).double() for k, v in cur.items()}
    total, sums, grads_o = so.loss_and_grads(prob, params64, weights, w_temp, names)
    d = {k: dev(v) for k, v in cur.items()}
    losses, grads = e.fit_eval(betas=d["betas"], log_beta_scales=d["log_beta_scales"], global_rotation=d["global_rotation"],
                               joint_rotations=d["joint_rotations"], trans=d["trans"], target_joints=dev(tg["tj"]),
                               target_visibility=dev(tg["vis"]), target_sil=None, weights=weights, w_temp=w_temp, window=window, want=names)
    l = losses.cpu().numpy().astype(np.float64)
    out = {"status": e.status(), "limit_oracle": sums["limit"], "limit_hip": float(l[8]),
           "total_rel": abs(l.sum() - float(total)) / abs(float(total))}
    for k in names:
        out["grad_%s_rel" % k] = rel(grads[k].cpu().numpy(), grads_o[k].numpy())
    return out


def case_u8_targets(M=8, S=128, window=4, iters=6, seed=29):
    """the same short fit with the target silhouettes resident as float32 and as bytes: every parameter and every loss term
    must come out bit-identical (binary masks: b / 255 is exactly 0 or 1)"""
    from smalify_amd import fitter as fit
    W = np.array(cfg.OPT_WEIGHTS).T
    e, prob, cur, tg = make_problem(M, S, window, seed)
    res = {}
    for storage in ("f32", "u8"):
        e.reset_raster_cache()
        f = fit.FusedFitter(e, tg["tj"], tg["vis"], tg["tsil"], window, True, cur["betas"], cur["log_beta_scales"], sil_storage=storage)
        for k in ("global_rotation", "joint_rotations", "trans"):
            f.p[k].copy_(dev(cur[k]))
        f.begin_stage(2)
        f.run_iterations(W[2][:6], float(W[2][6]), float(W[2][8]), 2, iters)
        res[storage] = (f.flat.cpu().numpy().copy(), f.losses.cpu().numpy().copy(), f.target_sil.dtype)
    return {"dtype_f32": str(res["f32"][2]), "dtype_u8": str(res["u8"][2]),
            "params_identical": bool(np.array_equal(res["f32"][0], res["u8"][0])),
            "losses_identical": bool(np.array_equal(res["f32"][1], res["u8"][1])), "status": e.status()}


FULL_SCHEDULE_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_full_schedule.npz")


def full_schedule_iterations(iters_scale):
    return [max(1, int(round(int(w[7]) * iters_scale))) for w in np.array(cfg.OPT_WEIGHTS).T]


def full_schedule_oracle(prob, cur, sched, trace=None):
    """the oracle's stage loop (optimize_to_joints.py:90-137) in float64 from the start `cur`: final per-term sums and
    parameters.  `trace`, if a list, receives the total of every iteration."""
    W = np.array(cfg.OPT_WEIGHTS).T
    params = {k: torch.from_numpy(v).double() for k, v in cur.items()}
    sums_o = {}
    for stage, w in enumerate(W):
        weights, w_temp, lr = w[:6].copy(), float(w[6]), float(w[8])
        names = so.trainable_names(stage)
        vis = so.stage0_visibility(prob.vis) if stage == 0 else None
        opt = so.Adam(so.PARAM_ORDER, lr=lr)
        for _ in range(sched[stage]):
            total, sums_o, grads = so.loss_and_grads(prob, params, weights, w_temp, names, visibility=vis)
            if trace is not None:
                trace.append(float(total))
            opt.step(params, grads)
    return {"sums": {k: float(v) for k, v in sums_o.items()}, "params": {k: v.numpy() for k, v in params.items()}}


def problem_fingerprint(cur, tg):
    """sha256 over the start parameters and targets of a problem: a cached oracle result belongs to exactly these inputs"""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(cur):
        h.update(np.ascontiguousarray(cur[k]).tobytes())
    for k in sorted(tg):
        h.update(np.ascontiguousarray(tg[k]).tobytes())
    return h.hexdigest()


def load_full_schedule_fixture(M, S, window, iters_scale, seed, cur, tg):
    """tests/golden/oracle_full_schedule.npz (written by tests/golden/make_oracle_full_schedule.py) holds the ORACLE's
    end state for one configuration -- minutes of float64 CPU work that the GPU box would otherwise repeat in every run.
    Used only when the configuration and the fingerprint of the problem's inputs match; anything else runs the oracle."""
    if not os.path.exists(FULL_SCHEDULE_FIXTURE):
        return None
    z = np.load(FULL_SCHEDULE_FIXTURE, allow_pickle=False)
    key = "M%d_S%d_w%d_scale%g_seed%d" % (M, S, window, iters_scale, seed)
    if str(z["config"]) != key or str(z["fingerprint"]) != problem_fingerprint(cur, tg):
        return None
    return {"sums": {k[4:]: float(z[k]) for k in z.files if k.startswith("sum_")},
            "params": {k[6:]: z[k] for k in z.files if k.startswith("param_")}}


def case_full_schedule(M=4, S=64, window=2, iters_scale=0.1, seed=21):
    """All four stages of the reference schedule (config.OPT_WEIGHTS: weights, learning rates, stage-0 freeze and torso
    keypoints, fresh Adam per stage), iteration counts scaled by `iters_scale`: FusedFitter on the GPU (one library call
    per stage) against the oracle's loss + autograd + Adam on the CPU from the same start.  Returns the final per-term
    losses of both, their relative differences, and the end-of-run relative L2 difference of every parameter tensor
    (checks (iii) and (iv))."""
    from smalify_amd import fitter as fit
    W = np.array(cfg.OPT_WEIGHTS).T
    e, prob, cur, tg = make_problem(M, S, window, seed)
    sched = full_schedule_iterations(iters_scale)
    orc = load_full_schedule_fixture(M, S, window, iters_scale, seed, cur, tg)
    oracle_source = "fixture"
    if orc is None:
        orc = full_schedule_oracle(prob, cur, sched)
        oracle_source = "live"
    sums_o, params = orc["sums"], {k: torch.from_numpy(v) for k, v in orc["params"].items()}
    # device loop
    f = fit.FusedFitter(e, tg["tj"], tg["vis"], tg["tsil"], window, True, cur["betas"], cur["log_beta_scales"])
    for k in ("global_rotation", "joint_rotations", "trans"):
        f.p[k].copy_(dev(cur[k]))
    for stage, w in enumerate(W):
        f.begin_stage(stage)
        f.run_iterations(w[:6].copy(), float(w[6]), float(w[8]), stage, sched[stage])
    l = f.losses.cpu().numpy().astype(np.float64)
    out = {"schedule": sched, "oracle_source": oracle_source, "status": e.status(), "final_total_oracle": float(sum(sums_o.values())), "final_total_hip": float(l.sum())}
    out["final_total_rel"] = abs(out["final_total_hip"] - out["final_total_oracle"]) / abs(out["final_total_oracle"])
    for i, nme in enumerate(eng.LOSS_NAMES):
        ref = float(sums_o.get(nme, 0.0))
        out["final_%s_oracle" % nme], out["final_%s_hip" % nme] = ref, float(l[i])
        out["final_%s_rel" % nme] = abs(l[i] - ref) / max(abs(ref), 1e-12)
    for k in ("betas", "log_beta_scales", "global_rotation", "joint_rotations", "trans"):
        out["param_%s_rel_l2" % k] = rel(f.p[k].cpu().numpy().reshape(params[k].shape), params[k].numpy())
    return out


def case_fit_family0_512(golden, M=2, S=512, window=2, stage=2, seed=43):
    """BASELINE config 5's ingredients in one evaluation: a non-unity shape family (20-dim SMAL cluster prior from the
    reference's golden data, per-frame (N,6) limb scales without a regulariser), 512 x 512 silhouettes, stage-2 weights."""
    W = np.array(cfg.OPT_WEIGHTS).T
    weights, w_temp = W[stage][:6].copy(), float(W[stage][6])
    md0 = synthetic.synthetic_model(seed=0, shape_family_id=0)
    om = so.OracleModel(md0)
    dm = eng.DeviceModel(md0)
    e = eng.Engine(dm, M, S)
    e.set_pose_prior(golden["pose_prec"], golden["pose_mean"], golden["pose_mask"])
    e.set_shape_prior(golden["fam0_prec"], golden["fam0_mean"])
    gt = random_pose(M, seed, z=1.6)
    cur = random_pose(M, seed, z=1.6)
    rs = np.random.RandomState(seed + 7)
    cur["global_rotation"] += (0.04 * rs.randn(M, 3)).astype(np.float32)
    cur["joint_rotations"] += (0.06 * rs.randn(M, 34, 3)).astype(np.float32)
    cur["trans"] += (0.02 * rs.randn(M, 3)).astype(np.float32)
    cur["log_beta_scales"] = (0.1 * rs.randn(M, 6)).astype(np.float32)                 # per-frame limb scales
    with torch.no_grad():
        theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
        vo, jo, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(gt["betas"], (M, 1))).double(),
                                       torch.from_numpy(theta).double(),
                                       torch.from_numpy(np.tile(gt["log_beta_scales"], (M, 1))).double())
        t = torch.from_numpy(gt["trans"]).double()[:, None]
        tj = so.project_points((jo + t)[:, so.CANONICAL], S).numpy() + rs.randn(M, 25, 2)
        tsil = (so.soft_silhouette(vo + t, om.faces, S) > 0.5).double().numpy()
    vis = (rs.rand(M, 25) < 0.85).astype(np.float32)
    prob = so.FitProblem(om, S, tj, vis, tsil, golden["pose_prec"], golden["pose_mean"], golden["pose_mask"],
                         golden["fam0_prec"], golden["fam0_mean"], window, use_unity_prior=False)
    names = ("betas", "log_beta_scales", "global_rotation", "trans", "joint_rotations")
    params64 = {k: torch.from_numpy(v).double() for k, v in cur.items()}
    total, sums, grads_o = so.loss_and_grads(prob, params64, weights, w_temp, names)
    d = {k: dev(v) for k, v in cur.items()}
    losses, grads = e.fit_eval(betas=d["betas"], log_beta_scales=d["log_beta_scales"], global_rotation=d["global_rotation"],
                               joint_rotations=d["joint_rotations"], trans=d["trans"], target_joints=dev(tj.astype(np.float32)),
                               target_visibility=dev(vis), target_sil=dev(tsil.astype(np.float32)), weights=weights,
                               w_temp=w_temp, window=window, want=names)
    l = losses.cpu().numpy().astype(np.float64)
    out = {"status": e.status(), "total_rel": abs(l.sum() - float(total)) / abs(float(total)),
           "sil_oracle": sums.get("sil_reproj", 0.0), "sil_hip": float(l[4])}
    for k in names:
        out["grad_%s_rel" % k] = rel(grads[k].cpu().numpy(), grads_o[k].numpy())
    return out


def case_config5_fit(family, M=2, S=512, window=2, iters=2, seed=47, z=1.9):
    """BASELINE config 5 as a fit, one shape family at a time (mixed families are independent fitters: replicas, no collective):
    512 x 512 silhouettes, limb scales on -- family 1 with the unity-style 26-dim prior and shared scales, every other family
    with its 20-dim SMAL cluster prior and per-frame (N,6) limb scales trained without a regulariser (reference
    smal_fitter.py:48-72, optimize_to_joints.py:108-109) -- `iters` stage-2 iterations of FusedFitter (losses + analytic
    gradients + Adam) against the oracle loop from the same start."""
    from smalify_amd import fitter as fit
    W = np.array(cfg.OPT_WEIGHTS).T
    stage = 2
    weights, w_temp, lr = W[stage][:6].copy(), float(W[stage][6]), float(W[stage][8])
    unity = family == 1
    # the stand-in's shape basis made mirror-symmetric like a real SMAL model's (round 5): every family's cluster mean then leaves
    # the template left / right balanced and all four families load through prepare_model, each with ITS template and ITS prior
    # (with the default basis families 2 and 3 stop where the reference stops, smal_basics.py:32-35)
    dd, data, sym = synthetic.synthetic_smal_dicts(seed=0, symmetric_basis=True)
    mdf = model_io.prepare_model(dd, data, sym, family)
    om = so.OracleModel(mdf)
    dm = eng.DeviceModel(mdf)
    e = eng.Engine(dm, M, S)
    pp = synthetic.synthetic_pose_prior()
    sp = synthetic.synthetic_shape_prior() if unity else model_io.family_shape_prior(data, family)
    e.set_pose_prior(*pp)
    e.set_shape_prior(*sp)
    gt = random_pose(M, seed, z=z)
    cur = random_pose(M, seed, z=z)
    rs = np.random.RandomState(seed + 7 + family)
    cur["global_rotation"] += (0.04 * rs.randn(M, 3)).astype(np.float32)
    cur["joint_rotations"] += (0.06 * rs.randn(M, 34, 3)).astype(np.float32)
    cur["trans"] += (0.02 * rs.randn(M, 3)).astype(np.float32)
    if not unity:
        cur["log_beta_scales"] = (0.1 * rs.randn(M, 6)).astype(np.float32)             # per-frame limb scales
        gt_ls = np.tile(gt["log_beta_scales"], (M, 1))
    else:
        gt_ls = np.tile(gt["log_beta_scales"], (M, 1))
    with torch.no_grad():
        theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
        vo, jo, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(gt["betas"], (M, 1))).double(), torch.from_numpy(theta).double(),
                                       torch.from_numpy(gt_ls).double())
        t = torch.from_numpy(gt["trans"]).double()[:, None]
        tj = so.project_points((jo + t)[:, so.CANONICAL], S).numpy() + rs.randn(M, 25, 2)
        tsil = (so.soft_silhouette(vo + t, om.faces, S) > 0.5).double().numpy()
    vis = (rs.rand(M, 25) < 0.85).astype(np.float32)
    prob = so.FitProblem(om, S, tj, vis, tsil, pp[0], pp[1], pp[2], sp[0], sp[1], window, use_unity_prior=unity)
    names = so.trainable_names(stage)
    params = {k: torch.from_numpy(v).double() for k, v in cur.items()}
    opt = so.Adam(so.PARAM_ORDER, lr=lr)
    hist_o = []
    for _ in range(iters):
        total, sums, grads = so.loss_and_grads(prob, params, weights, w_temp, names)
        opt.step(params, grads)
        hist_o.append(float(total))
    f = fit.FusedFitter(e, tj.astype(np.float32), vis, tsil.astype(np.float32), window, unity, cur["betas"],
                        cur["log_beta_scales"] if unity else None)
    for k in ("global_rotation", "joint_rotations", "trans") + (() if unity else ("log_beta_scales",)):
        f.p[k].copy_(dev(cur[k]).reshape(f.p[k].shape))
    f.begin_stage(stage)
    hist = []
    for _ in range(iters):
        f.step(weights, w_temp, lr, stage)
        hist.append(float(f.losses.double().sum().item()))
    out = {"status": e.status(), "family": family, "per_frame_scales": not unity, "sil_oracle": sums.get("sil_reproj", 0.0),
           "loss_rel_max": float(np.max(np.abs(np.array(hist) - np.array(hist_o)) / np.abs(np.array(hist_o))))}
    for k in ("betas", "log_beta_scales", "global_rotation", "joint_rotations", "trans"):
        out["param_%s_rel" % k] = rel(f.p[k].cpu().numpy().reshape(params[k].shape), params[k].numpy())
    return out
