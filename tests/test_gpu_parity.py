"""GPU parity tests: the HIP engine (through the C-ABI of libsmalfit.so) against the float64 oracle and
against the reference's own golden outputs.  Run with `pytest -m gpu` on an MI355X.

Tolerances (float32 engine vs float64 oracle, stated per the north-star's 1e-4 rel-L2 goal):
  LBS values 2e-5 rel-L2, LBS / fitter gradients 5e-4 rel-L2, loss terms 1e-4 relative,
  silhouette: max abs 2e-3 and < 0.5 % of pixels off by more than 1e-4 (a pixel/face pair exactly at the
  blur cut-off or at the K-th depth flips a 1e-4-sized contribution), silhouette gradients 1e-2 rel-L2.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_cases as pc  # noqa: E402


def test_library_loaded_is_in_tree():
    from smalify_amd import _lib
    lib = _lib.load()
    assert lib.smalfit_version() >= 1
    assert torch.cuda.is_available()


def test_rodrigues():
    m = pc.case_rodrigues()
    assert m["rodrigues_fwd_maxabs"] < 2e-6
    assert m["rodrigues_bwd_rel"] < 2e-5
    assert m["rodrigues_bwd_zero_row_abs"] < 1e-5


def test_adam():
    assert pc.case_adam()["adam_rel"] < 1e-6


@pytest.mark.parametrize("M,dense,with_scale", [(3, False, True), (3, True, False), (8, False, True), (8, True, False), (21, False, True)])
def test_lbs_forward_backward(M, dense, with_scale):
    """M = 8 and 21 take the matrix-core pose-blend kernel (frames in tiles of 16: a half-empty tile, a ragged second tile);
    dense = the reference's dense (V,35) weight matrices, i.e. 35 skinning weights per vertex"""
    m = pc.case_lbs(M, dense, with_scale)
    for k in ("lbs_verts_rel", "lbs_joints_rel", "lbs_Rs_rel", "lbs_vshaped_rel"):
        assert m[k] < 2e-5, (k, m[k])
    for k in ("lbs_dbeta_rel", "lbs_dtheta_rel") + (("lbs_dlogscale_rel",) if with_scale else ()):
        assert m[k] < 5e-4, (k, m[k])


@pytest.mark.parametrize("tag,window,stage", [("g6_stage0_w4", 4, 0), ("g6_stage1_w4", 4, 1),
                                              ("g6_stage1_w2", 2, 1), ("g6_stage1_w3", 3, 1)])
def test_fitter_against_reference_golden(golden, tag, window, stage):
    """loss terms and gradients of the HIP fitter vs outputs of the imported reference (no silhouette)."""
    m = pc.case_fit_golden(golden, tag, window, stage)
    assert m["golden_total_rel"] < 1e-4, m
    for k, v in m.items():
        if k.startswith("golden_loss_") or k == "golden_temporal_rel":
            assert v < 1e-4, (k, v)
        if k.startswith("golden_grad_"):
            assert v < 5e-4, (k, v)


def test_fitter_non_unity_family_against_reference_golden(golden):
    """shape family 0: 20-dim SMAL cluster prior, per-frame (N,6) limb scales trained without a regulariser"""
    m = pc.case_fit_golden(golden, "g6_family0_w4", 4, 1, family1=False)
    assert m["golden_total_rel"] < 1e-4, m
    for k, v in m.items():
        if k.startswith("golden_grad_"):
            assert v < 5e-4, (k, v)


@pytest.mark.parametrize("M,S,z,seed", [(2, 64, 1.45, 11), (1, 64, 0.0, 13), (1, 128, 1.3, 17)])
def test_renderer(M, S, z, seed):
    m = pc.case_render(M, S, z, seed)
    assert m["render_status"] == 0
    assert m["sil_maxabs"] < 2e-3, m
    assert m["sil_frac_gt_1e-4"] < 5e-3, m
    assert m["render_proj_maxabs_px"] < 1e-3, m
    # z = 0 is the head-on start with up to ~870 candidates per pixel: which faces make the K = 100 cut is
    # decided by depth comparisons of near-coplanar grazing faces, float32 vs float64 flips a few of them.
    assert m["render_dverts_rel"] < (5e-2 if z == 0.0 else 1e-2), m


@pytest.mark.parametrize("stage,window", [(0, 2), (1, 2), (2, 3)])
def test_fitter_full(stage, window):
    m = pc.case_fit(4, 64, window, stage)
    assert m["fit_status"] == 0
    assert m["fit_total_rel"] < 1e-4, m
    for k, v in m.items():
        if k.startswith("fit_grad_") and k.endswith("_rel"):
            assert v < 2e-3, (k, v, m)


def test_raster_cache_never_changes_results():
    """The rasteriser re-proves its cached per-pixel depth bounds from counts at every evaluation.  Engine A keeps its
    cache while the pose drifts (stale bounds, several steps), engine B forgets it before every evaluation: losses and
    gradients must agree to float32 summation noise, for the side view and for the head-on K-overflow view."""
    for z in (1.45, 0.0):
        m = pc.case_cache_consistency(z=z)
        assert m["loss_rel_max"] < 2e-6, m
        assert m["grad_rel_max"] < 2e-5, m
        assert m["status"] == 0


@pytest.mark.parametrize("M,S,window,iters", [(4, 64, 2, 8), (8, 128, 4, 5)])
def test_fit_loop_with_silhouette_follows_the_oracle(M, S, window, iters):
    """8 iterations of the whole loop with the silhouette term on (cached depth bounds included): losses and parameters
    against the oracle's loss + autograd + Adam.  north_star's bar for parameters is 1e-4 relative L2."""
    m = pc.case_sil_trajectory(M, S, window, iters)
    assert m["traj_status"] == 0
    assert m["traj_loss_rel_max"] < 1e-4, m
    for k, v in m.items():
        if k.endswith("_rel") and k != "traj_loss_rel_max":
            assert v < 1e-4, (k, v, m)


@pytest.mark.parametrize("scaled", [False, True])
def test_global_rigid_transformation_against_reference_golden(golden, scaled):
    """the free-standing batch_global_rigid_transformation drop-in (smalfit_global_rigid_transformation) against the
    reference's own outputs (tests/golden, G2), with and without limb scales"""
    import numpy as np
    from smalify_amd.smal_model.batch_lbs import batch_global_rigid_transformation, batch_rodrigues
    theta = torch.from_numpy(golden["g2_theta"]).cuda()
    Rs = batch_rodrigues(theta).reshape(3, 35, 3, 3)
    Js = torch.from_numpy(golden["g2_Js"]).cuda()
    ls = torch.from_numpy(golden["g2_ls"]).cuda() if scaled else None
    new_J, A = batch_global_rigid_transformation(Rs, Js, golden["parents"], betas_logscale=ls)
    tag = "scale" if scaled else "noscale"
    assert pc.rel(new_J.cpu().numpy(), golden["g2_newJ_" + tag]) < 2e-6
    assert pc.rel(A.cpu().numpy(), golden["g2_A_" + tag]) < 2e-6
    assert (A.cpu().numpy()[:, :, 3, :] == np.array([0, 0, 0, 1.0], np.float32)).all()
    with pytest.raises(NotImplementedError):
        batch_global_rigid_transformation(Rs, Js, golden["parents"], rotate_base=True)


def test_unclamped_edge_t_option_follows_the_oracle_with_the_same_flag():
    """SURVEY App. B's switch on a whole mesh: with SMALFIT_OPT_UNCLAMPED_EDGE_T the analytic gradient of a silhouette-stage
    evaluation equals the oracle's with EDGE_T_UNCLAMPED (custom backward of the point-segment distance), the loss terms do not
    move, and the two conventions do differ on this problem (so the option is not a no-op)"""
    from oracle import smal_oracle as so
    m_exact = pc.case_fit(4, 64, 2, 2)
    e, _, _, _ = pc.make_problem(4, 64, 2, 21, with_sil=True)
    e.set_option(e.OPT_UNCLAMPED_EDGE_T, 1)
    so.EDGE_T_UNCLAMPED = True
    try:
        m = pc.case_fit(4, 64, 2, 2)
        so.EDGE_T_UNCLAMPED = False
        m_cross = pc.case_fit(4, 64, 2, 2)          # HIP unclamped against the EXACT oracle: must NOT agree
    finally:
        so.EDGE_T_UNCLAMPED = False
        e.set_option(e.OPT_UNCLAMPED_EDGE_T, 0)
    keys = [k for k in m if k.startswith("fit_grad_") and k.endswith("_rel")]
    print("\nunclamped-t option, gradient rel-L2 HIP vs oracle: " + ", ".join("%s exact/exact %.1e  unclamped/unclamped %.1e  unclamped/exact %.1e" %
                                                                            (k[9:-4], m_exact[k], m[k], m_cross[k]) for k in keys))
    assert m["fit_status"] == 0 and m["fit_total_rel"] < 1e-4
    # (looser than the exact adjoint's 2e-3: at a face's THIRD vertex the edges a-c and b-c tie only up to rounding -- |p - c|^2 formed
    # from c - a and from (b - a) + (c - b) -- so which of the two takes the gradient is decided by the last bit, in float32 here, in
    # float64 in the oracle and in pytorch3d alike; the exact adjoint does not care, the unclamped one does.  The convention is
    # ill-conditioned there by construction.)
    for k in keys:
        assert m[k] < 1e-2, (k, m[k], m_exact[k])
    assert max(m_cross[k] for k in keys) > 3.0 * max(m[k] for k in keys), (m, m_cross)


def test_full_schedule(capsys):
    """All four stages (scaled to 15 / 40 / 60 / 80 iterations) on 4 frames at 64 x 64, HIP loop (one library call per
    stage) vs the oracle loop: the final loss terms agree (SURVEY section 7 check iii) and the end-of-run relative L2
    difference of every parameter tensor is reported and bounded by the float32 oracle's own drift from its float64 run on
    the same problem (check iv)."""
    m = pc.case_full_schedule(M=4, S=64, window=2, iters_scale=0.1)
    with capsys.disabled():
        print("\nfull schedule %s: final loss hip %.6f oracle %.6f (rel %.2e)" % (m["schedule"], m["final_total_hip"],
                                                                                 m["final_total_oracle"], m["final_total_rel"]))
        print("  per term rel: " + ", ".join("%s %.1e" % (k[6:-4], v) for k, v in m.items() if k.startswith("final_") and k.endswith("_rel") and k != "final_total_rel"))
        print("  end-of-run parameter rel-L2: " + ", ".join("%s %.2e" % (k[6:-7], v) for k, v in m.items() if k.startswith("param_")))
    assert m["status"] == 0
    # Yardsticks: the ORACLE ITSELF in float32 against its float64 run on this very problem (tests/oracle_float32_drift.py ->
    # tests/golden/oracle_full_schedule_f32_drift.json; Adam turns a gradient component whose sign differs in the last float32 bit
    # into a +-lr step, so 195 float32 iterations part ways with float64 on the flat directions of the objective whoever computes
    # them).  A float32 trajectory is ONE sample of a chaotic map: the file holds several draws (thread counts, and -- round 6 -- the
    # float32 loop started from an initial translation moved by one unit in the last place), and every bound below is FACTOR x the
    # largest draw of the SAME quantity.
    import json
    import os
    FACTOR = 2.0
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_full_schedule_f32_drift.json")))
    assert doc["config"]["schedule"] == list(m["schedule"]), (doc["config"], m["schedule"])
    # final objective: 3 x the largest draw (its draws: 6.6e-5 ... 5.2e-4; HIP: 3.0e-4 in round 5, 1.05e-3 with one of round 6's
    # rejected kernel variants -- a different summation order in the backward gather is a different draw)
    yard_total = max(d["loss_rel"] for d in doc["draws"])
    assert m["final_total_rel"] < max(1e-3, 3.0 * yard_total), (m["final_total_rel"], yard_total)
    # final loss terms (round 6: each against the float32 oracle's deviation of THAT term -- until then all eight shared one absolute
    # bound, the sum of the oracle's term deviations, which was vacuous for the small terms): FACTOR x the largest draw, with a floor
    # of 1e-3 of the term / 2e-5 of the objective for terms the oracle's draws happen to hit to the last digit
    with_terms = [d for d in doc["draws"] if "terms_abs_dev" in d]
    assert len(with_terms) >= 4, "run tests/oracle_float32_drift.py 2 <draw>: the yardstick file needs per-term draws"
    names = ("joint", "pose", "splay", "betas", "sil_reproj", "temp_joint", "temp_global", "temp_trans")
    bound = {k: max(FACTOR * max(d["terms_abs_dev"][k] for d in with_terms), 1e-3 * abs(m["final_%s_oracle" % k]), 2e-5 * abs(m["final_total_oracle"]))
             for k in names}
    with capsys.disabled():
        print("  per term |HIP - f64 oracle| (bound = %g x the float32 oracle's largest draw of that term): " % FACTOR +
              ", ".join("%s %.3f (%.3f)" % (k, abs(m["final_%s_hip" % k] - m["final_%s_oracle" % k]), bound[k]) for k in names))
    for k in names:
        assert abs(m["final_%s_hip" % k] - m["final_%s_oracle" % k]) <= bound[k], (k, m, bound[k])
    for k, v in m.items():
        if k.startswith("param_"):
            name = k[6:-7]
            yard = max(d[name] for d in doc["draws"])
            assert v <= FACTOR * yard, (k, v, yard)


def test_joint_limit_term():
    """the reference's disabled w_limit hinge, built behind the weight row it already has"""
    m = pc.case_fit_limits()
    assert m["status"] == 0
    assert m["limit_oracle"] > 0.05, m                    # the perturbed pose does leave the limits
    assert abs(m["limit_hip"] - m["limit_oracle"]) < 1e-5 * m["limit_oracle"], m
    assert m["total_rel"] < 1e-5, m
    for k, v in m.items():
        if k.startswith("grad_"):
            assert v < 5e-4, (k, v)


def test_byte_resident_targets_are_bit_identical():
    """device-resident u8 silhouette targets give the float32 path's bits"""
    m = pc.case_u8_targets()
    assert m["dtype_f32"] == "torch.float32" and m["dtype_u8"] == "torch.uint8", m
    assert m["status"] == 0 and m["params_identical"] and m["losses_identical"], m


def test_rebound_mask_and_targets_get_fresh_argument_blocks():
    """the cached argument blocks hold raw device pointers; the plan key covers the address of every tensor a caller may rebind
    (round-3 advisor finding): after `fitter.rotation_mask = <new tensor>` / new targets the next run must see the new tensor,
    exactly as a fitter built with it from the start does"""
    import numpy as np
    from smalify_amd import fitter as fit
    W = np.array(pc.cfg.OPT_WEIGHTS).T
    e, prob, cur, tg = pc.make_problem(4, 64, 2, seed=33)

    def run(rebind):
        e.reset_raster_cache()                           # (which float-equivalent path sums a pixel depends on the cache: same start for all)
        f = fit.FusedFitter(e, tg["tj"], tg["vis"], tg["tsil"], 2, True, cur["betas"], cur["log_beta_scales"])
        for k in ("global_rotation", "joint_rotations", "trans"):
            f.p[k].copy_(pc.dev(cur[k]))
        mask = torch.ones(34, 3, device="cuda")
        mask[5:9] = 0.0
        tj2 = f.target_joints + 1.5
        if not rebind:                                   # the new tensors from the start
            f.rotation_mask, f.target_joints = mask, tj2
        f.begin_stage(1)
        f.run_iterations(W[1][:6], float(W[1][6]), float(W[1][8]), 1, 2)
        if rebind:                                       # ... or bound after the block of this stage was built and cached
            f.rotation_mask, f.target_joints = mask, tj2
        f.run_iterations(W[1][:6], float(W[1][6]), float(W[1][8]), 1, 3)
        return f.losses.cpu().numpy().copy(), f.flat.cpu().numpy().copy()

    # reference behaviour: the same 5 iterations with the first two on the OLD tensors, built by hand without any cache reuse
    l_rebound, p_rebound = run(True)
    e.reset_raster_cache()
    f = fit.FusedFitter(e, tg["tj"], tg["vis"], tg["tsil"], 2, True, cur["betas"], cur["log_beta_scales"])
    for k in ("global_rotation", "joint_rotations", "trans"):
        f.p[k].copy_(pc.dev(cur[k]))
    f.begin_stage(1)
    f.run_iterations(W[1][:6], float(W[1][6]), float(W[1][8]), 1, 2)
    f._plan = None                                        # forget every cached block
    mask = torch.ones(34, 3, device="cuda")
    mask[5:9] = 0.0
    f.rotation_mask, f.target_joints = mask, f.target_joints + 1.5
    f.run_iterations(W[1][:6], float(W[1][6]), float(W[1][8]), 1, 3)
    assert np.array_equal(p_rebound, f.flat.cpu().numpy()) and np.array_equal(l_rebound, f.losses.cpu().numpy())
    # and the rebinding did change the fit (the test would be vacuous otherwise)
    l_new, p_new = run(False)
    assert not np.array_equal(p_rebound, p_new)
    assert e.status() == 0


def test_fused_fitter_keeps_its_joint_limits_on_a_shared_engine():
    """the joint-limit table is engine state: a fitter that opted in re-asserts its table when another fitter replaced or cleared it
    (round-3 advisor finding: the w_limit term used to drop out of the objective silently)"""
    import numpy as np
    from smalify_amd import fitter as fit, model_io
    W = np.array(pc.cfg.OPT_WEIGHTS).T
    e, prob, cur, tg = pc.make_problem(3, 64, 2, seed=35, with_sil=False)
    w = W[1][:6].copy()
    w[1] = 0.0
    f = fit.FusedFitter(e, tg["tj"], tg["vis"], tg["tsil"], 2, True, cur["betas"], cur["log_beta_scales"])
    f.p["joint_rotations"].copy_(pc.dev((0.6 * np.random.RandomState(3).randn(3, 34, 3)).astype(np.float32)))   # well outside the limits
    f.enable_joint_limits()
    f.begin_stage(1)
    f.evaluate(w, float(W[1][6]), 1)
    with_limits = float(f.losses[8])
    assert with_limits > 0.0
    e.clear_joint_limits()                                # what a limit-free SMALFitter on the same engine does
    f.evaluate(w, float(W[1][6]), 1)
    assert float(f.losses[8]) == with_limits
    lo, hi = model_io.joint_limit_table()
    e.set_joint_limits(lo - 10.0, hi + 10.0, owner="somebody else")    # ... or another fitter's (much wider) table
    f.run_iterations(w, float(W[1][6]), 0.0, 1, 1)                     # lr 0: the state does not move
    assert float(f.losses[8]) == with_limits
    assert e.status() == 0


def test_silhouette_loss_with_and_without_the_image_output():
    """raster_resolve_kernel sums the targets of a workgroup wholly outside the frame's active region in one wave when no silhouette
    image is asked for, and per thread otherwise (round-3 advisor finding): the reported sil_reproj loss may differ in its last
    bits between the two, never by more than float32 summation noise; gradients are identical"""
    import numpy as np
    W = np.array(pc.cfg.OPT_WEIGHTS).T
    e, prob, cur, tg = pc.make_problem(4, 64, 2, seed=37)
    d = {k: pc.dev(v) for k, v in cur.items()}
    out = {}
    for tag, sil_out in (("plain", None), ("image", torch.empty(4, 64, 64, device="cuda"))):
        e.reset_raster_cache()
        losses, grads = e.fit_eval(betas=d["betas"], log_beta_scales=d["log_beta_scales"], global_rotation=d["global_rotation"],
                                   joint_rotations=d["joint_rotations"], trans=d["trans"], target_joints=pc.dev(tg["tj"]),
                                   target_visibility=pc.dev(tg["vis"]), target_sil=pc.dev(tg["tsil"]), weights=W[2][:6].copy(),
                                   w_temp=float(W[2][6]), window=2, sil_out=sil_out)
        out[tag] = (losses.cpu().numpy().copy(), {k: v.cpu().numpy().copy() for k, v in grads.items()})
    a, b = out["plain"][0][4], out["image"][0][4]
    assert abs(a - b) <= 2e-6 * abs(b), (a, b)
    for k in out["plain"][1]:
        assert np.array_equal(out["plain"][1][k], out["image"][1][k]), k
    assert e.status() == 0
