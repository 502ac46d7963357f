"""Pins the CPU oracle (oracle/smal_oracle.py) against vectors produced by the imported reference
(tests/golden/reference_golden.npz, generator: tests/golden/make_golden.py).  CPU only.

Tolerances: the reference computes in float32, the oracle in float64, so differences are the
reference's own round-off: 2e-5 relative on values, 2e-4 relative (norm-wise) on gradients that pass
through the 34-step chain and 3889-vertex reductions.
"""
import os

import numpy as np
import pytest
import torch

from oracle import smal_oracle as so
from smalify_amd import config as cfg
from smalify_amd import model_io


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def test_synthetic_model_matches_generation_time_checksums(golden, synth_model):
    md = synth_model
    assert abs(md.v_template.astype(np.float64).sum() - golden["chk_v_template"][0]) < 1e-3
    assert abs(np.abs(md.v_template.astype(np.float64)).sum() - golden["chk_v_template"][1]) < 1e-3
    assert abs(md.posedirs.astype(np.float64).sum() - golden["chk_posedirs"][0]) < 1e-2
    assert abs(np.abs(md.shapedirs.astype(np.float64)).sum() - golden["chk_shapedirs"][1]) < 1e-2
    assert abs((md.weights.astype(np.float64) * np.arange(35)).sum() - golden["chk_weights"][0]) < 1e-2
    assert abs((md.J_regressor.astype(np.float64) * np.arange(35)).sum() - golden["chk_jreg"][0]) < 1e-3
    assert (md.parents == golden["parents"]).all()


def test_known_answers_from_reference_data(golden):
    # SURVEY §8c known-answer values (depend only on files shipped with the reference)
    assert np.allclose(golden["init_global_rotation"], -1.20919958, atol=1e-6)
    assert np.allclose(model_io.initial_global_rotation(), -1.20919958, atol=1e-7)
    assert abs(golden["g4_zero_mean"] - 0.80639112) < 1e-5
    assert abs(golden["g4_point1_mean"] - 91.9112549) < 1e-3
    assert abs(golden["g5_zero_betas_mean"] - 5848.07471) < 0.5
    assert np.allclose(golden["g5_init_ls"],
                       [-0.12309442, 0.27784678, -0.6320069, -0.26995066, -0.0148344, -0.3365687], atol=1e-6)


def test_rodrigues(golden):
    R = so.rodrigues(torch.from_numpy(golden["g1_theta"]).double()).numpy()
    assert np.abs(R - golden["g1_R"]).max() < 5e-6


@pytest.mark.parametrize("scaled", [False, True])
def test_kinematic_chain(golden, scaled):
    theta = torch.from_numpy(golden["g2_theta"]).double()
    Rs = so.rodrigues(theta).reshape(3, 35, 3, 3)
    Js = torch.from_numpy(golden["g2_Js"]).double()
    ls = torch.from_numpy(golden["g2_ls"]).double() if scaled else None
    parents = [int(p) for p in golden["parents"]]
    jw, a_r, a_t = so.kinematic_chain(Rs, Js, parents, ls)
    tag = "scale" if scaled else "noscale"
    A = golden["g2_A_" + tag]
    assert rel(jw.numpy(), golden["g2_newJ_" + tag]) < 2e-5
    assert rel(a_r.numpy(), A[:, :, :3, :3]) < 2e-5
    assert rel(a_t.numpy(), A[:, :, :3, 3]) < 2e-5
    assert (A[:, :, 3, :] == np.array([0, 0, 0, 1.0])).all()   # reference keeps the homogeneous row (0,0,0,1)


def test_smal_forward_and_grads(golden, synth_model):
    m = so.OracleModel(synth_model)
    beta = torch.from_numpy(golden["g3_beta"]).double().requires_grad_(True)
    theta = torch.from_numpy(golden["g3_theta"]).double().requires_grad_(True)
    ls = torch.from_numpy(golden["g3_ls"]).double().requires_grad_(True)
    verts, joints, Rs, v_shaped = so.smal_forward(m, beta, theta, ls)
    vsel = golden["g3_vsel"]
    assert rel(verts.detach().numpy()[:, vsel], golden["g3_verts"]) < 2e-5
    assert rel(joints.detach().numpy(), golden["g3_joints"]) < 2e-5
    assert rel(v_shaped.detach().numpy()[:, vsel], golden["g3_vshaped"]) < 2e-5
    assert rel(Rs.detach().numpy(), golden["g3_Rs"]) < 2e-5
    func = (verts[:, vsel] * torch.from_numpy(golden["g3_wv"]).double()).sum() + \
           (joints * torch.from_numpy(golden["g3_wj"]).double()).sum()
    func.backward()
    assert abs(func.item() - golden["g3_func"]) < 2e-4 * abs(golden["g3_func"]) + 1e-4
    assert rel(beta.grad.numpy(), golden["g3_dbeta"]) < 2e-4
    assert rel(theta.grad.numpy(), golden["g3_dtheta"]) < 2e-4
    assert rel(ls.grad.numpy(), golden["g3_dls"]) < 2e-4


def test_smal_zero_pose_gradient(golden, synth_model):
    """theta = 0 for every joint (state at the start of stage 1): finite values and gradients."""
    m = so.OracleModel(synth_model)
    theta = torch.zeros(1, 35, 3, dtype=torch.float64, requires_grad=True)
    verts, joints, _, _ = so.smal_forward(m, torch.zeros(1, 20, dtype=torch.float64), theta)
    (joints * torch.from_numpy(golden["g3_wj"][:1]).double()).sum().backward()
    assert rel(joints.detach().numpy(), golden["g3z_joints"]) < 2e-5
    assert rel(verts.detach().numpy()[:, golden["g3_vsel"]], golden["g3z_verts"]) < 2e-5
    # float32 reference loses digits in theta/||theta+1e-8|| at 0; 1e-3 is its own noise floor here
    assert rel(theta.grad.numpy(), golden["g3z_dtheta"]) < 1e-3


def test_pose_prior(golden):
    P, mu, mask = (torch.from_numpy(golden[k]).double() for k in ("pose_prec", "pose_mean", "pose_mask"))
    x = torch.from_numpy(golden["g4_x"]).double().requires_grad_(True)
    val = so.pose_prior_residual2(x, P, mu, mask)
    val.mean().backward()
    assert rel(val.detach().numpy(), golden["g4_val"]) < 2e-5
    assert rel(x.grad.numpy(), golden["g4_dx"]) < 2e-5
    assert np.abs(x.grad.numpy()[:, 0]).max() == 0.0     # global rotation never penalised (App. D.6)


def _problem(golden, md, window, family1=True):
    m = so.OracleModel(md)
    S = int(golden["g6_image_size"])
    N = golden["g6_target_joints"].shape[0]
    if family1:
        sp, sm = golden["unity_prec"], golden["unity_mean"]
    else:
        sp, sm = golden["fam0_prec"], golden["fam0_mean"]
    return so.FitProblem(m, S, golden["g6_target_joints"], golden["g6_visibility"],
                         np.zeros((N, S, S)), golden["pose_prec"], golden["pose_mean"], golden["pose_mask"],
                         sp, sm, window, use_unity_prior=family1)


def _params(golden, tag):
    return {k: torch.from_numpy(golden["%s_p_%s" % (tag, k)]).double()
            for k in ("global_rotation", "joint_rotations", "trans", "betas", "log_beta_scales")}


CASES = [("g6_stage0_w4", 4, 0, True), ("g6_stage1_w4", 4, 1, True), ("g6_stage1_w2", 2, 1, True),
         ("g6_stage1_w3", 3, 1, True), ("g6_family0_w4", 4, 1, False)]


@pytest.mark.parametrize("tag,window,stage,family1", CASES)
def test_fitter_loss_terms_and_grads(golden, synth_model, synth_model_family0, tag, window, stage, family1):
    md = synth_model if family1 else synth_model_family0
    p = _problem(golden, md, window, family1)
    params = _params(golden, tag)
    weights = golden["g6_w0"] if stage == 0 else golden["g6_w1"]
    w_temp = float(golden["g6_wtemp"][stage])
    vis = so.stage0_visibility(p.vis) if stage == 0 else None
    trainable = ("global_rotation", "trans") if stage == 0 else so.PARAM_ORDER
    total, sums, grads = so.loss_and_grads(p, params, weights, w_temp, trainable, vis)
    assert abs(total.item() - golden[tag + "_total"]) < 3e-5 * abs(golden[tag + "_total"])
    for k in ("joint", "pose", "splay", "betas"):
        key = "%s_term_%s" % (tag, k)
        if key in golden:
            assert abs(sums[k] - golden[key]) < 3e-5 * abs(golden[key]) + 1e-6, k
    t = golden[tag + "_temporal"]
    assert np.allclose([sums["temp_joint"], sums["temp_global"], sums["temp_trans"]], t, rtol=3e-5, atol=1e-7)
    for k in trainable:
        ref = golden["%s_g_%s" % (tag, k)]
        assert rel(grads[k].numpy(), ref) < 3e-4, (k, rel(grads[k].numpy(), ref))


def test_adam_trajectory_matches_reference_loop(golden, synth_model):
    """20 iterations of the reference's stage loop (6 of stage 0, 14 of stage 1, window 2, no silhouette).
    Short horizon because the trajectory is chaotic."""
    p = _problem(golden, synth_model, 2)
    N = p.N
    params = dict(
        betas=torch.from_numpy(golden["g5_init_betas"]).double(),
        log_beta_scales=torch.from_numpy(golden["g5_init_ls"]).double(),
        global_rotation=torch.from_numpy(np.tile(golden["init_global_rotation"], (N, 1))).double(),
        trans=torch.zeros(N, 3, dtype=torch.float64),
        joint_rotations=torch.zeros(N, 34, 3, dtype=torch.float64))
    W = np.array(cfg.OPT_WEIGHTS).T
    hist = []
    for stage, its in golden["g8_schedule"]:
        weights = golden["g6_w0"] if stage == 0 else golden["g6_w1"]
        trainable = so.trainable_names(int(stage))
        opt = so.Adam(so.PARAM_ORDER, lr=float(W[stage][8]))
        vis = so.stage0_visibility(p.vis) if stage == 0 else None
        for _ in range(int(its)):
            total, _, grads = so.loss_and_grads(p, params, weights, float(W[stage][6]), trainable, vis)
            opt.step(params, grads)
            hist.append(total.item())
        for k in so.PARAM_ORDER:
            ref = golden["g8_after_stage%d_%s" % (stage, k)]
            assert rel(params[k].numpy(), ref) < 2e-4, (stage, k, rel(params[k].numpy(), ref))
    assert np.allclose(hist, golden["g8_loss_history"], rtol=2e-4)


def test_joint_limit_term_known_answer(synth_model):
    """w_limit hinge (reference smal_fitter.py:146-151): flat inside the limits, |x - limit| outside, mean over (B, 34, 3)"""
    import torch
    from oracle import smal_oracle as so
    from smalify_amd import model_io, synthetic
    lo, hi = model_io.joint_limit_table()
    om = so.OracleModel(synth_model)
    pp, sp = synthetic.synthetic_pose_prior(), synthetic.synthetic_shape_prior()
    B = 2
    prob = so.FitProblem(om, 32, np.zeros((B, 25, 2)), np.zeros((B, 25)), np.zeros((B, 32, 32)), pp[0], pp[1], pp[2], sp[0], sp[1],
                         B, True, joint_limits=(lo, hi))
    jr = torch.zeros(B, 34, 3, dtype=torch.float64)
    jr[0, 6, 0] = 0.30           # LLeg1 x: limits +-0.05 -> 0.25 over
    jr[1, 24, 1] = -1.9          # Tail1 y: lower limit -1.5 -> 0.4 under
    jr[1, 33, 2] = 7.0           # an ear: not in the reference's table -> unconstrained
    params = dict(betas=torch.zeros(20, dtype=torch.float64), log_beta_scales=torch.zeros(6, dtype=torch.float64),
                  global_rotation=torch.zeros(B, 3, dtype=torch.float64), joint_rotations=jr, trans=torch.zeros(B, 3, dtype=torch.float64))
    _, terms = so.window_loss(prob, params, [0, 1], (0, 0, 0, 0, 10.0, 0))
    assert abs(float(terms["limit"]) - 10.0 * (0.25 + 0.4) / (B * 102)) < 1e-8      # the table is float32


def _smal_options_golden():
    import os
    return dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden_smal_options.npz")))


@pytest.mark.parametrize("tag", ["delv", "vtmpl", "both", "rs"])
def test_smal_call_options_against_the_reference(synth_model, tag):
    """del_v, per-call v_template and rotation-matrix theta of SMAL.__call__ (smal_torch.py:99-133): the oracle against
    values and gradients produced by the reference's own SMAL (tests/golden/make_golden_smal_options.py)"""
    import torch
    from oracle import smal_oracle as so
    g = _smal_options_golden()
    om = so.OracleModel(synth_model)
    t = lambda a: torch.from_numpy(np.asarray(a)).double().requires_grad_(True)  # noqa: E731
    beta, ls = t(g["beta"]), t(g["ls"])
    theta = t(g["Rs"]) if tag == "rs" else t(g["theta"])
    del_v = t(g["del_v"]) if tag in ("delv", "both") else None
    v_tmpl = t(g["v_template"]) if tag in ("vtmpl", "both") else None
    verts, joints, _, v_shaped = so.smal_forward(om, beta, theta, ls, del_v=del_v, v_template=v_tmpl)
    vsel = g["vsel"]
    func = (verts[:, vsel] * torch.from_numpy(g["wv"]).double()).sum() + (joints * torch.from_numpy(g["wj"]).double()).sum()
    func.backward()
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))  # noqa: E731
    assert rel(verts.detach().numpy()[:, vsel], g[tag + "_verts"]) < 2e-6
    assert rel(joints.detach().numpy(), g[tag + "_joints"]) < 2e-6
    assert rel(v_shaped.detach().numpy()[:, vsel], g[tag + "_vshaped"]) < 2e-6
    assert rel(beta.grad.numpy(), g[tag + "_dbeta"]) < 2e-5 and rel(ls.grad.numpy(), g[tag + "_dls"]) < 2e-5
    assert rel(theta.grad.numpy(), g[tag + ("_dRs" if tag == "rs" else "_dtheta")]) < 2e-5
    if del_v is not None:
        assert rel(del_v.grad.numpy()[:, vsel], g[tag + "_ddel_v"]) < 2e-5
    if v_tmpl is not None:
        assert rel(v_tmpl.grad.numpy()[vsel], g[tag + "_dv_template"]) < 2e-5


@pytest.mark.parametrize("scaled", [False, True])
def test_kinematic_chain_autograd_against_the_reference(scaled):
    """values and gradients of the reference's batch_global_rigid_transformation (its own autograd) vs the oracle's chain"""
    import torch
    from oracle import smal_oracle as so
    g = _smal_options_golden()
    tag = "chain_scale" if scaled else "chain_noscale"
    R = torch.from_numpy(g["chain_Rs"]).double().requires_grad_(True)
    J = torch.from_numpy(g["chain_Js"]).double().requires_grad_(True)
    L = torch.from_numpy(g["chain_ls"]).double().requires_grad_(True) if scaled else None
    g_t, g_r, a_t = so.kinematic_chain(R, J, [int(p) for p in g["chain_parents"]], L)
    wa = torch.from_numpy(g["chain_wa"]).double()
    ((g_t * torch.from_numpy(g["chain_wn"]).double()).sum() + (g_r * wa[:, :, :3, :3]).sum() + (a_t * wa[:, :, :3, 3]).sum()).backward()
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))  # noqa: E731
    assert rel(g_t.detach().numpy(), g[tag + "_newJ"]) < 2e-6
    assert rel(g_r.detach().numpy(), g[tag + "_A"][:, :, :3, :3]) < 2e-6 and rel(a_t.detach().numpy(), g[tag + "_A"][:, :, :3, 3]) < 2e-6
    assert rel(R.grad.numpy(), g[tag + "_dRs"]) < 2e-5 and rel(J.grad.numpy(), g[tag + "_dJs"]) < 2e-5
    if scaled:
        assert rel(L.grad.numpy(), g[tag + "_dls"]) < 2e-5


def test_full_schedule_fixture_is_current():
    """tests/golden/oracle_full_schedule.npz caches the oracle's end state of the scaled four-stage fit (minutes of float64
    CPU work) for tests/test_gpu_parity.py::test_full_schedule.  It must belong to the problem the test builds today
    (same inputs, bit for bit) and to today's oracle: the head of the loop is replayed here and compared with the stored
    trace of totals."""
    from tests import parity_cases as pc
    from tests.golden import make_oracle_full_schedule as gen
    assert os.path.exists(pc.FULL_SCHEDULE_FIXTURE), "run tests/golden/make_oracle_full_schedule.py"
    z = np.load(pc.FULL_SCHEDULE_FIXTURE, allow_pickle=False)
    prob, cur, tg = pc.make_problem_cpu(gen.M, gen.S, gen.WINDOW, gen.SEED)
    assert pc.load_full_schedule_fixture(gen.M, gen.S, gen.WINDOW, gen.SCALE, gen.SEED, cur, tg) is not None, \
        "fixture belongs to other inputs: regenerate it"
    sched = pc.full_schedule_iterations(gen.SCALE)
    assert list(z["schedule"]) == sched and len(z["trace"]) == sum(sched)
    head = [sched[0], 2, 0, 0]               # all of stage 0 (keypoints only) and two silhouette iterations of stage 1
    trace = []
    pc.full_schedule_oracle(prob, cur, head, trace)
    np.testing.assert_allclose(trace, z["trace"][:len(trace)], rtol=1e-9)
    # the stored per-term sums are those of the last evaluation
    np.testing.assert_allclose(sum(float(z[k]) for k in z.files if k.startswith("sum_")), z["trace"][-1], rtol=1e-9)


@pytest.mark.parametrize("case_name", ["config2", "config1"])
def test_config2_fixtures_are_current(case_name):
    """(config1, round 5: BASELINE config 1 as worded -- one image, all stages -- tests/golden/oracle_config1_*.npz, same generator)
    tests/golden/oracle_config2_{f64,f32,heads}.npz (BASELINE config 2's shape: 8 frames, 256 x 256, WINDOW_SIZE 8, the full
    150/400/600/800 schedule; hours of oracle CPU time, tests/golden/make_oracle_config2.py) belong to the problem
    tests/config2_case.py builds today and to today's oracle: same inputs bit for bit, and the head of the float64 loop --
    three keypoint iterations of stage 0, then the first silhouette iteration from the stored stage-1 start -- reproduces the
    stored trace."""
    import torch
    from tests import config2_case as cases
    from oracle import smal_oracle as so
    from smalify_amd import config as cfg
    c2 = cases.CASES[case_name]
    f64 = c2.load_fixture("f64")
    assert f64 is not None and f64["complete"], "run tests/golden/make_oracle_config2.py f64"
    md, tg = c2.targets()
    start = c2.initial_params()
    fp = c2.fingerprint(tg, start)
    assert f64["fingerprint"] == fp, "the float64 fixture belongs to other inputs: regenerate it"
    assert c2.fingerprint(f64["targets"], start) == fp            # the targets that travel with the fixture are these
    assert f64["schedule"] == c2.SCHEDULE and f64["trace"].shape == (sum(c2.SCHEDULE), len(c2.TERMS))
    f32 = c2.load_fixture("f32")
    assert f32 is not None and f32["complete"] and f32["fingerprint"] == fp, "run tests/golden/make_oracle_config2.py f32"
    heads = np.load(c2.fixture_path("heads"), allow_pickle=False)
    assert str(heads["fingerprint"]) == fp and all("stage%d_f32_trace" % s in heads.files for s in range(4)), \
        "run tests/golden/make_oracle_config2.py heads"
    prob = c2.problem(md, tg, torch.float64)
    trace, stage_start, _ = c2.oracle_schedule(prob, start, torch.float64, schedule=(3, 0, 0, 0))
    np.testing.assert_allclose(trace, f64["trace"][:3], rtol=1e-9, atol=1e-12)
    W = np.array(cfg.OPT_WEIGHTS).T
    params = {k: torch.from_numpy(v) for k, v in f64["stage_start"][1].items()}
    total, sums, _ = so.loss_and_grads(prob, params, W[1][:6].copy(), float(W[1][6]), so.trainable_names(1))
    ref = f64["trace"][c2.SCHEDULE[0]]
    np.testing.assert_allclose([sums.get(k, 0.0) for k in c2.TERMS], ref, rtol=1e-9, atol=1e-12)
    # the float32 run is a float32 run: it leaves the float64 trace, but not by much in the first iterations
    assert 0.0 < abs(f32["trace"][0].sum() - f64["trace"][0].sum()) / f64["trace"][0].sum() < 1e-5


@pytest.mark.parametrize("case", ["crop8", "config3"])
def test_eval_fixtures_are_current(case):
    """tests/golden/oracle_eval_<case>.npz (float64 losses + gradients of one epoch objective at BASELINE's own sizes: the
    crop-filling scene at 8 x 256^2 and the headline workload at 64 x 256^2; tests/eval_cases.py, tests/golden/make_oracle_eval.py)
    belong to today's problem and today's oracle: the targets are reproduced bit for bit (crop8: the oracle renders them again here;
    config3: their fingerprint), every state the fixture holds is the state tests/eval_cases.py builds (the HIP-made states: the
    committed dump), and the float64 terms of one state are recomputed (crop8: `near_gt`, ~15 s; config3: the keypoint / prior /
    temporal terms of `initial`, which need no rasteriser)."""
    import torch
    from tests import eval_cases as ec
    from oracle import smal_oracle as so
    fx, tg = ec.load_fixture(case), ec.load_targets(case)
    assert fx is not None and tg is not None, "run tests/golden/make_oracle_eval.py"
    st = ec.states(case)
    assert set(fx["states"]) == set(ec.CASES[case]["states"]), "states missing: rerun tests/golden/make_oracle_eval.py eval %s" % case
    assert fx["fingerprint"] == ec.fingerprint(tg, st)
    for name, s_ in fx["states"].items():
        assert s_["stage"] == ec.STATE_STAGE[name]
        for k in ec.PARAMS:
            np.testing.assert_array_equal(s_["params"][k], st[name][k])
        assert "terms_f32" in s_ and set(s_["grads"]) == set(so.trainable_names(s_["stage"]))
    if case == "crop8":
        again = ec.make_targets(case)
        for k in tg:
            np.testing.assert_array_equal(again[k], tg[k])
        torch.set_num_threads(4)
        terms, grads = ec.oracle_eval(ec.problem(case, tg), st["near_gt"], ec.STATE_STAGE["near_gt"])
        np.testing.assert_allclose(terms, fx["states"]["near_gt"]["terms"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(grads["trans"], fx["states"]["near_gt"]["grads"]["trans"], rtol=1e-7, atol=1e-10)
    else:
        prob = ec.problem(case, tg)
        weights, w_temp, _ = ec.stage_weights(1)
        p = {k: torch.from_numpy(np.asarray(v)).double() for k, v in st["initial"].items()}
        _, sums = so.epoch_loss(prob, p, weights, w_temp, with_sil=False)
        ref = dict(zip(ec.TERMS, fx["states"]["initial"]["terms"]))
        for k, v in sums.items():
            np.testing.assert_allclose(float(v), ref[k], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("case", ["config3", "crop64"])
def test_benchmark_targets_belong_to_the_ground_truth_draw(case):
    """tests/golden/eval_targets_{config3,crop64}.npz are what bench.py fits (its two scenes).  They are data rendered offline by the
    float64 oracle; the ground truth, the keypoint noise and the visibility are re-drawn at run time.  Pin the files to today's draw:
    keypoints (oracle LBS + projection + the seeded noise) and visibility bit for bit, and the hard silhouette of one frame rendered
    again (bench.py repeats the check with the engine before it fits them)."""
    import torch
    from tests import eval_cases as ec
    from oracle import smal_oracle as so
    from smalify_amd import synthetic
    tg = ec.load_targets(case)
    assert tg is not None, "run tests/golden/make_oracle_eval.py targets %s" % case
    c = ec.CASES[case]
    N, S = c["frames"], c["image_size"]
    gt = ec.ground_truth(case)
    om = so.OracleModel(synthetic.synthetic_model(seed=0, shape_family_id=1))
    with torch.no_grad():
        theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
        vo, jo, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(gt["betas"], (N, 1))).double(), torch.from_numpy(theta).double(),
                                       torch.from_numpy(np.tile(gt["log_beta_scales"], (N, 1))).double())
        t = torch.from_numpy(gt["trans"]).double()[:, None]
        noise, vis = synthetic.keypoint_noise_and_visibility(N)
        tj = (so.project_points((jo + t)[:, so.CANONICAL], S).numpy() + noise).astype(np.float32)
        frame = N // 2
        sil = (so.soft_silhouette((vo + t)[frame:frame + 1], om.faces, S) > 0.5).numpy().astype(np.uint8)[0]
    np.testing.assert_array_equal(tj, tg["tj"])
    np.testing.assert_array_equal(vis.astype(np.float32), tg["vis"])
    np.testing.assert_array_equal(sil, tg["tsil"][frame])
