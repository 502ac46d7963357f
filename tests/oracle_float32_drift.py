"""How far does float32 arithmetic alone carry a fit from its float64 twin?  The ORACLE's loop (loss + autograd + Adam, all
four stages at iters_scale 0.1 on the 4-frame 64x64 problem of tests/parity_cases.py::case_full_schedule) run once in
float64 and once in float32, CPU only:  python tests/oracle_float32_drift.py   (a few minutes).
Round-2 result (8 threads): final objective 6.6e-5 relative; parameters rel-L2 betas 1.5e-3, limb scales 3.4e-4, global
rotation 2.9e-3, joint rotations 4.9e-2, translation 1.1e-3 -- the yardstick for test_full_schedule's end-of-run numbers,
written to tests/golden/oracle_full_schedule_f32_drift.json."""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
from oracle import smal_oracle as so
from smalify_amd import config as cfg, synthetic, model_io
def random_pose(M, seed, scale=1.0, z=1.45):
    rs = np.random.RandomState(seed)
    init = model_io.initial_global_rotation()
    return dict(betas=(0.4 * rs.randn(20)).astype(np.float32), log_beta_scales=(0.15 * rs.randn(6)).astype(np.float32),
        global_rotation=(init[None] + 0.25 * scale * rs.randn(M, 3)).astype(np.float32),
        joint_rotations=(0.2 * scale * rs.randn(M, 34, 3)).astype(np.float32),
        trans=(np.array([0.03, -0.02, z])[None] + 0.03 * rs.randn(M, 3)).astype(np.float32))
M, S, window, seed = 4, 64, 2, 21
md = synthetic.synthetic_model(seed=0, shape_family_id=1)
pp, sp = synthetic.synthetic_pose_prior(), synthetic.synthetic_shape_prior()
gt = random_pose(M, seed); cur = random_pose(M, seed)
rs = np.random.RandomState(seed + 7)
cur["global_rotation"] += (0.05 * rs.randn(M, 3)).astype(np.float32)
cur["joint_rotations"] += (0.08 * rs.randn(M, 34, 3)).astype(np.float32)
cur["trans"] += (0.02 * rs.randn(M, 3)).astype(np.float32)
cur["betas"] += (0.1 * rs.randn(20)).astype(np.float32)
# optional second argument: a DRAW index d >= 1 -- the float32 loop starts from an initial translation moved by one unit in the last
# place (frame 0; axis d % 3, direction by d // 3), the float64 loop from the unperturbed one: float32 arithmetic's own spread on this
# problem (a 195-step Adam trajectory is one sample of a chaotic map, like tests/config2_case.py's f32b...e draws)
DRAW = int(sys.argv[2]) if len(sys.argv) > 2 else 0
om64 = so.OracleModel(md)
with torch.no_grad():
    theta = np.concatenate([gt["global_rotation"][:, None], gt["joint_rotations"]], 1)
    vo, jo, _, _ = so.smal_forward(om64, torch.from_numpy(np.tile(gt["betas"], (M, 1))).double(), torch.from_numpy(theta).double(),
                                   torch.from_numpy(np.tile(gt["log_beta_scales"], (M, 1))).double())
    t = torch.from_numpy(gt["trans"]).double()[:, None]
    tj = so.project_points((jo + t)[:, so.CANONICAL], S).numpy() + rs.randn(M, 25, 2)
    tsil = (so.soft_silhouette(vo + t, om64.faces, S) > 0.5).double().numpy()
vis = (rs.rand(M, 25) < 0.85).astype(np.float32); vis[:, [2, 5, 8]] = 1.0
W = np.array(cfg.OPT_WEIGHTS).T
sched = [max(1, int(round(int(w[7]) * 0.1))) for w in W]
res = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    om = so.OracleModel(md, dtype=dt)
    prob = so.FitProblem(om, S, tj, vis, tsil, pp[0], pp[1], pp[2], sp[0], sp[1], window, True, dtype=dt)
    params = {k: torch.from_numpy(v).to(dt) for k, v in cur.items()}
    if DRAW and dt == torch.float32:
        t0 = params["trans"].numpy().copy()
        ax = DRAW % 3
        t0[0, ax] = np.nextafter(t0[0, ax], np.float32(np.inf if (DRAW // 3) % 2 == 0 else -np.inf))
        params["trans"] = torch.from_numpy(t0)
    for stage, w in enumerate(W):
        names = so.trainable_names(stage)
        v0 = so.stage0_visibility(prob.vis) if stage == 0 else None
        opt = so.Adam(so.PARAM_ORDER, lr=float(w[8]))
        for _ in range(sched[stage]):
            total, sums, grads = so.loss_and_grads(prob, params, w[:6].copy(), float(w[6]), names, visibility=v0)
            opt.step(params, grads)
    res[name] = ({k: v.double().numpy() for k, v in params.items()}, float(total), dict(sums))
    print(name, "final total", float(total), flush=True)
drift = {}
for k in res["f64"][0]:
    a, b = res["f32"][0][k], res["f64"][0][k]
    drift[k] = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print("oracle f32 vs f64 param rel-L2", k, "%.2e" % drift[k])
drift["loss_rel"] = abs(res["f32"][1] - res["f64"][1]) / abs(res["f64"][1])
print("loss rel %.2e" % drift["loss_rel"])
# per-term final losses of both runs (round 5): the yardstick of test_full_schedule's per-term check -- every term of the HIP fit within
# FACTOR x the SUM of the float32 oracle's |term deviations| (one term's own deviation is a single heavy-tailed draw)
drift["terms_f64"] = {k: float(v) for k, v in res["f64"][2].items()}
drift["terms_abs_dev"] = {k: abs(float(res["f32"][2][k]) - float(v)) for k, v in res["f64"][2].items()}
print("per-term |f32 - f64|:", drift["terms_abs_dev"])
# the yardstick tests/test_gpu_parity.py::test_full_schedule reads (ORACLE output, like oracle_full_schedule.npz)
import json, os
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_full_schedule_f32_drift.json")
doc = json.load(open(out)) if os.path.exists(out) else {"config": dict(M=M, S=S, window=window, seed=seed, iters_scale=0.1, schedule=sched), "draws": []}
doc["draws"].append(dict({"source": "tests/oracle_float32_drift.py, %d threads%s" % (torch.get_num_threads(), (", initial trans[0, %d] moved by one ulp (draw %d)" % (DRAW % 3, DRAW)) if DRAW else "")}, **drift))
json.dump(doc, open(out, "w"), indent=1)
print("appended a draw to", out)
