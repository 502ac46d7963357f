"""Developer probe (GPU box): one complete 150/400/600/800 fit of the benchmark sequence on the crop-filling scene, nothing timed --
a workload for rocprofv3 (kernel stats, PMC passes: PMC_SCRIPT="tools/crop_fit.py" python tools/pmc_sq.py TAG) in the regime
most of a BADJA fit runs in.  usage: python tools/crop_fit.py [scene] [scale of the iteration counts]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from smalify_amd import engine as eng, synthetic, fitter as fit, config

scene = sys.argv[1] if len(sys.argv) > 1 else "crop"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
md = synthetic.synthetic_model(seed=0, shape_family_id=1)
e = eng.Engine(eng.DeviceModel(md), bench.NUM_FRAMES, bench.IMAGE_SIZE)
e.set_pose_prior(*synthetic.synthetic_pose_prior())
gt, tj, vis, tsil, sp = bench.build_problem(e, torch, scene)
e.set_shape_prior(*sp)
W = np.array(config.OPT_WEIGHTS).T
f = fit.FusedFitter(e, tj, vis, tsil, bench.WINDOW, True, sp[1][:20], sp[1][20:26])
for stage_id, its in enumerate(bench.SCHEDULE_ITERS):
    f.begin_stage(stage_id)
    f.run_iterations(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id, max(1, int(its * scale)))
torch.cuda.synchronize()
print("done, status", e.status(), "losses", f.losses.cpu().numpy().round(3).tolist())
