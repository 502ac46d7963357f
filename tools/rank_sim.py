"""Developer probe: what one rank of an N-GPU run costs without the collective -- the bench loop on the first 64/N
frames of the benchmark sequence (GPU box).  usage: python tools/rank_sim.py N [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from smalify_amd import engine as eng, synthetic, fitter as fit, config

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 390
md = synthetic.synthetic_model(seed=0, shape_family_id=1)
dm = eng.DeviceModel(md)
full = eng.Engine(dm, bench.NUM_FRAMES, bench.IMAGE_SIZE)
gt, tj, vis, tsil, sp = bench.build_problem(full, torch, "survey")
del full
torch.cuda.empty_cache()
per = bench.NUM_FRAMES // n
e = eng.Engine(dm, per, bench.IMAGE_SIZE)
e.set_pose_prior(*synthetic.synthetic_pose_prior())
e.set_shape_prior(*sp)
W = np.array(config.OPT_WEIGHTS).T
use_graph = os.environ.get("RANK_SIM_GRAPH") == "1"       # replay one captured iteration per step (needs a non-default stream)
if use_graph:
    e.set_graph(True)
    torch.cuda.set_stream(torch.cuda.Stream())
for rep in range(2):
    f = fit.FusedFitter(e, tj[:per], vis[:per], tsil[:per], bench.WINDOW, True, sp[1][:20], sp[1][20:26])
    sched = bench.scaled_schedule(steps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for stage_id, its in enumerate(sched):
        f.begin_stage(stage_id)
        f.run_iterations(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id, its)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("trans checksum %.9e betas checksum %.9e" % (float(f.p["trans"].double().sum()), float(f.p["betas"].double().sum())))
print(("graph " if use_graph else "") + "frames/rank %d: %.3f ms/step (host issue time %.3f ms/step) -> %.0f it/s if ranks were free of communication" %
      (per, dt / steps * 1e3, t_host / steps * 1e3, steps / dt))
