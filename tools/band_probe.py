"""Developer probe: per-stage hit rates of the rasteriser's depth-bound cache and section times (GPU box).
usage: python tools/band_probe.py [steps] [scene]
Needs a developer build of the library (the product compiles the probes out):
    tools/build_variant.sh probes -DSMALFIT_DEV_PROBES && SMALFIT_LIB=$PWD/smalify_amd/_variants/probes.so python tools/band_probe.py ..."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from smalify_amd import _lib, engine as eng, synthetic, fitter as fit, config


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 390
    scene = sys.argv[2] if len(sys.argv) > 2 else "survey"
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    e = eng.Engine(eng.DeviceModel(md), 64, 256)
    e.set_pose_prior(*synthetic.synthetic_pose_prior())
    lib = _lib.load()
    lib.smalfit_debug_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    gt, tj, vis, tsil, sp = bench.build_problem(e, torch, scene)
    e.set_shape_prior(*sp)
    W = np.array(config.OPT_WEIGHTS).T
    sched = bench.scaled_schedule(steps)
    out = (ctypes.c_int * 8)()
    for rep in range(2):                       # rep 0 = warm-up
        f = fit.FusedFitter(e, tj, vis, tsil, 8, True, sp[1][:20], sp[1][20:26])
        for stage_id, its in enumerate(sched):
            f.begin_stage(stage_id)
            lib.smalfit_debug_set(ctypes.c_int(8 if os.environ.get('PROBE_STATS', '1') == '1' else 0))
            lib.smalfit_debug_stats(e.handle, out)
            e.profile_begin(its, int(os.environ.get('PROBE_STRIDE', '1')))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(its):
                f.step(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / its
            sec = e.profile_end()
            lib.smalfit_debug_stats(e.handle, out)
            lib.smalfit_debug_set(ctypes.c_int(0))
            if rep:
                bounded, hit, bsum = out[1], out[2], out[3]
                print("stage %d its %4d  ms/it %.3f  bounded px/it %8.0f  hit %.3f  mean band %.1f  " %
                      (stage_id, its, dt * 1e3, bounded / its, hit / max(bounded, 1), bsum / max(hit, 1)),
                      "miss c>K %.3f band-full %.3f short %.3f" % tuple(out[i] / max(bounded, 1) for i in (4, 5, 6)),
                      {k: round(v[0] / max(v[1], 1), 4) for k, v in sec.items()}, flush=True)


if __name__ == "__main__":
    main()
