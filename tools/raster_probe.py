"""Developer probe: time smalfit_fit_eval sections under different debug flags / scenes (GPU box).
Needs a developer build of the library (the product compiles the probes out):
    tools/build_variant.sh probes -DSMALFIT_DEV_PROBES && SMALFIT_LIB=$PWD/smalify_amd/_variants/probes.so python tools/raster_probe.py ..."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from smalify_amd import _lib, engine as eng, synthetic, fitter as fit, config

def main():
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    dm = eng.DeviceModel(md)
    e = eng.Engine(dm, 64, 256)
    e.set_pose_prior(*synthetic.synthetic_pose_prior())
    lib = _lib.load()
    W = np.array(config.OPT_WEIGHTS).T
    for scene in os.environ.get("PROBE_SCENES", "survey,crop").split(","):
        gt, tj, vis, tsil, sp = bench.build_problem(e, torch, scene)
        e.set_shape_prior(*sp)
        f = fit.FusedFitter(e, tj, vis, tsil, 8, True, sp[1][:20], sp[1][20:26])
        t = lambda a: torch.as_tensor(a, device="cuda", dtype=torch.float32)
        f.p["global_rotation"].copy_(t(gt["global_rotation"])); f.p["joint_rotations"].copy_(t(gt["joint_rotations"]) + 0.02)
        f.p["trans"].copy_(t(gt["trans"]) + 0.01)
        for flags in [int(x) for x in os.environ.get("PROBE_FLAGS", "0,1,3").split(",")]:
            lib.smalfit_debug_set(ctypes.c_int(flags))
            for _ in range(3):
                f.evaluate(W[2][:6], float(W[2][6]), 2)
            e.profile_begin(20)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                if os.environ.get("PROBE_COLD") == "1":        # every evaluation selects every bounded pixel from scratch
                    e.reset_raster_cache()
                f.evaluate(W[2][:6], float(W[2][6]), 2)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            sec = e.profile_end()
            print(scene, "flags", flags, "eval ms %.3f" % (dt * 1e3), {k: round(v[0] / max(v[1], 1), 4) for k, v in sec.items()})
        lib.smalfit_debug_set(ctypes.c_int(0))

if __name__ == "__main__":
    main()
