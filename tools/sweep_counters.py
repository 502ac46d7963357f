"""Developer probe: how much of the two face sweeps' lane work is useful, per stage (GPU box).
    tools/build_variant.sh work -DSMALFIT_DEV_PROBES -DSMALFIT_WORK_STATS
    SMALFIT_LIB=$PWD/smalify_amd/_variants/work.so python tools/sweep_counters.py [steps] [scene]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from smalify_amd import _lib, engine as eng, synthetic, fitter as fit, config


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 195
    scene = sys.argv[2] if len(sys.argv) > 2 else "survey"
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    e = eng.Engine(eng.DeviceModel(md), 64, 256)
    e.set_pose_prior(*synthetic.synthetic_pose_prior())
    lib = _lib.load()
    lib.smalfit_debug_work_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    gt, tj, vis, tsil, sp = bench.build_problem(e, torch, scene)
    e.set_shape_prior(*sp)
    W = np.array(config.OPT_WEIGHTS).T
    sched = bench.scaled_schedule(steps)
    out = (ctypes.c_ulonglong * 32)()
    f = fit.FusedFitter(e, tj, vis, tsil, 8, True, sp[1][:20], sp[1][20:26])
    for stage_id, its in enumerate(sched):
        f.begin_stage(stage_id)
        lib.smalfit_debug_work_stats(out)
        for _ in range(its):
            f.step(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id)
        lib.smalfit_debug_work_stats(out)
        o = [float(x) / its for x in out]
        if o[0] == 0:
            continue
        print("stage %d (%d its), per iteration:" % (stage_id, its))
        print("  sweep: pairs %.2fM  pass-depth %.3f  pass-eval %.3f  near %.3f | wave-rounds %.0fk  lane eff %.3f | flush atomics %.2fM  blocks %.0f"
              % (o[0] / 1e6, o[1] / o[0], o[2] / o[0], o[3] / o[0], o[4] / 1e3, o[0] / max(o[4] * 64, 1), o[5] / 1e6, o[6] / 4))
        print("  sweep: near pairs outside the LDS window (one global atomic each) %.2fM = %.3f of the near pairs; workgroups whose rectangle exceeds the window %.3f"
              % (o[7] / 1e6, o[7] / max(o[3], 1), o[31] / max(o[6], 1)))
        print("  bwd:   pairs %.2fM  live-seed %.3f  pass-eval %.3f  in-K %.3f | wave-rounds(max trips) %.0fk lane eff %.3f | rounds %.0fk dead rounds %.3f | faces with a live pixel %.3f"
              % (o[8] / 1e6, o[9] / o[8], o[10] / o[8], o[11] / o[8], o[12] / 1e3, o[8] / max(o[12] * 64, 1), o[13] / 1e3,
                 o[14] / max(o[13], 1), o[15] / max(o[16], 1)), flush=True)
        print("  bwd faces: with a list %.0fk, box walk because the box exceeds 256 px %.0fk, box walk because the list overflowed / the selection flagged it %.0fk (their boxes hold %.2fM pixels)"
              % (o[21] / 1e3, o[22] / 1e3, o[23] / 1e3, o[24] / 1e6))
        print("  bwd in-K pairs by pixel alpha: <2^-12 %.3f  <2^-24 %.3f  <2^-40 %.3f | faces whose every in-K pixel has alpha <2^-24: %.3f  <2^-12: %.3f"
              % (o[17] / max(o[11], 1), o[18] / max(o[11], 1), o[19] / max(o[11], 1), o[29] / max(o[16], 1), o[30] / max(o[16], 1)))
        nw = max(o[6], 1)
        print("  bwd wave-cycles (avg per wave): rec load %.0f  pixel loop %.0f  reduce+store %.0f  (waves %.0f)"
              % (o[25] / max(o[28], 1), o[26] / max(o[28], 1), o[27] / max(o[28], 1), o[28]), flush=True)


if __name__ == "__main__":
    main()
