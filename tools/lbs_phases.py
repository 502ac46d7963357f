"""Developer probe (GPU box): where the latency-chain kernels of an iteration spend their cycles, per phase and frame count.
    tools/build_variant.sh phases -DSMALFIT_DEV_PROBES -DSMALFIT_PHASES
    SMALFIT_LIB=$PWD/smalify_amd/_variants/phases.so python tools/lbs_phases.py [steps]
Thread 0 of every workgroup records s_memtime between marks (shader cycles; ~11 % overhead on the marked waves).  Printed per kernel and
frame count: workgroups per launch, average cycles of a workgroup and of each phase (in us at 2.4 GHz in brackets), the longest workgroup."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from smalify_amd import _lib, engine as eng, synthetic, fitter as fit, config

KERNELS = [("lbs_head: pose block (per frame)", ["operands -> LDS", "Rodrigues", "tree walk", "A / G out"]),
           ("lbs_head: shape-blend block", []),
           ("skin_mfma (64 vertices x 16 frames)", ["side operands", "pose-blend GEMM", "skinning + camera"]),
           ("vertex_bwd (256 vertices x 1 frame)", []),
           ("lbs_bwd_mid: pose-blend adjoint block", []),
           ("lbs_bwd_mid: dA block (joint, frame)", []),
           ("chain_bwd: frame block", ["operands -> LDS", "dG / dJ set-up", "tree walk", "root, joint / scale adjoints", "Rodrigues adjoint + tails"]),
           ("chain_bwd: shape-blend adjoint rider", []),
           ("assemble: shape-gradient block", []),
           ("assemble: other role blocks", []),
           ("assemble: last block's tail", []),
           ("adam_segments (per workgroup)", [])]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 195
    lib = _lib.load()
    lib.smalfit_debug_phase_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    dm = eng.DeviceModel(md)
    full = eng.Engine(dm, bench.NUM_FRAMES, bench.IMAGE_SIZE)
    gt, tj, vis, tsil, sp = bench.build_problem(full, torch, "survey")
    W = np.array(config.OPT_WEIGHTS).T
    out = (ctypes.c_ulonglong * 192)()
    for per in (8, 64):
        e = full if per == bench.NUM_FRAMES else eng.Engine(dm, per, bench.IMAGE_SIZE)
        e.set_pose_prior(*synthetic.synthetic_pose_prior())
        e.set_shape_prior(*sp)
        f = fit.FusedFitter(e, tj[:per], vis[:per], tsil[:per], bench.WINDOW, True, sp[1][:20], sp[1][20:26])
        sched = bench.scaled_schedule(steps)
        lib.smalfit_debug_phase_stats(out)
        for stage_id, its in enumerate(sched):
            f.begin_stage(stage_id)
            f.run_iterations(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id, its)
        lib.smalfit_debug_phase_stats(out)
        o = np.array(list(out), dtype=np.float64).reshape(12, 16)
        print("== %d frames, %d iterations (schedule %s)" % (per, steps, sched))
        for k, (name, phases) in enumerate(KERNELS):
            n = o[k, 15]
            if n == 0:
                continue
            us = lambda c: c / 2400.0  # noqa: E731
            line = "  %-44s %6.1f workgroups / iteration, average %7.0f cycles (%5.2f us), longest %7.0f (%5.2f us)" % (
                name, n / steps, o[k, 13] / n, us(o[k, 13] / n), o[k, 14], us(o[k, 14]))
            print(line)
            for i, ph in enumerate(phases):
                print("      %-34s %7.0f cycles (%5.2f us)" % (ph, o[k, i] / n, us(o[k, i] / n)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
