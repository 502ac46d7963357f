#!/usr/bin/env python3
"""GPU side of the evaluation fixtures (tests/eval_cases.py): runs the HIP fit of a case from the reference's initial state
through the reference's 150/400/600/800 schedule on the case's oracle-rendered targets (tests/golden/eval_targets_<case>.npz) and
dumps the parameters it holds after stage 1 (`hip_stage1`) and at the end (`hip_final`): the states a real fit passes
through, which the float64 oracle then evaluates offline (tests/golden/make_oracle_eval.py eval <case>).

    python tools/dump_fit_states.py <case> [out.npz]       (default gpurun_out/hip_states_<case>.npz; copy to tests/golden/)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from smalify_amd import config as cfg, engine as eng, fitter as fit, synthetic
    from tests import eval_cases as ec
    case = sys.argv[1]
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "hip_states_%s.npz" % case)
    c = ec.CASES[case]
    tg = ec.load_targets(case)
    assert tg is not None, "tests/golden/eval_targets_%s.npz missing" % case
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    e = eng.Engine(eng.DeviceModel(md), c["frames"], c["image_size"])
    e.set_pose_prior(*synthetic.synthetic_pose_prior())
    sp = synthetic.synthetic_shape_prior()
    e.set_shape_prior(*sp)
    start = ec.initial_params(case)
    f = fit.FusedFitter(e, tg["tj"], tg["vis"], tg["tsil"].astype(np.float32), c["window"], True, start["betas"], start["log_beta_scales"])
    W = np.array(cfg.OPT_WEIGHTS).T
    out = {}
    after = {v: k for k, v in ec.HIP_STATE_AFTER.items() if k in c["states"]}
    for stage in range(4):
        f.begin_stage(stage)
        f.run_iterations(W[stage][:6], float(W[stage][6]), float(W[stage][8]), stage, int(W[stage][7]))
        torch.cuda.synchronize()
        print("stage", stage, "losses", f.losses.cpu().numpy()[:8].tolist(), "status", e.status(), flush=True)
        if stage + 1 in after:
            for k in ec.PARAMS:
                out["%s_%s" % (after[stage + 1], k)] = f.p[k].cpu().numpy().copy()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, sorted(out))


if __name__ == "__main__":
    main()
