#!/usr/bin/env python3
"""Developer measurement (GPU box) of the 3D mesh-fitting loop: iterations/s of Stage.step for a
batch of N target meshes, "default" scheme (trainer.py:115), all four loss terms, 3000 target points per mesh, and the
share of each of the five C-ABI calls of a step (torch events on the launch stream, separate short run).

    python tools/fit3d_bench.py [--meshes 1 8 32] [--iters 300]

Prints one JSON line per batch size.  Synthetic SMAL-topology model and targets (posed copies of it), as bench.py."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def targets_for(md, N, seed):
    """posed copies of the template: cheap stand-ins with the SMAL face list (no oracle involved)"""
    rs = np.random.RandomState(seed)
    base = np.asarray(md.v_template, np.float32)
    out = []
    for _ in range(N):
        a = 0.3 * rs.randn()
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        v = (base * (1.0 + 0.1 * rs.randn(3))) @ R.T + 0.01 * rs.randn(*base.shape)
        v = v - v.mean(0)
        out.append((v / np.abs(v).max()).astype(np.float32))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", type=int, nargs="+", default=[1, 8, 32])
    ap.add_argument("--iters", type=int, default=300)
    args = ap.parse_args()
    import torch
    from smalify_amd import synthetic
    from smalify_amd.fitter_3d import SMAL3DFitter, Stage, TargetMeshes

    md = synthetic.synthetic_model(seed=0, shape_family_id=-1)
    smal_data = synthetic.synthetic_smal_dicts(seed=0)[1]
    for N in args.meshes:
        fit = SMAL3DFitter(batch_size=N, shape_family=-1, model_data=md, smal_data=smal_data)
        tv = targets_for(md, N, seed=N)
        targets = TargetMeshes(tv, [np.asarray(md.faces)] * N)
        stage = Stage(args.iters, "default", fit, targets, lr=0.01, custom_lrs={"joint_rot": 0.005})
        for i in range(10):
            stage.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.iters):
            stage.step(10 + i)
        t_issue = time.perf_counter() - t0          # host time to enqueue everything (the GPU runs behind)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # share of the calls of one step
        e = fit._engine()
        betas, ls, theta = fit.betas.detach().contiguous(), fit.log_beta_scales.detach().contiguous(), fit._pack_theta()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        acc = np.zeros(4)
        reps = 50
        for r in range(reps):
            ev[0].record()
            lbs = e.lbs_forward(betas, theta, ls, want_Rs=False, want_v_shaped=False)[0]
            ev[1].record()
            pts = targets.sample(3000, 0, r)
            ev[2].record()
            o = stage._objective.eval(lbs, fit.trans.detach(), fit.deform_verts.detach(), pts, stage._weights())
            ev[3].record()
            e.lbs_backward(betas, theta, ls, o["dverts"], None)
            ev[4].record()
            torch.cuda.synchronize()
            acc += [ev[k].elapsed_time(ev[k + 1]) for k in range(4)]
        sec = dict(zip(("lbs_forward", "sample", "objective", "lbs_backward"), (acc / reps).tolist()))
        V, S = md.num_verts, 3000
        print(json.dumps({"meshes": N, "iterations_per_s": args.iters / dt, "ms_per_iteration": 1e3 * dt / args.iters,
                          "mesh_iterations_per_s": N * args.iters / dt,
                          "host_issue_ms_per_iteration": 1e3 * t_issue / args.iters, "section_ms": sec,
                          "chamfer_point_pairs_per_s": 2.0 * N * V * S / (sec["objective"] * 1e-3),
                          "final_total_loss": float(stage.last_terms[4])}))
        del stage, targets, fit


if __name__ == "__main__":
    main()
