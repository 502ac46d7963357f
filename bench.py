#!/usr/bin/env python3
"""Benchmark of the SMAL per-frame fitting hot path on MI355X.

metric  : fitter iterations/sec  (BASELINE.json) — one iteration = one epoch of the reference loop
          (smal_fitter/optimize_to_joints.py:113-137): LBS + projection + soft-silhouette render + all
          losses + temporal term + full gradient + Adam step over the whole batch.
workload: synthetic BADJA-shape sequence, 64 frames, 256x256, WINDOW_SIZE 8, shape family 1 with the
          unity-style shape prior, synthetic SMAL-topology model (V=3889, F=7774).  The K timed steps run
          the reference's 4-stage schedule (150:400:600:800 iterations, config.py:63-72) scaled to K
          iterations, each stage with its own weights / learning rate / fresh Adam state.
          With --steps 1950 the timed region is exactly one complete fit.

One process per GPU (python -m torch.distributed.run ... bench.py --gpus N): frames are sharded
contiguously across ranks (strong scaling), see smalify_amd/distributed.py.

Prints ONE JSON line on rank 0.

SMALFIT_BENCH_FORCE_DIST=1 (validation hook): initialise the process group and run the sharded step with its
all-gather even when WORLD_SIZE is 1, so that the RCCL path can be exercised on a single-GPU box
(python -m torch.distributed.run --nproc-per-node 1 ... bench.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NUM_FRAMES = 64
IMAGE_SIZE = 256
WINDOW = 8
SCHEDULE_ITERS = (150, 400, 600, 800)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


PROFILE_STRIDE = 8


def scaled_schedule(total):
    """split `total` iterations over the 4 stages in the reference's 150:400:600:800 proportion"""
    raw = [total * s / float(sum(SCHEDULE_ITERS)) for s in SCHEDULE_ITERS]
    its = [int(np.floor(r)) for r in raw]
    order = np.argsort([-(r - i) for r, i in zip(raw, its)])
    k = 0
    while sum(its) < total:
        its[order[k % 4]] += 1
        k += 1
    return its


def build_problem(engine, torch, scene):
    """ground-truth draw -> targets rendered by the engine itself (data synthesis, untimed)"""
    from smalify_amd import synthetic
    N, S = NUM_FRAMES, IMAGE_SIZE
    sp = synthetic.synthetic_shape_prior()
    gt = synthetic.ground_truth_params(N, seed=1234, mean_betas=sp[1][:20], mean_logscale=sp[1][20:26])
    if scene == "crop":           # animal fills the crop, as BADJA crops do (data_loader crop_to_silhouette)
        gt["trans"][:, 2] += 1.2
    dev = engine.device
    t = lambda a: torch.as_tensor(a, device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    sil = torch.empty(N, S, S, device=dev)
    proj = torch.empty(N, 25, 2, device=dev)
    engine.fit_eval(betas=t(gt["betas"]), log_beta_scales=t(gt["log_beta_scales"]),
                    global_rotation=t(gt["global_rotation"]), joint_rotations=t(gt["joint_rotations"]),
                    trans=t(gt["trans"]), target_joints=None, target_visibility=None, target_sil=None,
                    weights=(0, 0, 0, 0, 0, 0), w_temp=0.0, window=WINDOW, want=(), sil_out=sil, proj_out=proj)
    noise, vis = synthetic.keypoint_noise_and_visibility(N)
    target_joints = proj + t(noise)
    target_sil = (sil > 0.5).float()
    return gt, target_joints, t(vis), target_sil, sp


def cpu_baseline(md, pose_prior, shape_prior, target_joints, vis, target_sil, stage_weights, w_temp):
    """Times the oracle (CPU port of the same maths: torch float32, pair-list rasteriser, autograd backward, Adam) on a
    bounded sample of the same workload: one stage-2-type iteration over the first `nf` of the 64 frames, where nf is
    chosen from a 1-frame probe so that the sample costs roughly 10-15 s; extrapolated linearly to 64 frames.
    torch intra-op threads are capped at 16 (the oracle's tensors are small; more threads only add contention)."""
    import torch
    from oracle import smal_oracle as so
    from smalify_amd import model_io
    ncores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(ncores)
    om = so.OracleModel(md, dtype=torch.float32)

    def one_iteration(nf):
        prob = so.FitProblem(om, IMAGE_SIZE, target_joints[:nf], vis[:nf], target_sil[:nf], pose_prior[0], pose_prior[1],
                             pose_prior[2], shape_prior[0], shape_prior[1], min(WINDOW, nf), True, dtype=torch.float32)
        params = dict(betas=torch.from_numpy(shape_prior[1][:20].copy()),
                      log_beta_scales=torch.from_numpy(shape_prior[1][20:26].copy()),
                      global_rotation=torch.from_numpy(np.tile(model_io.initial_global_rotation(), (nf, 1))).float(),
                      trans=torch.zeros(nf, 3), joint_rotations=torch.zeros(nf, 34, 3))
        opt = so.Adam(so.PARAM_ORDER, lr=5e-4)
        t0 = time.perf_counter()
        total, sums, grads = so.loss_and_grads(prob, params, stage_weights, w_temp, so.PARAM_ORDER)
        opt.step(params, grads)
        dt = time.perf_counter() - t0
        assert sums.get("sil_reproj", 0.0) > 0.0, "silhouette term missing from the CPU baseline sample"
        return dt

    t1 = one_iteration(1)
    nf = int(max(1, min(NUM_FRAMES, 12.0 / max(t1, 1e-3))))
    dt = one_iteration(nf) if nf > 1 else t1
    per_iter_64 = dt * (NUM_FRAMES / nf)
    return {"value": 1.0 / per_iter_64, "unit": "iterations/s", "cores": ncores, "kind": "port",
            "sample": "oracle (torch CPU float32 restatement, pair-list rasteriser) on %d of 64 frames, 1 stage-2-type "
                      "iteration incl. backward + Adam: %.2f s (1-frame probe %.2f s); extrapolated x%.2f"
                      % (nf, dt, t1, NUM_FRAMES / nf)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=390)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="survey", choices=["survey", "crop"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout must carry exactly one JSON line: libraries write there too (RCCL prints a version banner when the
    # communicator is created), so file descriptor 1 points at stderr for the whole run and the line goes to the saved one
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from smalify_amd import config, distributed, engine as eng, fitter as fit, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (smalify_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("SMALFIT_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    dm = eng.DeviceModel(md)
    full_engine = eng.Engine(dm, NUM_FRAMES, IMAGE_SIZE)
    pose_prior = synthetic.synthetic_pose_prior()
    gt, tj, vis, tsil, shape_prior = build_problem(full_engine, torch, args.scene)
    lo, hi = distributed.shard_range(NUM_FRAMES, rank, world)
    if world > 1:
        del full_engine
        torch.cuda.empty_cache()
        engine = eng.Engine(dm, hi - lo, IMAGE_SIZE)
    else:
        engine = full_engine
    engine.set_pose_prior(*pose_prior)
    engine.set_shape_prior(*shape_prior)

    def new_fitter():
        f = fit.FusedFitter(engine, tj[lo:hi], vis[lo:hi], tsil[lo:hi], WINDOW, use_unity_prior=True,
                            mean_betas=shape_prior[1][:20], mean_log_scales=shape_prior[1][20:26])
        return distributed.ShardedFitter(f, rank, world, always_exchange=force_dist) if use_dist else f

    W = np.array(config.OPT_WEIGHTS).T

    def run(fitter, iters_per_stage, stage_seconds=None):
        for stage_id, its in enumerate(iters_per_stage):
            fitter.begin_stage(stage_id)
            t_stage = time.perf_counter()
            for _ in range(its):
                fitter.step(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id)
            if stage_seconds is not None:      # per-stage rates: one device sync per stage, 4 in the run
                torch.cuda.synchronize()
                stage_seconds.append(time.perf_counter() - t_stage)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (untimed): W iterations with the same stage mix ---------------------------------------
    run(new_fitter(), scaled_schedule(max(args.warmup, 4)))
    fitter = new_fitter()
    sched = scaled_schedule(args.steps)
    base = fitter.fitter if use_dist else fitter
    # HIP events on the launch stream inside the timed region, on every 8th iteration (an event record costs ~5 us
    # of stream time; all sections of every iteration would slow the measured loop by ~8 %)
    base.e.profile_begin(args.steps, PROFILE_STRIDE)
    sync()
    t0 = time.perf_counter()
    stage_seconds = []
    run(fitter, sched, stage_seconds)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    sections = base.e.profile_end()
    status = base.e.status()
    final_losses = base.losses.cpu().numpy().tolist()

    if rank == 0:
        V, F, S = md.num_verts, md.num_faces, IMAGE_SIZE
        nloc = hi - lo
        sec_ms = {k: (v[0] / v[1] if v[1] else None) for k, v in sections.items()}
        # dominant kernel: whichever rasteriser kernel has the largest average launch time in this run.  Algorithmic
        # bytes per launch: the forward rasteriser must read the projected vertices (12 V per frame),
        # the faces (12 F) and the target silhouette (4 S^2 per frame) and produce the per-pixel adjoint seed.
        algo_bytes = nloc * (4 * S * S + 12 * V) + 12 * F
        cand = {k: sec_ms[k] for k in ("raster_sweep", "raster_select", "raster_resolve", "raster_bwd") if sec_ms.get(k)}
        dom_name = max(cand, key=cand.get) if cand else None
        dom = cand.get(dom_name) if dom_name else None
        achieved = algo_bytes / (dom * 1e-3) / 1e9 if dom else None
        # HBM traffic per launch of that kernel from the rocprofv3 PMC passes committed under profiles/
        # (tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate runs; KiB, raw -- on gfx950 FETCH_SIZE
        # under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md, so this is a lower bound)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_summary.json")))
            kname = "smalfit::" + {"raster_sweep": "raster_sweep_kernel", "raster_select": "raster_select_kernel",
                                   "raster_resolve": "raster_resolve_kernel", "raster_bwd": "raster_bwd_kernel"}[dom_name]
            traffic = 1024.0 * (pmc["fetch"].get(kname, {}).get("avg_per_row", 0.0) + pmc["write"].get(kname, {}).get("avg_per_row", 0.0))
        except Exception:
            traffic = None
        out = {
            "metric": "fitter iterations/sec", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic BADJA-shape sequence: %d frames, %dx%d, WINDOW_SIZE %d, shape family 1, "
                                   "reference 4-stage schedule scaled to %d iterations %s, scene=%s"
                                   % (NUM_FRAMES, S, S, WINDOW, args.steps, sched, args.scene),
                       "frames": NUM_FRAMES, "image_size": S, "window": WINDOW, "parallelism": "frames/%d" % world},
            "per_stage_iterations_per_s": {"stage%d" % i: (sched[i] / stage_seconds[i] if stage_seconds[i] > 0 else None)
                                           for i in range(len(sched))},
            "roofline": {"bound": "hbm", "kernel": {"raster_sweep": "raster_sweep_kernel", "raster_select": "raster_select_kernel", "raster_resolve": "raster_resolve_kernel", "raster_bwd": "raster_bwd_kernel"}.get(dom_name), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": dom},
            "section_ms": sec_ms, "final_losses": dict(zip(eng.LOSS_NAMES, final_losses)), "status_bits": status,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(md, pose_prior, shape_prior, tj.cpu().numpy(), vis.cpu().numpy(),
                                               tsil.cpu().numpy(), W[2][:6], float(W[2][6]))
        line = json.dumps(out)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:                       # the JSON line is the last thing this process writes
        sys.stdout.flush()
        sys.stderr.flush()
        os.write(json_fd, (line + "\n").encode())


if __name__ == "__main__":
    main()
