"""GPU test of the frame-sharded loop: several processes on one GPU (gloo moves the 1 KB records), each with a FusedFitter on
its shard of the frames, against one process fitting all of them -- two halves; BASELINE config 4's partition (64 frames,
WINDOW_SIZE 8, 8 ranks x 8 frames, 7 interior halos); ONE frame of an 8-frame window per rank (the split north_star names);
ragged shards that cut through windows.  The collective itself is not the point here (RCCL is
exercised by bench.py --gpus N on a multi-GPU node); the point is that ShardedFitter + the HIP engine with halo frames
reproduce the unsharded fit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = 64                      # image size of the small partitions; BASELINE config 4's partition runs at its real 256 x 256
SCHEDULE = ((0, 3), (1, 3), (2, 4))


def _problem(n_frames, seed=21, size=SIZE):
    """start parameters and targets of a synthetic sequence, made ONCE by the parent and handed to every rank: targets are
    rendered by the engine itself from a ground-truth pose (the comparison here is sharded vs unsharded, not HIP vs oracle;
    the oracle's CPU rasteriser would take minutes for 64 frames in each of 9 processes)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_cases as pc
    from smalify_amd import engine as eng
    gt = pc.random_pose(n_frames, seed)
    cur = pc.random_pose(n_frames, seed)
    rs = np.random.RandomState(seed + 7)
    cur["global_rotation"] += (0.05 * rs.randn(n_frames, 3)).astype(np.float32)
    cur["joint_rotations"] += (0.08 * rs.randn(n_frames, 34, 3)).astype(np.float32)
    cur["trans"] += (0.02 * rs.randn(n_frames, 3)).astype(np.float32)
    cur["betas"] += (0.1 * rs.randn(20)).astype(np.float32)
    _, _, dm = pc.get_model()
    e = eng.Engine(dm, n_frames, size)
    sil = torch.empty(n_frames, size, size, device="cuda")
    proj = torch.empty(n_frames, 25, 2, device="cuda")
    d = {k: pc.dev(v) for k, v in gt.items()}
    e.fit_eval(betas=d["betas"], log_beta_scales=d["log_beta_scales"], global_rotation=d["global_rotation"],
               joint_rotations=d["joint_rotations"], trans=d["trans"], target_joints=None, target_visibility=None, target_sil=None,
               weights=(0, 0, 0, 0, 0, 0), w_temp=0.0, window=1, want=(), sil_out=sil, proj_out=proj)
    vis = (rs.rand(n_frames, 25) < 0.85).astype(np.float32)
    tg = dict(tj=(proj.cpu().numpy() + rs.randn(n_frames, 25, 2)).astype(np.float32), vis=vis,
              tsil=(sil > 0.5).float().cpu().numpy())
    assert e.status() == 0 and 0.02 < tg["tsil"].mean() < 0.9
    tg["size"] = size
    return cur, tg


def _run(fitter_factory, rank, world, n_frames, window, cur, tg):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_cases as pc
    from smalify_amd import config as cfg, distributed
    lo, hi = distributed.shard_range(n_frames, rank, world, window=window)
    f = fitter_factory(pc, cur, tg, lo, hi, n_frames, window)
    sf = distributed.ShardedFitter(f, rank, world) if world > 1 else f
    W = np.array(cfg.OPT_WEIGHTS).T
    for stage_id, its in SCHEDULE:
        sf.begin_stage(stage_id)
        for _ in range(its):
            sf.step(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id)
    return {k: v.detach().cpu().numpy().copy() for k, v in f.p.items()}


def _factory(pc, cur, tg, lo, hi, n_frames, window):
    from smalify_amd import engine as eng, fitter as fit, synthetic
    _, _, dm = pc.get_model()
    e = eng.Engine(dm, hi - lo, int(tg.get("size", SIZE)))
    e.set_pose_prior(*synthetic.synthetic_pose_prior())
    e.set_shape_prior(*synthetic.synthetic_shape_prior())
    f = fit.FusedFitter(e, tg["tj"][lo:hi], tg["vis"][lo:hi], tg["tsil"][lo:hi], window, True, cur["betas"], cur["log_beta_scales"],
                        frame_offset=lo, total_frames=n_frames)
    for k in ("global_rotation", "joint_rotations", "trans"):
        f.p[k].copy_(pc.dev(cur[k][lo:hi]))
    return f


def _worker(rank, world, port, q, backend, n_frames, window, cur, tg):
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":                       # RCCL: one GPU per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _run(_factory, rank, world, n_frames, window, cur, tg)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_unsharded_fit():
    _ranks("gloo", 2, 8, 4)


def test_config4_partition_eight_ranks_of_eight_frames():
    """BASELINE config 4's real partition AND image size on one GPU: 64 frames of 256 x 256, WINDOW_SIZE 8, 8 ranks x 8 frames
    (7 interior halos)"""
    _ranks("gloo", 8, 64, 8, size=256)


def test_sharded_fit_follows_the_oracle_loop():
    """the oracle leg: 8 frames, WINDOW_SIZE 4, 4 ranks x 2 frames at 64 x 64 -- the sharded HIP fit (halo frames, one record per
    iteration over the collective, shared parameters stepped from the rank-ordered gradient sum) against the ORACLE's unsharded
    loop (loss + autograd + Adam in float64 on the CPU, reference semantics optimize_to_joints.py:113-137,
    smal_fitter.py:107-190), not just against the unsharded HIP fit: parameters within north_star's 1e-4 relative L2."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_cases as pc
    from oracle import smal_oracle as so
    from smalify_amd import config as cfg
    n_frames, window, world = 8, 4, 4
    prob, cur, tg = pc.make_problem_cpu(n_frames, SIZE, window, seed=31)
    tg["size"] = SIZE
    W = np.array(cfg.OPT_WEIGHTS).T
    params = {k: torch.from_numpy(v).double() for k, v in cur.items()}
    for stage_id, its in SCHEDULE:
        names = so.trainable_names(stage_id)
        vis0 = so.stage0_visibility(prob.vis) if stage_id == 0 else None
        opt = so.Adam(so.PARAM_ORDER, lr=float(W[stage_id][8]))
        for _ in range(its):
            _, _, grads = so.loss_and_grads(prob, params, W[stage_id][:6].copy(), float(W[stage_id][6]), names, visibility=vis0)
            opt.step(params, grads)
    got = _spawn("gloo", world, n_frames, window, cur, tg)
    for k in ("global_rotation", "joint_rotations", "trans"):
        both = np.concatenate([got[r][k] for r in range(world)], 0)
        ref = params[k].numpy().reshape(both.shape)
        err = np.linalg.norm(both - ref) / np.linalg.norm(ref)
        assert err < 1e-4, (k, err)
    for k in ("betas", "log_beta_scales"):
        for r in range(1, world):
            assert np.array_equal(got[0][k], got[r][k]), (k, r)
        ref = params[k].numpy().reshape(got[0][k].shape)
        err = np.linalg.norm(got[0][k] - ref) / np.linalg.norm(ref)
        assert err < 1e-4, (k, err)


def test_one_frame_of_a_window_per_rank():
    """the split north_star names: the 8 frames of ONE window, one frame per rank -- every rank normalises by the window's
    8 frames, rank 0 owns the window's shape-prior term"""
    _ranks("gloo", 8, 8, 8)


def test_ragged_shards_cut_through_windows():
    """7 frames, WINDOW_SIZE 4 (windows of 4 and 3), 3 ranks holding 3 + 2 + 2 frames"""
    _ranks("gloo", 3, 7, 4)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL collective needs one GPU per rank")
def test_two_ranks_over_rccl_match_the_unsharded_fit():
    """the same comparison with the record travelling over RCCL (backend "nccl") between two GPUs"""
    _ranks("nccl", 2, 8, 4)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` outside a launcher starts N ranks itself and reports n_gpus = N: checked with N = 2 when a
    second GPU is visible; on a one-GPU box the launcher + RCCL initialisation + sharded step run with a world of one rank
    (SMALFIT_BENCH_FORCE_DIST)"""
    import json
    import subprocess
    n = 2 if torch.cuda.device_count() >= 2 else 1
    env = dict(os.environ, SMALFIT_BENCH_FORCE_DIST="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "12", "--warmup", "4", "--no-cpu-baseline"]
    if n == 1:                                   # exercise the launcher + RCCL initialisation with a world of one rank
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(29400 + os.getpid() % 500)] + cmd[1:]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 12 and d["value"] > 0 and d["status_bits"] == 0
    assert d["collective"].startswith("rccl"), d["collective"]      # the library called ncclAllGather itself (smalfit_rccl_allgather)


def test_sharded_loop_in_c_equals_the_python_loop():
    """smalfit_shard_run (one library call per stage, the collective enqueued from C through the host-callback adapter) against
    the round-3 host loop (one call before and one after a torch.distributed collective per iteration): the same launches in
    the same order -- identical bits.  World of one rank with the exchange forced on, gloo."""
    import subprocess
    code = """
import os, sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29900 + os.getpid() %% 90))
dist.init_process_group('gloo', rank=0, world_size=1)
import test_gpu_sharded as t
from smalify_amd import config as cfg, distributed
import parity_cases as pc
cur, tg = t._problem(4)
f = t._factory(pc, cur, tg, 0, 4, 4, 2)
sf = distributed.ShardedFitter(f, 0, 1, always_exchange=True)
W = np.array(cfg.OPT_WEIGHTS).T
for stage_id, its in t.SCHEDULE:
    sf.begin_stage(stage_id)
    if os.environ.get('SMALFIT_SHARD_PYTHON_LOOP') == '1':
        for _ in range(its):
            sf.step(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id)
    else:
        sf.run_iterations(W[stage_id][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id, its)
print('SHA', hashlib.sha256(f.flat.cpu().numpy().tobytes() + f.losses.cpu().numpy().tobytes()).hexdigest())
dist.destroy_process_group()
""" % (ROOT, ROOT)
    shas = []
    for loop in ("0", "1"):
        env = dict(os.environ, SMALFIT_SHARD_PYTHON_LOOP=loop)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        shas.append([l for l in out.stdout.splitlines() if l.startswith("SHA")][0])
    assert shas[0] == shas[1], shas


def _spawn(backend, world, n_frames, window, cur, tg):
    """-> {rank: final parameters} of `world` processes fitting their shards"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + 3 * world + n_frames
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend, n_frames, window, cur, tg)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        import queue as _queue
        for _ in range(240):
            try:
                r, val = q.get(timeout=1.0)
                got[r] = val
                if len(got) == world:
                    break
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert len(got) == world, "a rank died: exit codes %s" % [p.exitcode for p in procs]
    return got


def _ranks(backend, world, n_frames, window, size=SIZE):
    cur, tg = _problem(n_frames, size=size)
    single = _run(_factory, 0, 1, n_frames, window, cur, tg)
    torch.cuda.empty_cache()
    got = _spawn(backend, world, n_frames, window, cur, tg)
    for k in ("global_rotation", "joint_rotations", "trans"):
        both = np.concatenate([got[r][k] for r in range(world)], 0)
        assert both.shape == single[k].shape
        err = np.linalg.norm(both - single[k]) / np.linalg.norm(single[k])
        assert err < 2e-5, (k, err)
    for k in ("betas", "log_beta_scales"):
        for r in range(1, world):
            assert np.array_equal(got[0][k], got[r][k]), (k, r)   # shared parameters: identical bits on every rank
        err = np.linalg.norm(got[0][k] - single[k]) / np.linalg.norm(single[k])
        assert err < 2e-5, (k, err)
