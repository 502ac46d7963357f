"""CPU, world_size = 2 / 3 / 8, gloo: the frame-sharding layer (smalify_amd/distributed.py) must reproduce the
unsharded optimisation: halo exchange for the temporal pairs that straddle the shard boundary, all-reduce of
the shared shape gradient, identical Adam state on every rank -- for window-aligned shards (BASELINE config 4: 64 frames,
8 per rank), for ONE frame of an 8-frame window per rank (the split north_star names) and for ragged shards that cut
through windows.

The per-rank compute engine here is the ORACLE (torch CPU) wrapped in the local-fitter protocol that
FusedFitter implements on the GPU — the test targets the distributed logic, not the kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class OracleLocalFitter:
    """FusedFitter protocol on top of oracle.smal_oracle for frames [lo, hi) of a problem."""

    def __init__(self, prob_full, params_full, lo, hi, window):
        from oracle import smal_oracle as so
        self.so = so
        m = prob_full.m
        self.prob = so.FitProblem(m, prob_full.S, prob_full.tj[lo:hi].numpy(), prob_full.vis[lo:hi].numpy(),
                                  prob_full.sil[lo:hi].numpy(), prob_full.pose_prec.numpy(), prob_full.pose_mean.numpy(),
                                  prob_full.pose_mask.numpy(), prob_full.shape_prec.numpy(), prob_full.shape_mean.numpy(),
                                  window, True, frame_offset=lo, total_frames=prob_full.N)
        self.N = hi - lo
        self.p = {k: (v[lo:hi].clone() if v.shape[0] == prob_full.N and v.dim() > 1 else v.clone())
                  for k, v in params_full.items()}
        self.grads = {}
        self.halo_prev = self.halo_next = None
        self.losses = torch.zeros(8, dtype=torch.float64)
        self._shared = torch.zeros(26, dtype=torch.float64)

    def trainable(self, stage_id):
        return self.so.trainable_names(stage_id)

    def begin_stage(self, stage_id):
        self.opt = None
        self.stage_id = stage_id

    def boundary_records(self):
        idx = [0, self.N - 1]
        return torch.cat([self.p["global_rotation"][idx], self.p["joint_rotations"][idx].reshape(2, 102),
                          self.p["trans"][idx]], 1).contiguous()

    def shared_grad(self):
        return self._shared

    def evaluate(self, weights, w_temp, stage_id, want=None):
        so = self.so
        names = self.trainable(stage_id) if want is None else want
        leaf = {k: v.detach().clone().requires_grad_(k in names) for k, v in self.p.items()}
        total, _ = so.epoch_loss(self.prob, leaf, weights, w_temp)

        def pair(a, b):       # temporal terms of one adjacent pair, a = (theta 105 | trans 3) records
            return w_temp * (((a[:3] - b[:3]) ** 2).mean() + ((a[3:105] - b[3:105]) ** 2).mean()
                             + ((a[105:] - b[105:]) ** 2).mean())

        def rec(i):
            return torch.cat([leaf["global_rotation"][i], leaf["joint_rotations"][i].reshape(102), leaf["trans"][i]])

        if self.halo_next is not None:
            total = total + pair(rec(self.N - 1), self.halo_next)
        if self.halo_prev is not None:
            total = total + pair(self.halo_prev, rec(0))
        total.backward()
        self.grads = {k: leaf[k].grad for k in names}
        if "betas" in self.grads:
            self._shared = torch.cat([self.grads["betas"], self.grads["log_beta_scales"]]).contiguous()
        return self.losses

    def apply_adam(self, names, lr, advance=True):   # the oracle's Adam counts steps per tensor
        so = self.so
        if self.opt is None:
            self.opt = so.Adam(so.PARAM_ORDER, lr=lr)
        if "betas" in names:
            self.grads["betas"], self.grads["log_beta_scales"] = self._shared[:20].clone(), self._shared[20:].clone()
        self.opt.step(self.p, {k: self.grads[k] for k in names})


def _problem(N=4, window=2):
    sys.path.insert(0, ROOT)
    from oracle import smal_oracle as so
    from smalify_amd import model_io, synthetic
    md = synthetic.synthetic_model()
    om = so.OracleModel(md)
    rs = np.random.RandomState(5)
    S = 32
    pp, sp = synthetic.synthetic_pose_prior(), synthetic.synthetic_shape_prior()
    prob = so.FitProblem(om, S, rs.rand(N, 25, 2) * S, (rs.rand(N, 25) < 0.8).astype(np.float64), np.zeros((N, S, S)),
                         pp[0], pp[1], pp[2], sp[0], sp[1], window, True)
    init = model_io.initial_global_rotation()
    params = dict(betas=torch.from_numpy(sp[1][:20]).double(), log_beta_scales=torch.from_numpy(sp[1][20:26]).double(),
                  global_rotation=torch.from_numpy(np.tile(init, (N, 1)) + 0.1 * rs.randn(N, 3)).double(),
                  trans=torch.from_numpy(0.05 * rs.randn(N, 3)).double(),
                  joint_rotations=torch.from_numpy(0.1 * rs.randn(N, 34, 3)).double())
    return prob, params


def _worker(rank, world, port, out_q, N=4, window=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from smalify_amd import config as cfg, distributed
    torch.set_num_threads(1 if world > 2 else 2)
    prob, params = _problem(N, window)
    lo, hi = distributed.shard_range(prob.N, rank, world, window=window)
    f = distributed.ShardedFitter(OracleLocalFitter(prob, params, lo, hi, window), rank, world)
    W = np.array(cfg.OPT_WEIGHTS).T
    w1 = W[1][:6].copy()
    w1[1] = 0.0
    for stage_id, its in ((0, 2), (1, 3)):
        f.begin_stage(stage_id)
        if stage_id == 0:        # both entry points of the sharded loop: per iteration ...
            for _ in range(its):
                f.step(W[0][:6], float(W[stage_id][6]), float(W[stage_id][8]), stage_id)
        else:                    # ... and a whole stage at once (one library call with the HIP engine; a loop over step() for a
            f.run_iterations(w1, float(W[stage_id][6]), float(W[stage_id][8]), stage_id, its)    # local fitter without shard_run)
    proof = f.prove_world()                 # what bench.py --gpus N puts into its line: the collective spans `world` distinct ranks
    assert proof["ok"] and proof["rank_stamps"] == list(range(1, world + 1)) and proof["distinct_ranks"] == world, proof
    out_q.put((rank, (lo, hi), {k: v.numpy() for k, v in f.fitter.p.items()}))
    dist.barrier()
    dist.destroy_process_group()


# (frames, WINDOW_SIZE, ranks): window-aligned halves; BASELINE config 4's partition (64 frames, 8 per rank); ONE frame of
# an 8-frame window per rank; ragged shards 3 + 2 + 2 cutting through windows of 4 (the last window is short)
@pytest.mark.parametrize("N,window,world", [(4, 2, 2), (64, 8, 8), (8, 8, 8), (7, 4, 3)])
def test_sharded_fit_matches_single_process(N, window, world):
    sys.path.insert(0, ROOT)
    from oracle import smal_oracle as so
    from smalify_amd import config as cfg
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world + N
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, N, window)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    results, ranges = {}, {}
    try:
        for _ in range(600):
            try:
                r, rng, val = q.get(timeout=1.0)
                results[r], ranges[r] = val, rng
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
            if len(results) == world:
                break
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert len(results) == world, "a rank failed"
    assert all(p.exitcode == 0 for p in procs)
    assert [ranges[r][0] for r in range(world)] == [0] + [ranges[r][1] for r in range(world - 1)] and ranges[world - 1][1] == N
    # single-process oracle run of the same schedule
    prob, params = _problem(N, window)
    W = np.array(cfg.OPT_WEIGHTS).T
    w1 = W[1][:6].copy()
    w1[1] = 0.0
    for stage_id, its in ((0, 2), (1, 3)):
        opt = so.Adam(so.PARAM_ORDER, lr=float(W[stage_id][8]))
        for _ in range(its):
            _, _, grads = so.loss_and_grads(prob, params, W[0][:6] if stage_id == 0 else w1, float(W[stage_id][6]),
                                            so.trainable_names(stage_id))
            opt.step(params, grads)
    for k in ("betas", "log_beta_scales"):
        for r in range(world):
            assert np.array_equal(results[r][k], results[0][k]), (k, r)          # the same bits on every rank
            assert np.allclose(results[r][k], params[k].numpy(), rtol=1e-9, atol=1e-12), (k, r)
    for k in ("global_rotation", "trans", "joint_rotations"):
        got = np.concatenate([results[r][k] for r in range(world)], 0)
        assert np.allclose(got, params[k].numpy(), rtol=1e-9, atol=1e-12), k


def test_oracle_shards_add_up_to_the_sequence():
    """the shard-aware form of the oracle's epoch loss (FitProblem.frame_offset / total_frames) is a partition of the
    reference's sum over windows: per-term sums over any contiguous split equal the whole sequence's"""
    sys.path.insert(0, ROOT)
    from oracle import smal_oracle as so
    from smalify_amd import config as cfg, distributed
    prob, params = _problem(7, 4)
    W = np.array(cfg.OPT_WEIGHTS).T
    w1 = W[1][:6].copy()
    w1[1] = 0.0
    _, full = so.epoch_loss(prob, params, w1, 0.0)
    for world in (2, 3, 7):
        acc = {}
        for r in range(world):
            lo, hi = distributed.shard_range(7, r, world)
            part = so.FitProblem(prob.m, prob.S, prob.tj[lo:hi].numpy(), prob.vis[lo:hi].numpy(), prob.sil[lo:hi].numpy(),
                                 prob.pose_prec.numpy(), prob.pose_mean.numpy(), prob.pose_mask.numpy(), prob.shape_prec.numpy(),
                                 prob.shape_mean.numpy(), 4, True, frame_offset=lo, total_frames=7)
            pp = {k: (v[lo:hi] if v.dim() > 1 else v) for k, v in params.items()}
            _, sums = so.epoch_loss(part, pp, w1, 0.0)
            for k, v in sums.items():
                acc[k] = acc.get(k, 0.0) + float(v)
        for k in ("joint", "pose", "splay", "betas"):
            assert abs(acc[k] - float(full[k])) < 1e-9 * max(1.0, abs(float(full[k]))), (world, k, acc[k], float(full[k]))


def test_shard_range_validation():
    from smalify_amd import distributed
    assert distributed.shard_range(64, 3, 8, window=8) == (24, 32)
    assert [distributed.shard_range(8, r, 8, window=8) for r in range(8)] == [(r, r + 1) for r in range(8)]   # one frame per rank
    sizes = [hi - lo for lo, hi in (distributed.shard_range(61, r, 8, window=8) for r in range(8))]
    assert sizes == [8, 8, 8, 8, 8, 7, 7, 7] and distributed.shard_range(61, 7, 8)[1] == 61                   # ragged clip
    with pytest.raises(ValueError):
        distributed.shard_range(3, 0, 4)            # fewer frames than ranks
    with pytest.raises(ValueError):
        distributed.shard_range(64, 8, 8)


def test_world_of_one_with_the_collective_equals_the_plain_step():
    """ShardedFitter(always_exchange=True) on a world of one (bench.py's SMALFIT_BENCH_FORCE_DIST hook: the collective
    runs, there are no neighbours) must reproduce the short-cut path bit for bit"""
    sys.path.insert(0, ROOT)
    from smalify_amd import config as cfg, distributed
    torch.set_num_threads(2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(31500 + (os.getpid() % 2000)))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        W = np.array(cfg.OPT_WEIGHTS).T
        w1 = W[1][:6].copy()
        w1[1] = 0.0
        out = []
        for always in (False, True):
            prob, params = _problem()
            f = distributed.ShardedFitter(OracleLocalFitter(prob, params, 0, prob.N, 2), 0, 1, always_exchange=always)
            for stage_id, its in ((0, 2), (1, 2)):
                f.begin_stage(stage_id)
                for _ in range(its):
                    f.step(W[0][:6] if stage_id == 0 else w1, float(W[stage_id][6]), float(W[stage_id][8]), stage_id)
            assert (f.fitter.halo_prev is None) and (f.fitter.halo_next is None)
            out.append({k: v.numpy().copy() for k, v in f.fitter.p.items()})
        for k in out[0]:
            assert np.array_equal(out[0][k], out[1][k]), k
    finally:
        dist.destroy_process_group()
