#!/usr/bin/env python3
"""Writes tests/golden/oracle_config2_f64.npz / oracle_config2_f32.npz: the ORACLE's run of the reference's FULL 4-stage
schedule (150/400/600/800 iterations, reference config.py:63-72) on BASELINE.json config 2's shape -- 8 frames, 256 x 256,
WINDOW_SIZE 8 -- once in float64 and once in float32: the per-iteration per-term loss trace, the parameters at the start
of every stage and at the end.  tests/config2_case.py defines the problem; tests/test_gpu_config2.py compares the HIP loop
with the float64 run and uses the float32 run as the yardstick for what float32 arithmetic alone does to the end state.

This is a cache of ORACLE output (oracle/smal_oracle.py, pinned to the reference by tests/test_oracle_golden.py), not
reference output.  Hours of CPU: run in the build container, one process per precision:

    python tests/golden/make_oracle_config2.py f64 [threads]      (about 1.5-2 h on 3 threads)
    python tests/golden/make_oracle_config2.py f32 [threads]
    python tests/golden/make_oracle_config2.py f32b [threads]     (a second float32 draw: initial translation moved by 1e-7)
    python tests/golden/make_oracle_config2.py heads [threads]    (after f64: minutes; see heads())

Progress is checkpointed to /tmp/oracle_config2_<tag>.ckpt every 25 iterations (rerun to continue) and a partial fixture
(complete = False) is written every 100 iterations.
"""
import os
import pickle
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import config2_case as _cases    # noqa: E402

# the case the generator runs: SMALFIT_ORACLE_CASE = config2 (default) | config1 (ONE image, all stages, crop-filling scene;
# writes tests/golden/oracle_config1_{f64,f32,heads}.npz)
c2 = _cases.CASES[os.environ.get("SMALFIT_ORACLE_CASE", "config2")]


def write(tag, dtype, fp, tg, trace, stage_start, final, complete):
    out = {"trace": np.asarray(trace, np.float64), "schedule": np.array(c2.SCHEDULE), "fingerprint": np.array(fp),
           "dtype": np.array(str(dtype)), "complete": np.array(complete)}
    for s, p in stage_start.items():
        for k, v in p.items():
            out["stage%d_%s" % (s, k)] = v
    for k, v in (final or {}).items():
        out["final_" + k] = v
    if tag == "f64":                       # the targets travel with the float64 fixture (64 KB of packed bits)
        out["tj"], out["vis"], out["tsil_bits"] = tg["tj"], tg["vis"], np.packbits(tg["tsil"].reshape(-1))
    tmp = c2.fixture_path(tag) + ".tmp.npz"
    np.savez_compressed(tmp, **out)
    os.replace(tmp, c2.fixture_path(tag))


HEAD_ITERS = 24


def heads():
    """oracle_config2_heads.npz: the float32 ORACLE's first HEAD_ITERS iterations of every stage, started from the float64
    run's state at the start of that stage -- what float32 arithmetic alone does to the loss trace over the window in
    which tests/test_gpu_config2.py compares the HIP loop with the float64 trace (the yardstick of that comparison)."""
    import numpy as np
    from oracle import smal_oracle as so
    from smalify_amd import config as cfg
    f64 = c2.load_fixture("f64")
    md, tg = c2.targets()
    assert f64 is not None and f64["fingerprint"] == c2.fingerprint(tg, c2.initial_params())
    prob = c2.problem(md, tg, torch.float32)
    W = np.array(cfg.OPT_WEIGHTS).T
    out = {"fingerprint": np.array(f64["fingerprint"]), "head_iters": np.array(HEAD_ITERS)}
    for stage, start in sorted(f64["stage_start"].items()):
        params = {k: torch.from_numpy(np.asarray(v)).to(torch.float32) for k, v in start.items()}
        w = W[stage]
        names = so.trainable_names(stage)
        vis = so.stage0_visibility(prob.vis) if stage == 0 else None
        opt = so.Adam(so.PARAM_ORDER, lr=float(w[8]))
        rows = []
        for _ in range(HEAD_ITERS):
            total, sums, grads = so.loss_and_grads(prob, params, w[:6].copy(), float(w[6]), names, visibility=vis)
            rows.append([float(sums.get(k, 0.0)) for k in c2.TERMS])
            opt.step(params, grads)
        out["stage%d_f32_trace" % stage] = np.array(rows)
        print("heads: stage", stage, "done", flush=True)
    np.savez_compressed(c2.fixture_path("heads"), **out)
    print("wrote", c2.fixture_path("heads"))


def main():
    tag = sys.argv[1]
    if tag == "heads":
        torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
        return heads()
    torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    # f32b: a SECOND float32 draw -- the same float32 oracle loop from an initial translation moved by 1e-7 (one unit in the last
    # place of 1.0) -- because a float32 trajectory of ~2000 Adam steps is ONE sample of a chaotic map: the yardstick of a comparison
    # with the float64 run is the largest deviation any recorded float32 draw shows (tests/test_gpu_config2.py)
    # (f32c / f32d / f32e: further draws -- y + 1e-7, z + 1e-7, x - 1e-7)
    DRAWS = {"f32b": (0, 1e-7), "f32c": (1, 1e-7), "f32d": (2, 1e-7), "f32e": (0, -1e-7)}
    dtype = torch.float64 if tag == "f64" else torch.float32
    assert tag in ("f64", "f32") or tag in DRAWS, tag
    md, tg = c2.targets()
    start = c2.initial_params()
    fp = c2.fingerprint(tg, start)
    if tag in DRAWS:
        start = {k: v.copy() for k, v in start.items()}
        start["trans"][:, DRAWS[tag][0]] += np.float32(DRAWS[tag][1])
    prob = c2.problem(md, tg, dtype)
    ckpt = "/tmp/oracle_%s_%s.ckpt" % (c2.name, tag)
    state = None
    if os.path.exists(ckpt):
        saved = pickle.load(open(ckpt, "rb"))
        if saved["fp"] == fp:
            state = saved["state"]
            print("resuming at stage %d iteration %d" % (state["stage"], state["it"]), flush=True)
    t0 = time.time()

    def checkpoint(st):
        n = len(st["trace"])
        if n % 25 == 0:
            pickle.dump({"fp": fp, "state": st}, open(ckpt + ".tmp", "wb"))
            os.replace(ckpt + ".tmp", ckpt)
            print("%s  %4d / %d  total %.6f  (%.0f s)" % (tag, n, sum(c2.SCHEDULE), sum(st["trace"][-1]), time.time() - t0), flush=True)
        if n % 100 == 0:
            write(tag, dtype, fp, tg, st["trace"], st["stage_start"], None, False)

    trace, stage_start, final = c2.oracle_schedule(prob, start, dtype, checkpoint=checkpoint, state=state)
    write(tag, dtype, fp, tg, trace, stage_start, final, True)
    print("wrote", c2.fixture_path(tag), "final total", trace[-1].sum())


if __name__ == "__main__":
    main()
