#!/usr/bin/env python3
"""Writes tests/golden/oracle_full_schedule.npz: the ORACLE's end state (float64, CPU) of the scaled four-stage fit that
tests/test_gpu_parity.py::test_full_schedule compares the HIP loop with.

This is a cache of oracle output, not reference output: the oracle (oracle/smal_oracle.py) is pinned to the reference
by tests/test_oracle_golden.py; running its 195-iteration float64 loop takes minutes of CPU time, which the GPU box
would otherwise spend in every run of the GPU suite.  The file carries the configuration and a sha256 of the problem's
inputs; tests/parity_cases.load_full_schedule_fixture ignores it (and runs the oracle) when either differs, and
tests/test_oracle_golden.py::test_full_schedule_fixture_is_current replays the head of the loop against it on the CPU.

usage: python tests/golden/make_oracle_full_schedule.py          (about 6 minutes on 16 cores)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import parity_cases as pc       # noqa: E402

M, S, WINDOW, SCALE, SEED = 4, 64, 2, 0.1, 21


def main():
    prob, cur, tg = pc.make_problem_cpu(M, S, WINDOW, SEED)
    sched = pc.full_schedule_iterations(SCALE)
    trace = []
    res = pc.full_schedule_oracle(prob, cur, sched, trace)
    out = {"config": np.array("M%d_S%d_w%d_scale%g_seed%d" % (M, S, WINDOW, SCALE, SEED)),
           "fingerprint": np.array(pc.problem_fingerprint(cur, tg)), "schedule": np.array(sched), "trace": np.array(trace)}
    for k, v in res["sums"].items():
        out["sum_" + k] = np.array(v)
    for k, v in res["params"].items():
        out["param_" + k] = v
    np.savez_compressed(pc.FULL_SCHEDULE_FIXTURE, **out)
    print("wrote", pc.FULL_SCHEDULE_FIXTURE, "schedule", sched, "final total", sum(res["sums"].values()))


if __name__ == "__main__":
    main()
