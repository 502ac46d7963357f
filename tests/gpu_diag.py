#!/usr/bin/env python3
"""Run every HIP-vs-oracle parity case and print the error metrics (no assertions).
Usage on the GPU box:  python tests/gpu_diag.py [> gpurun_out/diag.txt]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from tests import parity_cases as pc  # noqa: E402


def run(name, fn, *a, **k):
    t = time.time()
    try:
        out = fn(*a, **k)
        print("== %s  (%.1fs)" % (name, time.time() - t))
        for key, v in out.items():
            print("   %-42s %s" % (key, ("%.4e" % v) if isinstance(v, float) else v))
    except Exception:
        print("== %s FAILED" % name)
        traceback.print_exc()
    sys.stdout.flush()


def main():
    golden = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_golden.npz")))
    run("rodrigues", pc.case_rodrigues)
    run("adam", pc.case_adam)
    run("lbs sparse", pc.case_lbs, 3, False, True)
    run("lbs dense no-scale", pc.case_lbs, 3, True, False)
    for tag, window, stage in (("g6_stage0_w4", 4, 0), ("g6_stage1_w4", 4, 1), ("g6_stage1_w2", 2, 1), ("g6_stage1_w3", 3, 1)):
        run("golden " + tag, pc.case_fit_golden, golden, tag, window, stage)
    run("render S=64 z=1.45", pc.case_render, 2, 64, 1.45, 11)
    run("render S=64 z=0 (K overflow)", pc.case_render, 1, 64, 0.0, 13)
    run("render S=128 z=1.3", pc.case_render, 1, 128, 1.3, 17)
    run("fit stage0", pc.case_fit, 4, 64, 2, 0)
    run("fit stage1", pc.case_fit, 4, 64, 2, 1)
    run("fit stage2 w3", pc.case_fit, 4, 64, 3, 2)


if __name__ == "__main__":
    main()
