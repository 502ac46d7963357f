#!/usr/bin/env python3
"""Writes the oracle evaluation fixtures of tests/eval_cases.py (BASELINE's own sizes; minutes of float64 CPU work each):

    python tests/golden/make_oracle_eval.py targets <case>            -> tests/golden/eval_targets_<case>.npz   (oracle-rendered targets)
    [GPU]  python tools/dump_fit_states.py <case>                      -> tests/golden/hip_states_<case>.npz     (states the HIP fit passes through: inputs)
    python tests/golden/make_oracle_eval.py eval <case> [threads]      -> tests/golden/oracle_eval_<case>.npz    (float64 losses + gradients per state,
                                                                                                                  and the oracle's own float32 evaluation)

This is a cache of ORACLE output (oracle/smal_oracle.py, pinned to the reference by tests/test_oracle_golden.py), not
reference output.  `eval` skips states it already holds for the same inputs (rerun after adding the hip_* states)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import eval_cases as ec       # noqa: E402


def main():
    what, case = sys.argv[1], sys.argv[2]
    torch.set_num_threads(int(sys.argv[3]) if len(sys.argv) > 3 else 4)
    if what == "targets":
        t0 = time.time()
        ec.save_targets(case, ec.make_targets(case))
        print("wrote", ec.targets_path(case), "%.0f s" % (time.time() - t0))
        return
    tg = ec.load_targets(case)
    assert tg is not None, "run `make_oracle_eval.py targets %s` first" % case
    st = ec.states(case)
    out = {"fingerprint": np.array(ec.fingerprint(tg, st))}
    old = np.load(ec.fixture_path(case), allow_pickle=False) if os.path.exists(ec.fixture_path(case)) else None
    probs = {}
    for name, params in st.items():
        stage = ec.STATE_STAGE[name]
        if old is not None and name + "_terms_f32" in old.files and all(
                np.array_equal(old["%s_p_%s" % (name, k)], params[k]) for k in ec.PARAMS):
            for key in old.files:
                if key.startswith(name + "_"):
                    out[key] = old[key]
            print(name, "kept", flush=True)
            continue
        out[name + "_stage"] = np.array(stage)
        for k in ec.PARAMS:
            out["%s_p_%s" % (name, k)] = params[k]
        for dtype, tkey, gkey in ((torch.float64, "_terms", "_g_"), (torch.float32, "_terms_f32", "_g32_")):
            if dtype not in probs:
                probs[dtype] = ec.problem(case, tg, dtype)
            t0 = time.time()
            terms, grads = ec.oracle_eval(probs[dtype], params, stage, dtype)
            out[name + tkey] = terms
            for k, g in grads.items():
                out[name + gkey + k] = g
            print(name, str(dtype), "total %.6f" % terms.sum(), "%.0f s" % (time.time() - t0), flush=True)
        tmp = ec.fixture_path(case) + ".tmp.npz"
        np.savez_compressed(tmp, **out)
        os.replace(tmp, ec.fixture_path(case))
    np.savez_compressed(ec.fixture_path(case) + ".tmp.npz", **out)
    os.replace(ec.fixture_path(case) + ".tmp.npz", ec.fixture_path(case))
    print("wrote", ec.fixture_path(case), "states:", list(st))


if __name__ == "__main__":
    main()
