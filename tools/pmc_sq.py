#!/usr/bin/env python3
"""Developer helper (GPU box): SQ / TCC counters of the rasteriser kernels via rocprofv3 PMC passes.

One rocprofv3 run per counter group (gfx950: 8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE do not fit one pass), never
combined with tracing domains other than --kernel-trace.  Counter names that `rocprofv3 -L` does not list are dropped
from a group instead of failing the pass.  Writes gpurun_out/<tag>/pmc_summary.json: per kernel, per counter, the
average per launch (summed over the dimension rows of one dispatch).

usage: python tools/pmc_sq.py TAG [bench args...]        (default bench args: --steps 39 --warmup 4 --no-cpu-baseline --no-crop)
       PMC_SCRIPT="tools/raster_probe.py" python tools/pmc_sq.py TAG      (another script of the repo instead of bench.py)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
GROUPS = [
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
     "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
     "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"],
    ["SQ_INSTS_SMEM", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA",
     "SQ_INSTS_FLAT", "SQ_LDS_ATOMIC_RETURN", "SQ_WAVE_CYCLES"],
    ["GRBM_GUI_ACTIVE", "FETCH_SIZE"],
    ["GRBM_GUI_ACTIVE", "WRITE_SIZE"],
]


def script_argv():
    """PMC_SCRIPT = "tools/x.py args...": the script path is taken relative to the repository (the passes run from /tmp)"""
    parts = os.environ["PMC_SCRIPT"].split()
    if not os.path.isabs(parts[0]):
        parts[0] = os.path.join(ROOT, parts[0])
    return parts


def main():
    global GROUPS
    if os.environ.get("PMC_GROUPS"):                 # e.g. PMC_GROUPS='[["TCP_TOTAL_CACHE_ACCESSES","TCP_TCC_READ_REQ"],["TA_BUSY"]]'
        GROUPS = json.loads(os.environ["PMC_GROUPS"])
    tag = sys.argv[1] if len(sys.argv) > 1 else "pmc_sq"
    bench_args = sys.argv[2:] or ["--steps", "39", "--warmup", "4", "--no-cpu-baseline", "--no-crop"]
    out = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    listing = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, cwd="/tmp", env=env).stdout
    open(os.path.join(out, "counters_list.txt"), "w").write(listing)
    summary = {}
    for gi, group in enumerate(GROUPS):
        names = [c for c in group if (c + " ") in listing or (c + "\n") in listing or ("Name: " + c) in listing or c in listing] or group
        d = os.path.join(out, "g%d" % gi)
        cmd = ["rocprofv3", "--pmc"] + names + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                                                sys.executable] + (script_argv() if os.environ.get("PMC_SCRIPT") else [os.path.join(ROOT, "bench.py")] + bench_args)
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env)
        open(os.path.join(out, "g%d.log" % gi), "w").write(" ".join(cmd) + "\n" + r.stdout[-2000:] + "\n" + r.stderr[-4000:])
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        ndisp = collections.defaultdict(set)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "").split("(")[0]
                if "smalfit" not in k:
                    continue
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                ndisp[k].add(row.get("Dispatch_Id"))
        for k, cs in agg.items():
            n = max(len(ndisp[k]), 1)
            e = summary.setdefault(k, {"launches": n})
            for c, v in cs.items():
                e[c] = v / n
    sys.path.insert(0, ROOT)
    import bench                                                   # the stamp bench.py checks before quoting `traffic` from this file
    what = os.environ.get("PMC_SCRIPT") or ("bench.py " + " ".join(bench_args))
    stamped = dict(summary, _note="rocprofv3 --pmc passes (one group of counters per run, --kernel-trace only) of `" + what + "`, made by tools/pmc_sq.py; "
                   "per kernel: average per launch; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE reports half of a wide coalesced read on gfx950, "
                   "MI355X_MICROARCH.md); kernel_source_sha = bench.kernel_source_sha() of the sources measured",
                   kernel_source_sha=bench.kernel_source_sha())
    json.dump(stamped, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    for k in sorted(summary, key=lambda k: -summary[k].get("SQ_WAVE_CYCLES", 0))[:8]:
        print(k, json.dumps({c: round(v, 1) for c, v in summary[k].items()}))


if __name__ == "__main__":
    main()
