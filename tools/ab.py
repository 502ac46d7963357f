#!/usr/bin/env python3
"""developer helper (GPU box): A/B of library variants built with tools/build_variant.sh (or by hand into smalify_amd/_variants/).
usage: python tools/ab.py [--steps K] [--reps R] [--scene survey|crop] NAME [NAME ...]      NAME = 'main' (smalify_amd/libsmalfit.so) or a variant name
Runs bench.py per variant (SMALFIT_LIB) and prints value / value_primed / per-stage rates / the raster sections and the
sha256 of the final parameters + losses -- equal hashes = bit-identical fits."""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
args = sys.argv[1:]
steps, reps, scene = 195, 1, "survey"
while args and args[0].startswith("--"):
    if args[0] == "--steps":
        steps = int(args[1])
    elif args[0] == "--reps":
        reps = int(args[1])
    elif args[0] == "--scene":
        scene = args[1]
    args = args[2:]
print("%-12s %8s %8s | %7s %7s %7s %7s | %6s %6s %6s %6s %6s %6s | %s" % ("variant", "cold", "primed", "st0", "st1", "st2", "st3", "sweep", "select", "bwd", "resolv", "lbsf", "lbsb", "state sha"))
for name in args:
    env = dict(os.environ)
    if name != "main":
        env["SMALFIT_LIB"] = os.path.join(ROOT, "smalify_amd", "_variants", name + ".so")
    for _ in range(reps):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--no-cpu-baseline", "--no-crop", "--scene", scene], env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            print("%-12s FAILED rc=%d %s" % (name, out.returncode, out.stderr[-400:]))
            continue
        d = json.loads(lines[0])
        st = d["per_stage_iterations_per_s_primed"]
        sm = d["section_ms"]
        us = lambda k: 1e3 * (sm.get(k) or 0.0)  # noqa: E731
        print("%-12s %8.1f %8.1f | %7.0f %7.0f %7.0f %7.0f | %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f | %s%s" % (
            name, d["value"], d["value_primed"], st["stage0"] or 0, st["stage1"] or 0, st["stage2"] or 0, st["stage3"] or 0,
            us("raster_sweep"), us("raster_select"), us("raster_bwd"), us("raster_resolve"), us("lbs_fwd"), us("lbs_bwd"),
            d["final_state_sha256"], "" if d["status_bits"] == 0 else "  STATUS %d" % d["status_bits"]), flush=True)
