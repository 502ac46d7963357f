#!/usr/bin/env python3
"""developer helper (no GPU needed): the vector-instruction mix of the rasteriser kernels' gfx950 assembly, priced with the issue costs
measured on MI355X (tools/ubench/valu_rate2.hip -> profiles/r4_ubench_valu_issue_costs.txt; cycles per wave-instruction per SIMD at
8 waves/SIMD):  plain float32 multiply-add / multiply / add / subtract, 32-bit integer add and moves 2.6;  the same with a
scalar-register operand, and every other vector ALU instruction (min / max / med3, converts, compares, selects, integer
multiply-adds, shift-adds, bit-field ops) 4.5;  64-bit shift-add / multiply-add 6.8;  packed float32 7.7;  transcendentals 8.4.

The mix is STATIC, weighted by loop nesting (a basic block inside d backward branches counts 8^d times): a stand-in for the dynamic
mix, which no counter reports.  Output: per kernel the weighted share of each class and the blended cycles per vector instruction --
bench.py multiplies SQ_INSTS_VALU (profiles/r5_pmc_summary*.json) by it for `roofline.issue`.

usage: python tools/isa_mix.py [out.json]        (default profiles/r5_isa_mix.json; stamped with bench.kernel_source_sha())"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
COST = {"full": 2.6, "half": 4.5, "u64": 6.8, "packed": 7.7, "trans": 8.4}
FULL = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32",
        "v_mac_f32", "v_mad_f32")
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")
U64 = ("v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32")


def classify(line):
    op = line.split()[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith("v_pk_"):
        return "packed"
    if base in TRANS:
        return "trans"
    if base in U64:
        return "u64"
    if base in FULL:
        operands = line[len(op):]
        if re.search(r"(^|[\s,\[-])(s\d+|s\[\d+:\d+\]|vcc|exec|ttmp\d+|m0)\b", operands):
            return "half"          # a float / integer add with a scalar-register operand issues at the half rate
        return "full"
    return "half"


def main():
    import bench
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r5_isa_mix.json")
    src = os.path.join(ROOT, "smalify_amd", "csrc", "smalfit_kernels.hip")
    asm = "/tmp/_isa_mix.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", asm, src],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-2000:])
    text = open(asm).read()
    doc = {"_note": __doc__.split("usage:")[0].strip(), "cost_cycles": COST, "kernel_source_sha": bench.kernel_source_sha(), "kernels": {}}
    for m in re.finditer(r"^(_Z\w*kernel\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.split("(")[0].strip()
        lines = [ln.strip() for ln in m.group(2).splitlines()]
        # loop nesting from backward branches: label positions, then depth[i] = number of (label <= i <= branch) intervals
        label_at = {ln[:-1]: i for i, ln in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:$", ln)}
        depth = [0] * len(lines)
        for i, ln in enumerate(lines):
            mm = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", ln)
            if mm:
                tgt = label_at.get(mm.group(1) or mm.group(2))
                if tgt is not None and tgt < i:
                    for j in range(tgt, i + 1):
                        depth[j] += 1
        w = {k: 0.0 for k in COST}
        n_valu = n_salu = 0
        for i, ln in enumerate(lines):
            if ln.startswith("v_") and not ln.startswith(("v_readfirstlane", "v_readlane", "v_writelane", "v_nop")):
                w[classify(ln)] += 8.0 ** min(depth[i], 4)
                n_valu += 1
            elif ln.startswith("s_") and not ln.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_load", "s_buffer_load")):
                n_salu += 1
        tot = sum(w.values())
        if tot == 0:
            continue
        doc["kernels"][name] = {"static_valu": n_valu, "static_salu": n_salu, "weighted_share": {k: round(v / tot, 4) for k, v in w.items()},
                                "blended_cycles_per_valu": round(sum(COST[k] * v for k, v in w.items()) / tot, 3)}
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(doc["kernels"].items()):
        if "raster" in k or "bbox" in k:
            print("%-40s valu %5d  blended %.2f cycles  %s" % (k, v["static_valu"], v["blended_cycles_per_valu"], v["weighted_share"]))


if __name__ == "__main__":
    main()
