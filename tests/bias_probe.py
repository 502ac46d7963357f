#!/usr/bin/env python3
"""Diagnostic (GPU box, no assertions): is the HIP silhouette / its adjoint BIASED -- an error of one sign at every pixel -- or
merely noisy?  The shared-shape gradient adds the frames' silhouette adjoints up; random float32 noise averages out in that sum, a
bias does not.  For one general pose at S x S:
  * alpha = 1 - sil per rim pixel (0.02 < alpha < 0.98): mean SIGNED relative deviation from the float64 oracle, HIP and float32 oracle
  * d(sum_pixels sil)/d(uniform scale about the centroid) and /d(translation): relative deviation from the float64 oracle
usage: python tests/bias_probe.py [S] [M]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import smal_oracle as so  # noqa: E402
from tests import parity_cases as pc  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    md, om, _ = pc.get_model()
    e, _, _ = pc.get_engine(8, S)
    for z in (1.45, 0.6):
        p = pc.random_pose(M, 11, z=z)
        theta = np.concatenate([p["global_rotation"][:, None], p["joint_rotations"]], 1)
        with torch.no_grad():
            vo, _, _, _ = so.smal_forward(om, torch.from_numpy(np.tile(p["betas"], (M, 1))).double(), torch.from_numpy(theta).double(),
                                          torch.from_numpy(np.tile(p["log_beta_scales"], (M, 1))).double())
        verts = (vo + torch.from_numpy(p["trans"]).double()[:, None]).float()
        res = {}
        for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
            v = verts.detach().clone().to(dt).requires_grad_(True)
            sil = so.soft_silhouette(v, om.faces, S)
            sil.sum().backward()
            res[name] = (sil.detach().double().numpy(), v.grad.double().numpy())
        vh = verts.detach().cuda().contiguous()
        sil_h, _ = e.render_forward(vh)
        dv_h = e.render_backward(vh, sil_h, torch.ones_like(sil_h)).double().cpu().numpy()
        res["hip"] = (sil_h.double().cpu().numpy(), dv_h)
        a64 = 1.0 - res["f64"][0]
        rim = (a64 > 0.02) & (a64 < 0.98)
        vn = verts.detach().double().numpy()
        c = vn.mean(1, keepdims=True)
        print("z = %.2f, %d x %d, %d frames, %d rim pixels, coverage %.3f" % (z, S, S, M, int(rim.sum()), float((res["f64"][0] > 0.5).mean())))
        for name in ("f32", "hip"):
            a = 1.0 - res[name][0]
            r = (a[rim] - a64[rim]) / a64[rim]
            g, g64 = res[name][1], res["f64"][1]
            dscale, dscale64 = (g * (vn - c)).sum(), (g64 * (vn - c)).sum()
            dtr, dtr64 = g.sum(1), g64.sum(1)
            print("  %-4s alpha: mean signed rel %+.2e  (rms %.2e, sem %.1e) | sum sil rel %+.2e | d/dscale rel %+.2e | d/dtrans rel-L2 %.2e (z comp. signed %+.2e) | dverts rel-L2 %.2e" %
                  (name, r.mean(), np.sqrt((r ** 2).mean()), r.std() / np.sqrt(r.size), (res[name][0].sum() - res["f64"][0].sum()) / res["f64"][0].sum(),
                   (dscale - dscale64) / abs(dscale64), np.linalg.norm(dtr - dtr64) / np.linalg.norm(dtr64),
                   ((dtr - dtr64)[:, 2] / np.abs(dtr64[:, 2])).mean(), np.linalg.norm(g - g64) / np.linalg.norm(g64)))


if __name__ == "__main__":
    main()
