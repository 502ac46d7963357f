"""Developer analysis (CPU, not a test): how often would the rasteriser's cached per-pixel depth bounds survive one
optimiser step, under different ways of carrying the bounds from one iteration to the next?

The HIP rasteriser keeps, per pixel with more than K = 100 candidates, two depths lo < z_K <= hi around the depth of the
K-th nearest candidate (smalify_amd/csrc/kernels_raster.inc, raster_select_kernel / raster_band_kernel).  The next
evaluation is exact without a new selection iff  #{z <= lo} <= K <= #{z <= hi}  and  #{lo < z <= hi} <= band capacity.
This script runs the oracle's fit on a few frames of the benchmark problem with bench.py's scaled schedule, records
every candidate depth per pixel per iteration, and replays that rule with the bounds
  none         left where they were (what the kernels do; reproduces the hit rates measured on the GPU),
  affine_dz    shifted by an affine function of the pixel fitted to the vertices' depth changes,
  warp_mean    looked up at the pixel the mesh's mean screen motion came from (+ affine_dz),
  warp_affine  the same with an affine screen motion fitted to the vertices,
  anchor_k     tied to the depth plane of the face that was the K-th nearest at the last selection,
  anchor_flat  tied to the flattest face within 12 ranks of the K-th.
Findings are summarised in DESIGN.md section 5.

usage: [SIM_VARIANTS=none,warp_affine] [SIM_FILL=60 SIM_CAP=64 SIM_HALF=8 SIM_TRIES=6] python tests/band_policy_sim.py [frames=2] [steps=20]
       (SIM_FILL: most entries a new band may hold, SIM_CAP: band list capacity, SIM_HALF: initial half-width in mean
       depth gaps of the K nearest, SIM_TRIES: halvings tried)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import smal_oracle as so                      # noqa: E402
from smalify_amd import config, model_io, synthetic       # noqa: E402
import bench                                              # noqa: E402

K = so.FACES_PER_PIXEL
S = 256
BAND_CAP = int(os.environ.get('SIM_CAP', '64'))


def candidates(verts, faces_np):
    """per frame: (pix sorted, z sorted within pixel) of every valid candidate, and the vertices' (x, y, z) in ndc"""
    xn, yn, zv = so.world_to_ndc(verts)
    x, y, z = xn.double().numpy(), yn.double().numpy(), zv.double().numpy()
    pix, fidx = so._candidate_pairs(x, y, faces_np, S, so.BLUR_RADIUS)
    f = faces_np[fidx]
    ax, ay, az = x[f[:, 0]], y[f[:, 0]], z[f[:, 0]]
    bx, by, bz = x[f[:, 1]], y[f[:, 1]], z[f[:, 1]]
    cx, cy, cz = x[f[:, 2]], y[f[:, 2]], z[f[:, 2]]
    px = 1.0 - (2.0 * (pix % S) + 1.0) / S
    py = 1.0 - (2.0 * (pix // S) + 1.0) / S
    area = (cx - ax) * (by - ay) - (cy - ay) * (bx - ax)
    den = area + so.K_EPS
    w0 = ((px - bx) * (cy - by) - (py - by) * (cx - bx)) / den
    w1 = ((px - cx) * (ay - cy) - (py - cy) * (ax - cx)) / den
    w2 = ((px - ax) * (by - ay) - (py - ay) * (bx - ax)) / den
    inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
    pz = w0 * az + w1 * bz + w2 * cz

    def seg(ux, uy, vx, vy):
        ex, ey = vx - ux, vy - uy
        l2 = np.maximum(ex * ex + ey * ey, 1e-30)
        t = np.clip(((px - ux) * ex + (py - uy) * ey) / l2, 0.0, 1.0)
        qx, qy = ux + t * ex - px, uy + t * ey - py
        return qx * qx + qy * qy
    dist = np.minimum(np.minimum(seg(ax, ay, bx, by), seg(ax, ay, cx, cy)), seg(bx, by, cx, cy))
    valid = (np.abs(area) > so.K_EPS) & (pz >= 0) & (inside | (dist < so.BLUR_RADIUS))
    # |d pz / d pixel| of the face's depth plane (ndc units per ndc unit)
    gzx = ((cz - az) * (by - ay) + (az - bz) * (cy - ay)) / den
    gzy = -((cz - az) * (bx - ax) + (az - bz) * (cx - ax)) / den
    gmag = np.hypot(gzx, gzy)
    pix, pz, fidx, gmag = pix[valid], pz[valid], fidx[valid], gmag[valid]
    order = np.lexsort((pz, pix))
    return pix[order], pz[order], np.stack([x, y, z], 1), fidx[order], gmag[order]


def plane_depth(xyz, faces_np, fid, pix):
    """depth of face fid's plane extrapolated to the centre of pixel pix (any pixel, inside the face or not)"""
    f = faces_np[fid]
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    ax, ay, az = x[f[:, 0]], y[f[:, 0]], z[f[:, 0]]
    bx, by, bz = x[f[:, 1]], y[f[:, 1]], z[f[:, 1]]
    cx, cy, cz = x[f[:, 2]], y[f[:, 2]], z[f[:, 2]]
    px = 1.0 - (2.0 * (pix % S) + 1.0) / S
    py = 1.0 - (2.0 * (pix // S) + 1.0) / S
    den = (cx - ax) * (by - ay) - (cy - ay) * (bx - ax) + so.K_EPS
    w0 = ((px - bx) * (cy - by) - (py - by) * (cx - bx)) / den
    w1 = ((px - cx) * (ay - cy) - (py - cy) * (ax - cx)) / den
    w2 = ((px - ax) * (by - ay) - (py - ay) * (bx - ax)) / den
    return w0 * az + w1 * bz + w2 * cz


def per_pixel(pix, z):
    """start offsets / counts of each pixel's run in the sorted candidate arrays"""
    cnt = np.bincount(pix, minlength=S * S)
    start = np.cumsum(cnt) - cnt
    return start, cnt


def count_le(pix_start, pix_cnt, z, thr):
    """#{candidates of pixel p with z <= thr[p]} for every pixel (z sorted within each pixel's run)"""
    out = np.zeros(S * S, np.int64)
    for p in np.nonzero(pix_cnt)[0]:
        s = pix_start[p]
        out[p] = np.searchsorted(z[s:s + pix_cnt[p]], thr[p], side="right")
    return out


def select_bounds(start, cnt, z, fill, fid=None, gmag=None, anchors=None):
    """the exact selection's new bounds: delta0 = 8 mean gaps of the K nearest, halved until the band holds <= fill"""
    lo = np.full(S * S, np.inf)
    hi = np.full(S * S, np.inf)
    zk = np.full(S * S, np.inf)
    for p in np.nonzero(cnt > K)[0]:
        zz = z[start[p]:start[p] + cnt[p]]
        k = zz[K - 1]
        zk[p] = k
        if anchors is not None:
            anchors["anchor_k"][p] = fid[start[p] + K - 1]
            a0, a1 = max(0, K - 1 - 12), min(cnt[p], K + 12)
            anchors["anchor_flat"][p] = fid[start[p] + a0 + int(np.argmin(gmag[start[p] + a0:start[p] + a1]))]
        delta = float(os.environ.get('SIM_HALF', '8')) * (k - zz[0]) / K
        chosen = False
        for _ in range(int(os.environ.get('SIM_TRIES', '6'))):
            a = np.searchsorted(zz, k - delta, side="right")
            b = np.searchsorted(zz, k + delta, side="right")
            if b - a <= fill and delta > 0:
                lo[p], hi[p] = k - delta, (np.inf if b == len(zz) else k + delta)
                chosen = True
                break
            delta *= 0.5
        if not chosen:
            nxt = zz[K] if len(zz) > K else k
            lo[p] = hi[p] = 0.5 * (k + nxt)
    return lo, hi, zk


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    torch.set_num_threads(16)
    md = synthetic.synthetic_model(seed=0, shape_family_id=1)
    om = so.OracleModel(md, dtype=torch.float32)
    faces_np = np.asarray(md["faces"] if isinstance(md, dict) else md.faces).astype(np.int64)
    pose_prior = synthetic.synthetic_pose_prior()
    sp = synthetic.synthetic_shape_prior()
    gt = synthetic.ground_truth_params(64, seed=1234, mean_betas=sp[1][:20], mean_logscale=sp[1][20:26])
    f0 = 8
    sel = slice(f0, f0 + nf)
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()      # noqa: E731
    theta = torch.cat([t32(gt["global_rotation"][sel])[:, None], t32(gt["joint_rotations"][sel])], 1)
    ls = t32(gt["log_beta_scales"])
    ls = ls[None].expand(nf, -1) if ls.dim() == 1 else ls[sel]
    verts, joints, _, _ = so.smal_forward(om, t32(gt["betas"])[None].expand(nf, -1), theta, ls)
    verts = verts + t32(gt["trans"][sel])[:, None]
    joints = joints + t32(gt["trans"][sel])[:, None]
    noise, vis = synthetic.keypoint_noise_and_visibility(64)
    tj = so.project_points(joints[:, so.CANONICAL], S) + t32(noise[sel])
    tsil = (so.soft_silhouette(verts, om.faces, S) > 0.5).float()
    prob = so.FitProblem(om, S, tj, t32(vis[sel]), tsil, pose_prior[0], pose_prior[1], pose_prior[2], sp[0], sp[1],
                         min(8, nf), True, dtype=torch.float32)
    params = dict(betas=t32(sp[1][:20].copy()), log_beta_scales=t32(sp[1][20:26].copy()),
                  global_rotation=t32(np.tile(model_io.initial_global_rotation(), (nf, 1))),
                  trans=torch.zeros(nf, 3), joint_rotations=torch.zeros(nf, 34, 3))
    W = np.array(config.OPT_WEIGHTS).T
    sched = bench.scaled_schedule(steps)

    def frame_verts(params):
        th = torch.cat([params["global_rotation"][:, None], params["joint_rotations"]], 1)
        l2 = params["log_beta_scales"]
        l2 = l2[None].expand(nf, -1) if l2.dim() == 1 else l2
        v, _, _, _ = so.smal_forward(om, params["betas"][None].expand(nf, -1), th, l2)
        return (v + params["trans"][:, None]).detach()

    # trajectory of vertex positions at every silhouette evaluation (the initial state is evaluated once: the priming call)
    traj = [("prime", frame_verts(params))]
    for stage, its in enumerate(sched):
        w = W[stage]
        opt = so.Adam(so.PARAM_ORDER, lr=float(w[8]))
        names = so.trainable_names(stage)
        visb = so.stage0_visibility(prob.vis) if stage == 0 else None
        for it in range(its):
            if w[1] > 0:
                traj.append(("s%d.%d" % (stage, it), frame_verts(params)))
            _, _, grads = so.loss_and_grads(prob, params, w[:6], float(w[6]), names, visb)
            opt.step(params, grads)
    print("schedule", sched, "silhouette evaluations", len(traj))

    FILL = int(os.environ.get('SIM_FILL', '60'))
    state = [None] * nf                 # per frame: dict(lo, hi) per variant
    prev = [None] * nf
    variants = tuple(os.environ.get("SIM_VARIANTS", "none,warp_affine").split(","))
    anchor = [None] * nf
    for label, verts in traj:
        row = {}
        for n in range(nf):
            pix, z, xyz, fid, gmag = candidates(verts[n], faces_np)
            start, cnt = per_pixel(pix, z)
            big = cnt > K
            new_anchor = {"anchor_k": np.full(S * S, -1, np.int64), "anchor_flat": np.full(S * S, -1, np.int64)}
            new_lo, new_hi, zk = select_bounds(start, cnt, z, FILL, fid, gmag, new_anchor)
            if state[n] is not None:
                dz = xyz[:, 2] - prev[n][:, 2]
                A = np.stack([np.ones(len(dz)), prev[n][:, 0], prev[n][:, 1]], 1)
                coef = np.linalg.lstsq(A, dz, rcond=None)[0]
                cc, rr = np.meshgrid(np.arange(S), np.arange(S))
                pxs = (1.0 - (2.0 * cc + 1.0) / S).ravel()
                pys = (1.0 - (2.0 * rr + 1.0) / S).ravel()
                adz = coef[0] + coef[1] * pxs + coef[2] * pys
                shift = {"none": np.zeros(S * S), "affine_dz": adz, "warp_mean": adz, "warp_affine": adz}
                for v in ("anchor_k", "anchor_flat"):
                    if v not in variants:
                        continue
                    a = anchor[n][v]
                    pp = np.nonzero(a >= 0)[0]
                    sh = np.zeros(S * S)
                    sh[pp] = plane_depth(xyz, faces_np, a[pp], pp) - plane_depth(prev[n], faces_np, a[pp], pp)
                    shift[v] = sh
                # screen motion of the vertices: x_new = M [1, x_old, y_old]; a pixel of the new image at x_new looks up the
                # old bounds at x_old (nearest pixel)
                Bm = np.stack([np.ones(len(dz)), xyz[:, 0], xyz[:, 1]], 1)
                inv = np.linalg.lstsq(Bm, prev[n][:, :2], rcond=None)[0]          # old position as affine function of the new
                mean_d = (prev[n][:, :2] - xyz[:, :2]).mean(0)
                src = {}
                for v, (ox, oy) in (("warp_mean", (pxs + mean_d[0], pys + mean_d[1])),
                                    ("warp_affine", (inv[0, 0] + inv[1, 0] * pxs + inv[2, 0] * pys, inv[0, 1] + inv[1, 1] * pxs + inv[2, 1] * pys))):
                    sc = np.clip(np.rint(((1.0 - ox) * S - 1.0) / 2.0), 0, S - 1).astype(np.int64)
                    sr = np.clip(np.rint(((1.0 - oy) * S - 1.0) / 2.0), 0, S - 1).astype(np.int64)
                    src[v] = sr * S + sc
                if n == 0:
                    print("   mean screen motion (px): %.2f %.2f" % (mean_d[0] * S / 2, mean_d[1] * S / 2), "rms vertex motion (px): %.2f"
                          % (np.sqrt(((prev[n][:, :2] - xyz[:, :2]) ** 2).sum(1).mean()) * S / 2), "rms dz %.4f" % np.sqrt((dz ** 2).mean()))
                for v in variants:
                    olo, ohi = state[n][v]
                    if v in src:
                        olo, ohi = olo[src[v]], ohi[src[v]]
                    lo = olo + shift[v]
                    hi = ohi + shift[v]
                    had = np.isfinite(olo)
                    c = count_le(start, cnt, z, lo)
                    ch = count_le(start, cnt, z, np.where(np.isfinite(hi), hi, 1e30))
                    b = ch - c
                    ok_b = had & (c <= K) & (K <= ch) & (b <= BAND_CAP)
                    ok_all = had & ~np.isfinite(hi) & (ch < K)
                    nob = ~had & (cnt <= K)
                    hit = ok_b | ok_all | nob
                    tracked = had | big
                    r = row.setdefault(v, [0, 0, 0, 0, 0])
                    r[0] += int((hit & tracked).sum()); r[1] += int(tracked.sum())
                    r[2] += int((had & (c > K)).sum()); r[3] += int((had & (c <= K) & (b > BAND_CAP)).sum())
                    r[4] += int((had & (c <= K) & (b <= BAND_CAP) & (ch < K) & np.isfinite(hi)).sum())
                    # bounds after this evaluation: hits re-centre on the new K-th keeping the half-width, misses are re-selected
                    half = 0.5 * (hi - lo)
                    keep = hit & had & big & np.isfinite(hi)
                    nl = np.where(keep, zk - half, new_lo)
                    nh = np.where(keep, zk + half, new_hi)
                    keep_inf = hit & had & big & ~np.isfinite(hi)
                    nl = np.where(keep_inf, lo, nl)
                    nh = np.where(keep_inf, hi, nh)
                    state[n][v] = (nl, nh)
                    if v in new_anchor and v in variants:
                        anchor[n][v] = np.where(hit & had & big, anchor[n][v], new_anchor[v])
            else:
                state[n] = {v: (new_lo, new_hi) for v in variants}
                anchor[n] = {v: new_anchor[v].copy() for v in new_anchor}
            prev[n] = xyz
        if row:
            print(label, " ".join("%s: hit %.1f%% (c>K %d, b>cap %d, beyond hi %d of %d)" % (v, 100.0 * r[0] / max(r[1], 1), r[2], r[3], r[4], r[1])
                                  for v, r in row.items()))


if __name__ == "__main__":
    main()
