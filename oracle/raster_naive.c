/* ORACLE (test infrastructure, never linked into the product): literal single-threaded float32
 * restatement of the rasteriser + blend the reference's Renderer runs on CPU.
 *
 * The arithmetic lives in the third-party dependency pytorch3d==0.2.5 (reference requirements.txt:60),
 * absent from /root/reference and not installable here.  This file restates that release's published
 * algorithm, call sites: reference smal_fitter/p3d_renderer.py:26-39 (settings), :65-66 (call):
 *   - RasterizeMeshesNaiveCpu: for every pixel, for every face: bbox test with sqrt(blur) margin,
 *     barycentrics (area + kEpsilon), depth test pz >= 0, squared point-triangle distance,
 *     keep the K = faces_per_pixel nearest in depth           
 *   - sigmoid_alpha_blend: alpha = prod_k (1 - sigmoid(-d_k / sigma)), silhouette = 1 - alpha
 *   - backward: exact gradient of the forward (distance through the nearest edge with clamped t); with unclamped_t != 0 the
 *     gradient with t left unclamped instead (last row: what some 0.2.x sources are recalled to do in
 *     PointLineDistanceBackward) -- same forward, same nearest edge (first minimum: a-b, a-c, b-c)
 * "parity unpinned": no vector produced by pytorch3d itself is available (see oracle/smal_oracle.py).
 *
 * Build: gcc -O2 -shared -fPIC oracle/raster_naive.c -o oracle/_build/libraster_naive.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define K_EPS 1e-8f

static float edge_fn(float px, float py, float ux, float uy, float wx, float wy) {
  return (px - ux) * (wy - uy) - (py - uy) * (wx - ux);
}

/* squared distance point -> segment (u,w); also returns clamped t (and the unclamped one; a degenerate edge: its end point w, t = 1) */
static float t_raw_last;
static float seg_dist2(float px, float py, float ux, float uy, float wx, float wy, float* t_out) {
  const float ex = wx - ux, ey = wy - uy;
  const float l2 = ex * ex + ey * ey;
  if (l2 <= K_EPS) {
    *t_out = 1.0f;
    t_raw_last = 1.0f;
    return (px - wx) * (px - wx) + (py - wy) * (py - wy);
  }
  float t = ((px - ux) * ex + (py - uy) * ey) / l2;
  t_raw_last = t;
  if (t < 0.0f) t = 0.0f;
  if (t > 1.0f) t = 1.0f;
  *t_out = t;
  const float qx = ux + t * ex - px, qy = uy + t * ey - py;
  return qx * qx + qy * qy;
}

/* v: (V,3) = (x_ndc, y_ndc, z_view); faces: (F,3).
 * Outputs (any may be NULL except sil): p2f (S,S,K) face ids (-1 = empty), zbuf, dists (signed), sil (S,S) */
void raster_naive_forward(const float* v, int V, const int* faces, int F, int S, float blur, int K,
                          float sigma, int* p2f, float* zbuf, float* dists, float* sil, int* max_candidates) {
  (void)V;
  const float r = sqrtf(blur);
  int* qf = (int*)malloc(sizeof(int) * (size_t)(K + 1));
  float* qz = (float*)malloc(sizeof(float) * (size_t)(K + 1));
  float* qd = (float*)malloc(sizeof(float) * (size_t)(K + 1));
  int maxc = 0;
  for (int yi = 0; yi < S; ++yi) {
    const float yf = 1.0f - (2.0f * (float)yi + 1.0f) / (float)S; /* +Y up: row 0 is the top */
    for (int xi = 0; xi < S; ++xi) {
      const float xf = 1.0f - (2.0f * (float)xi + 1.0f) / (float)S; /* +X left */
      int qn = 0, cand = 0;
      for (int f = 0; f < F; ++f) {
        const float* a = v + 3 * faces[3 * f];
        const float* b = v + 3 * faces[3 * f + 1];
        const float* c = v + 3 * faces[3 * f + 2];
        const float area = edge_fn(c[0], c[1], a[0], a[1], b[0], b[1]);
        if (area <= K_EPS && area >= -K_EPS) continue;
        const float zmax = fmaxf(a[2], fmaxf(b[2], c[2]));
        if (zmax < 0.0f) continue;
        const float xmin = fminf(a[0], fminf(b[0], c[0])) - r, xmax = fmaxf(a[0], fmaxf(b[0], c[0])) + r;
        const float ymin = fminf(a[1], fminf(b[1], c[1])) - r, ymax = fmaxf(a[1], fmaxf(b[1], c[1])) + r;
        if (xf < xmin || xf > xmax || yf < ymin || yf > ymax) continue;
        const float den = area + K_EPS;
        const float w0 = edge_fn(xf, yf, b[0], b[1], c[0], c[1]) / den;
        const float w1 = edge_fn(xf, yf, c[0], c[1], a[0], a[1]) / den;
        const float w2 = edge_fn(xf, yf, a[0], a[1], b[0], b[1]) / den;
        const float pz = w0 * a[2] + w1 * b[2] + w2 * c[2];
        if (pz < 0.0f) continue;
        float t;
        float d = seg_dist2(xf, yf, a[0], a[1], b[0], b[1], &t);
        float d2 = seg_dist2(xf, yf, a[0], a[1], c[0], c[1], &t);
        float d3 = seg_dist2(xf, yf, b[0], b[1], c[0], c[1], &t);
        if (d2 < d) d = d2;
        if (d3 < d) d = d3;
        const int inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
        if (!inside && d >= blur) continue;
        ++cand;
        const float sd = inside ? -d : d;
        /* bounded priority queue ordered by depth: insert, drop the farthest when over capacity */
        int i = qn;
        while (i > 0 && qz[i - 1] > pz) { qz[i] = qz[i - 1]; qf[i] = qf[i - 1]; qd[i] = qd[i - 1]; --i; }
        qz[i] = pz; qf[i] = f; qd[i] = sd;
        if (qn < K) ++qn;
      }
      if (cand > maxc) maxc = cand;
      float alpha = 1.0f;
      for (int k = 0; k < K; ++k) {
        const size_t o = ((size_t)yi * S + xi) * K + k;
        if (k < qn) {
          alpha *= 1.0f - 1.0f / (1.0f + expf(qd[k] / sigma)); /* 1 - sigmoid(-d/sigma) */
          if (p2f) p2f[o] = qf[k];
          if (zbuf) zbuf[o] = qz[k];
          if (dists) dists[o] = qd[k];
        } else {
          if (p2f) p2f[o] = -1;
          if (zbuf) zbuf[o] = -1.0f;
          if (dists) dists[o] = -1.0f;
        }
      }
      sil[(size_t)yi * S + xi] = 1.0f - alpha;
    }
  }
  if (max_candidates) *max_candidates = maxc;
  free(qf); free(qz); free(qd);
}

/* gradient of sum(grad_sil * sil) with respect to the 2-D NDC vertex positions: gv (V,2), zero-initialised
 * by the caller.  Uses the fragments (p2f, dists) of the forward. */
void raster_naive_backward(const float* v, const int* faces, int S, int K, float sigma, const int* p2f,
                           const float* dists, const float* grad_sil, double* gv, int unclamped_t) {
  for (int yi = 0; yi < S; ++yi) {
    const float yf = 1.0f - (2.0f * (float)yi + 1.0f) / (float)S;
    for (int xi = 0; xi < S; ++xi) {
      const float xf = 1.0f - (2.0f * (float)xi + 1.0f) / (float)S;
      const size_t base = ((size_t)yi * S + xi) * K;
      double alpha = 1.0;
      for (int k = 0; k < K && p2f[base + k] >= 0; ++k) alpha *= 1.0 - 1.0 / (1.0 + exp((double)dists[base + k] / sigma));
      const double g = grad_sil[(size_t)yi * S + xi];
      if (g == 0.0) continue;
      for (int k = 0; k < K && p2f[base + k] >= 0; ++k) {
        const int f = p2f[base + k];
        const double sd = dists[base + k];
        const double p = 1.0 / (1.0 + exp(sd / sigma));
        /* d sil / d sd = -alpha * p / sigma */
        const double gsd = g * (-alpha * p / sigma);
        const double gdist = (sd < 0.0) ? -gsd : gsd;
        const int ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
        const int idx[3][2] = {{ia, ib}, {ia, ic}, {ib, ic}};
        int best = 0;
        float bt = 0.0f, bd = 0.0f;
        for (int e = 0; e < 3; ++e) {
          float t;
          const float d = seg_dist2(xf, yf, v[3 * idx[e][0]], v[3 * idx[e][0] + 1], v[3 * idx[e][1]], v[3 * idx[e][1] + 1], &t);
          if (e == 0 || d < bd) { bd = d; bt = unclamped_t ? t_raw_last : t; best = e; }
        }
        const int iu = idx[best][0], iw = idx[best][1];
        const double qx = (double)xf - ((double)v[3 * iu] + bt * ((double)v[3 * iw] - v[3 * iu]));
        const double qy = (double)yf - ((double)v[3 * iu + 1] + bt * ((double)v[3 * iw + 1] - v[3 * iu + 1]));
        gv[2 * iu] += gdist * -2.0 * (1.0 - bt) * qx;
        gv[2 * iu + 1] += gdist * -2.0 * (1.0 - bt) * qy;
        gv[2 * iw] += gdist * -2.0 * bt * qx;
        gv[2 * iw + 1] += gdist * -2.0 * bt * qy;
      }
    }
  }
}
