// Per-element maths of the 3D mesh-fitting objective, shared by kernels_mesh3d.inc and by the
// test-only host shim (tests/host_mesh3d_shim.cpp).  SMALFIT_HD as in smalfit_math.h.
//
// Reference behaviour restated here: fitter_3d/trainer.py:205-227 (Stage.forward) calls PyTorch3D v0.2.5
//   sample_points_from_meshes  -> sample_face / barycentric_sample   (area-weighted face, sqrt-u barycentric map)
//   chamfer_distance           -> nearest_scan                       (squared distance, K = 1, ties -> lowest index)
//   mesh_edge_loss             -> vertex_ring                        (mean |v0 - v1|^2 over unique edges)
//   mesh_laplacian_smoothing   -> vertex_ring / laplacian_adjoint    (uniform weights: | mean of ring - v |)
//   mesh_normal_consistency    -> face_pair_eval                     (1 - cos of the normals either side of an edge,
//                                                                     cos = <n0,n1> rsqrt(max(|n0|^2 |n1|^2, 1e-16)))
#pragma once
#include <math.h>
#include <stdint.h>

#include "smalfit_math.h"

namespace smalfit {

constexpr float kCosEps2 = 1e-16f;   // torch cosine_similarity: eps = 1e-8, clamp on the product of squared norms

// ------------------------------------------------------------------------------------------------
// counter-based random numbers: Philox-4x32-10 (Salmon et al. 2011), one call per sample
// ------------------------------------------------------------------------------------------------
struct Philox4 {
  uint32_t v[4];
};

SMALFIT_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// uniform in [0, 1) with 24 random bits
SMALFIT_HD float unit_float(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// first face whose cumulative-area threshold exceeds r (thresholds: cumulative area / total * 2^32, non-decreasing);
// a zero-area face has the threshold of its predecessor and is never returned
SMALFIT_HD int sample_face(const uint32_t* thr, int F, uint32_t r) {
  int lo = 0, hi = F - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (thr[mid] > r) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// the barycentric map of pytorch3d's _rand_barycentric_coords
SMALFIT_HD void barycentric_sample(const float* a, const float* b, const float* c, float u, float v, float out[3]) {
  const float su = sqrtf(u);
  const float w0 = 1.0f - su, w1 = su * (1.0f - v), w2 = su * v;
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = w0 * a[k] + w1 * b[k] + w2 * c[k];
}

// ------------------------------------------------------------------------------------------------
// chamfer: nearest neighbour scan over the staged points p[begin .. end) (16-byte records: one LDS read per point);
// strict '<' over ascending indices keeps the lowest index among equal distances.  With OWNER the record also carries
// the index of the vertex that point chose in the other direction, and the scan accumulates sum (q - p_j) over the j
// with owner == self: the reverse-direction gradient gathered at the vertex instead of scattered from the points.
// ------------------------------------------------------------------------------------------------
struct alignas(16) ScanPoint {
  float x, y, z;
  int owner;
};

template <bool OWNER>
SMALFIT_HD void nearest_scan(float qx, float qy, float qz, const ScanPoint* p, int begin, int end, int index_base,
                             float& best, int& best_idx, int self, float g[3]) {
#pragma unroll 4
  for (int j = begin; j < end; ++j) {
    const ScanPoint s = p[j];
    const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    if (d2 < best) {
      best = d2;
      best_idx = index_base + j;
    }
    if (OWNER) {
      const float mk = s.owner == self ? 1.0f : 0.0f;   // branch-free: g + 1 * d and g + 0 * d are exact
      g[0] = fmaf(mk, dx, g[0]);
      g[1] = fmaf(mk, dy, g[1]);
      g[2] = fmaf(mk, dz, g[2]);
    }
  }
}

// lexicographic (distance, index) minimum: combines the scans of disjoint index ranges in any order
SMALFIT_HD void nearest_merge(float& best, int& best_idx, float d2, int idx) {
  if (d2 < best || (d2 == best && idx < best_idx)) {
    best = d2;
    best_idx = idx;
  }
}

// ------------------------------------------------------------------------------------------------
// one-ring of a vertex (neighbours through unique edges, ascending): everything the edge and Laplacian terms need
// ------------------------------------------------------------------------------------------------
struct RingEval {
  float edge_sum;      // sum over the ring of |v - u|^2 (every edge is seen from both ends: halve the total)
  float edge_grad[3];  // sum over the ring of (v - u):  d/dv of sum over incident edges |v - u|^2 = 2 * this
  float lap_norm;      // | mean(ring) - v |
  float lap_unit[3];   // (mean(ring) - v) / lap_norm, 0 when the norm is 0 (torch's subgradient of norm at 0)
};

SMALFIT_HD void vertex_ring(const float* verts /* [V][3] of one mesh */, int v, const int* nbr, int deg, RingEval& r) {
  const float x = verts[3 * v], y = verts[3 * v + 1], z = verts[3 * v + 2];
  float sx = 0.f, sy = 0.f, sz = 0.f, es = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
  for (int k = 0; k < deg; ++k) {
    const int u = nbr[k];
    const float ux = verts[3 * u], uy = verts[3 * u + 1], uz = verts[3 * u + 2];
    const float dx = x - ux, dy = y - uy, dz = z - uz;
    es += fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    gx += dx; gy += dy; gz += dz;
    sx += ux; sy += uy; sz += uz;
  }
  r.edge_sum = es;
  r.edge_grad[0] = gx; r.edge_grad[1] = gy; r.edge_grad[2] = gz;
  float lx = 0.f, ly = 0.f, lz = 0.f;
  if (deg > 0) {
    const float inv = 1.0f / (float)deg;
    lx = sx * inv - x; ly = sy * inv - y; lz = sz * inv - z;
  }
  const float n2 = fmaf(lz, lz, fmaf(ly, ly, lx * lx));
  const float n = sqrtf(n2);
  const float in = n > 0.f ? 1.0f / n : 0.f;
  r.lap_norm = n;
  r.lap_unit[0] = lx * in; r.lap_unit[1] = ly * in; r.lap_unit[2] = lz * in;
}

// d(sum_i |L v|_i) / d v_j = -unit_j [deg_j > 0] + sum over the ring i of unit_i / deg_i   (L = D^-1 A - I, symmetric pattern)
SMALFIT_HD void laplacian_adjoint(const float* unit /* [V][3] */, int v, const int* nbr, int deg, const int* nbr_off,
                                  float g[3]) {
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (deg > 0) {
    gx = -unit[3 * v]; gy = -unit[3 * v + 1]; gz = -unit[3 * v + 2];
  }
  for (int k = 0; k < deg; ++k) {
    const int i = nbr[k];
    const float inv = 1.0f / (float)(nbr_off[i + 1] - nbr_off[i]);
    gx = fmaf(unit[3 * i], inv, gx);
    gy = fmaf(unit[3 * i + 1], inv, gy);
    gz = fmaf(unit[3 * i + 2], inv, gz);
  }
  g[0] = gx; g[1] = gy; g[2] = gz;
}

// ------------------------------------------------------------------------------------------------
// normal consistency of the two faces (v0, v1, a) and (v0, v1, b) sharing the edge (v0, v1)
// returns 1 - cos; grad[r][k] = d(1 - cos) / d(vertex r), r = 0..3 for v0, v1, a, b
// ------------------------------------------------------------------------------------------------
SMALFIT_HD void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

SMALFIT_HD float face_pair_eval(const float* p0, const float* p1, const float* pa, const float* pb, float grad[4][3]) {
  float e[3], ea[3], eb[3], n0[3], n1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    e[k] = p1[k] - p0[k];
    ea[k] = pa[k] - p0[k];
    eb[k] = pb[k] - p0[k];
  }
  cross3(e, ea, n0);
  cross3(eb, e, n1);                     // -(e x eb)
  const float w12 = n0[0] * n1[0] + n0[1] * n1[1] + n0[2] * n1[2];
  const float w1 = n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2];
  const float w2 = n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2];
  const float prod = w1 * w2;
  const bool clamped = !(prod > kCosEps2);
  const float r = 1.0f / sqrtf(clamped ? kCosEps2 : prod);
  const float cosv = w12 * r;
  // d cos / d n0 = r n1 - w12 r^3 w2 n0 ; d cos / d n1 = r n0 - w12 r^3 w1 n1 (second terms vanish under the clamp)
  const float r3 = clamped ? 0.f : r * r * r * w12;
  float G0[3], G1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    G0[k] = -(r * n1[k] - r3 * w2 * n0[k]);      // adjoint of n0 for the value 1 - cos
    G1[k] = -(r * n0[k] - r3 * w1 * n1[k]);
  }
  // n0 = e x ea : d/de = ea x G0, d/dea = G0 x e ;  n1 = eb x e : d/deb = e x G1, d/de = G1 x eb
  float de0[3], dea[3], deb[3], de1[3];
  cross3(ea, G0, de0);
  cross3(G0, e, dea);
  cross3(e, G1, deb);
  cross3(G1, eb, de1);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float de = de0[k] + de1[k];
    grad[1][k] = de;
    grad[2][k] = dea[k];
    grad[3][k] = deb[k];
    grad[0][k] = -(de + dea[k] + deb[k]);
  }
  return 1.0f - cosv;
}

}  // namespace smalfit
