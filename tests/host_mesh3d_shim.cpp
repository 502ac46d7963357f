// TEST-ONLY host build (g++) of the per-element maths and the topology builder of the mesh-fitting objective
// (smalify_amd/csrc/mesh3d_math.h, mesh3d_topology.h), driven the way kernels_mesh3d.inc drives them -- same chunking,
// same slice split and merge, same gather tables -- so that the logic can be checked against the oracle on a machine
// without a GPU.  Never part of the product.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../smalify_amd/csrc/mesh3d_math.h"
#include "../smalify_amd/csrc/mesh3d_topology.h"

using namespace smalfit;

namespace {
constexpr int kChunk = 1024, kSlices = 4;   // kChamChunk and the 4 waves of mesh3d_chamfer_kernel

// one query of mesh3d_chamfer_kernel: chunks of 1024 staged points, each split over 4 slices, merged lexicographically
void chamfer_query(const float q[3], const float* others, int no, const int* owner, int self, float& best, int& bidx,
                   float g[3]) {
  float sb[kSlices];
  int si[kSlices];
  float sg[kSlices][3];
  for (int w = 0; w < kSlices; ++w) {
    sb[w] = INFINITY;
    si[w] = 0x7fffffff;
    sg[w][0] = sg[w][1] = sg[w][2] = 0.f;
  }
  std::vector<ScanPoint> sp(kChunk);
  for (int base = 0; base < no; base += kChunk) {
    const int cnt = std::min(kChunk, no - base);
    for (int i = 0; i < cnt; ++i) {
      sp[i].x = others[3 * (size_t)(base + i)];
      sp[i].y = others[3 * (size_t)(base + i) + 1];
      sp[i].z = others[3 * (size_t)(base + i) + 2];
      sp[i].owner = owner ? owner[base + i] : 0;
    }
    const int per = (cnt + 3) >> 2;
    for (int w = 0; w < kSlices; ++w) {
      const int b = std::min(w * per, cnt), e = std::min(b + per, cnt);
      if (owner) nearest_scan<true>(q[0], q[1], q[2], sp.data(), b, e, base, sb[w], si[w], self, sg[w]);
      else nearest_scan<false>(q[0], q[1], q[2], sp.data(), b, e, base, sb[w], si[w], self, sg[w]);
    }
  }
  best = sb[0];
  bidx = si[0];
  g[0] = sg[0][0]; g[1] = sg[0][1]; g[2] = sg[0][2];
  for (int w = 1; w < kSlices; ++w) {
    nearest_merge(best, bidx, sb[w], si[w]);
    for (int k = 0; k < 3; ++k) g[k] += sg[w][k];
  }
}
}  // namespace

extern "C" {

void hm3_philox(const uint32_t c[4], const uint32_t k[2], uint32_t out[4]) {
  const Philox4 r = philox4x32_10(c[0], c[1], c[2], c[3], k[0], k[1]);
  std::memcpy(out, r.v, sizeof(r.v));
}

// counts[4] = E, P, len(nbr), len(inc); call twice: first with null arrays for the sizes
int hm3_topology(int V, int F, const int* faces, int* counts, int* nbr_off, int* nbr, int* pairs, int* inc_off, int* inc) {
  MeshTopologyHost t;
  try {
    t = build_mesh_topology(V, F, faces);
  } catch (const std::exception&) {
    return 1;
  }
  counts[0] = t.E; counts[1] = t.P; counts[2] = (int)t.nbr.size(); counts[3] = (int)t.inc.size();
  if (nbr_off) std::copy(t.nbr_off.begin(), t.nbr_off.end(), nbr_off);
  if (nbr) std::copy(t.nbr.begin(), t.nbr.end(), nbr);
  if (pairs) std::copy(t.pairs.begin(), t.pairs.end(), pairs);
  if (inc_off) std::copy(t.inc_off.begin(), t.inc_off.end(), inc_off);
  if (inc) std::copy(t.inc.begin(), t.inc.end(), inc);
  return 0;
}

// the 6 launches of smalfit_mesh_objective_eval on the host, float32 throughout
int hm3_eval(int V, int F, const int* faces, int N, const float* lbs_verts, const float* trans, const float* deform,
             const float* points, int S, const float* weights, float* verts, float* losses, float* dverts,
             float* dtrans) {
  MeshTopologyHost t;
  try {
    t = build_mesh_topology(V, F, faces);
  } catch (const std::exception&) {
    return 1;
  }
  const float wc = std::max(weights[0], 0.f), we = std::max(weights[1], 0.f), wn = std::max(weights[2], 0.f),
              wl = std::max(weights[3], 0.f);
  for (size_t i = 0; i < (size_t)N * V * 3; ++i) {
    const int n = (int)(i / ((size_t)V * 3)), k = (int)(i % 3);
    verts[i] = lbs_verts[i] + trans[3 * n + k] + (deform ? deform[i] : 0.f);
  }
  std::vector<int> nn((size_t)N * S);
  std::vector<float> gcham((size_t)N * V * 3, 0.f), gedge((size_t)N * V * 3), unit((size_t)N * V * 3),
      gpair((size_t)N * std::max(t.P, 1) * 12);
  double scx = 0, scy = 0, sed = 0, slp = 0, snm = 0;
  for (int n = 0; n < N; ++n) {
    const float* vn = verts + (size_t)n * V * 3;
    const float* pn = points + (size_t)n * S * 3;
    if (wc > 0.f) {
      for (int i = 0; i < S; ++i) {
        float best, g[3];
        int bi;
        chamfer_query(pn + 3 * i, vn, V, nullptr, i, best, bi, g);
        nn[(size_t)n * S + i] = bi;
        scx += best;
      }
      for (int v = 0; v < V; ++v) {
        float best, g[3];
        int bi;
        chamfer_query(vn + 3 * v, pn, S, nn.data() + (size_t)n * S, v, best, bi, g);
        scy += best;
        const float cy = wc * 2.0f / ((float)V * (float)N), cx = wc * 2.0f / ((float)S * (float)N);
        for (int k = 0; k < 3; ++k) gcham[((size_t)n * V + v) * 3 + k] = cy * (vn[3 * v + k] - pn[3 * bi + k]) + cx * g[k];
      }
    }
    for (int v = 0; v < V; ++v) {
      RingEval r;
      vertex_ring(vn, v, t.nbr.data() + t.nbr_off[v], t.nbr_off[v + 1] - t.nbr_off[v], r);
      sed += r.edge_sum;
      slp += r.lap_norm;
      const float ce = we * 2.0f / ((float)t.E * (float)N);
      for (int k = 0; k < 3; ++k) {
        gedge[((size_t)n * V + v) * 3 + k] = ce * r.edge_grad[k];
        unit[((size_t)n * V + v) * 3 + k] = r.lap_unit[k];
      }
    }
    for (int p = 0; p < t.P; ++p) {
      const int* row = t.pairs.data() + 4 * (size_t)p;
      float grad[4][3];
      snm += face_pair_eval(vn + 3 * row[0], vn + 3 * row[1], vn + 3 * row[2], vn + 3 * row[3], grad);
      std::memcpy(gpair.data() + ((size_t)n * t.P + p) * 12, grad, sizeof(grad));
    }
    double tr[3] = {0, 0, 0};
    for (int v = 0; v < V; ++v) {
      float gl[3], gn[3] = {0.f, 0.f, 0.f};
      laplacian_adjoint(unit.data() + (size_t)n * V * 3, v, t.nbr.data() + t.nbr_off[v], t.nbr_off[v + 1] - t.nbr_off[v],
                        t.nbr_off.data(), gl);
      for (int s = t.inc_off[v]; s < t.inc_off[v + 1]; ++s)
        for (int k = 0; k < 3; ++k) gn[k] += gpair[(size_t)n * t.P * 12 + 3 * (size_t)t.inc[s] + k];
      const float cl = wl / ((float)V * (float)N), cn = t.P > 0 ? wn / ((float)t.P * (float)N) : 0.f;
      for (int k = 0; k < 3; ++k) {
        const size_t o = ((size_t)n * V + v) * 3 + k;
        dverts[o] = gcham[o] + gedge[o] + cl * gl[k] + cn * gn[k];
        tr[k] += dverts[o];
      }
    }
    for (int k = 0; k < 3; ++k) dtrans[3 * n + k] = (float)tr[k];
  }
  const float fn = (float)N;
  const float ch = wc > 0.f ? (float)(scx / ((double)S * fn) + scy / ((double)V * fn)) : 0.f;
  losses[0] = ch;
  losses[1] = (float)(sed / (2.0 * t.E * fn));
  losses[2] = t.P > 0 ? (float)(snm / ((double)t.P * fn)) : 0.f;
  losses[3] = (float)(slp / ((double)V * fn));
  losses[4] = wc * ch + we * losses[1] + wn * losses[2] + wl * losses[3];
  return 0;
}

// mesh3d_sample_kernel for one target mesh; also returns the chosen faces
int hm3_sample(int V, const float* verts, int F, const int* faces, int S, unsigned long long seed, unsigned iteration,
               int mesh_index, float* points, int* chosen) {
  std::vector<uint32_t> thr;
  try {
    thr = area_thresholds(V, verts, F, faces);
  } catch (const std::exception&) {
    return 1;
  }
  for (int i = 0; i < S; ++i) {
    const Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)mesh_index, iteration, 0u, (uint32_t)(seed & 0xFFFFFFFFull),
                                    (uint32_t)(seed >> 32));
    const int f = sample_face(thr.data(), F, r.v[0]);
    chosen[i] = f;
    barycentric_sample(verts + 3 * (size_t)faces[3 * f], verts + 3 * (size_t)faces[3 * f + 1],
                       verts + 3 * (size_t)faces[3 * f + 2], unit_float(r.v[1]), unit_float(r.v[2]), points + 3 * (size_t)i);
  }
  return 0;
}

}  // extern "C"
