// Host-side topology tables of one triangle mesh for the mesh-fitting objective: what PyTorch3D's Meshes object derives
// lazily (edges_packed, laplacian_packed, the face pairs of mesh_normal_consistency) built once, in plain C++, as CSR
// tables that let every gradient be GATHERED per vertex in a fixed order instead of scattered with atomics.
// Pure C++ (no HIP): also compiled by the test-only host shim.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace smalfit {

struct MeshTopologyHost {
  int V = 0, F = 0, E = 0, P = 0;
  std::vector<int> nbr_off;   // [V+1]  one-ring through unique edges
  std::vector<int> nbr;       // [2E]   neighbours of a vertex in ascending order
  std::vector<int> pairs;     // [P][4] (v0, v1, a, b): shared edge v0 < v1 and the two opposite vertices; sorted by edge,
                              //        faces of an edge in ascending face order; an edge with k faces gives k(k-1)/2 rows
  std::vector<int> inc_off;   // [V+1]  vertex -> the (pair * 4 + role) slots it occupies, ascending
  std::vector<int> inc;       // [4P]
};

inline MeshTopologyHost build_mesh_topology(int V, int F, const int* faces) {
  if (V <= 0 || F <= 0 || faces == nullptr) throw std::invalid_argument("mesh topology: empty mesh");
  struct HalfEdge {
    int lo, hi, opposite, face;
  };
  std::vector<HalfEdge> he;
  he.reserve((size_t)3 * F);
  for (int f = 0; f < F; ++f) {
    const int c[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
    for (int k = 0; k < 3; ++k)
      if (c[k] < 0 || c[k] >= V) throw std::invalid_argument("mesh topology: face index out of range");
    if (c[0] == c[1] || c[1] == c[2] || c[2] == c[0]) throw std::invalid_argument("mesh topology: degenerate face (repeated vertex)");
    for (int k = 0; k < 3; ++k) {
      const int p = c[k], q = c[(k + 1) % 3], r = c[(k + 2) % 3];
      he.push_back({std::min(p, q), std::max(p, q), r, f});
    }
  }
  std::sort(he.begin(), he.end(), [](const HalfEdge& a, const HalfEdge& b) {
    if (a.lo != b.lo) return a.lo < b.lo;
    if (a.hi != b.hi) return a.hi < b.hi;
    return a.face < b.face;
  });
  MeshTopologyHost t;
  t.V = V;
  t.F = F;
  std::vector<std::array<int, 2>> edges;
  for (size_t i = 0; i < he.size();) {
    size_t j = i;
    while (j < he.size() && he[j].lo == he[i].lo && he[j].hi == he[i].hi) ++j;
    edges.push_back({he[i].lo, he[i].hi});
    for (size_t x = i; x < j; ++x)
      for (size_t y = x + 1; y < j; ++y) {
        t.pairs.push_back(he[i].lo);
        t.pairs.push_back(he[i].hi);
        t.pairs.push_back(he[x].opposite);
        t.pairs.push_back(he[y].opposite);
      }
    i = j;
  }
  t.E = (int)edges.size();
  t.P = (int)(t.pairs.size() / 4);
  // one-ring CSR
  t.nbr_off.assign(V + 1, 0);
  for (const auto& e : edges) {
    ++t.nbr_off[e[0] + 1];
    ++t.nbr_off[e[1] + 1];
  }
  for (int v = 0; v < V; ++v) t.nbr_off[v + 1] += t.nbr_off[v];
  t.nbr.assign((size_t)2 * t.E, 0);
  {
    std::vector<int> fill(t.nbr_off.begin(), t.nbr_off.end() - 1);
    for (const auto& e : edges) {
      t.nbr[fill[e[0]]++] = e[1];
      t.nbr[fill[e[1]]++] = e[0];
    }
    for (int v = 0; v < V; ++v) std::sort(t.nbr.begin() + t.nbr_off[v], t.nbr.begin() + t.nbr_off[v + 1]);
  }
  // vertex -> pair slots
  t.inc_off.assign(V + 1, 0);
  for (size_t s = 0; s < t.pairs.size(); ++s) ++t.inc_off[t.pairs[s] + 1];
  for (int v = 0; v < V; ++v) t.inc_off[v + 1] += t.inc_off[v];
  t.inc.assign(t.pairs.size(), 0);
  {
    std::vector<int> fill(t.inc_off.begin(), t.inc_off.end() - 1);
    for (size_t s = 0; s < t.pairs.size(); ++s) t.inc[fill[t.pairs[s]]++] = (int)s;
  }
  return t;
}

// cumulative-area thresholds of one target mesh for sample_face(): thr[f] = floor(2^32 * area(0..f) / total), the last
// face with non-zero area and everything after it pinned to 0xFFFFFFFF.  Areas and the running sum in double.
inline std::vector<uint32_t> area_thresholds(int V, const float* verts, int F, const int* faces) {
  std::vector<double> cum((size_t)F);
  double total = 0.0;
  int last_nonzero = -1;
  for (int f = 0; f < F; ++f) {
    for (int k = 0; k < 3; ++k)
      if (faces[3 * f + k] < 0 || faces[3 * f + k] >= V) throw std::invalid_argument("target mesh: face index out of range");
    const float* a = verts + 3 * (size_t)faces[3 * f];
    const float* b = verts + 3 * (size_t)faces[3 * f + 1];
    const float* c = verts + 3 * (size_t)faces[3 * f + 2];
    const double ux = (double)b[0] - a[0], uy = (double)b[1] - a[1], uz = (double)b[2] - a[2];
    const double vx = (double)c[0] - a[0], vy = (double)c[1] - a[1], vz = (double)c[2] - a[2];
    const double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
    const double area = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
    if (area > 0.0) last_nonzero = f;
    total += area;
    cum[f] = total;
  }
  if (!(total > 0.0)) throw std::invalid_argument("target mesh: zero surface area");
  std::vector<uint32_t> thr((size_t)F);
  for (int f = 0; f < F; ++f) {
    const double x = std::floor(cum[f] / total * 4294967296.0);
    thr[f] = (f >= last_nonzero || x >= 4294967295.0) ? 0xFFFFFFFFu : (uint32_t)x;
  }
  return thr;
}

}  // namespace smalfit
