// TEST-ONLY host build of smalify_amd/csrc/smalfit_math.h (g++), so that the per-element maths the HIP
// kernels use can be checked against the oracle on a machine without a GPU.  Never part of the product.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../smalify_amd/csrc/smalfit_math.h"

using namespace smalfit;

extern "C" {

void hm_rodrigues(int n, const float* th, const float* G, float* R, float* dth) {
  for (int i = 0; i < n; ++i) {
    rodrigues_fwd(th + 3 * i, R + 9 * i);
    rodrigues_bwd(th + 3 * i, G + 9 * i, dth + 3 * i);
  }
}

// emulates raster_fwd_kernel + raster_bwd_kernel semantics without binning:
// sil (S,S), zthr (S,S); gv (V,2) = d sum(w*sil) / d ndc xy
void hm_raster(const float* v, int V, const int* faces, int F, int S, const float* w, float* sil, float* zthr,
               double* gv) {
  std::vector<FaceRec> recs(F);
  std::vector<char> ok(F);
  for (int f = 0; f < F; ++f) {
    const float* a = v + 3 * faces[3 * f];
    const float* b = v + 3 * faces[3 * f + 1];
    const float* c = v + 3 * faces[3 * f + 2];
    ok[f] = make_face_rec(a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], recs[f]);
  }
  const float inv_s = 1.0f / (float)S;
  std::vector<float> alpha_img((size_t)S * S);
  for (int row = 0; row < S; ++row)
    for (int col = 0; col < S; ++col) {
      const float px = pix_to_ndc(col, inv_s), py = pix_to_ndc(row, inv_s);
      std::vector<std::pair<float, float>> cand;
      for (int f = 0; f < F; ++f) {
        if (!ok[f]) continue;
        PixEval e;
        if (face_pixel_eval(recs[f], px, py, e)) cand.push_back({e.pz, e.d});
      }
      float zt = INFINITY;
      if ((int)cand.size() >= kFacesPerPixel) {
        std::vector<float> zs;
        for (auto& c : cand) zs.push_back(c.first);
        std::nth_element(zs.begin(), zs.begin() + kFacesPerPixel - 1, zs.end());
        zt = zs[kFacesPerPixel - 1];
      }
      float alpha = 1.0f;
      for (auto& c : cand)
        if (c.first <= zt) alpha *= one_minus_prob(c.second);
      sil[row * S + col] = 1.0f - alpha;
      zthr[row * S + col] = zt;
      alpha_img[row * S + col] = alpha;
    }
  for (int f = 0; f < F; ++f) {
    if (!ok[f]) continue;
    for (int row = 0; row < S; ++row)
      for (int col = 0; col < S; ++col) {
        const float g = -w[row * S + col] * alpha_img[row * S + col] * (1.0f / kSigma);
        if (g == 0.f) continue;
        PixEval e;
        if (!face_pixel_eval(recs[f], pix_to_ndc(col, inv_s), pix_to_ndc(row, inv_s), e)) continue;
        if (e.pz > zthr[row * S + col]) continue;
        const float gd = g * prob(e.d) * (e.inside ? -1.0f : 1.0f) * -2.0f;
        const float ku = 1.0f - e.tc, kw = e.tc;
        const float ca = (e.edge == 2) ? 0.f : ku;
        const float cb = (e.edge == 0) ? kw : ((e.edge == 2) ? ku : 0.f);
        const float cc = (e.edge == 0) ? 0.f : kw;
        const int ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
        gv[2 * ia] += ca * gd * e.qx; gv[2 * ia + 1] += ca * gd * e.qy;
        gv[2 * ib] += cb * gd * e.qx; gv[2 * ib + 1] += cb * gd * e.qy;
        gv[2 * ic] += cc * gd * e.qx; gv[2 * ic + 1] += cc * gd * e.qy;
      }
  }
}

// face_pixel_candidate (the forward sweep's / selection's form) against face_pixel_eval (the backward's form) on every
// (face, pixel) pair of a mesh: counts pairs on which the candidate decision or the bits of the signed distance differ
int hm_candidate_mismatches(const float* v, const int* faces, int F, int S, long long* pairs_out) {
  const float inv_s = 1.0f / (float)S;
  int bad = 0;
  long long pairs = 0;
  for (int f = 0; f < F; ++f) {
    const float* a = v + 3 * faces[3 * f];
    const float* b = v + 3 * faces[3 * f + 1];
    const float* c = v + 3 * faces[3 * f + 2];
    FaceRec r;
    if (!make_face_rec(a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], r)) continue;
    for (int row = 0; row < S; ++row)
      for (int col = 0; col < S; ++col) {
        const float px = pix_to_ndc(col, inv_s), py = pix_to_ndc(row, inv_s);
        PixEval e;
        float d2;
        const bool k1 = face_pixel_eval(r, px, py, e), k2 = face_pixel_candidate(r, px, py, d2);
        ++pairs;
        if (k1 != k2 || (k1 && (e.d != d2 || e.pz != face_pixel_depth(r, px, py)))) ++bad;
      }
  }
  *pairs_out = pairs;
  return bad;
}

// global_rigid_kernel on the host: one call of global_rigid_frame per frame
void hm_global_rigid(int n, const float* Rs, const float* Js, const int* parents, const float* logscale, float* newJ, float* A) {
  for (int i = 0; i < n; ++i)
    global_rigid_frame(Rs + (size_t)i * 315, Js + (size_t)i * 105, parents, logscale ? logscale + (size_t)i * 6 : nullptr,
                       newJ + (size_t)i * 105, A + (size_t)i * 560);
}

void hm_global_rigid_bwd(int n, const float* Rs, const float* Js, const int* parents, const float* logscale, const float* dnewJ,
                         const float* dA, float* dRs, float* dJs, float* dls) {
  std::vector<float> scratch(840);
  for (int i = 0; i < n; ++i)
    global_rigid_frame_bwd(Rs + (size_t)i * 315, Js + (size_t)i * 105, parents, logscale ? logscale + (size_t)i * 6 : nullptr,
                           dnewJ + (size_t)i * 105, dA + (size_t)i * 560, scratch.data(), scratch.data() + 420,
                           dRs + (size_t)i * 315, dJs + (size_t)i * 105, logscale ? dls + (size_t)i * 6 : nullptr);
}

void hm_camera(int n, const float* p, const float* g2, float* ndc, float* g3) {
  for (int i = 0; i < n; ++i) {
    world_to_ndc(p[3 * i], p[3 * i + 1], p[3 * i + 2], ndc[3 * i], ndc[3 * i + 1], ndc[3 * i + 2]);
    world_to_ndc_bwd(ndc[3 * i], ndc[3 * i + 1], ndc[3 * i + 2], g2[2 * i], g2[2 * i + 1], g3[3 * i], g3[3 * i + 1], g3[3 * i + 2]);
  }
}
}
