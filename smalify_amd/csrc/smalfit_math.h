// Per-element maths of the SMAL fitting path, shared by every kernel in smalfit_kernels.hip.
// Functions are SMALFIT_HD (= __host__ __device__ under hipcc, plain inline under g++) so that
// tests/host_math_check.cpp can finite-difference them on the CPU; the product only ever calls them
// from device code.
//
// Reference behaviour restated here (file:line into /root/reference):
//   rodrigues_fwd / _bwd        smal_model/batch_lbs.py:9-52       (angle = ||theta + 1e-8||)
//   chain step fwd / bwd        smal_model/batch_lbs.py:137-168
//   camera                      smal_fitter/p3d_renderer.py:22-23  (dist 2.7, OpenGL persp fov 60)
//   face_pixel_eval             pytorch3d 0.2.5 RasterizeMeshesNaiveCpu + sigmoid_alpha_blend
//                               (p3d_renderer.py:26-39,66)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SMALFIT_HD __host__ __device__ __forceinline__
#else
#define SMALFIT_HD inline
#endif

namespace smalfit {

// 1/x: v_rcp_f32 (1 ulp) in device code, IEEE division on the host (test shim)
SMALFIT_HD float recip(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

constexpr int kJoints = 35;
constexpr int kPoseFeat = 306;
constexpr int kModelJoints = 41;
constexpr int kKeypoints = 25;
constexpr int kLogScales = 6;

constexpr float kCamDist = 2.7f;
constexpr float kCamScale = 1.7320508075688772f;   // 1 / tan(30 deg)
constexpr float kSigma = 1e-4f;
constexpr float kBlur = 9.21024036697585e-4f;       // log(1/1e-4 - 1) * 1e-4
constexpr float kBlurSqrt = 0.030348377826444f;
constexpr float kEps = 1e-8f;                       // pytorch3d kEpsilon
constexpr int kFacesPerPixel = 100;

// ------------------------------------------------------------------------------------------------
// Rodrigues
// ------------------------------------------------------------------------------------------------
SMALFIT_HD void rodrigues_fwd(const float th[3], float R[9]) {
  const float ux = th[0] + 1e-8f, uy = th[1] + 1e-8f, uz = th[2] + 1e-8f;
  const float a = sqrtf(ux * ux + uy * uy + uz * uz);
  const float ia = 1.0f / a;
  const float rx = th[0] * ia, ry = th[1] * ia, rz = th[2] * ia;
  const float c = cosf(a), s = sinf(a), k = 1.0f - c;
  R[0] = c + k * rx * rx;      R[1] = k * rx * ry - s * rz; R[2] = k * rx * rz + s * ry;
  R[3] = k * ry * rx + s * rz; R[4] = c + k * ry * ry;      R[5] = k * ry * rz - s * rx;
  R[6] = k * rz * rx - s * ry; R[7] = k * rz * ry + s * rx; R[8] = c + k * rz * rz;
}

// dth = (dR/dth)^T G following the reference's computational graph (finite at th = 0).
SMALFIT_HD void rodrigues_bwd(const float th[3], const float G[9], float dth[3]) {
  const float ux = th[0] + 1e-8f, uy = th[1] + 1e-8f, uz = th[2] + 1e-8f;
  const float a = sqrtf(ux * ux + uy * uy + uz * uz);
  const float ia = 1.0f / a;
  const float r[3] = {th[0] * ia, th[1] * ia, th[2] * ia};
  const float c = cosf(a), s = sinf(a), k = 1.0f - c;
  // Gr = G r, GTr = G^T r
  float Gr[3], GTr[3];
  for (int i = 0; i < 3; ++i) {
    Gr[i] = G[3 * i] * r[0] + G[3 * i + 1] * r[1] + G[3 * i + 2] * r[2];
    GTr[i] = G[i] * r[0] + G[3 + i] * r[1] + G[6 + i] * r[2];
  }
  const float rGr = r[0] * Gr[0] + r[1] * Gr[1] + r[2] * Gr[2];
  const float trG = G[0] + G[4] + G[8];
  const float kv[3] = {G[7] - G[5], G[2] - G[6], G[3] - G[1]};     // d<G,[r]x>/dr
  const float dc = trG - rGr;
  const float ds = r[0] * kv[0] + r[1] * kv[1] + r[2] * kv[2];
  float dr[3];
  for (int i = 0; i < 3; ++i) dr[i] = k * (Gr[i] + GTr[i]) + s * kv[i];
  const float da = -s * dc + c * ds - (dr[0] * r[0] + dr[1] * r[1] + dr[2] * r[2]) * ia;
  dth[0] = dr[0] * ia + da * ux * ia;
  dth[1] = dr[1] * ia + da * uy * ia;
  dth[2] = dr[2] * ia + da * uz * ia;
}

// ------------------------------------------------------------------------------------------------
// 3x3 helpers (row-major)
// ------------------------------------------------------------------------------------------------
SMALFIT_HD void mat3_mul(const float A[9], const float B[9], float C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
SMALFIT_HD void mat3_vec(const float A[9], const float v[3], float o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
SMALFIT_HD void mat3T_vec(const float A[9], const float v[3], float o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}

// ------------------------------------------------------------------------------------------------
// kinematic chain of one frame, free-standing (batch_global_rigid_transformation, batch_lbs.py:75-170)
// ------------------------------------------------------------------------------------------------
// limb log-scale driving (joint j, axis a), or -1 (batch_lbs.py:107-121: legs 7..14, 17..24; tail 25..31; ears 33, 34)
SMALFIT_HD int limb_scale_index(int j, int a) {
  if (j >= 7 && j < 25 && j != 15 && j != 16) return a == 2 ? 0 : 1;
  if (j >= 25 && j < 32) return a == 0 ? 2 : 3;
  if (j == 33 || j == 34) return a == 1 ? 4 : (a == 2 ? 5 : -1);
  return -1;
}

// Rs [35][9], Js [35][3], parents [35] (parents[i] < i), logscale [6] or null -> newJ [35][3], A [35][16] (4x4 row-major).
// G_0 = [R_0 | J_0] (the root is not scaled), G_i = G_p [S_p^-1 R_i S_i | J_i - J_p]; newJ_i = G_i[:3,3];
// A_i = G_i with its last column replaced by t - G_i[:3,:3] J_i.  The chain is held in A itself (rows 0..2).
SMALFIT_HD void global_rigid_frame(const float* Rs, const float* Js, const int* parents, const float* logscale,
                                   float* newJ, float* A) {
  float es[6], ies[6];
  for (int k = 0; k < 6; ++k) {
    es[k] = logscale != nullptr ? expf(logscale[k]) : 1.0f;
    ies[k] = 1.0f / es[k];
  }
  for (int i = 0; i < kJoints; ++i) {
    float* G = A + 16 * i;
    if (i == 0) {
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) G[r * 4 + c] = Rs[r * 3 + c];
        G[r * 4 + 3] = Js[r];
      }
    } else {
      const int p = parents[i];
      const float* Gp = A + 16 * p;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
          const int si = limb_scale_index(i, c);
          const float sc = si >= 0 ? es[si] : 1.0f;
          float acc = 0.f;
          for (int q = 0; q < 3; ++q) {
            const int sp = limb_scale_index(p, q);
            const float isc = sp >= 0 ? ies[sp] : 1.0f;
            acc = fmaf(Gp[r * 4 + q], Rs[9 * i + q * 3 + c] * sc * isc, acc);
          }
          G[r * 4 + c] = acc;
        }
        float acc = Gp[r * 4 + 3];
        for (int q = 0; q < 3; ++q) acc = fmaf(Gp[r * 4 + q], Js[3 * i + q] - Js[3 * p + q], acc);
        G[r * 4 + 3] = acc;
      }
    }
    G[12] = 0.f; G[13] = 0.f; G[14] = 0.f; G[15] = 1.0f;
  }
  for (int i = 0; i < kJoints; ++i) {
    float* G = A + 16 * i;
    for (int r = 0; r < 3; ++r) {
      newJ[3 * i + r] = G[r * 4 + 3];
      float val = G[r * 4 + 3];
      for (int q = 0; q < 3; ++q) val = fmaf(-G[r * 4 + q], Js[3 * i + q], val);
      G[r * 4 + 3] = val;
    }
  }
}

// Adjoint of global_rigid_frame: dnewJ [35][3] and dA [35][16] (4x4 row-major; the constant bottom rows are ignored)
// -> dRs [35][9], dJs [35][3], dlogscale [6] (written only when logscale != null).  The chain is re-run forward into G
// (caller scratch, [35][12] = rows 0..2 of each 4x4), then walked in reverse joint order.
SMALFIT_HD void global_rigid_frame_bwd(const float* Rs, const float* Js, const int* parents, const float* logscale,
                                       const float* dnewJ, const float* dA, float* G /*[35][12] scratch*/,
                                       float* dG /*[35][12] scratch*/, float* dRs, float* dJs, float* dlogscale) {
  float es[6], ies[6], des[6];
  for (int k = 0; k < 6; ++k) {
    es[k] = logscale != nullptr ? expf(logscale[k]) : 1.0f;
    ies[k] = 1.0f / es[k];
    des[k] = 0.f;
  }
  auto sc = [&](int j, int a) { const int si = limb_scale_index(j, a); return si >= 0 ? es[si] : 1.0f; };
  auto isc = [&](int j, int a) { const int si = limb_scale_index(j, a); return si >= 0 ? ies[si] : 1.0f; };
  // forward chain (the same operations as global_rigid_frame)
  for (int i = 0; i < kJoints; ++i) {
    float* g = G + 12 * i;
    if (i == 0) {
      for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) g[r * 4 + c] = Rs[r * 3 + c]; g[r * 4 + 3] = Js[r]; }
    } else {
      const int p = parents[i];
      const float* gp = G + 12 * p;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
          float acc = 0.f;
          for (int q = 0; q < 3; ++q) acc = fmaf(gp[r * 4 + q], Rs[9 * i + q * 3 + c] * sc(i, c) * isc(p, q), acc);
          g[r * 4 + c] = acc;
        }
        float acc = gp[r * 4 + 3];
        for (int q = 0; q < 3; ++q) acc = fmaf(gp[r * 4 + q], Js[3 * i + q] - Js[3 * p + q], acc);
        g[r * 4 + 3] = acc;
      }
    }
  }
  // A_i = [G.R | G.t - G.R J_i], newJ_i = G.t
  for (int i = 0; i < kJoints; ++i) {
    const float* da = dA + 16 * i;
    const float* g = G + 12 * i;
    float* dg = dG + 12 * i;
    for (int r = 0; r < 3; ++r) {
      for (int q = 0; q < 3; ++q) dg[r * 4 + q] = da[r * 4 + q] - da[r * 4 + 3] * Js[3 * i + q];
      dg[r * 4 + 3] = da[r * 4 + 3] + dnewJ[3 * i + r];
    }
    for (int q = 0; q < 3; ++q) {
      float acc = 0.f;
      for (int r = 0; r < 3; ++r) acc = fmaf(-g[r * 4 + q], da[r * 4 + 3], acc);
      dJs[3 * i + q] = acc;
    }
  }
  for (int i = kJoints - 1; i >= 1; --i) {
    const int p = parents[i];
    const float* gp = G + 12 * p;
    const float* dg = dG + 12 * i;
    float* dgp = dG + 12 * p;
    // dR' = G_p.R^T dG_i.R ;  dj = G_p.R^T dG_i.t
    float dRp[9], dj[3];
    for (int q = 0; q < 3; ++q) {
      for (int c = 0; c < 3; ++c) dRp[q * 3 + c] = gp[0 * 4 + q] * dg[0 * 4 + c] + gp[1 * 4 + q] * dg[1 * 4 + c] + gp[2 * 4 + q] * dg[2 * 4 + c];
      dj[q] = gp[0 * 4 + q] * dg[3] + gp[1 * 4 + q] * dg[7] + gp[2 * 4 + q] * dg[11];
    }
    for (int r = 0; r < 3; ++r) {
      for (int q = 0; q < 3; ++q) {
        // dG_p.R[r][q] += sum_c dG_i.R[r][c] R'[q][c] + dG_i.t[r] (J_i - J_p)[q]
        float acc = dg[r * 4 + 3] * (Js[3 * i + q] - Js[3 * p + q]);
        for (int c = 0; c < 3; ++c) acc = fmaf(dg[r * 4 + c], Rs[9 * i + q * 3 + c] * sc(i, c) * isc(p, q), acc);
        dgp[r * 4 + q] += acc;
      }
      dgp[r * 4 + 3] += dg[r * 4 + 3];
    }
    for (int q = 0; q < 3; ++q) { dJs[3 * i + q] += dj[q]; dJs[3 * p + q] -= dj[q]; }
    for (int q = 0; q < 3; ++q)
      for (int c = 0; c < 3; ++c) {
        const float rp = Rs[9 * i + q * 3 + c];
        dRs[9 * i + q * 3 + c] = dRp[q * 3 + c] * sc(i, c) * isc(p, q);
        const int si = limb_scale_index(i, c), sp = limb_scale_index(p, q);
        // R'[q][c] = R[q][c] s_i[c] / s_p[q]:  d/ds_i[c] = R / s_p[q],  d/ds_p[q] = -R' / s_p[q]
        if (si >= 0) des[si] += dRp[q * 3 + c] * rp * isc(p, q);
        if (sp >= 0) des[sp] -= dRp[q * 3 + c] * rp * sc(i, c) * isc(p, q) * isc(p, q);
      }
  }
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) dRs[r * 3 + c] = dG[r * 4 + c];
    dJs[r] += dG[r * 4 + 3];
  }
  if (logscale != nullptr && dlogscale != nullptr)
    for (int k = 0; k < 6; ++k) dlogscale[k] = des[k] * es[k];
}

// ------------------------------------------------------------------------------------------------
// camera: world -> (x_ndc, y_ndc, z_view) and its adjoint
// ------------------------------------------------------------------------------------------------
SMALFIT_HD void world_to_ndc(float x, float y, float z, float& xn, float& yn, float& zv) {
  zv = kCamDist - z;
  const float iz = kCamScale / zv;
  xn = -x * iz;
  yn = y * iz;
}
SMALFIT_HD void world_to_ndc_bwd(float xn, float yn, float zv, float gxn, float gyn,
                                 float& gx, float& gy, float& gz) {
  const float iz = 1.0f / zv;
  gx = -kCamScale * iz * gxn;
  gy = kCamScale * iz * gyn;
  gz = (xn * gxn + yn * gyn) * iz;
}

// ------------------------------------------------------------------------------------------------
// soft-silhouette per (pixel, face) evaluation
// ------------------------------------------------------------------------------------------------
// Face record, 20 floats, laid out for five 16-byte LDS broadcasts.
struct FaceRec {
  float ax, ay, e1x, e1y;        // a, e1 = b - a
  float e2x, e2y, e3x, e3y;      // e2 = c - a, e3 = c - b
  float il1, il2, il3, area;     // 1/|e|^2 per edge (0 when degenerate), signed area E(c; a, b)
  float t01, t02, t03, inv_den;  // t offsets (1 when the edge is degenerate: distance to its end point)
  float pz0, gzx, gzy, pad;      // interpolated depth as a plane: pz = pz0 + gzx (px - ax) + gzy (py - ay)
};

// interpolated view-space depth of face r at pixel offset (dx, dy) = p - a.  This exact expression is the
// definition of pz everywhere (inclusion test, K-nearest selection, backward), so comparisons agree bitwise.
SMALFIT_HD float face_depth(const FaceRec& r, float dx, float dy) { return fmaf(r.gzy, dy, fmaf(r.gzx, dx, r.pz0)); }

// returns false when the face is culled as a whole (degenerate area or entirely behind the camera)
SMALFIT_HD bool make_face_rec(float ax, float ay, float az, float bx, float by, float bz,
                              float cx, float cy, float cz, FaceRec& r) {
  r.ax = ax; r.ay = ay;
  r.e1x = bx - ax; r.e1y = by - ay;
  r.e2x = cx - ax; r.e2y = cy - ay;
  r.e3x = cx - bx; r.e3y = cy - by;
  const float l1 = r.e1x * r.e1x + r.e1y * r.e1y;
  const float l2 = r.e2x * r.e2x + r.e2y * r.e2y;
  const float l3 = r.e3x * r.e3x + r.e3y * r.e3y;
  r.il1 = l1 > kEps ? recip(l1) : 0.0f;  r.t01 = l1 > kEps ? 0.0f : 1.0f;
  r.il2 = l2 > kEps ? recip(l2) : 0.0f;  r.t02 = l2 > kEps ? 0.0f : 1.0f;
  r.il3 = l3 > kEps ? recip(l3) : 0.0f;  r.t03 = l3 > kEps ? 0.0f : 1.0f;
  r.area = r.e2x * r.e1y - r.e2y * r.e1x;            // E(c; a, b) = (c-a) x (b-a)
  r.inv_den = recip(r.area + kEps);
  // pz = w0 az + w1 bz + w2 cz with barycentrics over (area + eps) is affine in the pixel
  const float da = cz - az, db = az - bz;
  r.gzx = r.inv_den * (da * r.e1y + db * r.e2y);
  r.gzy = -r.inv_den * (da * r.e1x + db * r.e2x);
  r.pz0 = r.inv_den * az * r.area;
  r.pad = 0.0f;
  const float zmax = fmaxf(az, fmaxf(bz, cz));
  return (fabsf(r.area) > kEps) && (zmax >= 0.0f);
}

// depth of face r at pixel centre (px, py): the same expression face_pixel_eval uses (cheap pre-test)
SMALFIT_HD float face_pixel_depth(const FaceRec& r, float px, float py) { return face_depth(r, px - r.ax, py - r.ay); }

struct PixEval {
  float d;        // signed squared distance (negative inside)
  float pz;       // interpolated view-space depth at the pixel
  float qx, qy;   // p - closest point on the nearest edge
  float tc;       // clamped parameter on that edge
  int edge;       // 0: a-b, 1: a-c, 2: b-c
  bool inside;
};

// true when the face contributes to pixel centre (px, py) (pytorch3d naive rasteriser inclusion test).
// Every multiply-add is an explicit fmaf and no other a*b+c pattern is left for the compiler to contract: the
// two-pixel packed form in the kernels (face_pixel_eval2) performs the same IEEE operations in the same order, so the
// two agree bitwise (the K-nearest bookkeeping relies on every kernel seeing the same candidate set and depths).
SMALFIT_HD bool face_pixel_eval(const FaceRec& r, float px, float py, PixEval& o) {
  const float dx = px - r.ax, dy = py - r.ay;
  const float c1 = fmaf(dx, r.e1y, -(dy * r.e1x));   // E(p; a, b)
  const float c2 = fmaf(dx, r.e2y, -(dy * r.e2x));   // -E(p; c, a)
  const float w2 = c1 * r.inv_den;
  const float w1 = -c2 * r.inv_den;
  const float w0 = ((c2 - c1) + r.area) * r.inv_den; // E(p; b, c) / (area + eps)
  o.inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
  const float pz = face_depth(r, dx, dy);
  o.pz = pz;
  // edge a-b
  const float t1 = fminf(fmaxf(fmaf(fmaf(dy, r.e1y, dx * r.e1x), r.il1, r.t01), 0.0f), 1.0f);
  const float q1x = fmaf(-t1, r.e1x, dx), q1y = fmaf(-t1, r.e1y, dy);
  const float d1 = fmaf(q1y, q1y, q1x * q1x);
  // edge a-c
  const float t2 = fminf(fmaxf(fmaf(fmaf(dy, r.e2y, dx * r.e2x), r.il2, r.t02), 0.0f), 1.0f);
  const float q2x = fmaf(-t2, r.e2x, dx), q2y = fmaf(-t2, r.e2y, dy);
  const float d2 = fmaf(q2y, q2y, q2x * q2x);
  // edge b-c
  const float ex = dx - r.e1x, ey = dy - r.e1y;
  const float t3 = fminf(fmaxf(fmaf(fmaf(ey, r.e3y, ex * r.e3x), r.il3, r.t03), 0.0f), 1.0f);
  const float q3x = fmaf(-t3, r.e3x, ex), q3y = fmaf(-t3, r.e3y, ey);
  const float d3 = fmaf(q3y, q3y, q3x * q3x);
  float dist = d1; o.qx = q1x; o.qy = q1y; o.tc = t1; o.edge = 0;
  if (d2 < dist) { dist = d2; o.qx = q2x; o.qy = q2y; o.tc = t2; o.edge = 1; }
  if (d3 < dist) { dist = d3; o.qx = q3x; o.qy = q3y; o.tc = t3; o.edge = 2; }
  o.d = o.inside ? -dist : dist;
  return (pz >= 0.0f) && (o.inside || dist < kBlur);
}

// The forward sweep's form of face_pixel_eval: the candidate decision and the signed squared distance only.  The same IEEE
// operations in the same order produce t, q and the three edge distances (so d and the decision agree bitwise with
// face_pixel_eval, which the selection and the backward use); the minimum over the edges and the inside test are taken
// with min3 instead of compare/select chains -- identical values, a third fewer instructions around them.
SMALFIT_HD bool face_pixel_candidate(const FaceRec& r, float px, float py, float& d_out) {
  const float dx = px - r.ax, dy = py - r.ay;
  const float c1 = fmaf(dx, r.e1y, -(dy * r.e1x));
  const float c2 = fmaf(dx, r.e2y, -(dy * r.e2x));
  const float w2 = c1 * r.inv_den;
  const float w1 = -c2 * r.inv_den;
  const float w0 = ((c2 - c1) + r.area) * r.inv_den;
  const bool inside = fminf(w0, fminf(w1, w2)) > 0.0f;
  const float pz = face_depth(r, dx, dy);
  const float t1 = fminf(fmaxf(fmaf(fmaf(dy, r.e1y, dx * r.e1x), r.il1, r.t01), 0.0f), 1.0f);
  const float q1x = fmaf(-t1, r.e1x, dx), q1y = fmaf(-t1, r.e1y, dy);
  const float d1 = fmaf(q1y, q1y, q1x * q1x);
  const float t2 = fminf(fmaxf(fmaf(fmaf(dy, r.e2y, dx * r.e2x), r.il2, r.t02), 0.0f), 1.0f);
  const float q2x = fmaf(-t2, r.e2x, dx), q2y = fmaf(-t2, r.e2y, dy);
  const float d2 = fmaf(q2y, q2y, q2x * q2x);
  const float ex = dx - r.e1x, ey = dy - r.e1y;
  const float t3 = fminf(fmaxf(fmaf(fmaf(ey, r.e3y, ex * r.e3x), r.il3, r.t03), 0.0f), 1.0f);
  const float q3x = fmaf(-t3, r.e3x, ex), q3y = fmaf(-t3, r.e3y, ey);
  const float d3 = fmaf(q3y, q3y, q3x * q3x);
  const float dist = fminf(d1, fminf(d2, d3));
  d_out = inside ? -dist : dist;
  return (pz >= 0.0f) && (inside || dist < kBlur);
}

// 1 - p = sigmoid(d / sigma)
SMALFIT_HD float one_minus_prob(float d) { return recip(1.0f + expf(-d * (1.0f / kSigma))); }
// p = sigmoid(-d / sigma)
SMALFIT_HD float prob(float d) { return recip(1.0f + expf(d * (1.0f / kSigma))); }

// log2(1 - p) = log2 sigmoid(x), x = d / sigma = min(x,0) log2(e) - log2(1 + 2^(-|x| log2 e)).
// Forming 1 + t in float costs at most 6e-8 absolute in the logarithm, i.e. <= 6e-8 relative in alpha per
// candidate -- the same as one rounded multiplication.  Clamped at -256 (alpha underflows long before).
SMALFIT_HD float log2_one_minus_prob(float d) {
  const float x2 = d * (1.4426950408889634f / kSigma);
  const float ls2 = fminf(x2, 0.0f) - log2f(1.0f + exp2f(-fabsf(x2)));
  return fmaxf(ls2, -256.0f);
}

// pixel centre in NDC (both image axes flipped)
// one fused multiply-add: 1 - (2 i + 1)/S = fma(i, -2/S, 1 - 1/S)  (exact for power-of-two sizes; every kernel uses this
// one definition, so pixel centres agree bitwise between them)
SMALFIT_HD float pix_to_ndc(int i, float inv_s) { return fmaf((float)i, -2.0f * inv_s, 1.0f - inv_s); }

}  // namespace smalfit
