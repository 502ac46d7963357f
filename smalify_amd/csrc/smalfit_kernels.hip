// HIP kernels of the SMAL fitting engine for gfx950 (MI355X / CDNA4).
//
// Data layout in HBM (all float32 unless noted):
//   model bases are *planar*:  vt[3][Vp], sd[nb][3][Vp], pd[306][3][Vp]  (Vp = V rounded up to 256,
//   zero padded) so that a wavefront of 64 consecutive vertices issues fully coalesced 256-byte loads;
//   per-frame vertex buffers are planar too: x[n][3][Vp].
//   skin weights / joint regressor are stored sparse (ELL by vertex, CSC by joint) — the dense (V,35)
//   matrices of the reference (smal_torch.py:78-96) are mostly zeros; a dense matrix is simply an ELL
//   with 35 entries per row, the code path is the same and the sums run in the same joint order.
//
// One translation unit: this file holds the shared device helpers and includes the kernels by subsystem
// (kernels_lbs_forward.inc, kernels_raster.inc, kernels_color.inc, kernels_lbs_backward.inc, kernels_mesh3d.inc), then
// the host side (smalfit_launch.inc, smalfit_mesh3d.inc).
//
// Kernel -> reference map (file:line into /root/reference):
//   lbs_head_kernel     pose blocks: batch_lbs.py:33-52 (Rodrigues), :105-129 (limb scales), :131-168 (chain, A);
//                       shape blocks: smal_torch.py:115 (+ :125-128 through the precomputed J0 + JS beta);
//                       prior block: smal_fitter.py:162-171
//   skin_mfma_kernel / skin_kernel   smal_torch.py:138-163 (pose blend, W*A, skinning) + renderer camera transform
//   joints_kernel       smal_torch.py:171-184
//   loss_kernel         smal_fitter.py:129-132,140-160,177-190 ; pose_prior_35.py:117-124
//   face_bbox / raster_sweep / raster_resolve / raster_band / raster_select / raster_bwd
//                       p3d_renderer.py:26-39,65-66 (pytorch3d rasterize_meshes + sigmoid_alpha_blend)
//   vertex_bwd, lbs_bwd_mid, chain_bwd, assemble    autograd of the above (optimize_to_joints.py:136)
//   adam_kernel         optimize_to_joints.py:96,137 (torch.optim.Adam, betas=(0.5,0.999))
//   mesh3d_*            fitter_3d/trainer.py:205-227 (pytorch3d sample_points_from_meshes, chamfer_distance, mesh_edge_loss,
//                       mesh_normal_consistency, mesh_laplacian_smoothing) and their adjoints
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smalfit_internal.h"
#include "mesh3d_math.h"

namespace smalfit {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of `v` (blockDim.x multiple of 64, <= 1024); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* red /* >= 16 floats of LDS */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// butterfly reduce-scatter: after folding with bits 32..1 every lane holds the wave-wide sums of two accumulators
// (indices 2*lane, 2*lane+1) -- 126 shuffles instead of 128 full wave reductions; the summation order is fixed
template <int HALF>
__device__ __forceinline__ void fold_accumulators(float* acc, int lane, int bit) {
#pragma unroll
  for (int j = 0; j < HALF; ++j) {
    const bool up = (lane & bit) != 0;
    const float keep = up ? acc[HALF + j] : acc[j];
    const float send = up ? acc[j] : acc[HALF + j];
    acc[j] = keep + __shfl_xor(send, bit, 64);
  }
}

// wave-wide sums of NV (16 or 32) per-lane values with a butterfly reduce-scatter: lane l returns the total of value
// (l * NV) >> 6.  NV - 1 + log2(64 / NV) shuffles instead of 6 NV; fixed order.
template <int NV>
__device__ __forceinline__ float wave_sums(float* v, int lane) {
  static_assert(NV == 16 || NV == 32, "16 or 32 values");
  fold_accumulators<NV / 2>(v, lane, 32);
  fold_accumulators<NV / 4>(v, lane, 16);
  fold_accumulators<NV / 8>(v, lane, 8);
  fold_accumulators<NV / 16>(v, lane, 4);
  if (NV == 32) fold_accumulators<NV / 32>(v, lane, 2);
  float r = v[0];
  if (NV == 16) r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;
}

__device__ __forceinline__ int frame_window_size(int n, const WinMap& w) {
  // size of the window local frame n belongs to: the sequence's frames are grouped into consecutive windows of
  // `window` frames, the last one may be ragged (optimize_to_joints.py:119-120)
  const int start = ((n + w.offset) / w.window) * w.window;
  const int rem = w.total - start;
  return rem < w.window ? rem : w.window;
}

// Phase timers of the latency-chain kernels (developer builds: tools/build_variant.sh NAME -DSMALFIT_DEV_PROBES -DSMALFIT_PHASES;
// tools/lbs_phases.py): thread 0 of a workgroup adds the shader cycles (s_memtime) between two marks to g_phase[kernel][phase] and
// counts the workgroup in [kernel][15]; [kernel][14] holds the longest workgroup.  Compiled out of the product.
#ifdef SMALFIT_PHASES
__device__ unsigned long long g_phase[12][16];
#define PHASE_MARK(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PHASE_ADD(k, i, t0, t1) do { if (threadIdx.x == 0) atomicAdd(&g_phase[k][i], (t1) - (t0)); } while (0)
#define PHASE_END(k, t0, t1) do { if (threadIdx.x == 0) { atomicAdd(&g_phase[k][15], 1ull); atomicMax(&g_phase[k][14], (t1) - (t0)); atomicAdd(&g_phase[k][13], (t1) - (t0)); } } while (0)
#else
#define PHASE_MARK(var)
#define PHASE_ADD(k, i, t0, t1)
#define PHASE_END(k, t0, t1)
#endif
enum { PH_HEAD_POSE = 0, PH_HEAD_SHAPE, PH_SKIN, PH_VERTEX_BWD, PH_MID_PB, PH_MID_DA, PH_CHAIN, PH_CHAIN_RIDER, PH_ASM_BETA, PH_ASM_ELEM, PH_ASM_LAST, PH_ADAM };

#include "kernels_lbs_forward.inc"
#include "kernels_raster.inc"
#include "kernels_color.inc"
#include "kernels_lbs_backward.inc"
#include "kernels_mesh3d.inc"

}  // namespace smalfit

#include "smalfit_launch.inc"
#include "smalfit_mesh3d.inc"
