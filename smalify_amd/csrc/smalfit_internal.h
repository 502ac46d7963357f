// Internal declarations shared by the kernels and the host-side engine. Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "smalfit_math.h"

#define SMALFIT_STATUS_BIN_OVERFLOW 1   // a frame's candidate-list pool overflowed

namespace smalfit {

// kinematic tree by depth: joints of one level are independent, so the chain and its adjoint take `nlev` steps
// (10 for SMAL) instead of 34.  children lists are in DESCENDING joint order: accumulating a parent's adjoint from its
// children in that order reproduces the summation order of a plain reverse loop over the joints.
constexpr int kTreeMaxPass = 16, kTreeMaxChildren = 4;
struct TreeLevels {
  unsigned char nlev;
  unsigned char lvl_off[36];     // level L owns lvl_joint[lvl_off[L] .. lvl_off[L+1])
  unsigned char lvl_joint[35];
  unsigned char child_off[36];   // joint j owns child_idx[child_off[j] .. child_off[j+1])
  unsigned char child_idx[35];
  // The walks as a flat schedule (round 6): pass k handles up to five joints of one level (a wave = 5 joints x 12 lanes), levels in
  // ascending order.  With it a lane reads its joint / parent / children of EVERY pass before the walk starts (independent loads, one
  // latency) instead of chasing level offset -> joint -> parent -> child list through LDS inside every pass.  `fast` = the tree fits
  // (at most kTreeMaxPass passes, kTreeMaxChildren children per joint: SMAL's needs 12 and 4); deeper / bushier trees take the
  // table-driven loops as before.
  unsigned char fast, npass;
  unsigned char pass_joint[kTreeMaxPass][8];    // [pass][slot 0..4] joint, 255 = idle slot
  unsigned char pass_parent[kTreeMaxPass][8];
  unsigned char pass_nchild[kTreeMaxPass][8];
  unsigned char pass_child[kTreeMaxPass][8][kTreeMaxChildren];   // in the order of child_idx (descending joint index)
};

// device-resident model constants (pointers into one allocation owned by smalfit_model)
struct ModelDev {
  int V, Vp, F, NBall;
  const float* vt;        // [3][Vp]
  const float* sd;        // [NBall][3][Vp]
  const float* pd;        // [306][3][Vp]
  int Kw;                 // skin weights, ELL by vertex
  const int* w_j;         // [Kw][Vp]
  const float* w_val;     // [Kw][Vp]
  const int* wc_off;      // skin weights, CSC by joint [36]
  const int* wc_v;
  const float* wc_val;
  const int* jr_off;      // joint regressor, CSC by joint [36]
  const int* jr_v;
  const float* jr_val;
  int Kj;                 // joint regressor, ELL by vertex
  const int* jrv_j;       // [Kj][Vp]
  const float* jrv_val;   // [Kj][Vp]
  const float* Jt;        // [105]  rest joints at beta = 0
  const float* JS;        // [105][NBall]  d(rest joints)/d(beta)
  const int* parents;     // [35]
  const int* faces;       // [F][3]
  const int* vf_off;      // [V+1] vertex -> incident (face*3+corner)
  const int* vf_idx;
  const int* scale_idx;   // [105] log-scale index per (joint, axis) or -1
  int landmarks[6];
  TreeLevels tree;
};

// Where the M frames of an evaluation sit in their sequence: frames are grouped into consecutive windows of `window`
// frames counted from the START OF THE SEQUENCE (optimize_to_joints.py:119-120, the last window may be ragged), and the
// reference's per-window normalisers 1/(B 50), 1/(B 105), 1/(B S^2) (smal_fitter.py:144,157,173) use the size B of the
// window a frame belongs to.  An evaluation may hold any contiguous part of the sequence -- a whole sequence
// (offset 0, total M), a shard of it, or a single frame of an 8-frame window (one frame per GPU).
struct WinMap {
  int window;   // WINDOW_SIZE
  int offset;   // index of local frame 0 in the sequence
  int total;    // frames in the whole sequence
};

struct LossArgs {
  int M, S;
  WinMap win;
  const float* theta;      // [M][105] masked
  const float* trans;      // [M][3]
  const float* joints;     // [M][41][3] (untranslated)
  const int* canon;        // [25]
  const float* tj;         // [M][25][2] (row, col)
  const float* vis;        // [M][25]
  float w_j2d, w_pose, w_splay, w_temp;
  float w_limit;           // joint-limit hinge (smal_fitter.py:146-151)
  const float* lim_min;    // [102] lower / upper limit per joint-rotation component
  const float* lim_max;
  const float* pose_prec;  // [105][105]
  const float* pose_mean;  // [105]
  const float* pose_mask;  // [105]
  const float* halo_prev;  // [108] neighbour frame before frame 0 (theta 105 | trans 3) or null
  const float* halo_next;  // [108] neighbour frame after frame M-1 or null
  float* proj_out;         // [M][25][2] or null
  float* dth_direct;       // [M][105]
  float* dJ41;             // [M][41][3]
  float* dtr_direct;       // [M][3]
  float* loss_part;        // [M][8]
};

struct AssembleArgs {
  int M, S, T, nb, NBall, nblk_beta, nvt;
  WinMap win;
  int betas_shared, ls_shared;
  float w_sil;
  const float* dbeta_part;
  const float* dJrest;
  const float* dbetaJ;     // [M][NBall] d beta through the rest joints (chain_bwd_kernel)
  int ngrp_beta;           // frame groups of the shape-blend adjoint (dbeta_block)
  const float* JS;
  const float* gb_prior;
  const float* gls_prior;
  const float* dls;
  const float* dtheta;
  const float* gmask;
  const float* rmask;
  const float* dtr_direct;
  const float* dtr_part;
  const float* loss_part;
  const float* loss_betas;
  const float* tile_loss;
  const long long* qloss;  // weighted silhouette loss of the queued pixels, one 2^-40 fixed-point partial per band / select block
  int nqblk;
  float* lpart;            // [kAsmLoss] partial sums of the silhouette loss (assemble_kernel)
  long long* qpart;        // [kAsmLoss] integer partial sums of the queue kernels' loss
  int* counter;            // arrival counter of assemble_kernel's blocks (zero between launches)
  float* g_betas;
  float* g_ls;
  float* g_grot;
  float* g_jrot;
  float* g_trans;
  float* losses;
};

}  // namespace smalfit
