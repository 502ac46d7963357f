/* smalfit.h — C-ABI of the MI355X-native SMAL fitting engine (libsmalfit.so, built by hipcc for gfx950).
 *
 * The reference (benjiebob/SMALify @ /root/reference) has no FFI layer: its hot path is Python calling
 * torch / pytorch3d.  The boundary a maintainer would bind is therefore the set of Python call sites
 * listed in SURVEY.md §8b; every entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - every tensor argument is a DEVICE pointer to contiguous row-major float32 (int32 for indices)
 *     unless the parameter is documented as "host"
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     except smalfit_engine_status()
 *   - functions return 0 on success, non-zero on failure; smalfit_last_error() describes the failure
 *   - handles are thread-compatible, not thread-safe; no ownership of caller buffers is taken
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails
 */
#ifndef SMALFIT_H_
#define SMALFIT_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smalfit_model smalfit_model;
typedef struct smalfit_engine smalfit_engine;

#define SMALFIT_NUM_JOINTS 35
#define SMALFIT_NUM_MODEL_JOINTS 41
#define SMALFIT_NUM_KEYPOINTS 25
#define SMALFIT_NUM_LOSS_TERMS 9 /* joint, pose, splay, betas, sil_reproj, temp_joint, temp_global, temp_trans, limit */

#define SMALFIT_STATUS_BIN_OVERFLOW 1 /* reserved */

/* ABI version of this header.  smalfit_version() returns the version the library was built from; a caller compares the
 * two once at load time (smalify_amd/_lib.py does) -- the argument structs below are passed by pointer and grow at the
 * tail from version to version.  History: 1 = round 1; 2 = 9 loss terms (losses must hold SMALFIT_NUM_LOSS_TERMS floats),
 * smalfit_fit_args gained target_sil_u8 / w_limit; 3 = smalfit_fit_args.struct_size (first field), frame_offset,
 * total_frames; smalfit_engine_clear_joint_limits, smalfit_shard_local_step; 4 = smalfit_shard_run (the sharded loop of a
 * whole stage in one call, the collective supplied by the host as a function pointer), smalfit_rccl_allgather;
 * 5 = smalfit_engine_set_option. */
#define SMALFIT_ABI_VERSION 5
int smalfit_version(void);
const char* smalfit_last_error(void);

/* ---- model constants -------------------------------------------------------------------------
 * replaces: SMAL.__init__ tensors            reference smal_model/smal_torch.py:36-96
 * All pointers are HOST arrays prepared by the caller (smalify_amd/model_io.py does the pickle
 * parsing, family mean and symmetry alignment exactly as the reference). */
typedef struct smalfit_model_desc {
  int num_verts;            /* V (3889)                                   */
  int num_faces;            /* F (7774)                                   */
  int num_betas;            /* rows of shapedirs (41)                     */
  const float* v_template;  /* (V,3)                                      */
  const float* shapedirs;   /* (num_betas, 3V), column = 3*v + axis       */
  const float* posedirs;    /* (306, 3V)                                  */
  const float* J_regressor; /* (V,35) dense                               */
  const float* weights;     /* (V,35) dense                               */
  const int* parents;       /* (35), parents[0] = -1, parents[i] < i      */
  const int* faces;         /* (F,3)                                      */
} smalfit_model_desc;

int smalfit_model_create(const smalfit_model_desc* desc, smalfit_model** out);
void smalfit_model_destroy(smalfit_model* model);

/* ---- engine = workspace in HBM for up to max_frames frames at image_size^2 ----------------------
 * image_size <= 1024 (the reference renders 256 or 512, config.py IMG_RES; larger sizes are rejected: the rasteriser's
 * pixel walk is exact in float32 up to there) */
int smalfit_engine_create(smalfit_model* model, int max_frames, int image_size, smalfit_engine** out);
void smalfit_engine_destroy(smalfit_engine* engine);
/* synchronises `stream`, returns and clears the sticky status bits (SMALFIT_STATUS_*) */
int smalfit_engine_status(smalfit_engine* engine, void* stream, int* status_bits);
/* The rasteriser keeps, per pixel, the depth bounds of its last exact K-nearest selection and re-proves them from
 * counts at every evaluation (results never depend on that state, only the time does).  This call forgets the
 * bounds, e.g. before fitting an unrelated sequence with the same engine.  No counterpart in the reference. */
int smalfit_engine_reset_raster_cache(smalfit_engine* engine, void* stream);

/* optional: time sections of smalfit_fit_eval with HIP events recorded on the caller's stream.
 * profile_begin arms up to max_evals timed evaluations, taking every `stride`-th evaluation (an event record costs
 * ~5 us of stream time on MI355X, so timing every evaluation would slow the loop it measures by several percent);
 * profile_end synchronises `stream`, and returns the summed milliseconds and the number of timed evaluations per
 * section. */
#define SMALFIT_NUM_SECTIONS 7
#define SMALFIT_SEC_LBS_FWD 0        /* lbs_head (pose, shape blend, shape prior) + skin + joints kernels */
#define SMALFIT_SEC_RASTER_SWEEP 1   /* raster_sweep_kernel alone                                  */
#define SMALFIT_SEC_RASTER_SELECT 2  /* raster_select_kernel alone (K-nearest selection)          */
#define SMALFIT_SEC_RASTER_BWD 3     /* raster_bwd_kernel alone                                   */
#define SMALFIT_SEC_LBS_BWD 4        /* vertex, mid-stage (dA, pose-blend, shape-blend) and chain adjoints */
#define SMALFIT_SEC_RASTER_RESOLVE 5 /* raster_resolve_kernel + raster_band_kernel                  */
#define SMALFIT_SEC_RASTER_BBOX 6    /* face_bbox_kernel (per-face pixel boxes and records)        */
int smalfit_engine_profile_begin(smalfit_engine* engine, int max_evals, int stride);
int smalfit_engine_profile_end(smalfit_engine* engine, void* stream, float* ms_total, int* counts);

/* replaces: Prior.__init__ data              reference smal_fitter/priors/pose_prior_35.py:51-92
 * host arrays: prec (105,105), mean (105), mask (105) */
int smalfit_engine_set_pose_prior(smalfit_engine* engine, const float* prec, const float* mean,
                                  const float* mask);
/* replaces: LimitPrior().min_values / max_values viewed as (N_POSE, 3)   reference smal_fitter/smal_fitter.py:76-79,
 * priors/joint_limits_prior.py:39-104 (commented out upstream: its table has 32 joints x 3 where view(34, 3) expects 34;
 * smalify_amd/model_io.py::joint_limit_table completes it).  host arrays: min (34,3), max (34,3), min <= max */
int smalfit_engine_set_joint_limits(smalfit_engine* engine, const float* min_values, const float* max_values);
/* back to the reference's behaviour (term commented out): w_limit is ignored again */
int smalfit_engine_clear_joint_limits(smalfit_engine* engine);
/* Engine options (no counterpart in the reference: they select between readings of pytorch3d 0.2.5 that cannot be told apart
 * without a PyTorch3D-produced vector).
 *   SMALFIT_OPT_UNCLAMPED_EDGE_T  value 0 (default) / 1.  replaces: rasterize_meshes_backward behind
 *     MeshRasterizer / SoftSilhouetteShader, reference smal_fitter/p3d_renderer.py:33-39.  0 = the exact gradient of the
 *     forward's point-segment distance (edge parameter t clamped to [0, 1]); 1 = the gradient with t left unclamped (the edge
 *     treated as an infinite line in PointLineDistanceBackward), as some 0.2.x sources are recalled to do.  The two differ only
 *     at pixels whose nearest feature of the nearest edge is a vertex.  Applies to smalfit_fit_eval / smalfit_fit_run /
 *     smalfit_shard_run / smalfit_render_backward; the forward pass is the same either way. */
#define SMALFIT_OPT_UNCLAMPED_EDGE_T 1
int smalfit_engine_set_option(smalfit_engine* engine, int option, int value);
/* replaces: betas_prec / mean_betas          reference smal_fitter/smal_fitter.py:48-69
 * host arrays: prec (dim,dim), mean (dim); dim = 26 (unity prior: betas|log scales) or <= 20 */
int smalfit_engine_set_shape_prior(smalfit_engine* engine, const float* prec, const float* mean, int dim);

/* ---- SMAL.__call__ ------------------------------------------------------------------------------
 * replaces: SMAL.__call__(beta, theta, betas_logscale=...)   reference smal_model/smal_torch.py:99-189
 * beta (M,nb) theta (M,35,3) logscale (M,6) or NULL -> verts (M,V,3) joints (M,41,3)
 * optional outputs (NULL to skip): Rs (M,35,3,3), v_shaped (M,V,3) */
int smalfit_lbs_forward(smalfit_engine* engine, void* stream, int M, int nb, const float* beta,
                        const float* theta, const float* logscale, float* verts, float* joints,
                        float* Rs, float* v_shaped);
/* adjoint of the above for upstream gradients dverts (M,V,3) and/or djoints (M,41,3) (either may be
 * NULL): dbeta (M,nb), dtheta (M,35,3), dlogscale (M,6) or NULL.  Recomputes the forward internally. */
int smalfit_lbs_backward(smalfit_engine* engine, void* stream, int M, int nb, const float* beta,
                         const float* theta, const float* logscale, const float* dverts,
                         const float* djoints, float* dbeta, float* dtheta, float* dlogscale);

/* the same call with every option of the reference's signature (smal_torch.py:99):
 *   SMAL.__call__(beta, theta, trans=None, del_v=None, betas_logscale=None, get_skin=True, v_template=None)
 * theta as (M,35,3) axis-angles OR Rs as (M,35,3,3) rotation matrices (:132-133, `len(theta.shape) == 4`);
 * v_offset (M,V,3) is added to the shaped template: del_v, plus (v_template - the model's template) for a per-call
 * template (:107-122).  trans is a plain addition after the call.  Forward reads the inputs and writes verts / joints
 * (+ Rs_out, v_shaped when not NULL); backward recomputes the forward and reads dverts / djoints (either may be NULL),
 * writing whichever of dbeta, dtheta (axis-angle input) or dRs (matrix input), dlogscale, dv_offset is not NULL. */
typedef struct smalfit_lbs_args {
  int num_frames, num_betas;
  const float* beta;       /* (M,nb)                          */
  const float* theta;      /* (M,35,3) or NULL when Rs given  */
  const float* Rs;         /* (M,35,3,3) or NULL              */
  const float* logscale;   /* (M,6) or NULL                   */
  const float* v_offset;   /* (M,V,3) or NULL                 */
  float *verts, *joints, *Rs_out, *v_shaped;            /* forward outputs: (M,V,3), (M,41,3), (M,35,3,3), (M,V,3) */
  const float *dverts, *djoints;                        /* backward inputs                                          */
  float *dbeta, *dtheta, *dRs, *dlogscale, *dv_offset;  /* backward outputs                                         */
} smalfit_lbs_args;
int smalfit_lbs_forward_ex(smalfit_engine* engine, void* stream, const smalfit_lbs_args* args);
int smalfit_lbs_backward_ex(smalfit_engine* engine, void* stream, const smalfit_lbs_args* args);

/* ---- batch_rodrigues ----------------------------------------------------------------------------
 * replaces: batch_rodrigues(theta)                           reference smal_model/batch_lbs.py:33-52 */
int smalfit_rodrigues(void* stream, int count, const float* theta, float* R);
int smalfit_rodrigues_backward(void* stream, int count, const float* theta, const float* dR, float* dtheta);

/* replaces: batch_global_rigid_transformation(Rs, Js, parent, betas_logscale=...)   reference smal_model/batch_lbs.py:75-170
 * Rs (count,35,3,3)  Js (count,35,3)  parents: host int[35] with parents[i] < i (parents[0] ignored)
 * logscale (count,6) or NULL -> new_J (count,35,3), A (count,35,4,4).  (Inside the fitting path the chain and its
 * adjoint are part of smalfit_lbs_forward / _backward and smalfit_fit_eval.) */
int smalfit_global_rigid_transformation(void* stream, int count, const float* Rs, const float* Js,
                                        const int* parents /*host*/, const float* logscale, float* new_J, float* A);
/* its adjoint (autograd of the reference's function): d_new_J (count,35,3), d_A (count,35,4,4) -> dRs (count,35,3,3),
 * dJs (count,35,3), dlogscale (count,6) (ignored when logscale is NULL); scratch: count * 840 floats of device memory */
int smalfit_global_rigid_transformation_backward(void* stream, int count, const float* Rs, const float* Js,
                                                 const int* parents /*host*/, const float* logscale, const float* d_new_J,
                                                 const float* d_A, float* scratch, float* dRs, float* dJs, float* dlogscale);

/* ---- Renderer.forward ---------------------------------------------------------------------------
 * replaces: Renderer.forward(vertices, points, faces)        reference smal_fitter/p3d_renderer.py:61-74
 * verts (M,V,3) world space -> sil (M,S,S) soft silhouette (sigma 1e-4, blur log(9999)*1e-4, K=100)
 * points (M,P,3) -> proj_points (M,P,2) as (row, col) screen coordinates; points may be NULL */
int smalfit_render_forward(smalfit_engine* engine, void* stream, int M, const float* verts,
                           const float* points, int P, float* sil, float* proj_points);
/* Renderer.forward(render_texture=True), colour branch (p3d_renderer.py:41-59,70-72): hard rasterisation
 * (blur_radius 0, faces_per_pixel 1) + HardPhongShader, one point light at (0,0,3), constant vertex colour `rgb`
 * (host, 3 floats in [0,1]; the reference uses config.MESH_COLOR / 255), white background.  Visualisation only: no
 * gradient.  verts [M][V][3] world space with the translation applied; image [M][3][S][S]. */
int smalfit_render_color(smalfit_engine* engine, void* stream, int num_frames, const float* verts, const float* rgb /*host*/,
                         float* image);

/* adjoint wrt verts given the saved silhouette and dL/dsil (M,S,S) */
int smalfit_render_backward(smalfit_engine* engine, void* stream, int M, const float* verts,
                            const float* sil, const float* dsil, float* dverts);
int smalfit_project_points_backward(void* stream, int count, int image_size, const float* points,
                                    const float* dproj, float* dpoints);

/* ---- SMALFitter.forward + get_temporal + backward, fused ------------------------------------------
 * replaces: SMALFitter.forward / get_temporal and the autograd backward of their sum
 *           reference smal_fitter/smal_fitter.py:107-190, smal_fitter/optimize_to_joints.py:117-136
 * Evaluates M consecutive frames, grouped into windows of `window` frames (the last may be ragged),
 * with the reference's per-window normalisers, and writes every loss term plus the gradient of
 * their sum with respect to each parameter tensor. */
typedef struct smalfit_fit_args {
  unsigned struct_size;           /* sizeof(smalfit_fit_args) of the caller's header; checked by the library */
  int num_frames;                 /* M                                                          */
  int window;                     /* WINDOW_SIZE; normalisers use the size of each frame's window (see frame_offset) */
  int logscale_mode;              /* 0: no limb scales, 1: shared (6,), 2: per frame (M,6)       */
  int temporal;                   /* include the temporal term over these M frames              */
  int shape_prior_dim;            /* 0 = use the dim given to set_shape_prior                    */
  float w_j2d, w_sil, w_betas, w_pose, w_splay, w_temp;
  const float* betas;             /* (20,) shared                                               */
  const float* log_beta_scales;   /* (6,) or (M,6) or NULL                                      */
  const float* global_rotation;   /* (M,3)                                                      */
  const float* joint_rotations;   /* (M,34,3)                                                   */
  const float* trans;             /* (M,3)                                                      */
  const float* global_mask;       /* (3,)   or NULL (= ones)                                    */
  const float* rotation_mask;     /* (34,3) or NULL (= ones)                                    */
  const float* target_joints;     /* (M,25,2) (row, col)                                        */
  const float* target_visibility; /* (M,25) float {0,1}                                         */
  const float* target_sil;        /* (M,S,S); may be NULL when w_sil == 0                       */
  const float* halo_prev;         /* (108,) masked theta(105)|trans(3) of the frame before frame 0, or NULL */
  const float* halo_next;         /* (108,) of the frame after frame M-1, or NULL               */
  float* losses;                  /* (9,) see SMALFIT_NUM_LOSS_TERMS                            */
  float* g_betas;                 /* (20,)           or NULL                                    */
  float* g_log_beta_scales;       /* (6,) / (M,6)    or NULL                                    */
  float* g_global_rotation;       /* (M,3)           or NULL                                    */
  float* g_joint_rotations;       /* (M,34,3)        or NULL                                    */
  float* g_trans;                 /* (M,3)           or NULL                                    */
  float* sil_out;                 /* (M,S,S) rendered silhouettes or NULL                       */
  float* proj_out;                /* (M,25,2) projected keypoints or NULL                       */
  float* verts_out;               /* (M,V,3) translated vertices or NULL                        */
  const unsigned char* target_sil_u8; /* (M,S,S) target silhouettes as bytes, t = b / 255 (what the 8-bit masks of
                                     data_loader.py:43 hold); used instead of target_sil when not NULL          */
  float w_limit;                  /* joint-limit hinge weight (OPT_WEIGHTS row 5).  Ignored -- like the reference, whose
                                     term is commented out while its weight table says 100 -- until
                                     smalfit_engine_set_joint_limits has been called                          */
  /* Position of these M frames in their sequence.  The reference groups the frames of a SEQUENCE into consecutive windows
   * of WINDOW_SIZE (optimize_to_joints.py:119-120, last one ragged) and normalises each window's terms by its size
   * (smal_fitter.py:144,157,173).  An evaluation may hold any contiguous part of the sequence -- a shard, or ONE frame of
   * an 8-frame window (one frame per GPU): local frame n is sequence frame frame_offset + n, its window is
   * [w, min(w + window, total_frames)) with w = ((frame_offset + n) / window) * window.  The shape-prior term (once per
   * window in the reference) is owned by the evaluation that holds the window's first frame.
   * Zeros = the M frames are the whole sequence. */
  int frame_offset;               /* index of local frame 0 in the sequence                     */
  int total_frames;               /* frames in the whole sequence; 0 = frame_offset + num_frames */
} smalfit_fit_args;

int smalfit_fit_eval(smalfit_engine* engine, void* stream, const smalfit_fit_args* args);

/* ---- the epoch loop: loss + backward + optimizer.step(), `iterations` times in one call ---------------------------
 * replaces: the body of the epoch loop                        reference smal_fitter/optimize_to_joints.py:113-137
 *   optimizer.zero_grad(); acc_loss = sum over windows of model(...) + temporal; acc_loss.backward(); optimizer.step()
 * with optimizer = torch.optim.Adam(model.parameters(), lr, betas=(0.5, 0.999)) created per stage (:96).
 * The fit parameters, their gradient and the two Adam moments are four flat device buffers of one layout (the caller
 * chooses it; the pointers inside smalfit_fit_args point into `param` / `grad`); the trainable tensors of the stage are
 * up to four [begin, end) ranges of that layout and are updated by ONE kernel launch per iteration.
 * step = optimiser steps already taken in this stage; when it is 0 the moments are taken as zero and not read, so a new
 * stage needs no fill.  Everything is enqueued on `stream`; nothing synchronises. */
typedef struct smalfit_adam_args {
  float* param;        /* flat parameters                                   */
  float* grad;         /* their gradient (written by the evaluation)        */
  float* exp_avg;      /* Adam first moment                                 */
  float* exp_avg_sq;   /* Adam second moment                                */
  int num_segments;    /* 0..4 trainable ranges of the flat layout          */
  int seg_begin[4];
  int seg_end[4];
  float lr, beta1, beta2, eps;
  int step;            /* steps already taken by this stage's optimiser     */
} smalfit_adam_args;
int smalfit_fit_run(smalfit_engine* engine, void* stream, const smalfit_fit_args* args, const smalfit_adam_args* adam,
                    int iterations);
/* enable != 0: smalfit_fit_run captures one iteration (the evaluation's kernels + the Adam launch) into a HIP graph the
 * first time it sees a set of arguments on a (non-default) stream and replays it `iterations` times; the Adam step count
 * then lives in a device counter and the bias corrections are formed on the device.  Off by default. */
int smalfit_engine_set_graph(smalfit_engine* engine, int enable);
/* optimizer.step() alone on the ranges (t = adam->step + 1) */
int smalfit_adam_segments(void* stream, const smalfit_adam_args* adam);

/* ---- frame-sharded fitting (one process per GPU; no counterpart in the reference) --------------------
 * Per iteration a rank evaluates its frames, steps its per-frame parameters, and contributes one record to an
 * all-gather: record = [partial gradient of the shared parameters (num_shared) | masked theta(105)|trans(3) of its first
 * frame | of its last frame] -- the partial shape gradient and the neighbours' temporal halo of the next iteration. */
/* one rank's side of an iteration up to the collective, in ONE call: the evaluation (smalfit_fit_eval), Adam on the
 * per-frame ranges (`adam_local`, t = step + 1) and the record (smalfit_shard_record with the parameters / masks of `args`)
 * written straight into the collective's send buffer */
int smalfit_shard_local_step(smalfit_engine* engine, void* stream, const smalfit_fit_args* args,
                             const smalfit_adam_args* adam_local, int num_shared, const float* shared_grad,
                             float* record /*(num_shared + 216)*/);
int smalfit_shard_record(void* stream, int num_shared, const float* shared_grad, int num_frames,
                         const float* global_rotation, const float* joint_rotations, const float* trans,
                         const float* global_mask /*(3,)*/, const float* rotation_mask /*(34,3)*/,
                         float* record /*(num_shared + 216)*/);
/* after the all-gather: grad[0:num_shared] = sum over ranks (in rank order: the same bits on every rank) of the gathered
 * partial gradients, then Adam (t = adam->step + 1) on the first num_trainable of them.  gathered: (world_size, record_stride) */
int smalfit_shard_reduce_step(void* stream, int world_size, int record_stride, const float* gathered, int num_shared,
                              int num_trainable, const smalfit_adam_args* adam);

/* The sharded loop of a whole stage in ONE call (round 4; until then a host loop made one call before and one after the
 * collective of every iteration): `iterations` x [ smalfit_shard_local_step -> all-gather of the record -> smalfit_shard_reduce_step ],
 * everything enqueued on `stream`, nothing synchronises, no host code between two iterations.  The library links against no
 * communication library: the collective is the caller's, handed over as a function that enqueues
 *     all-gather of `count` floats per rank:  send (count) -> recv (world_size x count, rank-major)   on `stream`
 * and returns 0 on success.  With RCCL that is smalfit_rccl_allgather below on the process group's own communicator (same
 * stream as the kernels: no event hand-over per iteration); any other transport (gloo in the tests) is a host callback.
 * `gathered` is persistent: args->halo_prev / halo_next point INTO it (row rank-1, floats [num_shared+108, num_shared+216) and
 * row rank+1, floats [num_shared, num_shared+108)), so the all-gather of iteration i delivers the temporal halo of iteration
 * i+1 in place; the caller fills it once before the first iteration (one stand-alone all-gather of the boundary records).
 * adam_local: the per-frame ranges; adam_shared: the same flat buffers (its ranges are ignored: the first
 * num_trainable_shared floats are stepped from the rank-ordered sum); both with the same `step`. */
typedef int (*smalfit_allgather_fn)(void* ctx, const float* send, float* recv, int count, void* stream);
typedef struct smalfit_shard_args {
  unsigned struct_size;        /* sizeof(smalfit_shard_args) of the caller's header */
  int world_size, rank;
  int num_shared;              /* shared floats at the head of the flat gradient (20, or 26 with shared limb scales) */
  int num_trainable_shared;    /* how many of them this stage trains (0 in stage 0) */
  const float* shared_grad;    /* this rank's partial gradient of the shared parameters (num_shared), written by the evaluation */
  float* record;               /* (num_shared + 216): the collective's send buffer */
  float* gathered;             /* (world_size, num_shared + 216): its receive buffer */
  smalfit_allgather_fn allgather;
  void* allgather_ctx;
} smalfit_shard_args;
int smalfit_shard_run(smalfit_engine* engine, void* stream, const smalfit_fit_args* args, const smalfit_adam_args* adam_local,
                      const smalfit_adam_args* adam_shared, const smalfit_shard_args* shard, int iterations);
/* a smalfit_allgather_fn over RCCL without linking it: ctx points at {ncclComm_t comm; address of ncclAllGather} -- both
 * belong to the caller's process (torch.distributed's communicator and the librccl.so it loaded) */
typedef struct smalfit_rccl_ctx {
  void* comm;                  /* ncclComm_t */
  void* nccl_all_gather;       /* ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) */
} smalfit_rccl_ctx;
int smalfit_rccl_allgather(void* ctx, const float* send, float* recv, int count, void* stream);

/* ---- Prior.__call__ ---------------------------------------------------------------------------------
 * replaces: Prior.__call__(x)                                 reference smal_fitter/priors/pose_prior_35.py:112-124
 * x (N,105) -> out (N,105) = (((x - mean) prec) * mask)^2 with the prior given to set_pose_prior */
int smalfit_pose_prior(smalfit_engine* engine, void* stream, int N, const float* x, float* out);
int smalfit_pose_prior_backward(smalfit_engine* engine, void* stream, int N, const float* x, const float* dout,
                                float* dx);

/* ---- SMALFitter.get_temporal ----------------------------------------------------------------------------
 * replaces: SMALFitter.get_temporal(w_temp) and its backward    reference smal_fitter/smal_fitter.py:177-190
 * losses (3,) = joint, global, trans; gradients wrt the raw (unmasked) parameters; any g_* may be NULL */
int smalfit_temporal(smalfit_engine* engine, void* stream, int N, float w_temp, const float* global_rotation,
                     const float* joint_rotations, const float* trans, const float* global_mask,
                     const float* rotation_mask, float* losses, float* g_global_rotation,
                     float* g_joint_rotations, float* g_trans);

/* ---- fitter_3d: SMAL-to-mesh objective -----------------------------------------
 * replaces: Stage.forward / Stage.step                       reference fitter_3d/trainer.py:205-241
 * and the PyTorch3D v0.2.5 calls inside it (sample_points_from_meshes, chamfer_distance, mesh_edge_loss,
 * mesh_normal_consistency, mesh_laplacian_smoothing("uniform")) together with their autograd.
 *
 * smalfit_mesh_objective: the deforming source meshes, all of one topology (faces: host, (F,3) int32; every face
 * must have 3 distinct vertices).  Holds the edge / one-ring / face-pair tables and the work buffers for up to
 * max_meshes meshes and max_points target points per mesh. */
typedef struct smalfit_mesh_objective smalfit_mesh_objective;
typedef struct smalfit_mesh_targets smalfit_mesh_targets;
#define SMALFIT_NUM_MESH_LOSS_TERMS 5 /* chamfer, edge, normal, laplacian (unweighted), weighted total */

int smalfit_mesh_objective_create(int num_verts, int num_faces, const int* faces /*host*/, int max_meshes,
                                  int max_points, smalfit_mesh_objective** out);
void smalfit_mesh_objective_destroy(smalfit_mesh_objective* objective);
/* unique edges and pairs of faces sharing an edge (Meshes.edges_packed; the pair table of mesh_normal_consistency) */
int smalfit_mesh_objective_counts(const smalfit_mesh_objective* objective, int* num_edges, int* num_face_pairs);
/* verts = lbs_verts + trans + deform_verts (SMAL3DFitter.forward, trainer.py:94-108; deform_verts may be NULL), then
 * losses[0..3] = chamfer(points, verts), edge, normal, laplacian; losses[4] = sum of weights[i] * term over the terms
 * with weights[i] > 0 (weights: host, order w_chamfer, w_edge, w_normal, w_laplacian; trainer.py:31,203).
 * lbs_verts (N,V,3)  trans (N,3)  deform_verts (N,V,3)  points (N,S,3) (ignored when w_chamfer <= 0)
 * -> verts_out (N,V,3) or NULL, losses (5, device), dverts (N,V,3) = d total / d verts (also the gradient of
 * deform_verts), dtrans (N,3).  Chain dverts through smalfit_lbs_backward for the SMAL parameters. */
int smalfit_mesh_objective_eval(smalfit_mesh_objective* objective, void* stream, int num_meshes,
                                const float* lbs_verts, const float* trans, const float* deform_verts,
                                const float* points, int num_points, const float* weights /*host*/,
                                float* verts_out, float* losses, float* dverts, float* dtrans);

/* smalfit_mesh_targets: the target meshes, packed (all arguments host): vert_counts / face_counts (num_meshes),
 * verts (sum V,3), faces (sum F,3) with indices local to each mesh.
 * replaces: the Meshes object of fitter_3d/utils.py:253 as far as sample_points_from_meshes needs it */
int smalfit_mesh_targets_create(int num_meshes, const int* vert_counts, const int* face_counts, const float* verts,
                                const int* faces, smalfit_mesh_targets** out);
void smalfit_mesh_targets_destroy(smalfit_mesh_targets* targets);
/* replaces: sample_points_from_meshes(target_meshes, num_points)   trainer.py:209
 * face ~ area, barycentric (1 - sqrt u, sqrt u (1 - v), sqrt u v); Philox-4x32-10 keyed by `seed`, counter
 * (sample, mesh, iteration): the same (seed, iteration) always gives the same points.  points (N,S,3) device. */
int smalfit_mesh_targets_sample(smalfit_mesh_targets* targets, void* stream, int num_points,
                                unsigned long long seed, unsigned int iteration, float* points);

/* ---- Stage.step in one call ------------------------------------------------------------------------------------
 * replaces: one iteration of Stage.run                       reference fitter_3d/trainer.py:229-241,257-262
 *   new_src_verts = smal_3d_fitter(); loss = forward(...); loss.backward(); optimizer.step()
 * = SMAL forward on [global_rot | joint_rot], target points sampled (or taken from `points`), the four-term objective,
 * its gradient back through the SMAL model, torch.optim.Adam on every parameter whose learning rate is > 0 (the
 * parameter groups of trainer.py:113-119 with their custom learning rates) -- 14 kernel launches, no host
 * synchronisation.  All pointers are device pointers unless noted.  log_beta_scales is read, never trained
 * (requires_grad=False in the reference, trainer.py:64-65). */
typedef struct smalfit_fit3d_args {
  int num_meshes;                  /* N */
  int num_betas;                   /* columns of betas (20) */
  int num_points;                  /* S target points per mesh (3000 in the reference) */
  float* betas;                    /* (N,num_betas) */
  const float* log_beta_scales;    /* (N,6) or NULL */
  float* global_rot;               /* (N,3) */
  float* joint_rot;                /* (N,34,3) */
  float* trans;                    /* (N,3) */
  float* deform_verts;             /* (N,V,3) or NULL (then treated as zero and not trainable) */
  /* Adam: learning rate per parameter (<= 0: frozen in this stage) and its state (exp_avg, exp_avg_sq), same shapes */
  float lr_betas, lr_global_rot, lr_joint_rot, lr_trans, lr_deform_verts;
  float *m_betas, *v_betas, *m_global_rot, *v_global_rot, *m_joint_rot, *v_joint_rot, *m_trans, *v_trans,
        *m_deform_verts, *v_deform_verts;
  float beta1, beta2, eps;         /* torch defaults 0.9, 0.999, 1e-8 (trainer.py:194) */
  int adam_t;                      /* 1-based step count of this stage's optimiser */
  float weights[4];                /* w_chamfer, w_edge, w_normal, w_laplacian; <= 0 skips the term */
  const float* points;             /* (N,S,3) target points, or NULL: sample them from the target meshes */
  unsigned long long seed;         /* sampler key and counter, see smalfit_mesh_targets_sample */
  unsigned int iteration;
  float* points_out;               /* optional (N,S,3): the target points this step used */
  float* losses;                   /* (5): chamfer, edge, normal, laplacian, weighted total -- before the update */
  float* verts_out;                /* optional (N,V,3): the vertices the loss was evaluated at */
} smalfit_fit3d_args;
/* `targets` may be NULL when args->points is given or w_chamfer <= 0 */
int smalfit_fit3d_step(smalfit_engine* engine, smalfit_mesh_objective* objective, smalfit_mesh_targets* targets,
                       void* stream, const smalfit_fit3d_args* args);

/* ---- torch.optim.Adam.step -----------------------------------------------------------------------
 * replaces: torch.optim.Adam(lr, betas=(0.5, 0.999)).step()  reference smal_fitter/optimize_to_joints.py:96,137
 * t = 1-based step count; eps outside the bias-corrected sqrt, as torch does */
int smalfit_adam_step(void* stream, int count, float* param, const float* grad, float* exp_avg,
                      float* exp_avg_sq, float lr, float beta1, float beta2, float eps, int t);

#ifdef __cplusplus
}
#endif
#endif /* SMALFIT_H_ */
