/* libuuo_hip.so -- C ABI of the MI355X (gfx950) SMPL-to-marker fitting hot path.
 *
 * The reference (NicholasMilef/UUO-Mocap) is pure Python and has no FFI layer; its boundary for this
 * path is a set of Python callables.  This header is the C-ABI a binding for those
 * callables sits on: plain pointers and sizes, no torch types.  Each entry point cites the reference
 * interface it replaces (paths relative to the reference checkout).  The Python mirror of the
 * reference's operator surface (the uuo_mocap_amd package) binds these symbols with ctypes; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, a negative errno-style code otherwise; the message for the
 *    calling thread is available from uuo_last_error().
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Nothing synchronises the
 *    device unless the comment says so; no hidden allocation happens inside the *_eval / kernel-only
 *    calls (they use the workspace created by uuo_fit_create).
 *  - pointers named `d_*` are device pointers (HBM), `h_*` host pointers.  All floats are fp32,
 *    row-major, contiguous.  Rotations are 3x3 row-major (9 floats).
 *  - the library owns no threads and keeps no global state besides the last-error string.
 */
#ifndef UUO_HIP_H
#define UUO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UUO_NUM_JOINTS 24
#define UUO_NUM_BETAS 10
#define UUO_NUM_POSE_FEATS 207
#define UUO_NUM_EXTRA_JOINTS 21

typedef struct uuo_model uuo_model_t; /* device copies of the SMPL tables */
typedef struct uuo_fit uuo_fit_t;     /* per-sequence workspace (F frames, M markers) */

const char* uuo_last_error(void);
/* 3 since uuo_problem_t grew `w_soft` / `soft_tau` (2) and `n_corners` / `d_bary` (3), round 4; a binding checks it before it
 * hands structures over */
int uuo_abi_version(void);

/* ---- model ------------------------------------------------------------------------------------
 * Replaces smplx.create("./body_models/", model_type="smpl", gender="neutral", batch_size=1) as used
 * by SmplInference.__init__ (src/video_mocap/utils/smpl.py:22-27).  Host arrays in smplx's layouts:
 * v_template [V,3], shapedirs [V,3,10], posedirs [207,V*3], J_regressor [24,V], lbs_weights [V,24],
 * parents [24] (parents[0] ignored), extra_joint_vids [21] (VertexJointSelector ids). */
int uuo_model_create(const float* h_v_template, const float* h_shapedirs, const float* h_posedirs,
                     const float* h_J_regressor, const float* h_lbs_weights, const int64_t* h_parents,
                     const int64_t* h_extra_joint_vids, int num_verts, uuo_model_t** out);
int uuo_model_destroy(uuo_model_t* model);
int uuo_model_num_verts(const uuo_model_t* model);

/* ---- SMPL forward (materialising) ---------------------------------------------------------------
 * Replaces SmplInference.forward (src/video_mocap/utils/smpl.py:29-50) = smplx SMPL.forward with
 * pose2rot=False: d_poses [F,23,9], d_betas [betas_rows,10] with betas_rows in {1,F}, d_root [F,9],
 * d_trans [F,3] or NULL.  Outputs d_verts [F,V,3] and d_joints [F,45,3] (either may be NULL). */
int uuo_smpl_forward(uuo_model_t* model, void* stream, int F, const float* d_poses, const float* d_betas,
                     int betas_rows, const float* d_root, const float* d_trans, float* d_verts,
                     float* d_joints);

/* Backward of uuo_smpl_forward for given upstream gradients (what torch autograd computes through smplx.lbs when a
 * caller differentiates SmplInference.forward itself, e.g. a user-written closure; reference utils/smpl.py:29-50).
 *   d_up_verts  [F,6890,3] dL/dvertices or NULL,  d_up_joints [F,45,3] dL/djoints or NULL (at least one)
 *   d_g_poses [F,23,3,3], d_g_betas [F,10] (per frame: sum the rows if betas had one row), d_g_root [F,1,3,3],
 *   d_g_trans [F,3];  d_scratch: F*24 floats of device scratch.  Asynchronous on `stream`.
 * With d_up_verts the backward is dense (every vertex carries a gradient): v_posed and the transposed 207 x 20670 blend
 * contraction run on the matrix pipe (csrc/dense_bwd.hip); the library keeps ~80 MB of scratch per stream and frame count
 * for it (allocated on first use, freed with the model). */
int uuo_smpl_backward(uuo_model_t* model, void* stream, int F, const float* d_poses, const float* d_betas,
                      int betas_rows, const float* d_root, const float* d_trans, const float* d_up_verts,
                      const float* d_up_joints, float* d_g_poses, float* d_g_betas, float* d_g_root, float* d_g_trans,
                      float* d_scratch);

/* ---- K=1 nearest neighbour ------------------------------------------------------------------------
 * Replaces pytorch3d.ops.knn_points(K=1) as reached through pytorch3d.loss.chamfer_distance from
 * weighted_chamfer_distance (src/video_mocap/losses/chamfer_distance.py:5-21) and
 * markers_utils.py:471-475,575-579: for every query x[n,i] the first index of the minimum of
 * ((dx*dx)+(dy*dy))+(dz*dz) over y[n,:] (separately rounded, no FMA).  d_x [N,P1,3], d_y [N,P2,3];
 * d_y_subset (optional, [P2s] int32) restricts/reorders the candidates: candidate c is
 * y[n, d_y_subset[c]] and the reported index is c.  Outputs d_dist [N,P1], d_idx [N,P1] (int32). */
int uuo_nn_argmin(void* stream, int N, int P1, int P2, const float* d_x, const float* d_y,
                  const int32_t* d_y_subset, int P2s, float* d_dist, int32_t* d_idx, void* d_workspace_u64);

/* ---- marker placement -----------------------------------------------------------------------------
 * Replaces compute_nearest_points (src/video_mocap/optimization.py:402-642; use_mean, granularity
 * "full"): idx[m] = argmin_v mean_{f valid} ||verts[f,v]-markers[f,m]||, numpy fp32 semantics
 * (sequential-in-f accumulation, first index).  d_valid [F] uint8.  Output d_idx [M] int32. */
int uuo_assign_mean_argmin(void* stream, int F, int M, int V, const float* d_verts, const float* d_markers,
                           const uint8_t* d_valid, int32_t* d_idx, void* d_workspace_u64);

/* ---- rigidity matrix of the marker segmentation ---------------------------------------------------
 * Replaces the double loop of segment_rigid (src/video_mocap/markers/markers_utils.py:254-259):
 *   d_std[i*M+j] = np.std(np.linalg.norm(points[:, i] - points[:, j], axis=-1))     d_points [F,M,3] float32
 * in numpy's float32 arithmetic and numpy's summation order (pairwise summation, 8 accumulators per 128-element block,
 * 8192-element reduction chunks), so the values -- which the average-linkage clustering cuts at 5 mm -- are bit-equal
 * to the reference's.  d_std [M,M] float32 (the reference stores them in a float64 matrix: exact).  Asynchronous. */
int uuo_rigid_distance_std(void* stream, int F, int M, const float* d_points, float* d_std);

/* ---- EXTENSION: soft-assignment (soft-min) nearest neighbour ----------------------------------------
 * Not reference behaviour (the reference's chamfer term is the hard K=1 minimum above; BASELINE's north star names a
 * soft assignment).  softmin[n,i] = -tau log sum_j exp(-|x[n,i]-y[n,j]|^2 / tau); d_dmin / d_sumexp ([N,P1]) are the
 * exact minimum and the normaliser sum_j exp((dmin - d2)/tau), kept for the backward.  Backward: gradients of
 * sum_i grad_softmin[n,i] softmin[n,i] with respect to x (d_gx [N,P1,3]) and y (d_gy [N,P2,3]); either may be NULL.
 * d_ws: N*P1 uint64. */
int uuo_soft_nn_forward(void* stream, int N, int P1, int P2, const float* d_x, const float* d_y, float tau,
                        float* d_softmin, float* d_dmin, float* d_sumexp, void* d_ws);
int uuo_soft_nn_backward(void* stream, int N, int P1, int P2, const float* d_x, const float* d_y, float tau,
                         const float* d_dmin, const float* d_sumexp, const float* d_grad_softmin, float* d_gx,
                         float* d_gy);

/* ---- barycentric marker placement ------------------------------------------------------------------
 * Replaces igl.signed_distance + trimesh.triangles.points_to_barycentric in compute_nearest_points with
 * compute_locations.use_barycentric (src/video_mocap/optimization.py:494-500,519-523): for every query
 * d_points[f,m] the closest point on the triangle mesh (d_verts[f] [V,3], d_faces [NF,3] int32).
 * Outputs, all [F,M,...]: d_dist (unsigned Euclidean distance), d_face (winning face, lowest index on
 * exact ties, -1 if no face is usable), d_closest [.,3] and d_bary [.,3] (trimesh "cramer" coordinates of
 * the closest point with respect to the face's corners in d_faces order).  fp32 throughout. */
int uuo_mesh_closest_points(void* stream, int F, int M, int V, int NF, const float* d_verts,
                            const int32_t* d_faces, const float* d_points, float* d_dist, int32_t* d_face,
                            float* d_closest, float* d_bary);

/* ---- stage problems -------------------------------------------------------------------------------
 * One closure evaluation (forward + backward) of the three L-BFGS stages, on flat parameter vectors
 * packed exactly as the reference hands them to torch.optim.LBFGS:
 *   UUO_STAGE_CHAMFER  closure_stage_chamfer   (optimization.py:187-275)  x = [trans 3F | z F | betas 10 | pose 207F]
 *   UUO_STAGE_MARKER   closure_stage_marker_pose (optimization.py:329-394) x = [pose 207F | betas 10 | root 9F | trans 3F]
 *   UUO_STAGE_PART     closure_fit_subtree     (markers/markers_utils.py:454-562) x = [z 1 | trans 3F | betas 10]
 */
enum { UUO_STAGE_CHAMFER = 0, UUO_STAGE_MARKER = 1, UUO_STAGE_PART = 2 };

typedef struct {
  int32_t stage;             /* UUO_STAGE_* */
  int32_t F, M;              /* frames, markers */
  const float* d_markers;    /* [F,M,3]; exact zeros = missing (optimization.py:703-715) */
  const float* d_o_pose;     /* [F,23,9]: prior target (chamfer, marker) / the fixed body pose (part) */
  const float* d_o_betas;    /* [10] prior target */
  const float* d_root;       /* [F,9] fixed root orientation (chamfer, part); unused for marker */
  const int32_t* d_assign;   /* marker stage: [M] vertex id per marker column */
  const int32_t* d_subset;   /* part stage: [n_subset] vertex ids (candidate order = tie order) */
  int32_t n_subset;
  float w_data;              /* full_chamfer / marker / chamfer weight */
  float w_pose;              /* reg_pose_body weight (0 = term absent) */
  float w_betas;             /* reg_betas weight (0 = term absent) */
  float marker_distance;     /* MARKER_DISTANCE, utils/settings.py:1 */
  uint64_t pose_cache_id;    /* part stage: non-zero = the body pose d_o_pose is constant for every evaluation that
                                carries this id, so the pose-corrective blend (207 x 20670 contraction, 70 % of the
                                forward's arithmetic) is computed once and re-used; 0 = recompute every evaluation */
  /* EXTENSION (not reference behaviour; BASELINE's north star names a soft-assignment Chamfer distance, configs[2] a
   * "soft-assignment path" of hmr_part.yaml): the data term's hard minimum over the vertices joined or replaced by the soft
   * minimum  -tau log sum_v exp(-|x - v|^2 / tau).  w_soft = 0: the reference's term alone.
   *   UUO_STAGE_PART    loss_data = (1 / (F M)) sum_{f,m} (w_data min_v d2 + w_soft softmin) over the candidate's vertices
   *                     (markers/markers_utils.py:471-475); fused closure k_part_soft: needs pose_cache_id != 0 and M <= 16
   *   UUO_STAGE_CHAMFER loss_data = (1 / sum mask) sum_{f,m} mask_fm (w_data min_v d2 + w_soft softmin) over all vertices
   *                     (losses/chamfer_distance.py:5-21); soft-min kernels + the dense backward on the matrix pipe; not
   *                     available inside lock-step batches (uuo_batch_*) */
  float w_soft;              /* stages.<stage>.losses.soft_chamfer (0 = absent) */
  float soft_tau;            /* temperature in m^2 (> 0 when w_soft != 0) */
  /* Marker stage on a three-corner (barycentric) placement -- compute_locations.use_barycentric of the reference
   * (optimization.py:494-523; the closure's virtual markers are `placement @ vertices`, :345-351, every row of the placement
   * holding the barycentric coordinates of a surface point at its face's three corners).  n_corners = 3: d_assign is
   * [M][3] vertex ids, d_bary [M][3] their weights, virtual marker m = sum_k d_bary[m][k] v[d_assign[m][k]].
   * n_corners = 0 or 1: the one-hot placement of the shipped configs, d_assign [M], d_bary unused. */
  int32_t n_corners;
  const float* d_bary;
} uuo_problem_t;

int uuo_fit_create(uuo_model_t* model, int F, int M, uuo_fit_t** out);
int uuo_fit_destroy(uuo_fit_t* fit);
/* number of parameters of a stage at (F): 211F+10 / 219F+10 / 3F+11 */
int uuo_problem_num_params(const uuo_problem_t* p);

/* One closure evaluation at d_x: writes loss to d_loss[0] and the flat gradient to d_grad.
 * d_nn_idx (optional, [F,M] int32) receives the nearest-vertex assignment of the chamfer/part data term
 * (vertex ids for chamfer, candidate positions for part).  Asynchronous on `stream`. */
int uuo_closure_eval(uuo_fit_t* fit, void* stream, const uuo_problem_t* p, const float* d_x, float* d_loss,
                     float* d_grad, int32_t* d_nn_idx);

/* ---- L-BFGS ----------------------------------------------------------------------------------------
 * Replaces torch.optim.LBFGS(params, max_iter, tolerance_grad, tolerance_change, lr,
 * line_search_fn="strong_wolfe").step(closure) as constructed at optimization.py:176-183,319-326 and
 * markers/markers_utils.py:428-435 (history 100, max_eval = max_iter*5/4, c1 1e-4, c2 0.9; torch 2.10
 * semantics incl. max_ls = max_eval - evals).  d_x is updated in place.  Synchronises `stream`
 * (one small read-back per closure evaluation). */
typedef struct {
  int32_t max_iter;
  int32_t history_size;     /* 100 */
  float lr;
  float tolerance_grad;
  float tolerance_change;
  int32_t max_eval;         /* <=0: max_iter*5/4 */
  int32_t verbose;          /* print "<name> <iteration> <loss>" per closure like the reference's verbose flag */
} uuo_lbfgs_options_t;

typedef struct {
  int32_t n_iter;
  int32_t n_eval;
  float first_loss;
  float final_loss;
  int32_t stop_reason;      /* 0 max_iter, 1 max_eval, 2 grad tol, 3 step tol, 4 loss tol, 5 gtd, 6 initial grad tol */
  float device_ms;          /* HIP-event time of the whole solve */
} uuo_lbfgs_stats_t;

/* optional per-closure callback (mirrors iter_fn / verbose prints, optimization.py:259-272,378-391): called on
 * the host, from the thread that called uuo_lbfgs_solve, after each closure evaluation with (user, evaluation
 * index, loss, device pointer of the parameter vector that was evaluated -- valid, and ordered after the
 * evaluation on the solve's stream, until the callback returns). */
typedef void (*uuo_eval_callback_t)(void* user, int eval_index, float loss, const float* d_x_eval);

int uuo_lbfgs_solve(uuo_fit_t* fit, void* stream, const uuo_problem_t* p, float* d_x,
                    const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats, uuo_eval_callback_t cb,
                    void* cb_user);

/* EXTENSION (BASELINE configs[3] "shared beta over xGMI"; NOT reference behaviour -- the reference fits every sequence with
 * its own betas, test/test.py:57-112): uuo_lbfgs_solve called together by the `world` ranks of a group, one stage problem
 * (same stage) per rank, as ONE joint L-BFGS problem whose 10 betas are shared by all ranks' sequences.  Each rank keeps its
 * own parameters and a replica of the betas on its own device; the driver is uuo_lbfgs_solve's, and all that crosses the
 * ranks goes through `gather` -- called on the host, from the calling thread, at the same points on every rank:
 *   once at the start (the betas themselves: rank 0's values win), once per closure evaluation (16 doubles + a status word: loss, g.d,
 *   gradient norms of the rank's own parameters, max|d|, its 10 betas-gradient entries) and once per iteration (the new
 *   Gram rows of the history, 627 doubles).  `gather(user, mine, n, all)` must fill all[r*n .. r*n+n) with rank r's `mine`
 *   for r = 0..world-1 (an all_gather: RCCL / gloo through torch.distributed in the Python mirror) and return 0.  Every rank
 *   reduces the gathered tables in rank order, so all ranks take bit-identical decisions and end with bit-identical betas.
 * With world = 1 the result is bit-identical to uuo_lbfgs_solve (chamfer and marker stages). */
typedef int (*uuo_gather_fn)(void* user, const double* mine, int n, double* all);
typedef struct {
  uuo_gather_fn gather;
  void* user;
  int32_t rank, world;
} uuo_shared_t;
int uuo_lbfgs_solve_shared(uuo_fit_t* fit, void* stream, const uuo_problem_t* p, float* d_x,
                           const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats, const uuo_shared_t* shared,
                           uuo_eval_callback_t cb, void* cb_user);

/* Node-local transport for uuo_shared_t.gather: a mailbox in POSIX shared memory (csrc/mailbox.hip).  One table per
 * concurrent lane of solves (name: "/something", the same on every rank; rank 0 creates it, the others wait for it);
 * uuo_mailbox_gather is a uuo_gather_fn whose `user` is the mailbox: rank r copies its message into its row and reads the
 * other rows as their sequence words arrive -- no system call, collective library or interpreter on the path of the
 * 17 doubles a closure evaluation exchanges.  Messages are <= 640 doubles.  `timeout_s` (<= 0: 120 s) bounds every wait. */
typedef struct uuo_mailbox uuo_mailbox_t;
int uuo_mailbox_open(const char* name, int32_t rank, int32_t world, double timeout_s, uuo_mailbox_t** out);
int uuo_mailbox_close(uuo_mailbox_t* mb);
int uuo_mailbox_gather(void* mailbox, const double* mine, int n, double* all);
int uuo_mailbox_stats(uuo_mailbox_t* mb, unsigned long long* gathers, unsigned long long* nanoseconds);

/* The same driver for a closure composed on the host (the reference's optional objectives: the 2D reprojection fit,
 * utils/hmr_utils.py:170-425 (step at :367) -- which also has a fused closure of its own, uuo_reprojection_* above; the chamfer / marker / part stages with velocity, ground, foot-contact,
 * per-part or reprojection terms, optimization.py:187-275,329-394, markers/markers_utils.py:454-562): replaces
 * torch.optim.LBFGS(params, ..., line_search_fn="strong_wolfe").step(closure) over the flat parameter vector d_x
 * (n floats, torch's order = the order of the params list).  `closure(user, stream, d_x_eval, d_loss, d_grad)` must
 * enqueue on `stream` the work that writes the loss (1 float) and the gradient (n floats) at d_x_eval and return 0;
 * a non-zero return aborts the solve with that code.  The history, the two-loop products, the line search and every
 * termination test run on the device / in the library as in uuo_lbfgs_solve. */
typedef int (*uuo_closure_fn)(void* user, void* stream, const float* d_x_eval, float* d_loss, float* d_grad);
int uuo_lbfgs_minimize(void* stream, int n, float* d_x, const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats,
                       uuo_closure_fn closure, void* user, uuo_eval_callback_t cb, void* cb_user);
/* ---- the 2D-prior fit as one fused closure ----------------------------------------------------------------
 * Replaces the closure of optim_reprojection (utils/hmr_utils.py:281-365: two SMPL forwards, perspective_projection
 * :14-52, masked key-point L2 :321 and a one-directional chamfer_distance :333 per evaluation) and, with
 * uuo_reprojection_solve, the torch.optim.LBFGS(...).step(closure) at :367 for ONE yaw hypothesis.  Parameter vector, in the
 * order of the reference's params list (:276-279):  x = [yaw 1 | body translation 3F (HMR axes) | camera translation 3 |
 * betas 10 (in the list but detached, :218,292: zero gradient, never moves)]  = 3F + 14 floats.
 * The body pose, the shape and the HMR root orientation are constants of this solve, so the caller runs ONE forward
 * (uuo_smpl_forward: HMR pose, the solve's betas, the HMR root orientation, zero translation) and hands its joints and
 * vertices over; the closure needs no skinning (uuo_mocap_amd/csrc/reprojection.hip). */
typedef struct {
  int32_t F, M, V, J;        /* frames, markers, vertices, joints per frame (SMPL + vertex joints: 45; at most 64) */
  const float* d_markers;    /* [F,M,3] */
  const float* d_joints0;    /* [F,J,3] joints of that forward (joint 0 = the pelvis the root rotation turns about) */
  const float* d_verts0;     /* [F,V,3] vertices of that forward */
  const float* d_kp_target;  /* [F,J,2] HMR key points (NaN already replaced by 0, :236) */
  const float* d_mask;       /* [F] 1 = frame has a valid HMR camera (:240) */
  float focal[2];            /* mean focal length / 256 (:258) */
  float center[2];           /* camera centre (zeros in HMR 2.0, hmr_utils.py:96) */
  float w_reprojection, w_chamfer; /* stages.reprojection_part.losses */
} uuo_reprojection_problem_t;
typedef struct uuo_reprojection uuo_reprojection_t;
int uuo_reprojection_create(const uuo_reprojection_problem_t* p, uuo_reprojection_t** out);
int uuo_reprojection_destroy(uuo_reprojection_t* h);
int uuo_reprojection_num_params(const uuo_reprojection_problem_t* p); /* 3F + 14 */
/* One closure evaluation at d_x: loss to d_loss[0], gradient to d_grad[3F+14]; optional outputs: d_kp [F,J,2] the projected
 * key points (+0.5, as the reference's joints_2d), d_nn_idx [F,M] the nearest vertex of every marker.  Asynchronous. */
int uuo_reprojection_eval(uuo_reprojection_t* h, void* stream, const float* d_x, float* d_loss, float* d_grad,
                          float* d_kp, int32_t* d_nn_idx);
/* uuo_lbfgs_solve for this closure.  The reference returns quantities of the LAST closure evaluation (its `nonlocal`
 * temporaries, :296-299,383-425), which is not necessarily the accepted point: d_x_last (optional, 3F+14) and d_kp_last
 * (optional, [F,J,2]) receive that evaluation's parameter vector and key points. */
int uuo_reprojection_solve(uuo_reprojection_t* h, void* stream, float* d_x, const uuo_lbfgs_options_t* opt,
                           uuo_lbfgs_stats_t* stats, float* d_x_last, float* d_kp_last, uuo_eval_callback_t cb,
                           void* cb_user);

/* How the solver's host threads wait for the device's reports (one 192-byte block in pinned memory per closure evaluation).
 * Default: they spin -- lowest latency, one CPU per solve in flight.  spin_polls >= 0: after that many polls a wait sleeps
 * sleep_ns nanoseconds at a time (for hosts whose CPU quota is smaller than the number of solves in flight: a throttled
 * cgroup stalls every thread of the process); spin_polls < 0 restores pure spinning.  Process-wide. */
int uuo_set_wait_policy(int spin_polls, int sleep_ns);

/* device -> device copy of `bytes` bytes ordered on `stream` (closures written in Python move the evaluated point and
 * the gradient between their own tensors and the driver's vectors with it) */
int uuo_copy_device(void* stream, void* d_dst, const void* d_src, size_t bytes);

/* ---- lock-step batches of independent solves ---------------------------------------------------------
 * The reference solves the candidate body parts of find_best_part_fits one after the other
 * (markers/markers_utils.py:416-610: one torch.optim.LBFGS(...).step(closure) per sub-tree, 202 of them for a 10-marker
 * limb) and likewise the yaw hypotheses (multimodal.py:462-574).  They are independent problems of one stage and one
 * (F, M): a batch steps up to B of them together -- one round = one closure evaluation of every live problem, every
 * kernel of the round launched ONCE for all of them -- with the decisions and the arithmetic of uuo_lbfgs_solve
 * (results are bit-identical to solving each problem alone).  Part-stage problems of one batch share the body pose
 * (d_o_pose) and hence one pose-blend cache.  uuo_batch_solve synchronises `stream`; stats[i].device_ms is 0. */
typedef struct uuo_batch uuo_batch_t;
int uuo_batch_create(uuo_model_t* model, int stage, int F, int M, int B, uuo_batch_t** out);
int uuo_batch_destroy(uuo_batch_t* batch);
int uuo_batch_solve(uuo_batch_t* batch, void* stream, const uuo_problem_t* problems /* [nb] */,
                    float* const* d_xs /* [nb] device vectors, updated in place */, int nb,
                    const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats /* [nb] */);
/* Ranking scores of part-stage candidates at d_xs (normally the vectors uuo_batch_solve has just solved; same problems, same
 * batch): h_scores[i] = pytorch3d chamfer_distance(markers_subset, vertices[:, subset_i]) in BOTH directions (mean / mean,
 * squared L2) -- what find_best_part_fits ranks its candidates by (markers/markers_utils.py:566-579, one SMPL forward and two
 * K=1 searches per candidate there).  One batched forward + one score kernel for all candidates; synchronises `stream`. */
int uuo_batch_part_scores(uuo_batch_t* batch, void* stream, const uuo_problem_t* problems, float* const* d_xs, int nb,
                          float* h_scores /* [nb] host */);

/* device -> host copy of n floats, ordered on `stream`, complete on return (for uuo_eval_callback_t users that only
 * hold the raw pointer, e.g. the iter_fn adapter). */
int uuo_copy_to_host(void* stream, const float* d_src, float* h_dst, int n);

/* ---- benchmarking helpers (used by bench.py for the roofline object) -------------------------------
 * average device milliseconds per closure evaluation over `iters` back-to-back evaluations, measured with
 * HIP events on `stream`; dominant_kernel_only 1: the fp32 skinning kernel alone (k_skin2), 2: the skinning kernel of the chamfer
 * closure's search alone (k_skin3, fp16 matrix pipe on split operands), one event pair per launch. */
int uuo_time_closure(uuo_fit_t* fit, void* stream, const uuo_problem_t* p, const float* d_x, int iters,
                     int dominant_kernel_only, float* ms_per_eval);

#ifdef __cplusplus
}
#endif
#endif /* UUO_HIP_H */
