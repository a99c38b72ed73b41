// Shared declarations for libuuo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/uuo_hip.h"

#define UUO_KP 224        // padded K of the augmented blend GEMM: 207 pose features | 10 betas | 7 zero (56 MFMA K-steps)
#define UUO_KB 208        // padded K of the per-vertex transposed posedirs rows
#define UUO_FT 16         // frames per MFMA row tile (v_mfma_f32_16x16x4_f32)
#define UUO_MAX_DEPTH 10  // SMPL tree depth is 9
#define UUO_LEVEL_W 5     // joints per tree level the level-parallel sweeps support (SMPL: 1,3,3,3,5,3,2,2,2)
#define UUO_FP 24         // floats per frame of the closure's partial-sum block
#define UUO_SK16_ASCALE 128.0f  // k_skin3: the pose features / betas are split as halves of 128 x (|R - I| <= 2, |beta| < 511 stay in range)
#define UUO_PRE 96        // floats per frame of the soft part closure's record (k_part_soft -> k_bwd_part): 0 weighted data-loss
                          // sum, 1..3 d trans, 4..13 d beta (blend-shape path), 14 torque_z about trans_f, 16..87 joint forces [24][3]

void uuo_set_error(const std::string& msg);

// Ablation / comparison knobs (kernel variants, phase cut-offs) and the uuo_debug_* hooks exist only in the debug
// flavour of the library (libuuo_hip_debug.so, built with -DUUO_DEBUG_HOOKS and loaded by tests/ and tools/ only).
// The shipped libuuo_hip.so reads no environment variable and exports nothing outside include/uuo_hip.h.
#ifdef UUO_DEBUG_HOOKS
#include <cstdlib>
#define UUO_ENV_INT(name, dflt) (std::getenv(name) ? std::atoi(std::getenv(name)) : (dflt))
#else
#define UUO_ENV_INT(name, dflt) (dflt)
#endif

#define UUO_HIP_CHECK(expr)                                                                      \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      uuo_set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" +  \
                    std::to_string(__LINE__));                                                   \
      return -5;                                                                                 \
    }                                                                                            \
  } while (0)

#define UUO_REQUIRE(cond, msg)        \
  do {                                \
    if (!(cond)) {                    \
      uuo_set_error(msg);             \
      return -22;                     \
    }                                 \
  } while (0)

// the argument struct of every batched kernel (lock-step batches, below) starts with its own grid extent
// A pointer read from an argument struct that itself lives in device memory (the lock-step kernels' `batch[...]`) is a
// generic pointer to the compiler: its loads become flat_load, which count on the LDS counter as well and so serialise
// with the kernel's LDS traffic.  Everything this library hands to a kernel is global memory (device or pinned host), so
// the pointer members of the argument structs carry that in their type when compiled for the device: uuo_gptr<T> is a
// T* in size, layout and use (it converts from and to T*), and the loads through it - also through pointers derived
// from it in inlined callees - are global_load, as in the kernels that take their pointers as direct arguments.
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define UUO_GLOBAL __attribute__((address_space(1)))
#else
#define UUO_GLOBAL
#endif
template <typename T>
struct uuo_gptr {
  UUO_GLOBAL T* p;
  uuo_gptr() = default;
  __host__ __device__ uuo_gptr(T* q) : p((UUO_GLOBAL T*)q) {}
  __host__ __device__ operator T*() const { return (T*)p; }
  __host__ __device__ T* get() const { return (T*)p; }
  __host__ __device__ T* operator->() const { return (T*)p; }
};
static_assert(sizeof(uuo_gptr<const float>) == sizeof(void*), "uuo_gptr must be a plain pointer in memory");
struct UuoGridHdr {
  int gx, gy;
};

// Small per-model constant tables read by every frame kernel.
struct UuoTree {
  int parent[UUO_NUM_JOINTS];
  int depth[UUO_NUM_JOINTS];
  int max_depth;
  int nchild[UUO_NUM_JOINTS];
  int child[UUO_NUM_JOINTS][4];  // children in ascending joint order (deterministic backward sweep)
  // joints by depth: the kinematic sweeps walk one depth per step with one lane per (joint of the level, matrix entry);
  // SMPL has at most 5 joints per level (5 x 12 entries = 60 lanes of one wave)
  int level_n[UUO_MAX_DEPTH];
  int level_j[UUO_MAX_DEPTH][UUO_LEVEL_W];
  int level_p[UUO_MAX_DEPTH][UUO_LEVEL_W];  // parent of level_j[d][k]
  float Jt[UUO_NUM_JOINTS][3];      // J_regressor . v_template
  float JS[UUO_NUM_JOINTS][3][10];  // J_regressor . shapedirs
  int extra_vids[UUO_NUM_EXTRA_JOINTS];
};

struct uuo_model {
  int V = 0;    // vertices
  int VP = 0;   // padded to a multiple of 128
  int nnz = 0;  // max non-zero skin weights per vertex (<= 4: uuo_model_create refuses anything else)
  // device tables
  float* P3 = nullptr;    // [3][VP/16][14][64][4]  blend basis (posedirs rows then shapedirs rows) in MFMA-operand order:
                          //   per (coord, 16-vertex unit, group of 4 K-steps) one 1-KB block = lane l's 4 B values
  float* vt3 = nullptr;   // [3][VP]           template, coordinate-planar
  void* P16 = nullptr;    // [3][VP/16][7][2][64][8] halfs: the same basis split into two fp16 planes (hi, lo) of basis * skin16_bscale,
                          //   in v_mfma_f32_16x16x32_f16 operand order (k_skin3: the chamfer closure's search); slot t of lane l in
                          //   K-step s is P3's (group 2 s + (t >> 2), component t & 3) of the same lane
  float skin16_inv = 0.f; // 1 / (UUO_SK16_ASCALE * skin16_bscale): what k_skin3 multiplies its accumulators with
  float* PT = nullptr;    // [V][3][UUO_KB]    per-vertex posedirs rows (backward / gather-LBS)
  float* ST = nullptr;    // [V][3][10]        shapedirs
  float* vt = nullptr;    // [V][3]
  int* Wi = nullptr;      // [VP][4]           sparse joint ids (ascending), -1 padded -> 0 weight
  float* Ww = nullptr;    // [VP][4]
  UuoTree* tree = nullptr;  // device copy
  UuoTree h_tree;
  // dense backward (dense_bwd.hip): the transposed blend contraction on the matrix pipe
  float* PB = nullptr;    // [VP/16][14][3][64][4]  augmented basis as the B operand of d pose-feature = d v_posed . P^T:
                          //   lane l's 4 values of (unit u, feature tile jt, K group g) = Baug[16 jt + (l&15)][coordinate 16 g + 4 t + (l>>4) of the unit]
  int* JLoff = nullptr;   // [25] joint j's (vertex, weight) pairs are entries JLoff[j] .. JLoff[j+1] of JLv / JLw (vertices ascending)
  int* JLv = nullptr;
  float* JLw = nullptr;
  // scratch of the dense backward, one set per stream (as `fwd`)
  struct BwdScratch {
    std::mutex mu;
    int capF = 0;
    float* pfaT = nullptr;
    float* A = nullptr;
    float* frames = nullptr;
    struct UuoDenseWs* ws = nullptr;
    float* gcopy = nullptr;  // [F][V][3] upstream gradient + the vertex-picked joints' (only when up_joints is given)
  };
  std::map<hipStream_t, BwdScratch> bwd;
  // scratch of uuo_smpl_forward, one set per stream; `mu` serialises callers that hold the same stream handle
  struct FwdScratch {
    std::mutex mu;
    int cap = 0;
    float* pfaT = nullptr;
    float* A = nullptr;
    float* jp = nullptr;
  };
  std::map<hipStream_t, FwdScratch> fwd;
  std::mutex fwd_mutex;
};

// How the frame kernels obtain the 24 rotations and the shape for frame f.
enum { UUO_ROOT_RAW = 0, UUO_ROOT_GS = 1, UUO_ROOT_Z_GS = 2, UUO_ROOT_ZSHARED = 3 };

struct UuoPoseSrc {
  uuo_gptr<const float> body;   // [F,23,9]
  int norm_body;       // Gram-Schmidt (rotation_6d round trip) on the body rotations
  uuo_gptr<const float> root;   // [F,9]
  int root_mode;       // UUO_ROOT_*
  uuo_gptr<const float> z;      // [F] (Z_GS) or [1] (ZSHARED)
  uuo_gptr<const float> betas;  // [10] or [F,10]
  int betas_stride;    // 0 or 10
  uuo_gptr<const float> trans;  // [F,3] or null
};

struct uuo_fit {
  uuo_model* model = nullptr;
  int F = 0, M = 0, nFT = 0;
  int n_max = 0;  // 219F+10
  // closure workspace
  float* pfaT = nullptr;            // [nFT][14][64][4]: A operand (pose features | betas) in MFMA-operand order
  void* pfa16 = nullptr;            // [nFT][7][2][64][8] halfs: the same operand * UUO_SK16_ASCALE as two fp16 planes (k_skin3)
  float* A = nullptr;               // [nFT*UUO_FT][24][12]
  float* verts = nullptr;           // [F][V][3]
  float* bbox = nullptr;            // [F][ceil(V/16)][6] per-unit bounding boxes (lo xyz, hi xyz), written by k_skin
  void* slab = nullptr;             // the one device allocation every buffer below (but pose_cache) is carved from
  float* part_sb = nullptr;         // [V][8] per-vertex constants of a part-stage candidate (k_pose_prep -> k_part_fwd)
  float* soft_pre = nullptr;        // [F][UUO_PRE] per-frame sums of the soft part closure (k_part_soft -> k_bwd_part)
  struct UuoDenseWs* dense = nullptr;  // soft chamfer closure (extension): workspace of the dense backward, its vertex gradient
  float* soft_gV = nullptr;            // [F][V][3] and [4][F][M] floats of soft-min scratch; allocated on first use
  float* soft_sm = nullptr;
  float* dbg_verts = nullptr;          // debug flavour, UUO_SKIN_F16_CHECK: the fp32 kernel's vertices and boxes beside k_skin3's (first use)
  float* bary_items = nullptr;         // marker stage on a three-corner placement: [F][3 M][3] corner items + [F] loss sums (first use)
  int* nn_flags = nullptr;          // [F][8] survivor counts of the pruned nearest-neighbour search (debug / tests)
  unsigned long long* nn = nullptr; // [F][M] packed (dist bits << 32 | idx)
  float* frame_part = nullptr;      // [F][UUO_FP]: loss, dz, pose sq, dbeta[10], gradient statistics
  float* frames = nullptr;          // [F][sizeof(FrameLds)/4]: per-frame rotations / joints / world transforms left by
                                    // k_pose_prep for the backward kernel of the same closure
  float* pose_cache = nullptr;      // [F][V][3] template + pose-corrective offsets of a constant body pose (part stage),
                                    // allocated on first use
  unsigned long long pose_cache_id = 0;  // the problem id the cache was built for (0 = none)
  float* zeros16 = nullptr;         // 16 zero floats (betas of the cache build)
  float* mask = nullptr;            // [F][M] 0/1
  float* scalars = nullptr;         // device scalars block (see solver)
  float* vecs = nullptr;            // one work vector of n_max floats (timing helper gradient)
  void* lbws = nullptr;             // L-BFGS workspace (lbfgs_driver.hip), created on first solve
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float mask_sum = 0.f;  // host copy of sum(mask) (chamfer normaliser), refreshed by uuo_ensure_mask
  bool shared_pose_cache = false;  // pose_cache belongs to a uuo_batch (not freed with the fit)
};

// ---- dense backward of the skinning (dense_bwd.hip) ---------------------------------------------------------------------
// Given dL/dvertices for EVERY vertex (a soft assignment, or a caller differentiating SmplInference.forward), the backward
// is no longer a gather of <= M items per frame: d pose-feature = d v_posed . P^T is the transposed 207 x 20 670 contraction
// and runs on the matrix pipe like the forward.  Workspace for F frames:
struct UuoDenseWs {
  int F = 0, nFT = 0;
  float* A_id = nullptr;   // [nFT*16][24][12] identity skinning matrices (v_posed = the forward blend with them)
  float* vp = nullptr;     // [F][V][3] v_posed = template + pose-corrective + shape blend
  float* dvpT = nullptr;   // [nFT][VP/16][3][64][4] d v_posed in MFMA A-operand order (frames past F: zeros)
  float* part = nullptr;   // [UUO_DPF_NCB][F][UUO_KP] partial d[pose-feature | beta] per vertex chunk
  float* pre = nullptr;    // [F][UUO_PREG]: 0 data-loss sum, 1..3 d trans, 16..303 d A [24][12]
};
#define UUO_DPF_NCB 27     // vertex chunks of the transposed contraction (one block per (frame tile, chunk))
#define UUO_PREG 304
int uuo_dense_ws_create(const uuo_model* m, hipStream_t s, int F, UuoDenseWs** out);
void uuo_dense_ws_destroy(UuoDenseWs* ws);
// pfaT / A: this evaluation's operand tiles and skinning matrices (k_pose_prep); gV [F][V][3].  Fills ws->pre (but entry 0)
// and ws->part; the caller then runs k_bwd_sparse with BwdArgs.pre = ws->pre, dpf_part = ws->part.
// have_vp: ws->vp already holds v_posed of this evaluation (written by the forward's k_skin2 with vp_out)
int uuo_dense_backward(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A, const float* gV, UuoDenseWs* ws,
                       bool have_vp = false);
// the chamfer stage's data term with a soft assignment (nn_kernels.hip); sm: 4 F M floats of scratch
int uuo_launch_soft_chamfer(hipStream_t s, int F, int M, int V, const float* markers, const float* verts, const float* mask,
                            float mask_sum, const unsigned long long* keys, float w_hard, float w_soft, float tau, float* sm,
                            float* gV, float* pre, int pre_stride, const float* bbox);

// ---- kernel launchers (defined in the .hip files) --------------------------------------------------
int uuo_launch_pose_prep(const uuo_model* m, hipStream_t s, int F, const UuoPoseSrc& src, float* pfaT, float* A,
                         float* joints_posed, float* frames = nullptr, const int32_t* sb_subset = nullptr, int sb_ns = 0,
                         float* sb_out = nullptr, void* pfa16 = nullptr);
int uuo_launch_part_fwd(const uuo_model* m, hipStream_t s, int F, int P1, const float* cache, const float* sb, const float* A,
                        const float* trans, const int32_t* subset, int n_subset, const float* markers,
                        unsigned long long* packed);
int uuo_launch_part_soft(const uuo_model* m, hipStream_t s, int F, int P1, const float* cache, const float* sb, const float* A,
                         const float* trans, const int32_t* subset, int n_subset, const float* markers,
                         unsigned long long* packed, float* pre, float w_hard, float w_soft, float tau);
// vp_out (optional, with bbox): v_posed [F][V][3] as well (the dense backward's input: saves its own skinning launch)
int uuo_launch_skin(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A,
                    const float* trans, float* verts, float* bbox, float* vp_out = nullptr);
// the skinning of a closure's SEARCH on the fp16 matrix pipe (k_skin3): both operands split into two fp16 planes, three products,
// fp32 accumulation; returns -22 when the launch geometry does not fit (the caller falls back to uuo_launch_skin)
int uuo_launch_skin16(const uuo_model* m, hipStream_t s, int F, const void* pfa16, const float* A, const float* trans, float* verts,
                      float* bbox);
int uuo_launch_skin_cached(const uuo_model* m, hipStream_t s, int F, const float* cache, const float* A,
                           const float* betas, const float* trans, const int32_t* subset, int n_subset, float* verts,
                           float* bbox_compact = nullptr);
int uuo_launch_identity_transforms(hipStream_t s, int count, float* A);
int uuo_launch_nn_cull(hipStream_t s, int F, int M, int V, int nunits, const float* markers, const float* verts,
                       const float* bbox, unsigned long long* packed, int* flags);
int uuo_launch_joints45(const uuo_model* m, hipStream_t s, int F, const float* joints_posed, const float* verts,
                        float* joints45);
int uuo_launch_nn(hipStream_t s, int N, int P1, int P2, const float* x, const float* y, const int32_t* ysub,
                  int P2s, unsigned long long* packed);
int uuo_launch_nn_unpack(hipStream_t s, int count, const unsigned long long* packed, float* dist, int32_t* idx);
int uuo_launch_assign(hipStream_t s, int F, int M, int V, const float* verts, const float* markers,
                      const uint8_t* valid, int32_t* idx, unsigned long long* packed);
int uuo_launch_mask(hipStream_t s, int F, int M, const float* markers, float* mask, float* mask_sum_dev);
int uuo_launch_bwd(const uuo_fit* fit, hipStream_t s, const uuo_problem_t& p, const UuoPoseSrc& src,
                   const float* x, float* grad, float inv_count);
int uuo_launch_finalize(const uuo_fit* fit, hipStream_t s, const uuo_problem_t& p, const float* x, float* grad,
                        float* loss);

// closure internals shared with the solver (closure.hip)
int uuo_validate_problem(const uuo_fit* fit, const uuo_problem_t* p);
int uuo_ensure_mask(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p);
int uuo_prepare_pose_cache(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x);
// forward half of a stage closure at d_x (pose preparation, skinning, nearest-neighbour search): recorded when a batch records
int uuo_closure_forward_at(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x);
// two-directional chamfer sums of part-stage candidates after uuo_closure_forward_at (closure.hip)
struct PartScoreArgs {
  UuoGridHdr h;
  int F, M, V, ns;
  uuo_gptr<const float> markers;             // [F][M][3]
  uuo_gptr<const float> verts;               // [F][V][3], valid at the subset's vertices
  uuo_gptr<const int32_t> subset;            // [ns]
  uuo_gptr<const unsigned long long> nn;     // [F][M] packed (squared distance bits << 32 | candidate)
  uuo_gptr<double> out;                      // [F][2]: sum_m d2(marker -> nearest subset vertex), sum_c d2(subset vertex -> nearest marker)
};
int uuo_launch_part_scores(hipStream_t s, const void* d_args, int count, int F);
// Optional zero-copy report of a closure evaluation: the finalize kernel copies the 80-byte block that starts 8 bytes
// before d_stats (the solver's {max|d| bits, pad, out[9]}) into `host` (pinned, device-visible) and then publishes
// `seq` in host[10]; the solver polls that word instead of enqueueing a copy and synchronising the stream.
struct UuoEvalReport {
  unsigned long long* host = nullptr;  // [24]: 0..9 the block, 10 the sequence word, 11..13 max|g| / sum|g| / g.g over the problem's
                                       // own parameters (all but the betas), 14..23 its shape gradient (as doubles)
  unsigned long long seq = 0;
};
// `compact`: gradient and direction in the solver's compact packing (closure.hip stage_layout); the parameters d_x always in
// the reference's
int uuo_closure_eval_impl(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x, float* d_loss,
                          float* d_grad, int32_t* d_nn_idx, const float* d_dir, double* d_stats,
                          const UuoEvalReport* report = nullptr, bool compact = false);
// compact solver index -> index in the reference's parameter packing: up to four runs, each either dense (full = fo + i) or
// "6 of 9" (full = fo + 9 (i / 6) + i % 6: the first two rows of row-major 3x3 rotations).  nseg == 0: the identity.
struct UuoIndexMap {
  int nseg;
  int cb[5];   // first compact index of each run; cb[nseg] = n_act
  int fo[4];   // offset of the run in the full packing
  int k69[4];  // run kind
  int n_act, n_full;
  __host__ __device__ int full(int c) const {
    if (nseg == 0) return c;
    int sg = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) sg += (q < nseg && c >= cb[q]) ? 1 : 0;
    const int rel = c - cb[sg];
    if (!k69[sg]) return fo[sg] + rel;
    const int rot = (int)(((unsigned long long)(unsigned)rel * 0xAAAAAAABull) >> 34);  // rel / 6, exact for every 32-bit rel
    return fo[sg] + 9 * rot + (rel - 6 * rot);
  }
};
int uuo_stage_compactable(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x, bool* compact);
UuoIndexMap uuo_stage_index_map(const uuo_problem_t* p, bool compact);

// ---- lock-step batches (uuo_batch_*: batch.hip) ------------------------------------------------------------------------
// Every kernel of the solve path exists in two launch forms over ONE device body: k_X(XArgs) for a single problem and
// k_X_b(const XArgs* batch) where blockIdx.z picks the problem (blocks outside a problem's own grid extent exit at once).
// While a batch is being stepped, launches are not issued but RECORDED per problem (uuo_recorder != nullptr); the
// scheduler then merges the records of all live problems by kind, in the canonical order below -- which is the order the
// kinds occur in inside any one problem's round, so per-problem ordering is preserved by the in-order stream -- and
// issues one launch per kind with the argument structs staged in device memory.
enum {
  UUO_OP_AXPY_ACCEPT = 0,  // x <- x + t d of the accepted step (end of an iteration)
  UUO_OP_COPY,             // gradient hand-over copy (rare)
  UUO_OP_DOTS,
  UUO_OP_SMALL,
  UUO_OP_DIR,
  UUO_OP_NEG,
  UUO_OP_AXPY,             // line-search trial point
  UUO_OP_POSE_PREP,
  UUO_OP_SKIN,             // k_skin / k_skin2: whole-GPU kernels, issued one problem after the other
  UUO_OP_SKIN_CACHED,
  UUO_OP_PART_FWD,         // k_part_fwd: skinning of a candidate's vertices fused with the nearest-vertex search
  UUO_OP_PART_SOFT,        // k_part_soft: the same with a soft assignment (extension) + the dense backward's per-frame sums
  UUO_OP_FILL,             // hipMemsetAsync
  UUO_OP_NN,
  UUO_OP_NN_FEWQ,
  UUO_OP_NN_CULL,
  UUO_OP_BWD,
  UUO_OP_BWD_PART,         // k_bwd_part: the part stage on its cached pose blend
  UUO_OP_FIN,
  UUO_OP_COUNT
};
#define UUO_OP_ARG_MAX 640
struct UuoOpRec {
  int op;
  int gx, gy;
  unsigned nbytes;
  alignas(16) unsigned char args[UUO_OP_ARG_MAX];
};
struct UuoRecorder {
  std::vector<UuoOpRec> ops;
};
extern thread_local UuoRecorder* uuo_recorder;
template <class A>
inline bool uuo_record(int op, int gx, int gy, const A& a) {
  if (!uuo_recorder) return false;
  static_assert(sizeof(A) <= UUO_OP_ARG_MAX, "argument struct too large for an op record");
  UuoOpRec r;
  r.op = op;
  r.gx = gx;
  r.gy = gy;
  r.nbytes = (unsigned)sizeof(A);
  std::memcpy(r.args, &a, sizeof(A));
  uuo_recorder->ops.push_back(r);
  return true;
}
// Host-only bookkeeping of a batch's pinned argument blob (two regions: one per stepping group / stream).  A flush stages
// its structs from the START of its region and copies them to the device asynchronously, so before the next flush may
// overwrite the region that copy must have executed: inside a solve this is implied (the round's reports follow the copy
// on the stream: report_arrived), otherwise the caller has to synchronise the region's stream first (begin_flush says
// so).  Structs that must coexist with the last flush's (the score kernel's) are appended BEHIND it and never need a
// wait.  The GPU memory fault of round 2 (two processes on one device) was this state machine done wrong; it has no
// device code and is unit-tested on the CPU through uuo_debug_staging_script (tests/test_abi.py).
struct UuoStaging {
  size_t region_cap = 0;
  size_t used[2] = {0, 0};           // bytes of each region holding structs whose copy may still be pending
  bool pending[2] = {false, false};  // the region's last host-to-device copy may not have executed yet
  // a new flush of `bytes` (> 0) at the start of region r; returns true iff the region's stream must be synchronised first
  bool begin_flush(int r, size_t bytes) {
    if (bytes == 0) return false;
    const bool need_sync = pending[r];
    pending[r] = true;
    used[r] = bytes;
    return need_sync;
  }
  // `bytes` behind everything region r holds; false if they do not fit.  *off is relative to the region's start.
  bool append(int r, size_t bytes, size_t* off) {
    const size_t at = (used[r] + 255) / 256 * 256;
    if (at + bytes > region_cap) return false;
    *off = at;
    used[r] = at + bytes;
    pending[r] = true;
    return true;
  }
  void report_arrived(int r) { pending[r] = false; }  // a kernel enqueued after the region's copy has reported
  // every stream that used the blob has been synchronised (a sequence of uuo_batch_part_scores calls with no solve in
  // between would otherwise keep appending behind `used` until the region overflowed)
  void synchronized() {
    pending[0] = pending[1] = false;
    used[0] = used[1] = 0;
  }
};

#define UUO_BATCH_PICK(ArgsT, batch)                                      \
  const ArgsT a = (batch)[blockIdx.z];                                    \
  if ((int)blockIdx.x >= a.h.gx || (int)blockIdx.y >= a.h.gy) return;
// batched launch of `count` records of one kind (d_args: count structs of that kind, contiguous in device memory);
// each translation unit serves the kinds whose kernels it defines and returns 1 for the others
int uuo_batched_launch_smpl(int op, hipStream_t s, const void* d_args, int count, int gx, int gy);
int uuo_replay_skin_call(hipStream_t s, const void* h_args);
int uuo_batched_launch_nn(int op, hipStream_t s, const void* d_args, int count, int gx, int gy);
int uuo_batched_launch_closure(int op, hipStream_t s, const void* d_args, int count, int gx, int gy);

// reprojection.hip: the handle of a 2D-prior fit and one evaluation of its fused closure
struct uuo_reprojection {
  uuo_reprojection_problem_t p;
  double* part = nullptr;              // [F][8] per-frame partial sums of an evaluation
  unsigned long long* keys = nullptr;  // [F][4][M] the vertex slices' nearest-vertex keys (same allocation as `part`)
};
// one evaluation of the fused 2D-prior closure (uuo_reprojection_eval without the argument checks)
int uuo_reprojection_eval_impl(uuo_reprojection* h, hipStream_t s, const float* d_x, float* d_loss, float* d_grad,
                               float* d_kp, int32_t* d_nn_idx);
