// Internal declarations of the L-BFGS solver (device state, kernel argument structs, launchers, host driver types).
// Translation units: lbfgs_kernels.hip (every k_lb_* kernel and its launcher), lbfgs_driver.hip (workspace, the host driver
// lbfgs_run that mirrors torch.optim.LBFGS.step, the uuo_lbfgs_* / uuo_fit_* entry points), batch.hip (lock-step batches:
// coroutine scheduler, uuo_batch_*), solver_debug.hip (debug flavour only: cross-check kernels, self-tests, timing hooks).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "uuo_common.h"

#define LB_MAXH 104                 // history capacity (slots); history_size <= LB_MAXH - 4
#define LB_ROWS (2 * LB_MAXH + 1)   // S slots, Y slots, g
#define LB_MAXCHUNK 32              // element chunks of the history dot kernel (partials reduced unrolled)
#define LB_NVEC 10

struct LbDev {                       // device-resident optimiser state
  double SY[LB_MAXH * LB_MAXH];      // s_i . y_j by slot
  double YY[LB_MAXH * LB_MAXH];      // y_i . y_j by slot
  double cs[LB_MAXH], cy[LB_MAXH];   // direction coefficients by slot
  double cg;
  double Hdiag;
  double gg;
  int head, count;
  unsigned dmax_bits;
  int pad;
  double out[16];                    // read-back block (see LbOut)
  double W[LB_MAXH * LB_MAXH];       // inverse of U = upper triangle of S.Y^T over the window, by slot (k_lb_small_inv)
};

struct LbOut {  // layout of LbDev::out
  double loss, gtd_new, gmax, g1, gg, gtd_dir, accepted, dmax, ys;
};


// History layout.  S and Y are stored in column blocks of LB_CW floats: element i of slot j lives at
//   (i / LB_CW) * (capL * LB_CW + LB_CBPAD) + j * LB_CW + i % LB_CW            (capL = allocated slots)
// (see lbfgs_kernels.hip for why)
#define LB_CW 512
#define LB_CBPAD 64  // floats of padding after each column block's slots: region stride = odd multiple of 256 B, so
                     // that blocks working on different column blocks at the same slot do not camp on the same
                     // memory channels (101 x 2 KB alone is 808 x 256 B)
#define LB_CBSTRIDE(capL) ((size_t)(capL) * LB_CW + LB_CBPAD)
__device__ __host__ __forceinline__ size_t lb_hist_off(int slot, int i, int capL) {
  return (size_t)(i / LB_CW) * LB_CBSTRIDE(capL) + (size_t)slot * LB_CW + (i % LB_CW);
}
#define LB_DRS 16                        // row splits of k_lb_dots (grid y)
#define LB_DQ 4  // slot ranges per 256-column strip of k_lb_direction (one wave each)

// ---- kernel argument structs (single launch form k_X(XArgs) and lock-step form k_X_b(const XArgs*))
struct LbNegArgs {
  UuoGridHdr h;
  int n;
  uuo_gptr<const float> g;
  uuo_gptr<float> d;
  uuo_gptr<const float> x;
  float t;
  uuo_gptr<float> xt;
  UuoIndexMap map;
};
struct LbAxpyArgs {
  UuoGridHdr h;
  int n;
  uuo_gptr<const float> x;
  float t;
  uuo_gptr<const float> d;
  uuo_gptr<float> o;
  UuoIndexMap map;
};
struct LbDotsArgs {
  UuoGridHdr h;
  int n, cap, capL, head, count, cand;
  uuo_gptr<float> S;
  uuo_gptr<float> Y;
  uuo_gptr<const float> g;
  uuo_gptr<const float> gp;
  uuo_gptr<const float> d;
  float t;
  int ncb, gcb;
  uuo_gptr<double> part;
  int skip_lo, skip_hi;  // elements left out of the dot products (shared betas on ranks > 0); empty otherwise
};
struct LbSmallArgs {
  UuoGridHdr h;
  int nchunks, cap, hist, cand;
  uuo_gptr<const double> part;
  uuo_gptr<LbDev> st;
  int stop;
  uuo_gptr<const double> rd_in;  // shared-betas solves: the Gram rows already summed over chunks AND ranks; null otherwise
};
struct LbDirArgs {
  UuoGridHdr h;
  int n, cap, capL;
  uuo_gptr<const float> S;
  uuo_gptr<const float> Y;
  uuo_gptr<const float> g;
  uuo_gptr<LbDev> st;
  uuo_gptr<float> d;
  uuo_gptr<const float> x;
  float t;
  uuo_gptr<float> xt;
  UuoIndexMap map;
};
struct LbCopyArgs {  // UUO_OP_COPY: device-to-device copy of n floats
  UuoGridHdr h;
  uuo_gptr<float> dst;
  uuo_gptr<const float> src;
  size_t bytes;
};

// ---- launchers (lbfgs_kernels.hip).  The single-problem forms; a lock-step batch launches the recorded structs of all its
// problems through uuo_batched_launch_lbfgs (returns 1 for kinds whose kernels live in another translation unit).
void uuo_lb_launch_init(hipStream_t s, LbDev* st);
void uuo_lb_launch_neg(hipStream_t s, dim3 grid, const LbNegArgs& a);
void uuo_lb_launch_axpy(hipStream_t s, dim3 grid, const LbAxpyArgs& a);
void uuo_lb_launch_dots(hipStream_t s, dim3 grid, const LbDotsArgs& a);
void uuo_lb_launch_small(hipStream_t s, const LbSmallArgs& a);
void uuo_lb_launch_direction(hipStream_t s, dim3 grid, const LbDirArgs& a);
void uuo_lb_launch_rows(hipStream_t s, int nchunks, int cap, int cand, const double* part, const LbDev* st, double* host_rows,
                        unsigned long long seq);
void uuo_lb_launch_stats(hipStream_t s, int nstat, int n, const float* g, const float* d, double* part, const float* loss,
                         LbDev* st);
int uuo_batched_launch_lbfgs(int op, hipStream_t s, const void* d_args, int count, int gx, int gy);

// ---- host driver (lbfgs_driver.hip)
// How a host thread waits for a report word in pinned memory (closure evaluations, Gram rows, lock-step rounds).  Default:
// spin (pause) -- lowest latency, one CPU per waiting thread.  uuo_set_wait_policy(spin_polls, sleep_ns) makes every wait
// sleep `sleep_ns` at a time once it has polled `spin_polls` times: for hosts whose CPU quota is smaller than the number
// of solves in flight (a throttled cgroup stalls ALL threads of the process for the rest of the scheduler period).
extern std::atomic<int> g_wait_spin_polls, g_wait_sleep_ns;
#include <sys/prctl.h>
struct UuoWaiter {
  unsigned long polls = 0, slow = 0;
  // one relaxation step of a polling loop; true when it is time for the loop's slow checks (stream query, wall clock)
  bool tick() {
    const int sp = g_wait_spin_polls.load(std::memory_order_relaxed);
    ++polls;
    if (sp >= 0 && polls > (unsigned long)sp) {
      static thread_local bool slack_set = false;
      if (!slack_set) {  // the default timer slack (50 us) would round every short sleep up
        (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
        slack_set = true;
      }
      timespec ts{0, (long)g_wait_sleep_ns.load(std::memory_order_relaxed)};
      nanosleep(&ts, nullptr);
      return (++slow & 0xFFF) == 0;
    }
    __builtin_ia32_pause();
    return (polls & 0xFFFFF) == 0;
  }
};

struct Objective {
  int n = 0;        // coordinates of the solver (gradient, direction, history)
  int n_full = 0;   // floats of the parameter vector the closure is evaluated at (0: same as n)
  UuoIndexMap map;  // solver coordinate -> parameter index (nseg = 0: the identity)
  Objective() { std::memset(&map, 0, sizeof(map)); }
  bool fused_stats = false;  // eval() also writes {loss, g.d, max|g|, sum|g|, g.g} to stats_dev
  virtual int eval(hipStream_t s, const float* x, float* loss_dev, float* grad, const float* dir, double* stats_dev,
                   const UuoEvalReport* report) = 0;
  virtual ~Objective() {}
};

struct StageObjective : Objective {
  uuo_fit* fit;
  const uuo_problem_t* p;
  bool compact = false;  // gradient / direction in the compact packing (closure.hip stage_layout); set by stage_objective_init
  int eval(hipStream_t s, const float* x, float* loss_dev, float* grad, const float* dir, double* stats_dev,
           const UuoEvalReport* report) override {
    return uuo_closure_eval_impl(fit, s, p, x, loss_dev, grad, nullptr, dir, stats_dev, report, compact);
  }
};
// decides the packing of one stage solve (one small read-back: uuo_stage_compactable) and sizes the objective accordingly
inline int stage_objective_init(StageObjective& obj, uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x) {
  obj.fit = fit;
  obj.p = p;
  obj.fused_stats = true;
  bool compact = false;
  const int rc = uuo_stage_compactable(fit, s, p, d_x, &compact);
  if (rc) return rc;
  obj.compact = compact;
  obj.map = uuo_stage_index_map(p, compact);
  obj.n_full = uuo_problem_num_params(p);
  obj.n = compact ? obj.map.n_act : obj.n_full;
  return 0;
}

// ---------------------------------------------------------------------------------------------------- workspace
struct LbWs {
  int n_cap = 0, cap = 0;  // vector length capacity, history slots
  float* S = nullptr;
  float* Y = nullptr;
  float* vecs = nullptr;  // LB_NVEC work vectors
  double* part = nullptr;
  LbDev* st = nullptr;
  float* loss_dev = nullptr;
  double* h_out = nullptr;  // pinned, device-visible: read-back block + sequence word
  void* slab = nullptr;     // the one device allocation the pointers above are carved from
  unsigned long long seq = 0;
  int nchunks = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // shared-betas solves only (created on first use): pinned staging of the Gram rows / the summed shape gradient
  double* h_rows = nullptr;     // [LB_ROWS*3 + 1]: this rank's rows, written by k_lb_rows, + its sequence word
  double* h_joint = nullptr;    // [2][LB_ROWS*3]: the rows summed over the ranks, on their way to rd_joint (two slots)
  float* h_gb = nullptr;        // [4][16]: the summed shape gradient on its way into the gradient vector (four slots)
  double* rd_joint = nullptr;   // device [LB_ROWS*3]
  unsigned long long row_seq = 0, joint_slot = 0, gb_slot = 0;
};

int lbws_destroy(LbWs* w);
int lbws_create(int n, int hist, LbWs** out, bool sync = true);
int fit_create_impl(uuo_model_t* model, int F, int M, uuo_fit_t** out, bool sync);

// lock-step batches: while a batch steps this problem its launches are recorded (uuo_recorder) and `evaluate` hands control
// back to the batch scheduler instead of polling; the scheduler resumes the solve when the evaluation has reported
extern thread_local void (*g_batch_yield)(void);

// Shared-betas solves (uuo_lbfgs_solve_shared, EXTENSION): this rank's problem is one block of a joint problem whose shape
// vector x[off .. off + cnt) is replicated on every rank.  The driver below is unchanged but for three exchanges, each ONE
// rank-ordered gather through the caller's hook, after which every rank holds the same numbers and decides the same:
//   * after every closure evaluation {loss, g.d, own-parameter gradient statistics, max|d|, local shape gradient}: the
//     shape gradient is summed and written back into the gradient vector before anything reads it, the line search sees the
//     joint loss / g.d / norms;
//   * per iteration the new Gram rows (k_lb_rows): summed in rank order and handed to k_lb_small_inv (ranks > 0 leave the
//     replicated range out of their dot products, so every entry of the joint vector is counted once);
//   * once, at the start, the shape vector itself (rank 0's values win: the replicas must be bit-identical).
struct SharedCtx {
  uuo_gather_fn gather = nullptr;
  void* user = nullptr;
  int rank = 0, world = 1;
  int off = 0, cnt = 0;     // the shared entries in the SOLVER's packing (gradient, direction, dot products)
  int off_x = 0;            // and in the parameter vector
  std::vector<double> all;  // gather target [world][n]
  std::vector<double> send, recv;  // the messages with their status words
};

int lbfgs_run(LbWs* w, hipStream_t s, Objective& obj, float* d_x, const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats,
              uuo_eval_callback_t cb, void* cb_user, SharedCtx* sh = nullptr);

#ifdef UUO_DEBUG_HOOKS
// solver_debug.hip: the two earlier formulations of the coefficient step, kept as cross-checks of k_lb_small_inv
// (kind 1 = k_lb_small_ref: 2 x k serial steps; kind 2 = k_lb_small: 16 x 16 block steps)
void uuo_debug_launch_small(int kind, hipStream_t s, int nchunks, int cap, int hist, int cand, const double* part, LbDev* st,
                            int stop);
#endif
