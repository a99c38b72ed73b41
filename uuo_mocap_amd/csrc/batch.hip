// ---------------------------------------------------------------------------------------------------- lock-step batches
// B independent L-BFGS problems of one stage and one (F, M) -- the candidate body parts of find_best_part_fits
// (reference markers/markers_utils.py:416-610 solves them one after the other) or the yaw hypotheses of
// multimodal_video_mocap (multimodal.py:462-574) -- stepped together: one ROUND = one closure evaluation of every live
// problem, every kernel of the round launched once for all of them (grid z = problem).  Each problem runs the unchanged
// lbfgs_run (same decisions, same arithmetic, same kernels' bodies: bit-identical to solving it alone) as a coroutine on
// its own stack; where the single-problem driver would poll for its evaluation's report it yields to the scheduler, which
// merges the launches the live problems recorded, stages their argument structs with one host-to-device copy, issues them
// in the canonical order of uuo_common.h and waits for every report.  A problem that converged simply stops taking part.
#include "lbfgs.h"

#include <sys/mman.h>
#include <ucontext.h>

// a coroutine's stack: 1 MB of private pages below a PROT_NONE guard page (lbfgs_run calls into the HIP runtime -- lazy code
// object loading on a first launch, error strings -- from it: an overflow must fault, not run into a neighbour's heap)
struct CoStack {
  static constexpr size_t kGuard = 4096, kBytes = 1024 * 1024;
  void* base = nullptr;
  bool alloc() {
    void* p = mmap(nullptr, kGuard + kBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (p == MAP_FAILED) return false;
    if (mprotect(p, kGuard, PROT_NONE) != 0) {  // stacks grow down: the guard sits at the low end
      munmap(p, kGuard + kBytes);
      return false;
    }
    base = p;
    return true;
  }
  void* sp() const { return (char*)base + kGuard; }
  CoStack() = default;
  CoStack(const CoStack&) = delete;
  CoStack& operator=(const CoStack&) = delete;
  ~CoStack() {
    if (base) munmap(base, kGuard + kBytes);
  }
};

struct BatchCo {
  ucontext_t ctx;
  CoStack stack;
  UuoRecorder rec;
  StageObjective obj;
  LbWs* w = nullptr;
  float* d_x = nullptr;
  const uuo_lbfgs_options_t* opt = nullptr;
  uuo_lbfgs_stats_t* stats = nullptr;
  hipStream_t s = nullptr;
  int rc = 0;
  bool done = false, started = false, waiting = false;
};

struct uuo_batch {
  uuo_model* model = nullptr;
  int stage = 0, F = 0, M = 0, B = 0;
  std::vector<uuo_fit*> fits;
  std::vector<LbWs*> ws;
  float* pose_cache = nullptr;  // part stage: the one pose-corrective blend all candidates share
  unsigned char* h_blob = nullptr;  // pinned staging of one round's argument structs
  unsigned char* d_blob = nullptr;
  size_t blob_cap = 0;
  int lb_n = 0, lb_hist = 0;
  UuoStaging staging;          // which parts of the blob may still be waiting for their host-to-device copy (uuo_common.h)
  double* d_scores = nullptr;  // uuo_batch_part_scores: [nb][F][2] per-frame sums
  double* h_scores = nullptr;
  size_t score_cap = 0;
  hipStream_t s2 = nullptr;  // the second stepping group's stream (forked from / joined to the caller's stream)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

static thread_local ucontext_t g_sched_ctx;
static thread_local BatchCo* g_cur_co = nullptr;

static void batch_yield_impl() {
  BatchCo* c = g_cur_co;
  c->waiting = true;
  swapcontext(&c->ctx, &g_sched_ctx);
}

static void batch_co_entry(unsigned lo, unsigned hi) {
  BatchCo* c = reinterpret_cast<BatchCo*>(((unsigned long long)hi << 32) | (unsigned long long)lo);
  c->rc = lbfgs_run(c->w, c->s, c->obj, c->d_x, c->opt, c->stats, nullptr, nullptr);
  c->done = true;
  c->waiting = false;
  swapcontext(&c->ctx, &g_sched_ctx);
}

extern "C" int uuo_batch_destroy(uuo_batch_t* b) {
  if (!b) return 0;
  for (uuo_fit* f : b->fits) uuo_fit_destroy(f);  // also frees the fit's own L-BFGS workspace (fit->lbws)
  if (b->pose_cache) (void)hipFree(b->pose_cache);
  if (b->h_blob) (void)hipHostFree(b->h_blob);
  if (b->d_blob) (void)hipFree(b->d_blob);
  if (b->d_scores) (void)hipFree(b->d_scores);
  if (b->h_scores) (void)hipHostFree(b->h_scores);
  if (b->s2) (void)hipStreamDestroy(b->s2);
  if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
  if (b->ev_join) (void)hipEventDestroy(b->ev_join);
  delete b;
  return 0;
}

extern "C" int uuo_batch_create(uuo_model_t* model, int stage, int F, int M, int B, uuo_batch_t** out) {
  UUO_REQUIRE(model && out, "uuo_batch_create: null argument");
  UUO_REQUIRE(stage >= 0 && stage <= 2 && F > 0 && M > 0 && B > 0 && B <= 4096, "uuo_batch_create: bad stage / sizes");
  uuo_batch* b = new uuo_batch();
  b->model = model;
  b->stage = stage;
  b->F = F;
  b->M = M;
  b->B = B;
  int rc = 0;
  for (int i = 0; i < B && rc == 0; ++i) {
    uuo_fit* f = nullptr;
    rc = fit_create_impl(model, F, M, &f, false);
    if (rc == 0) b->fits.push_back(f);
  }
  if (rc == 0 && stage == UUO_STAGE_PART) {
    if (hipMalloc((void**)&b->pose_cache, (size_t)F * model->V * 3 * sizeof(float)) != hipSuccess) {
      uuo_set_error("uuo_batch_create: pose cache allocation failed");
      rc = -12;
    } else {
      for (uuo_fit* f : b->fits) {
        f->pose_cache = b->pose_cache;
        f->shared_pose_cache = true;
      }
    }
  }
  b->blob_cap = 2 * ((size_t)B * 12 * UUO_OP_ARG_MAX + 4096);  // two halves: one per stepping group
  b->staging.region_cap = b->blob_cap / 2;
  if (rc == 0 && hipHostMalloc((void**)&b->h_blob, b->blob_cap, hipHostMallocDefault) != hipSuccess) rc = -12;
  if (rc == 0 && hipMalloc((void**)&b->d_blob, b->blob_cap) != hipSuccess) rc = -12;
  if (rc == 0 && hipStreamCreateWithFlags(&b->s2, hipStreamNonBlocking) != hipSuccess) rc = -5;
  if (rc == 0 && hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming) != hipSuccess) rc = -5;
  if (rc == 0 && hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming) != hipSuccess) rc = -5;
  if (rc == 0 && hipDeviceSynchronize() != hipSuccess) rc = -5;  // the workspaces' zero fills ran on the null stream
  if (rc != 0) {
    if (rc == -12) uuo_set_error("uuo_batch_create: allocation failed");
    uuo_batch_destroy(b);
    return rc;
  }
  *out = b;
  return 0;
}

// one round's recorded launches of all problems, merged by kind
// `i0 .. i1` = the problems of one stepping group, `region` = which half of the staging blob the group owns
static int batch_flush(uuo_batch* b, hipStream_t s, std::vector<BatchCo>& cos, int i0, int i1, int region) {
  // validate the per-problem order and count bytes
  size_t off[UUO_OP_COUNT], cnt[UUO_OP_COUNT], width[UUO_OP_COUNT];
  int gx[UUO_OP_COUNT], gy[UUO_OP_COUNT];
  for (int k = 0; k < UUO_OP_COUNT; ++k) off[k] = cnt[k] = width[k] = 0, gx[k] = gy[k] = 0;
  bool any = false;
  for (int i = i0; i < i1; ++i) {
    int last = -1;
    for (const UuoOpRec& r : cos[i].rec.ops) {
      UUO_REQUIRE(r.op >= last && (r.op > last || r.op == UUO_OP_COPY), "batch: a problem recorded its launches out of the canonical order");
      last = r.op;
      cnt[r.op] += 1;
      UUO_REQUIRE(width[r.op] == 0 || width[r.op] == r.nbytes, "batch: argument structs of one kind differ in size");
      width[r.op] = r.nbytes;
      if (r.gx > gx[r.op]) gx[r.op] = r.gx;
      if (r.gy > gy[r.op]) gy[r.op] = r.gy;
      any = true;
    }
  }
  if (!any) return 0;
  size_t total = 0;
  for (int k = 0; k < UUO_OP_COUNT; ++k) {
    if (k == UUO_OP_COPY || k == UUO_OP_SKIN) continue;  // issued one by one from the host copies
    off[k] = total;
    total += (cnt[k] * width[k] + 255) / 256 * 256;
  }
  const size_t region_cap = b->blob_cap / 2, region_off = (size_t)region * region_cap;
  unsigned char* h_blob = b->h_blob + region_off;
  unsigned char* d_blob = b->d_blob + region_off;
  UUO_REQUIRE(total <= region_cap, "batch: argument staging buffer too small");
  // A flush overwrites its region of the pinned blob from the start, so the region's previous (asynchronous) host-to-device
  // copy must have executed.  Inside a solve that is implied -- every round waits for its evaluations' reports, which follow
  // the copy on the stream; a flush that follows another one with no such wait synchronises first (UuoStaging).
  if (b->staging.begin_flush(region, total))
    UUO_HIP_CHECK(hipStreamSynchronize(s));  // (a region is only ever used on one stream between two joins)
  size_t fill[UUO_OP_COUNT];
  for (int k = 0; k < UUO_OP_COUNT; ++k) fill[k] = 0;
  for (int i = i0; i < i1; ++i)
    for (const UuoOpRec& r : cos[i].rec.ops) {
      if (r.op == UUO_OP_COPY || r.op == UUO_OP_SKIN) continue;
      std::memcpy(h_blob + off[r.op] + fill[r.op] * width[r.op], r.args, r.nbytes);
      fill[r.op] += 1;
    }
  if (total) UUO_HIP_CHECK(hipMemcpyAsync(d_blob, h_blob, total, hipMemcpyHostToDevice, s));
  for (int k = 0; k < UUO_OP_COUNT; ++k) {
    if (cnt[k] == 0) continue;
    if (k == UUO_OP_COPY) {
      for (int i = i0; i < i1; ++i)
        for (const UuoOpRec& r : cos[i].rec.ops)
          if (r.op == UUO_OP_COPY) {
            const LbCopyArgs* c = reinterpret_cast<const LbCopyArgs*>(r.args);
            UUO_HIP_CHECK(hipMemcpyAsync(c->dst, c->src, c->bytes, hipMemcpyDeviceToDevice, s));
          }
      continue;
    }
    if (k == UUO_OP_SKIN) {
      for (int i = i0; i < i1; ++i)
        for (const UuoOpRec& r : cos[i].rec.ops)
          if (r.op == UUO_OP_SKIN) {
            const int rc = uuo_replay_skin_call(s, r.args);
            if (rc) return rc;
          }
      continue;
    }
    const void* da = d_blob + off[k];
    const int n_ = (int)cnt[k];
    int rc = 1;
    rc = uuo_batched_launch_lbfgs(k, s, da, n_, gx[k], gy[k]);  // each translation unit serves the kinds it defines
    if (rc == 1) rc = uuo_batched_launch_smpl(k, s, da, n_, gx[k], gy[k]);
    if (rc == 1) rc = uuo_batched_launch_nn(k, s, da, n_, gx[k], gy[k]);
    if (rc == 1) rc = uuo_batched_launch_closure(k, s, da, n_, gx[k], gy[k]);
    UUO_REQUIRE(rc != 1, "batch: no batched kernel for a recorded launch kind");
    if (rc) return rc;
    UUO_HIP_CHECK(hipGetLastError());
  }
  for (int i = i0; i < i1; ++i) cos[i].rec.ops.clear();
  return 0;
}

extern "C" int uuo_batch_solve(uuo_batch_t* b, void* stream, const uuo_problem_t* problems, float* const* d_xs, int nb,
                               const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats) {
  UUO_REQUIRE(b && problems && d_xs && opt && stats, "uuo_batch_solve: null argument");
  UUO_REQUIRE(nb >= 1 && nb <= b->B, "uuo_batch_solve: more problems than the batch was created for");
  UUO_REQUIRE(opt->max_iter > 0, "uuo_batch_solve: max_iter must be positive");
  UUO_REQUIRE(uuo_recorder == nullptr, "uuo_batch_solve: batches do not nest");
  hipStream_t s = (hipStream_t)stream;
  const int hist = opt->history_size > 0 ? opt->history_size : 100;
  int rc = 0;
  for (int i = 0; i < nb; ++i) {
    UUO_REQUIRE(problems[i].stage == b->stage, "uuo_batch_solve: every problem must be of the batch's stage");
    rc = uuo_validate_problem(b->fits[i], &problems[i]);
    if (rc) return rc;
    UUO_REQUIRE(d_xs[i] != nullptr, "uuo_batch_solve: null parameter vector");
  }
  const int n_params = uuo_problem_num_params(&problems[0]);
  // per-problem optimiser workspaces (kept with the fits; re-created when the history grows)
  for (int i = 0; i < nb; ++i) {
    LbWs* w = (LbWs*)b->fits[i]->lbws;
    if (!w || w->cap < hist + 1 || w->n_cap < n_params) {
      if (w) {
        UUO_HIP_CHECK(hipStreamSynchronize(s));
        lbws_destroy(w);
      }
      b->fits[i]->lbws = nullptr;
      rc = lbws_create(n_params, hist, &w, false);
      if (rc) return rc;
      b->fits[i]->lbws = w;
    }
  }
  UUO_HIP_CHECK(hipDeviceSynchronize());  // zero fills of new workspaces (null stream) before the first round
  // marker masks (one small read-back each) and the shared pose cache, outside record mode
  if (b->stage != UUO_STAGE_PART) {  // the part stage's chamfer term is unmasked (markers_utils.py:471-475)
    for (int i = 0; i < nb; ++i) {
      rc = uuo_ensure_mask(b->fits[i], s, &problems[i]);
      if (rc) return rc;
    }
  }
  if (b->stage == UUO_STAGE_PART && problems[0].pose_cache_id != 0) {
    for (int i = 1; i < nb; ++i)
      UUO_REQUIRE(problems[i].d_o_pose == problems[0].d_o_pose && problems[i].pose_cache_id != 0,
                  "uuo_batch_solve: part-stage problems of one batch share the body pose (d_o_pose) and its cache");
    b->fits[0]->pose_cache_id = 0;
    rc = uuo_prepare_pose_cache(b->fits[0], s, &problems[0], d_xs[0]);
    if (rc) return rc;
    for (int i = 0; i < nb; ++i) b->fits[i]->pose_cache_id = problems[i].pose_cache_id;
  }

  std::vector<BatchCo> cos(nb);
  for (int i = 0; i < nb; ++i) {
    BatchCo& c = cos[i];
    UUO_REQUIRE(c.stack.alloc(), "uuo_batch_solve: could not map a coroutine stack");
    {
      const int orc = stage_objective_init(c.obj, b->fits[i], s, &problems[i], d_xs[i]);
      if (orc) return orc;
    }
    c.w = (LbWs*)b->fits[i]->lbws;
    c.d_x = d_xs[i];
    c.opt = opt;
    c.stats = &stats[i];
    c.s = s;
    std::memset(&stats[i], 0, sizeof(stats[i]));
    getcontext(&c.ctx);
    c.ctx.uc_stack.ss_sp = c.stack.sp();
    c.ctx.uc_stack.ss_size = CoStack::kBytes;
    c.ctx.uc_link = &g_sched_ctx;
    const unsigned long long pv = (unsigned long long)reinterpret_cast<uintptr_t>(&c);
    makecontext(&c.ctx, (void (*)())batch_co_entry, 2, (unsigned)(pv & 0xFFFFFFFFull), (unsigned)(pv >> 32));
  }
  const double eval_timeout_s = (double)UUO_ENV_INT("UUO_LBFGS_EVAL_TIMEOUT_S", 60);
  struct YieldScope {  // whatever path leaves this function, the thread is out of batch mode afterwards
    YieldScope() { g_batch_yield = batch_yield_impl; }
    ~YieldScope() {
      g_batch_yield = nullptr;
      uuo_recorder = nullptr;
      g_cur_co = nullptr;
    }
  } yield_scope;
  int result = 0;
  // Two stepping groups (halves of the batch), each on its own stream: while the kernels of one group's round run, the
  // host steps the other group's coroutines and stages their launches (the ~1 ms of host work per round of a 200-problem
  // batch hides behind the kernels instead of adding to them), and the latency-bound tail of a round (finalize, the
  // solver's small kernels, the launch gaps between them) overlaps the other group's wide kernels on the GPU.  The
  // second stream is forked from the caller's and joined to it before returning.  Each group owns half of the staging blob.
  const int ngroups = nb >= 8 ? 2 : 1;
  hipStream_t gs[2] = {s, ngroups == 2 ? b->s2 : s};
  if (ngroups == 2) {
    UUO_HIP_CHECK(hipEventRecord(b->ev_fork, s));
    UUO_HIP_CHECK(hipStreamWaitEvent(b->s2, b->ev_fork, 0));
    for (int i = nb / 2; i < nb; ++i) cos[i].s = b->s2;
  }
  const int gbeg[2] = {0, ngroups == 2 ? nb / 2 : nb}, gend[2] = {ngroups == 2 ? nb / 2 : nb, nb};
  auto step_group = [&](int g) -> int {  // every live problem of the group to its next evaluation (or to its end)
    for (int i = gbeg[g]; i < gend[g]; ++i) {
      BatchCo& c = cos[i];
      if (c.done) continue;
      c.waiting = false;
      uuo_recorder = &c.rec;
      g_cur_co = &c;
      swapcontext(&g_sched_ctx, &c.ctx);
      uuo_recorder = nullptr;
      g_cur_co = nullptr;
      if (c.done && c.rc) return c.rc;
    }
    return batch_flush(b, gs[g], cos, gbeg[g], gend[g], g);
  };
  auto wait_group = [&](int g) -> int {  // the reports of the group's problems that are in an evaluation
    bool waited_any = false;
    timespec t_start;
    clock_gettime(CLOCK_MONOTONIC, &t_start);
    for (int i = gbeg[g]; i < gend[g]; ++i) {
      BatchCo& c = cos[i];
      if (c.done || !c.waiting) continue;
      waited_any = true;
      unsigned long long* rep_words = reinterpret_cast<unsigned long long*>(c.w->h_out);
      UuoWaiter waiter;
      while (__atomic_load_n(&rep_words[10], __ATOMIC_ACQUIRE) != c.w->seq) {
        if (waiter.tick()) {
          const hipError_t q = hipStreamQuery(gs[g]);
          if (q != hipErrorNotReady && __atomic_load_n(&rep_words[10], __ATOMIC_ACQUIRE) != c.w->seq) {
            uuo_set_error(std::string("batch: an evaluation did not report: ") + hipGetErrorString(q));
            return -5;
          }
          timespec t_now;
          clock_gettime(CLOCK_MONOTONIC, &t_now);
          const double waited = (double)(t_now.tv_sec - t_start.tv_sec) + 1e-9 * (double)(t_now.tv_nsec - t_start.tv_nsec);
          if (waited > eval_timeout_s) {
            uuo_set_error("batch: a round of evaluations did not finish within " + std::to_string((int)eval_timeout_s) + " s");
            return -62;
          }
        }
      }
    }
    // a report of this group arrived: its flush's copy (enqueued before the kernels that reported) has executed
    if (waited_any) b->staging.report_arrived(g);
    return 0;
  };
  auto group_live = [&](int g) {
    for (int i = gbeg[g]; i < gend[g]; ++i)
      if (!cos[i].done) return true;
    return false;
  };
  for (int g = 0; g < ngroups && result == 0; ++g) result = step_group(g);
  while (result == 0 && (group_live(0) || (ngroups == 2 && group_live(1)))) {
    for (int g = 0; g < ngroups && result == 0; ++g) {
      if (!group_live(g)) continue;
      result = wait_group(g);
      if (result == 0) result = step_group(g);
    }
  }
  if (ngroups == 2) {  // join: whatever follows on the caller's stream sees both groups' results
    if (hipEventRecord(b->ev_join, b->s2) == hipSuccess) (void)hipStreamWaitEvent(s, b->ev_join, 0);
  }
  if (result == 0) {
    UUO_HIP_CHECK(hipStreamSynchronize(s));  // (problems that ended in the last round flushed their final copies there)
  } else {
    (void)hipStreamSynchronize(s);  // unfinished coroutines are abandoned with their stacks; nothing of theirs is in flight
    if (ngroups == 2) (void)hipStreamSynchronize(b->s2);
  }
  b->staging.synchronized();  // both paths: the second stream was joined to s (or synchronised) before s was waited for
  return result;
}

// Ranking scores of the part-stage candidates of a batch at their (solved) parameter vectors: one batched forward of all of
// them (pose preparation, cached-blend skinning of each candidate's vertices, nearest-vertex search) and one score kernel.
// h_scores[i] = chamfer(markers -> vertices) + chamfer(vertices -> markers), pytorch3d's two-directional mean / mean.
extern "C" int uuo_batch_part_scores(uuo_batch_t* b, void* stream, const uuo_problem_t* problems, float* const* d_xs, int nb,
                                     float* h_scores) {
  UUO_REQUIRE(b && problems && d_xs && h_scores, "uuo_batch_part_scores: null argument");
  UUO_REQUIRE(b->stage == UUO_STAGE_PART && nb >= 1 && nb <= b->B, "uuo_batch_part_scores: not a part-stage batch / too many problems");
  UUO_REQUIRE(uuo_recorder == nullptr, "uuo_batch_part_scores: batches do not nest");
  hipStream_t s = (hipStream_t)stream;
  const int F = b->F;
  std::vector<BatchCo> cos(nb);  // only their recorders are used
  int rc = 0;
  for (int i = 0; i < nb && rc == 0; ++i) {
    rc = uuo_validate_problem(b->fits[i], &problems[i]);
    if (rc) break;
    UUO_REQUIRE(b->fits[i]->pose_cache_id == problems[i].pose_cache_id && problems[i].pose_cache_id != 0,
                "uuo_batch_part_scores: call after uuo_batch_solve of the same problems (shared pose cache)");
    uuo_recorder = &cos[i].rec;
    rc = uuo_closure_forward_at(b->fits[i], s, &problems[i], d_xs[i]);
    uuo_recorder = nullptr;
  }
  if (rc) return rc;
  rc = batch_flush(b, s, cos, 0, nb, 0);
  if (rc) return rc;
  const size_t out_doubles = (size_t)nb * F * 2;
  if (b->score_cap < out_doubles) {
    if (b->d_scores) (void)hipFree(b->d_scores);
    if (b->h_scores) (void)hipHostFree(b->h_scores);
    b->d_scores = nullptr;
    b->h_scores = nullptr;
    UUO_HIP_CHECK(hipMalloc((void**)&b->d_scores, out_doubles * sizeof(double)));
    UUO_HIP_CHECK(hipHostMalloc((void**)&b->h_scores, out_doubles * sizeof(double), hipHostMallocDefault));
    b->score_cap = out_doubles;
  }
  // The forward's argument structs sit at the start of region 0 of the pinned blob and their host-to-device copy may not
  // have executed yet (it is asynchronous): the score kernel's structs go BEHIND them, never over them.
  size_t score_off = 0;
  UUO_REQUIRE(b->staging.append(0, (size_t)nb * sizeof(PartScoreArgs), &score_off), "uuo_batch_part_scores: staging buffer too small");
  PartScoreArgs* ha = reinterpret_cast<PartScoreArgs*>(b->h_blob + score_off);
  for (int i = 0; i < nb; ++i) {
    PartScoreArgs a;
    a.h.gx = F;
    a.h.gy = 1;
    a.F = F;
    a.M = problems[i].M;
    a.V = b->model->V;
    a.ns = problems[i].n_subset;
    a.markers = problems[i].d_markers;
    a.verts = b->fits[i]->verts;
    a.subset = problems[i].d_subset;
    a.nn = b->fits[i]->nn;
    a.out = b->d_scores + (size_t)i * F * 2;
    ha[i] = a;
  }
  UUO_HIP_CHECK(hipMemcpyAsync(b->d_blob + score_off, b->h_blob + score_off, (size_t)nb * sizeof(PartScoreArgs),
                               hipMemcpyHostToDevice, s));
  rc = uuo_launch_part_scores(s, b->d_blob + score_off, nb, F);
  if (rc) return rc;
  UUO_HIP_CHECK(hipMemcpyAsync(b->h_scores, b->d_scores, out_doubles * sizeof(double), hipMemcpyDeviceToHost, s));
  UUO_HIP_CHECK(hipStreamSynchronize(s));
  b->staging.synchronized();
  for (int i = 0; i < nb; ++i) {
    double cx = 0.0, cy = 0.0;
    const double* o = b->h_scores + (size_t)i * F * 2;
    for (int f = 0; f < F; ++f) {
      cx += o[2 * f] / (double)problems[i].M;
      cy += o[2 * f + 1] / (double)problems[i].n_subset;
    }
    h_scores[i] = (float)(cx / (double)F + cy / (double)F);
  }
  return 0;
}

