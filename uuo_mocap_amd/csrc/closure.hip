// Stage closures: forward + hand-derived sparse backward of the three L-BFGS problems of the fit
// (reference optimization.py:187-275 chamfer, :329-394 marker, markers/markers_utils.py:454-562 part).
// Only the <= M vertices a frame's markers touch carry gradient, so the backward is a gather-LBS over those
// vertices instead of the reference's dense autograd GEMMs.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "frame_math.h"

#define UUO_STAGE_UPSTREAM 3  // internal: not a fitting stage (uuo_smpl_backward)

struct FinArgs {
  UuoGridHdr h;
  int stage, F;
  uuo_gptr<const float> frame_part;
  uuo_gptr<const float> betas;
  uuo_gptr<const float> o_betas;
  double closs;   // data-term coefficient on the summed per-frame values
  double cpose;   // w_pose / (F*207)
  double cbetas;  // w_betas / 10
  uuo_gptr<float> g_betas;
  uuo_gptr<float> g_z;  // part stage
  uuo_gptr<float> loss;
  uuo_gptr<const float> dir_betas;  // optional direction entries of the shared parameters
  uuo_gptr<const float> dir_z;
  uuo_gptr<double> stats;           // optional [5]: loss, g.d, max|g|, sum|g|, g.g of the whole gradient
  uuo_gptr<unsigned long long> rep_host;  // optional zero-copy report (UuoEvalReport)
  unsigned long long rep_seq;
};

struct BwdArgs {
  UuoGridHdr h;  // grid extent of this problem (lock-step batches: uuo_common.h)
  // model
  uuo_gptr<const float> PT;
  uuo_gptr<const float> ST;
  uuo_gptr<const float> vt;
  uuo_gptr<const int> Wi;
  uuo_gptr<const float> Ww;
  uuo_gptr<const UuoTree> tree;
  int V;
  // frame inputs
  UuoPoseSrc src;
  int stage, F, M;
  int stop;  // ablation only
  uuo_gptr<const float> markers;
  uuo_gptr<const float> mask;
  uuo_gptr<const unsigned long long> nn;
  uuo_gptr<const int> assign;
  uuo_gptr<const int> subset;
  uuo_gptr<const float> raw_pose;  // optimised raw body rotations (chamfer, marker) or null
  uuo_gptr<const float> o_pose;    // prior target
  uuo_gptr<const float> raw_root;  // marker stage: optimised raw root
  float cg;               // 2*w_data / normaliser
  float cpose;            // 2*w_pose / (F*207)
  float d0;
  // outputs
  uuo_gptr<float> g_pose;
  uuo_gptr<float> g_root;
  uuo_gptr<float> g_z;
  uuo_gptr<float> g_trans;
  uuo_gptr<const float> dir;   // optional: current search direction (same packing as the gradient) for the fused g.d
  int off_pose, off_root, off_z, off_trans;  // section offsets inside the flat gradient / direction vector (-1 = absent)
  int gs_pose, gs_root;  // floats per rotation in the gradient / direction vector: 9 (the reference's packing) or 6 (the
                         // solver's compact packing: the third rows, whose gradient is identically zero, left out)
  uuo_gptr<const float> frames;  // optional: FrameLds of every frame as left by k_pose_prep of this closure
  uuo_gptr<const float> C;       // part stage (k_bwd_part): the cached template + pose-corrective blend [F][V][3]
  uuo_gptr<const float> pre;     // sums of a DENSE backward formed by other kernels: the item loop is skipped and the kinematic tail
                                 // runs on them (null: the sparse gather).  Part stage (k_bwd_part): [F][UUO_PRE] left by k_part_soft
                                 // (soft assignment, extension).  General kernel: [F][UUO_PREG] left by uuo_dense_backward (dense_bwd.hip),
  uuo_gptr<const float> dpf_part;  // with the partials [UUO_DPF_NCB][F][UUO_KP] of d [pose-feature | beta] of its matrix-pipe contraction
  // item mode (k_bwd_items: the marker stage on a three-corner placement): item mm of a frame is vertex assign[mm] with the
  // upstream gradient up_items[f][mm] (M = 3 x markers; k_bary_fwd formed them), the frame's data-loss sum comes with them
  uuo_gptr<const float> up_items;   // [F][M][3]
  uuo_gptr<const float> item_loss;  // [F]
  // upstream-gradient mode (stage UUO_STAGE_UPSTREAM, SmplInference.forward's backward): the items are ALL vertices
  // (+ the 21 vertex-picked joints) with dL/dv given, instead of markers with a residual
  uuo_gptr<const float> up_verts;   // [F][V][3] or null
  uuo_gptr<const float> up_joints;  // [F][45][3] or null
  uuo_gptr<float> g_betas_frame;    // [F][10]
  uuo_gptr<float> frame_part;  // [F][UUO_FP]: 0 data-loss sum, 1 dz (part), 2 pose prior sq sum, 4..13 dbeta,
                      //              16 g.d, 17 sum|g|, 18 g.g, 19 max|g| over this frame's gradient entries
  // fused finalize (k_bwd_sparse): the block that finishes LAST sums the per-frame partials and reports (see bwd_body's tail)
  uuo_gptr<unsigned> fin_counter;   // blocks of this launch that have published their partials (null: k_finalize follows)
  FinArgs fin;
};

// ----------------------------------------------------------------------------------------------------
// K_D  finalize: sums the per-frame partials in a fixed order (double accumulators), adds the shape
// prior, writes the loss and the shared-parameter gradients (betas; z for the part stage).
// ----------------------------------------------------------------------------------------------------

// NT threads (a multiple of 32 that divides 1024).  The sums are formed exactly as by the 1024-thread kernel whatever NT is:
// 32 groups of frames x 32 components, group g takes frames g, g + 32, ...; with NT < 1024 a thread forms the sums of
// 1024 / NT groups one after the other.  COHERENT: the partials were written by OTHER blocks of the running kernel (fused
// finalize at the end of k_bwd_sparse): they are read with agent-scope loads (sc1: not served from this XCD's L2) and the
// report is written without a cache write-back (see bwd_body's tail).
template <int NT, bool COHERENT>
__device__ __forceinline__ void finalize_body(const FinArgs& a) {
  __builtin_amdgcn_s_setprio(3);  // latency-bound kernel: do not queue behind co-resident MFMA waves
  __shared__ double sh[32][32];
  const int tid = threadIdx.x;
  const int comp = tid & 31;  // 32 groups of frames, components 0..UUO_FP-1
  // thread 0's inputs for the tail (shape prior, direction entries, the part of the report block the direction kernels
  // left) are on their way while the partials are summed
  float pb[10], po[10], pd[10], pdz = 0.f;
  unsigned long long rep_pre[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < 10; ++l) {
      pb[l] = a.betas[l];
      po[l] = a.o_betas[l];
      pd[l] = a.dir_betas ? a.dir_betas[l] : 0.f;
    }
    if (a.dir_z) pdz = a.dir_z[0];
    if (a.stats && a.rep_host) {
      const unsigned long long* blk = reinterpret_cast<const unsigned long long*>(a.stats.get()) - 1;
      rep_pre[0] = blk[0];
#pragma unroll
      for (int i = 6; i < 10; ++i) rep_pre[i - 5] = blk[i];
    }
  }
#pragma unroll
  for (int gi = 0; gi < 1024 / NT; ++gi) {
    const int grp = (tid >> 5) + gi * (NT / 32);
    double acc = 0.0;
    if (comp < UUO_FP) {
      const bool is_max = (comp == 19);
      for (int f0 = grp; f0 < a.F; f0 += 32 * 8) {  // 8 independent loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + 32 * u;
          const float* src = a.frame_part.get() + (size_t)f * UUO_FP + comp;
          if constexpr (COHERENT)
            v[u] = (f < a.F) ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
          else
            v[u] = (f < a.F) ? *src : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = is_max ? fmax(acc, (double)v[u]) : acc + (double)v[u];
      }
    }
    sh[grp][comp] = acc;
  }
  __syncthreads();
  if (tid < 32) {
    double s = 0.0;
    if (tid == 19) {
      for (int g = 0; g < 32; ++g) s = fmax(s, sh[g][tid]);
    } else {
      for (int g = 0; g < 32; ++g) s += sh[g][tid];
    }
    sh[0][tid] = s;
  }
  __syncthreads();
  if (tid == 0) {
    double bsq = 0.0, sd = sh[0][16], s1 = sh[0][17], s2 = sh[0][18], sm = sh[0][19];
    for (int l = 0; l < 10; ++l) {
      const double diff = (double)pb[l] - (double)po[l];
      bsq += diff * diff;
      const float gb = (float)(sh[0][4 + l] + 2.0 * a.cbetas * diff);
      a.g_betas[l] = gb;
      if (a.dir_betas) sd += (double)gb * (double)pd[l];
      s1 += fabs((double)gb);
      s2 += (double)gb * (double)gb;
      sm = fmax(sm, fabs((double)gb));
    }
    const float lossf = (float)(a.closs * sh[0][0] + a.cpose * sh[0][2] + a.cbetas * bsq);
    a.loss[0] = lossf;
    // the statistics of this problem's OWN parameters (everything but the shape vector), for solves that share the betas
    // with other ranks (uuo_lbfgs_solve_shared: the betas' gradient is summed over the ranks before it enters any norm)
    double own1 = sh[0][17], own2 = sh[0][18], ownm = sh[0][19];
    float gb_local[10];
#pragma unroll
    for (int l = 0; l < 10; ++l) gb_local[l] = (float)(sh[0][4 + l] + 2.0 * a.cbetas * ((double)pb[l] - (double)po[l]));
    if (a.stage == UUO_STAGE_PART) {
      const float gz = (float)sh[0][1];
      a.g_z[0] = gz;
      if (a.dir_z) sd += (double)gz * (double)pdz;
      s1 += fabs((double)gz);
      s2 += (double)gz * (double)gz;
      sm = fmax(sm, fabs((double)gz));
      own1 += fabs((double)gz);
      own2 += (double)gz * (double)gz;
      ownm = fmax(ownm, fabs((double)gz));
    }
    if (a.stats) {
      a.stats[0] = (double)lossf;
      a.stats[1] = sd;
      a.stats[2] = sm;
      a.stats[3] = s1;
      a.stats[4] = s2;
      if (a.rep_host) {
        // the solver's read-back block {max|d| bits, pad, out[9]} starts one word before stats; words 1..5 are
        // the values just written, the rest was left by the direction kernels of this iteration
        // (the block is {max|d| bits, stats[0..4], four more words}; nothing is read back from device memory here)
        const double five[5] = {(double)lossf, sd, sm, s1, s2};
        // COHERENT (fused finalize): the block is written with system-scope stores (pinned host memory: they go straight out)
        // and published after they have been acknowledged -- no __threadfence_system(), i.e. no write-back of this XCD's
        // whole L2 (which holds other solves' freshly skinned vertices) per evaluation
#define FIN_REP(i_, v_)                                                                                              \
  do {                                                                                                               \
    if constexpr (COHERENT)                                                                                          \
      __hip_atomic_store(a.rep_host.get() + (i_), (unsigned long long)(v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
    else                                                                                                             \
      a.rep_host[i_] = (unsigned long long)(v_);                                                                     \
  } while (0)
        FIN_REP(0, rep_pre[0]);
#pragma unroll
        for (int i = 0; i < 5; ++i) FIN_REP(1 + i, __double_as_longlong(five[i]));
#pragma unroll
        for (int i = 6; i < 10; ++i) FIN_REP(i, rep_pre[i - 5]);
        // words 11..23: own-parameter statistics and this problem's shape gradient (read by shared-betas solves only)
        FIN_REP(11, __double_as_longlong(ownm));
        FIN_REP(12, __double_as_longlong(own1));
        FIN_REP(13, __double_as_longlong(own2));
#pragma unroll
        for (int l = 0; l < 10; ++l) FIN_REP(14 + l, __double_as_longlong((double)gb_local[l]));
#undef FIN_REP
        if constexpr (COHERENT) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(&a.rep_host[10], a.rep_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
          __threadfence_system();
          __hip_atomic_store(&a.rep_host[10], a.rep_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// K_C  one block (4 waves) per frame.  Phase 1: waves stride over the frame's markers; per marker the
// wave gathers the touched vertex (posedirs rows PT[v] split over lanes), re-skins it, forms dL/dv and
// accumulates d(pose feature) in registers, dA in wave-private LDS, dbeta/dtrans/loss in registers.
// Phase 2: joints on lanes -- A -> G, reverse kinematic sweep (parents pull from children in a fixed
// order, so the result is deterministic), Gram-Schmidt backward, priors, parameter gradients.
// ----------------------------------------------------------------------------------------------------
#ifdef UUO_DEBUG_HOOKS
// debug flavour only: shader-clock stamps of wave 0 at the phase boundaries of bwd_body (tools/bwd_phases.py --stamps)
#define BWD_NSTAMP 12
__device__ unsigned long long g_bwd_stamps[4096 * BWD_NSTAMP];
#define BWD_STAMP(i)                                                                                          \
  do {                                                                                                        \
    if (a.stop == 9 && threadIdx.x == 0 && blockIdx.x < 4096)                                                 \
      g_bwd_stamps[blockIdx.x * BWD_NSTAMP + (i)] = __builtin_amdgcn_s_memtime();                              \
  } while (0)
#else
#define BWD_STAMP(i) do {} while (0)
#endif
#define BWD_NW 4  // waves per frame block: one per SIMD, so up to three blocks share a CU at 168 VGPRs.  (6 waves -- 24 item
                  // slots, 3 rounds for M = 50 instead of 4 -- place 2,2,1,1 waves on the SIMDs and a second block no longer
                  // fits at 3 waves per SIMD: 300 blocks then run in two rounds on 256 CUs, 24.8 -> 38.5 us.  8 waves need
                  // <= 128 VGPRs for two blocks per CU and spill: 43 -> 73 us.)
#define BWD_SLOTS (BWD_NW * 4)  // (wave, 16-lane group) pairs: items in flight per block
// PART (with SPARSE): the part stage with its cached pose blend.  The body pose is a constant there, so nothing of the
// pose-feature path exists - no posedirs gather (2.5 KB per item), no feature gradient, no body-rotation epilogue - and
// the posed-template vertex is read from the cache k_part_fwd searched: v_posed = C[f][v] + S[v] . beta.
// NWV = waves per block.  One wave per frame (part stage: <= 16 items, the tail's steps never use more than 60 lanes)
// leaves the per-block latency about where it is and lets four times as many frames be resident.
// Fused finalize (round 4): BUILT, MEASURED, NOT THE DEFAULT.  With UUO_FIN_FUSED the block of
// k_bwd_sparse that finishes last does k_finalize's work (bit-identical sums, tests/test_gpu_parity.py) without any
// cache-flushing fence.  Alternating runs, 3 x 9 sequences each: 5.29 vs 5.32 M frame-evaluations/s with three sequences in
// flight (k_finalize hides behind other chains' kernels anyway) and 328-336 vs 320-323 ms for one sequence alone -- the
// write-through stores at the end of every block and the last block's uncached reads cost more than the 2-3 us launch gap
// they replace (profiles/r4_ab_finalize_separate_vs_fused.log).  Compiled into the debug flavour (UUO_FIN_UNFUSED=0 selects
// it) and into builds with -DUUO_FIN_FUSED=1 only; the product kernel is round 3's.
#ifndef UUO_FIN_FUSED
#define UUO_FIN_FUSED 0
#endif
#if UUO_FIN_FUSED || defined(UUO_DEBUG_HOOKS)
#define UUO_FIN_FUSED_BUILT 1
// per-frame partials go out with agent-scope stores (write-through: the finalize block may run on another XCD)
#define BWD_FP_STORE(i_, v_) \
  __hip_atomic_store(a.frame_part.get() + (size_t)f * UUO_FP + (i_), (float)(v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define UUO_FIN_FUSED_BUILT 0
#define BWD_FP_STORE(i_, v_) a.frame_part[(size_t)f * UUO_FP + (i_)] = (float)(v_)
#endif
// DENSE (general kernel only): the sums of a dense backward come from dense_bwd.hip (BwdArgs.pre / dpf_part); a separate
// instantiation, so that the sparse kernel of the fitted stages keeps its register allocation (168 VGPRs, no spill)
// ITEMS (general kernel only): the items are (vertex, upstream gradient) pairs handed over by another kernel; another separate
// instantiation, for the same reason
template <bool PART = false, int NWV = BWD_NW, bool DENSE = false, bool ITEMS = false>
__device__ __forceinline__ void bwd_body(const BwdArgs& a) {
  static_assert(!(PART && DENSE), "the part stage's dense sums come through the runtime `pre` pointer");
  static_assert(!ITEMS || (!PART && !DENSE), "item mode belongs to the general sparse kernel");
  static_assert(NWV == BWD_NW || (PART && NWV == 1), "one-wave blocks exist for the part stage only");
  constexpr int NT = NWV * 64, SLOTS = NWV * 4;  // threads per block, (wave, 16-lane group) item slots
  // latency-bound kernel of a solve chain: do not queue behind co-resident MFMA waves.  (Not the one-wave part-stage form:
  // a batch launches tens of thousands of those, and at a raised priority they starve the other group's solver kernels.)
  if constexpr (NWV == BWD_NW) __builtin_amdgcn_s_setprio(2);
  __shared__ FrameLds L;
  __shared__ float sA[UUO_NUM_JOINTS * 12];
  __shared__ float spf[UUO_KB];
  __shared__ float w_dA[SLOTS][UUO_NUM_JOINTS * 12];  // private accumulators: one per (wave, 16-lane group)
  __shared__ float w_dpf[PART ? 1 : SLOTS][PART ? 256 : UUO_KB];  // (the shape-gradient tail borrows 240 entries from row 0)
  __shared__ float w_red[SLOTS][16];
  __shared__ float sdA[UUO_NUM_JOINTS * 12];
  __shared__ float sdpf[UUO_KB];
  __shared__ float red[16];
  __shared__ float sdGR[UUO_NUM_JOINTS][9], sdGt[UUO_NUM_JOINTS][3], sdJ[UUO_NUM_JOINTS][3], sdR[UUO_NUM_JOINTS][9];
  __shared__ float spsq[UUO_NUM_JOINTS];
  __shared__ float sstat[28][4];  // per writer: g.d, sum|g|, g.g, max|g| (0..22 body joints, 23 root/z, 24..26 transl)
  __shared__ int s_lvl_n[UUO_MAX_DEPTH], s_lvl_j[UUO_MAX_DEPTH][UUO_LEVEL_W], s_lvl_p[UUO_MAX_DEPTH][UUO_LEVEL_W];
  __shared__ int s_nch[UUO_NUM_JOINTS], s_ch[UUO_NUM_JOINTS][4];

  const int f = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int F = a.F, M = a.M;
  BWD_STAMP(0);
  {  // the tree's tables for the kinematic tail: fetched now (one round trip hidden behind the item loop), read from LDS there
    const UuoTree* tr_ = a.tree;
    if (tid < UUO_MAX_DEPTH) s_lvl_n[tid] = tr_->level_n[tid];
    if (tid < UUO_MAX_DEPTH * UUO_LEVEL_W) {
      (&s_lvl_j[0][0])[tid] = (&tr_->level_j[0][0])[tid];
      (&s_lvl_p[0][0])[tid] = (&tr_->level_p[0][0])[tid];
    }
    if constexpr (NWV >= 4) {
      if (tid >= 64 && tid < 64 + UUO_NUM_JOINTS) s_nch[tid - 64] = tr_->nchild[tid - 64];
      if (tid >= 128 && tid < 128 + UUO_NUM_JOINTS * 4) (&s_ch[0][0])[tid - 128] = (&tr_->child[0][0])[tid - 128];
    } else {
      if (tid < UUO_NUM_JOINTS) s_nch[tid] = tr_->nchild[tid];
      for (int i = tid; i < UUO_NUM_JOINTS * 4; i += NT) (&s_ch[0][0])[i] = (&tr_->child[0][0])[i];
    }
  }
  if (a.frames) {  // block-uniform
    constexpr int NW = sizeof(FrameLds) / 4;
    float* dst_l = reinterpret_cast<float*>(&L);
    for (int i = tid; i < NW; i += NT) dst_l[i] = a.frames[(size_t)f * NW + i];
    __syncthreads();
  } else {
    frame_forward(a.src, a.tree, f, L);
  }
  if (tid < UUO_NUM_JOINTS) frame_skin_matrix(L, tid, sA + tid * 12);
  if (!PART && tid < UUO_KB) {
    float v = 0.f;
    if (tid < UUO_NUM_POSE_FEATS) {
      const int j = 1 + tid / 9, e = tid % 9;
      v = L.R[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    spf[tid] = v;
  }
  for (int i = tid; i < SLOTS * UUO_NUM_JOINTS * 12; i += NT) (&w_dA[0][0])[i] = 0.f;
  // w_dpf is written whole by plain stores at the end of the item loop; w_red's used entries likewise (every slot stores
  // its 14 sums)
  for (int i = tid; i < 28 * 4; i += NT) (&sstat[0][0])[i] = 0.f;
  __syncthreads();

  if (a.stop == 1) return;
  BWD_STAMP(1);
  // soft part closure: k_part_soft has already summed the (dense) vertex gradients of this frame -- the joint forces are the
  // translation columns of dA, the rotation blocks are not needed (the yaw's gradient comes as a torque, below)
  bool use_pre = DENSE;
  if constexpr (PART) use_pre = a.pre.get() != nullptr;  // block-uniform
  if (use_pre) {
    if constexpr (PART) {
      const float* pr = a.pre.get() + (size_t)f * UUO_PRE;
      for (int i = tid; i < UUO_NUM_JOINTS * 12; i += NT) {
        const int jj = i / 12, e = i - jj * 12;
        sdA[i] = ((e & 3) == 3) ? pr[16 + jj * 3 + (e >> 2)] : 0.f;
      }
      if (tid < 14) red[tid] = pr[tid];
    } else if constexpr (DENSE) {
      // dense backward (dense_bwd.hip): d A and d trans per frame from k_dA, d [pose-feature | beta] as the vertex chunks'
      // partials of k_dpf, summed here in chunk order
      const float* pr = a.pre.get() + (size_t)f * UUO_PREG;
      for (int i = tid; i < UUO_NUM_JOINTS * 12; i += NT) sdA[i] = pr[16 + i];
      if (tid < 4) red[tid] = pr[tid];
      if (tid < UUO_NUM_POSE_FEATS + UUO_NUM_BETAS) {  // 217 of the block's 256 threads
        float acc = 0.f;
        const float* pp = a.dpf_part.get() + (size_t)f * UUO_KP + tid;
        for (int cb = 0; cb < UUO_DPF_NCB; ++cb) acc += pp[(size_t)cb * F * UUO_KP];
        if (tid < UUO_NUM_POSE_FEATS) sdpf[tid] = acc;
        else red[4 + tid - UUO_NUM_POSE_FEATS] = acc;
      } else if (tid == UUO_NUM_POSE_FEATS + UUO_NUM_BETAS) {
        sdpf[UUO_NUM_POSE_FEATS] = 0.f;  // (padding entry of the 208-float row)
      }
    }
  } else if constexpr (!DENSE) {
  float tr[3] = {0.f, 0.f, 0.f};
  if (a.src.trans) {
    tr[0] = a.src.trans[(size_t)f * 3];
    tr[1] = a.src.trans[(size_t)f * 3 + 1];
    tr[2] = a.src.trans[(size_t)f * 3 + 2];
  }
  {
    // Four items per wave, one per 16-lane group: the per-item scalar work (blended transform, residual, dv) used
    // to be computed redundantly by all 64 lanes for ONE item at a time; now the four DPP rows of a wave carry four
    // items, each sub-lane owning 13 of the 208 pose-blend rows (k = sl + 16 t), and the sums are row reductions
    // that stay inside the DPP row (no v_readlane).  M = 50 -> 4 rounds per block instead of 13 per wave.
    const int gq = lane >> 4, sl = lane & 15;
    const int slot = wave * 4 + gq;  // 0..15
    float fk[13];
#pragma unroll
    for (int t = 0; t < 13; ++t) fk[t] = PART ? 0.f : spf[sl + 16 * t];
    const float beta_s = (sl < 10) ? L.beta[sl] : 0.f;
    float acc_pf[13];
#pragma unroll
    for (int t = 0; t < 13; ++t) acc_pf[t] = 0.f;
    float acc_db16 = 0.f, acc_dt16 = 0.f, acc_loss16 = 0.f;
    float* my_dA = &w_dA[slot][0];
    struct Item16 {
      float p[3][13];
      float x0, x1, x2, wgt, d2, vt0, vt1, vt2, st0, st1, st2;
      int4 wi;
      float4 ww;
    };
    auto fetch16 = [&](int m, Item16& q) {
      // items past M (or masked out) are processed with weight 0 on vertex 0: every contribution is scaled by it
      const bool in = m < M;
      const int mm = in ? m : 0;
      float wgt = in ? 1.f : 0.f;
      float d2 = 0.f;
      int vi;
      float up0 = 0.f, up1 = 0.f, up2 = 0.f;
      if constexpr (ITEMS) {
        vi = a.assign[mm];
        const float* pu = a.up_items + ((size_t)f * M + mm) * 3;
        up0 = pu[0]; up1 = pu[1]; up2 = pu[2];
      } else if (a.stage == UUO_STAGE_UPSTREAM) {
        if (mm < a.V) {
          vi = mm;
          if (a.up_verts) {
            const float* pu = a.up_verts + ((size_t)f * a.V + mm) * 3;
            up0 = pu[0]; up1 = pu[1]; up2 = pu[2];
          }
        } else {  // joints 24..44 of SMPL.forward are vertices picked by id
          vi = a.tree->extra_vids[mm - a.V];
          const float* pu = a.up_joints + ((size_t)f * 45 + UUO_NUM_JOINTS + (mm - a.V)) * 3;
          up0 = pu[0]; up1 = pu[1]; up2 = pu[2];
        }
      } else if (a.stage == UUO_STAGE_MARKER) {
        wgt *= a.mask[(size_t)f * M + mm];
        vi = a.assign[mm];
      } else {
        const unsigned long long key = a.nn[(size_t)f * M + mm];
        d2 = __uint_as_float((unsigned)(key >> 32));
        vi = (int)(unsigned)(key & 0xFFFFFFFFull);
        if (a.stage == UUO_STAGE_PART)
          vi = a.subset[vi];
        else
          wgt *= a.mask[(size_t)f * M + mm];
      }
      if ((unsigned)vi >= (unsigned)a.V) vi = 0;  // never happens for a completed search; keeps the gather in bounds
      q.wgt = wgt;
      q.d2 = d2;
      if (ITEMS || a.stage == UUO_STAGE_UPSTREAM) {
        q.x0 = up0; q.x1 = up1; q.x2 = up2;
      } else {
        const float* px = a.markers + ((size_t)f * M + mm) * 3;
        q.x0 = px[0]; q.x1 = px[1]; q.x2 = px[2];
      }
      if constexpr (PART) {  // template + pose-corrective offsets of this frame, from the cache
        const float* pc = a.C + ((size_t)f * a.V + vi) * 3;
        q.vt0 = pc[0];
        q.vt1 = pc[1];
        q.vt2 = pc[2];
      } else {
        const float* pt = a.PT + (size_t)vi * 3 * UUO_KB + sl;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int t = 0; t < 13; ++t) q.p[c][t] = pt[c * UUO_KB + 16 * t];
        q.vt0 = a.vt[(size_t)vi * 3];
        q.vt1 = a.vt[(size_t)vi * 3 + 1];
        q.vt2 = a.vt[(size_t)vi * 3 + 2];
      }
      const float* ps = a.ST + (size_t)vi * 30 + (sl < 10 ? sl : 0);
      q.st0 = (sl < 10) ? ps[0] : 0.f;
      q.st1 = (sl < 10) ? ps[10] : 0.f;
      q.st2 = (sl < 10) ? ps[20] : 0.f;
      q.wi = *reinterpret_cast<const int4*>(a.Wi + (size_t)vi * 4);
      q.ww = *reinterpret_cast<const float4*>(a.Ww + (size_t)vi * 4);
    };
    auto row_sum = [](float v) {  // sum over the 16 lanes of the DPP row, left on every lane of the row
      v += dpp_rot<0x128>(v);
      v += dpp_rot<0x124>(v);
      v += dpp_rot<0x122>(v);
      v += dpp_rot<0x121>(v);
      return v;
    };
    const int rounds = (M + SLOTS - 1) / SLOTS;
    Item16 cur;
    for (int r = 0; r < rounds; ++r) {
      fetch16(slot + SLOTS * r, cur);
      const float wgt = cur.wgt, d2 = cur.d2;
      float vp[3];
      if constexpr (PART) {
        vp[0] = cur.vt0 + row_sum(cur.st0 * beta_s);
        vp[1] = cur.vt1 + row_sum(cur.st1 * beta_s);
        vp[2] = cur.vt2 + row_sum(cur.st2 * beta_s);
      } else {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < 13; ++t) {
          s0 = fmaf(cur.p[0][t], fk[t], s0);
          s1 = fmaf(cur.p[1][t], fk[t], s1);
          s2 = fmaf(cur.p[2][t], fk[t], s2);
        }
        vp[0] = row_sum(s0) + (cur.vt0 + row_sum(cur.st0 * beta_s));
        vp[1] = row_sum(s1) + (cur.vt1 + row_sum(cur.st1 * beta_s));
        vp[2] = row_sum(s2) + (cur.vt2 + row_sum(cur.st2 * beta_s));
      }
      float T[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = 0.f;
      const int wj[4] = {cur.wi.x, cur.wi.y, cur.wi.z, cur.wi.w};
      const float ww[4] = {cur.ww.x, cur.ww.y, cur.ww.z, cur.ww.w};
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float4* pa4 = reinterpret_cast<const float4*>(sA + wj[n] * 12);
        const float4 r0 = pa4[0], r1 = pa4[1], r2 = pa4[2];
        const float w = ww[n];
        T[0] = fmaf(w, r0.x, T[0]); T[1] = fmaf(w, r0.y, T[1]); T[2] = fmaf(w, r0.z, T[2]); T[3] = fmaf(w, r0.w, T[3]);
        T[4] = fmaf(w, r1.x, T[4]); T[5] = fmaf(w, r1.y, T[5]); T[6] = fmaf(w, r1.z, T[6]); T[7] = fmaf(w, r1.w, T[7]);
        T[8] = fmaf(w, r2.x, T[8]); T[9] = fmaf(w, r2.y, T[9]); T[10] = fmaf(w, r2.z, T[10]); T[11] = fmaf(w, r2.w, T[11]);
      }
      const float vx = fmaf(T[2], vp[2], fmaf(T[1], vp[1], T[0] * vp[0])) + T[3] + tr[0];
      const float vy = fmaf(T[6], vp[2], fmaf(T[5], vp[1], T[4] * vp[0])) + T[7] + tr[1];
      const float vz = fmaf(T[10], vp[2], fmaf(T[9], vp[1], T[8] * vp[0])) + T[11] + tr[2];
      const float dx = cur.x0 - vx, dy = cur.x1 - vy, dz = cur.x2 - vz;
      float g[3];
      float loss_item;
      if (ITEMS || a.stage == UUO_STAGE_UPSTREAM) {
        loss_item = 0.f;
        g[0] = wgt * cur.x0; g[1] = wgt * cur.x1; g[2] = wgt * cur.x2;
      } else if (a.stage == UUO_STAGE_MARKER) {
        const float rr = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float e = rr - a.d0;
        loss_item = wgt * (e * e);
        const float sc = (rr > 0.f) ? (-a.cg * wgt * e / rr) : 0.f;
        g[0] = sc * dx; g[1] = sc * dy; g[2] = sc * dz;
      } else {
        loss_item = wgt * d2;
        const float sc = -a.cg * wgt;
        g[0] = sc * dx; g[1] = sc * dy; g[2] = sc * dz;
      }
      float dvp[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) dvp[c] = fmaf(T[8 + c], g[2], fmaf(T[4 + c], g[1], T[c] * g[0]));
      if constexpr (!PART) {
#pragma unroll
        for (int t = 0; t < 13; ++t)
          acc_pf[t] += fmaf(cur.p[2][t], dvp[2], fmaf(cur.p[1][t], dvp[1], cur.p[0][t] * dvp[0]));
      }
      acc_db16 += fmaf(cur.st2, dvp[2], fmaf(cur.st1, dvp[1], cur.st0 * dvp[0]));  // sub-lanes 0..9: d beta (direct path)
      if (sl < 12) {
        const int rr_ = sl >> 2, cc_ = sl & 3;
        const float gr = (rr_ == 0) ? g[0] : ((rr_ == 1) ? g[1] : g[2]);
        const float pc = (cc_ == 0) ? vp[0] : ((cc_ == 1) ? vp[1] : ((cc_ == 2) ? vp[2] : 1.f));
        const float val = gr * pc;
#pragma unroll
        for (int n = 0; n < 4; ++n) my_dA[wj[n] * 12 + sl] += ww[n] * val;
      }
      acc_dt16 += (sl == 0) ? g[0] : ((sl == 1) ? g[1] : g[2]);  // sub-lanes 0..2
      acc_loss16 += loss_item;                                     // sub-lane 0 is the one that is stored
    }
    if constexpr (!PART) {
#pragma unroll
      for (int t = 0; t < 13; ++t) w_dpf[slot][sl + 16 * t] = acc_pf[t];
    }
    if (sl == 0) w_red[slot][0] = acc_loss16;
    if (sl < 3) w_red[slot][1 + sl] = acc_dt16;
    if (sl < 10) w_red[slot][4 + sl] = acc_db16;
  }

  if (a.stop == 2) return;
  BWD_STAMP(2);
  // ---- block reduction over the accumulation slots (fixed order -> deterministic)
  __syncthreads();
  BWD_STAMP(3);
  if constexpr (!PART) {
    if (tid < UUO_KB) {
      float acc = w_dpf[0][tid];
#pragma unroll
      for (int w = 1; w < SLOTS; ++w) acc += w_dpf[w][tid];
      sdpf[tid] = acc;
    }
  }
  for (int i = tid; i < UUO_NUM_JOINTS * 12; i += NT) {
    float acc = w_dA[0][i];
#pragma unroll
    for (int w = 1; w < SLOTS; ++w) acc += w_dA[w][i];
    sdA[i] = acc;
  }
  if (tid < 14) {
    float acc = w_red[0][tid];
#pragma unroll
    for (int w = 1; w < SLOTS; ++w) acc += w_red[w][tid];
    red[tid] = acc;
  }
  }  // !use_pre
  __syncthreads();

  if (a.stop == 3) return;
  BWD_STAMP(4);
  // ---- phase 2: the kinematic chain backwards.  Entries 0..8 of a joint are the 3x3 of dG_j^R, entries 9..11 the 3 of
  // dG_j^t.  The tree is swept leaves -> root one depth per step: wave 0 holds the (joint, entry) pairs of the parents at
  // depth d and PULLS their children's finished totals in ascending child order (no float atomics: bit-reproducible),
  // wave 1 holds the pairs of the children at depth d + 1 and turns the same finished totals into the local-rotation
  // gradient dR_c = G_p^R^T dG_c^R and the joint gradients.  SMPL has <= 5 joints per level, so each role fits 60 lanes.
  // Each pair's arithmetic is what the 24-lane version did serially (same operands, same order): bit-identical to it; a
  // step is two LDS round trips and a few FMAs instead of hundreds of dependent instructions on one lane.
  const UuoTree* tree = a.tree;
  const int j = tid;
  // direction entries this thread will need for the fused g.d: fetched here, AFTER the item loop (nine registers that are
  // not live through it: at the 168-register budget they were nine spills inside the loop), their round trip overlaps the sweep
  float dpre[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (a.dir) {
    if (tid >= 1 && tid < UUO_NUM_JOINTS && a.off_pose >= 0) {
      const float* pd = a.dir + a.off_pose + ((size_t)f * 23 + (tid - 1)) * a.gs_pose;
#pragma unroll
      for (int e = 0; e < 9; ++e) dpre[e] = (e < 6 || a.gs_pose == 9) ? pd[e] : 0.f;
    } else if (tid == 0) {
      if (a.stage == UUO_STAGE_CHAMFER) dpre[0] = a.dir[a.off_z + f];
      if (a.stage == UUO_STAGE_MARKER) {
        const float* pd = a.dir + a.off_root + (size_t)f * a.gs_root;
#pragma unroll
        for (int e = 0; e < 6; ++e) dpre[e] = pd[e];
      }
    } else if (tid >= 32 && tid < 35) {
      dpre[0] = a.dir[a.off_trans + (size_t)f * 3 + (tid - 32)];
    }
  }

  // this thread's rotation inputs for the epilogue: issued here so their round trip overlaps the sweep
  float raw_pre[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, po_pre[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (!PART && j >= 1 && j < UUO_NUM_JOINTS && a.g_pose) {
    const float* pr = a.raw_pose + ((size_t)f * 23 + (j - 1)) * 9;
#pragma unroll
    for (int e = 0; e < 9; ++e) raw_pre[e] = pr[e];
    if (a.cpose != 0.f) {
      const float* po = a.o_pose + ((size_t)f * 23 + (j - 1)) * 9;
#pragma unroll
      for (int e = 0; e < 9; ++e) po_pre[e] = po[e];
    }
  }
  for (int t = tid; t < UUO_NUM_JOINTS * 12; t += NT) {  // 288 pairs on 256 threads
    const int jj = t / 12, e = t - jj * 12;
    if (e < 9) {
      const int r = e / 3, c = e - r * 3;
      sdGR[jj][e] = sdA[jj * 12 + r * 4 + c] - sdA[jj * 12 + r * 4 + 3] * L.J[jj][c];
    } else {
      const int r = e - 9;
      // joints 0..23 of SMPL.forward are the world translations G_j^t (+ transl): their upstream gradient enters here
      const float uj = (a.stage == UUO_STAGE_UPSTREAM && a.up_joints) ? a.up_joints[((size_t)f * 45 + jj) * 3 + r] : 0.f;
      const float dAt0 = sdA[jj * 12 + 3], dAt1 = sdA[jj * 12 + 7], dAt2 = sdA[jj * 12 + 11];
      sdGt[jj][r] = sdA[jj * 12 + r * 4 + 3] + uj;
      sdJ[jj][r] = -(fmaf(L.GR[jj][6 + r], dAt2, fmaf(L.GR[jj][3 + r], dAt1, L.GR[jj][r] * dAt0)));
    }
  }
  __syncthreads();
  BWD_STAMP(5);
  {
    const int rl = tid & 63, k = rl / 12, e = rl - k * 12;
    const int max_depth = tree->max_depth;
    for (int d = max_depth - 1; d >= 0; --d) {
     // the two roles of a step touch disjoint entries (parents: their own totals; children: their dR / dJ), so one wave
     // can take them one after the other
     for (int role = (NWV >= 2 ? (tid >> 6) : 0); role < 2; role += (NWV >= 2 ? 2 : 1)) {
      if (role == 0 && rl < 12 * UUO_LEVEL_W && k < s_lvl_n[d]) {  // pull from the children (depth d + 1, totals final)
        const int jj = s_lvl_j[d][k];
        const int nch = s_nch[jj];
        const int c0 = s_ch[jj][0], c1 = s_ch[jj][1], c2 = s_ch[jj][2], c3 = s_ch[jj][3];
        if (e < 9) {
          const int r = e / 3, aa = e - r * 3;
          float acc = sdGR[jj][e];
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) {
            if (ci < nch) {
              const int c = (ci == 0) ? c0 : ((ci == 1) ? c1 : ((ci == 2) ? c2 : c3));
              float add = fmaf(sdGR[c][r * 3 + 2], L.R[c][aa * 3 + 2],
                               fmaf(sdGR[c][r * 3 + 1], L.R[c][aa * 3 + 1], sdGR[c][r * 3] * L.R[c][aa * 3]));
              add = fmaf(sdGt[c][r], L.J[c][aa] - L.J[jj][aa], add);
              acc += add;
            }
          }
          sdGR[jj][e] = acc;
        } else {
          const int r = e - 9;
          float acc_t = sdGt[jj][r], acc_j = sdJ[jj][r];
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) {
            if (ci < nch) {
              const int c = (ci == 0) ? c0 : ((ci == 1) ? c1 : ((ci == 2) ? c2 : c3));
              acc_t += sdGt[c][r];
              acc_j -= fmaf(L.GR[jj][6 + r], sdGt[c][2], fmaf(L.GR[jj][3 + r], sdGt[c][1], L.GR[jj][r] * sdGt[c][0]));
            }
          }
          sdGt[jj][r] = acc_t;
          sdJ[jj][r] = acc_j;
        }
      } else if (role == 1 && rl < 12 * UUO_LEVEL_W && k < s_lvl_n[d + 1]) {  // own totals final, parent's G^R an input
        const int jj = s_lvl_j[d + 1][k], p = s_lvl_p[d + 1][k];
        if (e < 9) {
          const int aa = e / 3, bb = e - aa * 3;
          sdR[jj][e] = fmaf(L.GR[p][6 + aa], sdGR[jj][6 + bb], fmaf(L.GR[p][3 + aa], sdGR[jj][3 + bb], L.GR[p][aa] * sdGR[jj][bb]));
        } else {
          const int aa = e - 9;
          sdJ[jj][aa] += fmaf(L.GR[p][6 + aa], sdGt[jj][2], fmaf(L.GR[p][3 + aa], sdGt[jj][1], L.GR[p][aa] * sdGt[jj][0]));
        }
      }
     }
      __syncthreads();
    }
  }
  if (tid < 12) {  // the root: its world transform is its local one
    if (tid < 9) sdR[0][tid] = sdGR[0][tid];
    else sdJ[0][tid - 9] += sdGt[0][tid - 9];
  }
  __syncthreads();

  BWD_STAMP(6);
  // shape gradient of this frame: direct (blend shapes) + joint path (240 threads: one (joint, beta) pair each)
  for (int t = tid; t < 240; t += NT) {
    const int jj = t / 10, l = t - jj * 10;
    w_dpf[0][t] = fmaf(tree->JS[jj][2][l], sdJ[jj][2], fmaf(tree->JS[jj][1][l], sdJ[jj][1], tree->JS[jj][0][l] * sdJ[jj][0]));
  }
  __syncthreads();
  if (tid < 10) {
    float acc = red[4 + tid];
    for (int jj = 0; jj < UUO_NUM_JOINTS; ++jj) acc += w_dpf[0][jj * 10 + tid];
    BWD_FP_STORE(4 + tid, acc);
    if (a.stage == UUO_STAGE_UPSTREAM) a.g_betas_frame[(size_t)f * 10 + tid] = acc;
  }
  BWD_STAMP(7);
  // body rotations
  if (j >= 1 && j < UUO_NUM_JOINTS) {
    float psq = 0.f;
    if (!PART && a.g_pose) {
      float dR[9], raw[9], gout[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        dR[e] = sdR[j][e] + sdpf[(j - 1) * 9 + e];
        raw[e] = raw_pre[e];
      }
      if (a.src.norm_body) {
        float da[6];
        gs6d_backward(raw, dR, da);
#pragma unroll
        for (int e = 0; e < 6; ++e) gout[e] = da[e];
        gout[6] = gout[7] = gout[8] = 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 9; ++e) gout[e] = dR[e];
      }
      if (a.cpose != 0.f) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          const float diff = raw[e] - po_pre[e];
          gout[e] = fmaf(a.cpose, diff, gout[e]);
          psq = fmaf(diff, diff, psq);
        }
      }
      float* pg = a.g_pose + ((size_t)f * 23 + (j - 1)) * a.gs_pose;
      float sd = 0.f, s1 = 0.f, s2 = 0.f, sm = 0.f;
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        if (e < 6 || a.gs_pose == 9) pg[e] = gout[e];  // (compact packing: entries 6..8 are exact zeros and have no slot)
        sd = fmaf(gout[e], dpre[e], sd);
        s1 += fabsf(gout[e]);
        s2 = fmaf(gout[e], gout[e], s2);
        sm = fmaxf(sm, fabsf(gout[e]));
      }
      sstat[j - 1][0] = sd; sstat[j - 1][1] = s1; sstat[j - 1][2] = s2; sstat[j - 1][3] = sm;
    }
    spsq[j] = psq;
  }
  if (j == 0) {
    spsq[0] = 0.f;
    // d(Rz(z) root)/dz = [[-s,-c,0],[c,-s,0],[0,0,0]] root
    const float cz = L.Rz[0], sz = L.Rz[2];
    if (a.stage == UUO_STAGE_UPSTREAM) {
      float* pg = a.g_root + (size_t)f * 9;  // the root rotation is an input as it stands: dL/dR_0
#pragma unroll
      for (int e = 0; e < 9; ++e) pg[e] = sdR[0][e];
    } else if (a.stage == UUO_STAGE_CHAMFER) {
      float da[6];
      gs6d_backward(L.Mroot, sdR[0], da);
      const float* r0 = a.src.root + (size_t)f * 9;
      float dzv = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dzv = fmaf(da[c], (-sz * r0[c] - cz * r0[3 + c]), dzv);
        dzv = fmaf(da[3 + c], (cz * r0[c] - sz * r0[3 + c]), dzv);
      }
      a.g_z[f] = dzv;
      sstat[23][0] = dzv * dpre[0];
      sstat[23][1] = fabsf(dzv); sstat[23][2] = dzv * dzv; sstat[23][3] = fabsf(dzv);
    } else if (a.stage == UUO_STAGE_MARKER) {
      float da[6];
      gs6d_backward(a.raw_root + (size_t)f * 9, sdR[0], da);
      float* pg = a.g_root + (size_t)f * a.gs_root;
      float sd = 0.f, s1 = 0.f, s2 = 0.f, sm = 0.f;
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        pg[e] = da[e];
        sd = fmaf(da[e], dpre[e], sd);
        s1 += fabsf(da[e]);
        s2 = fmaf(da[e], da[e], s2);
        sm = fmaxf(sm, fabsf(da[e]));
      }
      if (a.gs_root == 9) pg[6] = pg[7] = pg[8] = 0.f;
      sstat[23][0] = sd; sstat[23][1] = s1; sstat[23][2] = s2; sstat[23][3] = sm;
    } else {
      const float* r0 = a.src.root + (size_t)f * 9;
      float dzv = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dzv = fmaf(sdR[0][c], (-sz * r0[c] - cz * r0[3 + c]), dzv);
        dzv = fmaf(sdR[0][3 + c], (cz * r0[c] - sz * r0[3 + c]), dzv);
      }
      if constexpr (PART) {
        // d v / d z = e_z x (v - root joint - trans): torque about trans_f (k_part_soft) moved to the root joint J_0 = G_0^t
        if (use_pre) dzv = a.pre[(size_t)f * UUO_PRE + 14] - (L.Gt[0][0] * red[2] - L.Gt[0][1] * red[1]);
      }
      BWD_FP_STORE(1, dzv);
    }
  }
  if (tid >= 32 && tid < 35 && a.g_trans) {
    const int c = tid - 32;
    float gt = red[1 + c];
    if (a.stage == UUO_STAGE_UPSTREAM && a.up_joints) {
      for (int jj = 0; jj < UUO_NUM_JOINTS; ++jj) gt += a.up_joints[((size_t)f * 45 + jj) * 3 + c];
    }
    a.g_trans[(size_t)f * 3 + c] = gt;
    sstat[24 + c][0] = gt * dpre[0];
    sstat[24 + c][1] = fabsf(gt); sstat[24 + c][2] = gt * gt; sstat[24 + c][3] = fabsf(gt);
  }
  BWD_STAMP(8);
  __syncthreads();
  BWD_STAMP(9);
  if (tid < 4) {  // fixed-order sum of the writers' statistics
    float acc = 0.f;
    if (tid < 3) {
      for (int w = 0; w < 27; ++w) acc += sstat[w][tid];
    } else {
      for (int w = 0; w < 27; ++w) acc = fmaxf(acc, sstat[w][3]);
    }
    BWD_FP_STORE(16 + tid, acc);
  }
  if (tid == 0) {
    float ps = 0.f;
    for (int jj = 1; jj < UUO_NUM_JOINTS; ++jj) ps += spsq[jj];
    if constexpr (ITEMS) BWD_FP_STORE(0, red[0] + a.item_loss[f]);  // (the items carry gradients; their frame's loss sum comes beside them)
    else BWD_FP_STORE(0, red[0]);
    BWD_FP_STORE(2, ps);
    if (a.stage != UUO_STAGE_PART) BWD_FP_STORE(1, 0.f);
  }
  BWD_STAMP(10);
#if UUO_FIN_FUSED_BUILT
  if constexpr (!PART) {
    // ---- fused finalize.  Every block has published its partials with write-through stores; it waits for their
    // acknowledgement, counts itself in, and the block that arrives LAST does what k_finalize did (same sums in the same
    // order: bit-identical) and reports to the host.  No release fence anywhere: a device-scope fence on this part writes back
    // the XCD's whole L2 (round 1 measured the fenced version: the single chain 8 us shorter, four chains 3 % slower); the
    // partials alone are made visible, by the scope bits of their own stores and loads.
    if (a.fin_counter) {
      __shared__ unsigned s_arrived;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        s_arrived = __hip_atomic_fetch_add(a.fin_counter.get(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (s_arrived == (unsigned)a.h.gx - 1u) {
        if (tid == 0) __hip_atomic_store(a.fin_counter.get(), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
        finalize_body<NT, true>(a.fin);
      }
    }
  }
#endif
}

// The general backward kernel (<= 4 skin weights per vertex: SMPL; uuo_model_create refuses anything else) is
// held to 168 VGPRs = 3 waves per SIMD: k_skin2 keeps 2 x 168 of the 512 registers of every SIMD for the whole of its
// run, and a backward block of another yaw hypothesis / sequence can only start beside it if it fits in the remaining
// 176 (at 238 registers it waited for the skinning kernel to drain: +1.7 % fit throughput, same single-stream time).
__global__ __launch_bounds__(BWD_NW * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_bwd_sparse(BwdArgs a) {
  bwd_body<false>(a);
}
__global__ __launch_bounds__(BWD_NW * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_bwd_sparse_b(const BwdArgs* __restrict__ batch) {
  UUO_BATCH_PICK(BwdArgs, batch)
  bwd_body<false>(a);
}
// the kinematic tail alone, on the sums of a dense backward (dense_bwd.hip)
__global__ __launch_bounds__(BWD_NW * 64) void k_bwd_dense(BwdArgs a) { bwd_body<false, BWD_NW, true>(a); }
// the sparse kernel on (vertex, upstream gradient) items (the marker stage on a three-corner placement, k_bary_fwd before it)
__global__ __launch_bounds__(BWD_NW * 64) void k_bwd_items(BwdArgs a) { bwd_body<false, BWD_NW, false, true>(a); }

// ----------------------------------------------------------------------------------------------------
// Marker stage on a three-corner (barycentric) placement (reference optimization.py:345-351 with the placement of
// compute_nearest_points' use_barycentric branch, :494-523): virtual marker vm = sum_k b_k v[i_k], loss term
// w (|x - vm| - d0)^2.  The residual needs all three corners, so a forward pass re-skins the 3 M corner vertices of a frame
// (gather-LBS: the arithmetic of the backward kernel's item loop, four corners at a time per wave), forms vm, the term
// and d loss / d vm, and hands b_k d loss / d vm to the sparse backward as per-corner items (k_bwd_items).
// Block = frame (4 waves); 16-lane group = one marker at a time, its corners one after the other.
// ----------------------------------------------------------------------------------------------------
struct BaryFwdArgs {
  uuo_gptr<const float> PT, ST, vt, Ww;
  uuo_gptr<const int> Wi;
  uuo_gptr<const UuoTree> tree;
  int V, F, M;  // M = markers
  UuoPoseSrc src;
  uuo_gptr<const float> markers, mask;
  uuo_gptr<const int> assign3;   // [M][3]
  uuo_gptr<const float> bary;    // [M][3]
  float cg, d0;
  uuo_gptr<float> frames;        // [F][FrameLds]: left for k_bwd_items of the same evaluation
  uuo_gptr<float> up_items;      // [F][3 M][3]
  uuo_gptr<float> item_loss;     // [F]
};
__global__ __launch_bounds__(BWD_NW * 64) void k_bary_fwd(BaryFwdArgs a) {
  constexpr int NT = BWD_NW * 64, SLOTS = BWD_NW * 4;
  __shared__ FrameLds L;
  __shared__ float sA[UUO_NUM_JOINTS * 12];
  __shared__ float spf[UUO_KB];
  __shared__ float sloss[SLOTS];
  const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int M = a.M;
  frame_forward(a.src, a.tree, f, L);  // (ends with a barrier)
  if (tid < UUO_NUM_JOINTS) frame_skin_matrix(L, tid, sA + tid * 12);
  if (tid < UUO_KB) {
    float v = 0.f;
    if (tid < UUO_NUM_POSE_FEATS) {
      const int j = 1 + tid / 9, e = tid % 9;
      v = L.R[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    spf[tid] = v;
  }
  __syncthreads();
  {  // the frame's state for the backward kernel of this evaluation
    constexpr int NW = sizeof(FrameLds) / 4;
    const float* src_l = reinterpret_cast<const float*>(&L);
    for (int i = tid; i < NW; i += NT) a.frames[(size_t)f * NW + i] = src_l[i];
  }
  float tr[3] = {0.f, 0.f, 0.f};
  if (a.src.trans) {
    tr[0] = a.src.trans[(size_t)f * 3];
    tr[1] = a.src.trans[(size_t)f * 3 + 1];
    tr[2] = a.src.trans[(size_t)f * 3 + 2];
  }
  const int gq = lane >> 4, sl = lane & 15;
  const int slot = wave * 4 + gq;
  float fk[13];
#pragma unroll
  for (int t = 0; t < 13; ++t) fk[t] = spf[sl + 16 * t];
  const float beta_s = (sl < 10) ? L.beta[sl] : 0.f;
  auto row_sum = [](float v) {
    v += dpp_rot<0x128>(v);
    v += dpp_rot<0x124>(v);
    v += dpp_rot<0x122>(v);
    v += dpp_rot<0x121>(v);
    return v;
  };
  float acc_loss = 0.f;
  const int rounds = (M + SLOTS - 1) / SLOTS;
  for (int r = 0; r < rounds; ++r) {
    const int m = slot + SLOTS * r;
    const bool in = m < M;
    const int mm = in ? m : 0;
    const float wgt = in ? a.mask[(size_t)f * M + mm] : 0.f;
    float vm[3] = {0.f, 0.f, 0.f}, bk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int vi = a.assign3[mm * 3 + k];
      if ((unsigned)vi >= (unsigned)a.V) vi = 0;
      bk[k] = a.bary[mm * 3 + k];
      const float* pt = a.PT + (size_t)vi * 3 * UUO_KB + sl;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        s0 = fmaf(pt[16 * t], fk[t], s0);
        s1 = fmaf(pt[UUO_KB + 16 * t], fk[t], s1);
        s2 = fmaf(pt[2 * UUO_KB + 16 * t], fk[t], s2);
      }
      const float* ps = a.ST + (size_t)vi * 30 + (sl < 10 ? sl : 0);
      const float st0 = (sl < 10) ? ps[0] : 0.f, st1 = (sl < 10) ? ps[10] : 0.f, st2 = (sl < 10) ? ps[20] : 0.f;
      float vp[3];
      vp[0] = row_sum(s0) + (a.vt[(size_t)vi * 3] + row_sum(st0 * beta_s));
      vp[1] = row_sum(s1) + (a.vt[(size_t)vi * 3 + 1] + row_sum(st1 * beta_s));
      vp[2] = row_sum(s2) + (a.vt[(size_t)vi * 3 + 2] + row_sum(st2 * beta_s));
      const int4 wi = *reinterpret_cast<const int4*>(a.Wi + (size_t)vi * 4);
      const float4 w4 = *reinterpret_cast<const float4*>(a.Ww + (size_t)vi * 4);
      const int wj[4] = {wi.x, wi.y, wi.z, wi.w};
      const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
      float T[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float* pa = sA + wj[n] * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = fmaf(ww[n], pa[e], T[e]);
      }
      const float vx = fmaf(T[2], vp[2], fmaf(T[1], vp[1], T[0] * vp[0])) + T[3] + tr[0];
      const float vy = fmaf(T[6], vp[2], fmaf(T[5], vp[1], T[4] * vp[0])) + T[7] + tr[1];
      const float vz = fmaf(T[10], vp[2], fmaf(T[9], vp[1], T[8] * vp[0])) + T[11] + tr[2];
      vm[0] = fmaf(bk[k], vx, vm[0]);
      vm[1] = fmaf(bk[k], vy, vm[1]);
      vm[2] = fmaf(bk[k], vz, vm[2]);
    }
    const float* px = a.markers + ((size_t)f * M + mm) * 3;
    const float dx = px[0] - vm[0], dy = px[1] - vm[1], dz = px[2] - vm[2];
    const float rr = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float e = rr - a.d0;
    acc_loss += wgt * (e * e);
    const float sc = (rr > 0.f) ? (-a.cg * wgt * e / rr) : 0.f;
    if (in && sl < 9) {  // sub-lane 3 k + c writes component c of corner k's item
      const int k = sl / 3, c = sl - 3 * k;
      const float gc = sc * ((c == 0) ? dx : ((c == 1) ? dy : dz));
      const float b = (k == 0) ? bk[0] : ((k == 1) ? bk[1] : bk[2]);
      a.up_items[((size_t)f * 3 * M + (size_t)m * 3 + k) * 3 + c] = b * gc;
    }
  }
  if (sl == 0) sloss[slot] = acc_loss;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int s = 0; s < SLOTS; ++s) t += sloss[s];
    a.item_loss[f] = t;
  }
}
// part stage on its cached pose blend: a fraction of the registers and two thirds of the LDS of the general kernel
// two forms: one wave per frame for <= 16 markers (the candidate search: four items per pass), four waves per frame above
// that (hmr_full.yaml: 50 markers on the full skeleton would be 13 passes of one wave)
__global__ __launch_bounds__(64) void k_bwd_part1(BwdArgs a) { bwd_body<true, 1>(a); }
__global__ __launch_bounds__(64) void k_bwd_part1_b(const BwdArgs* __restrict__ batch) {
  UUO_BATCH_PICK(BwdArgs, batch)
  bwd_body<true, 1>(a);
}
__global__ __launch_bounds__(BWD_NW * 64) void k_bwd_part(BwdArgs a) { bwd_body<true>(a); }
__global__ __launch_bounds__(BWD_NW * 64) void k_bwd_part_b(const BwdArgs* __restrict__ batch) {
  UUO_BATCH_PICK(BwdArgs, batch)
  bwd_body<true>(a);
}

__global__ __launch_bounds__(1024) void k_finalize(FinArgs a) { finalize_body<1024, false>(a); }
__global__ __launch_bounds__(1024) void k_finalize_b(const FinArgs* __restrict__ batch) {
  UUO_BATCH_PICK(FinArgs, batch)
  finalize_body<1024, false>(a);
}

// batched launches of this file's kernels (uuo_common.h): 0 = launched, 1 = not one of mine, < 0 = error
int uuo_batched_launch_closure(int op, hipStream_t s, const void* d_args, int count, int gx, int gy) {
  if (op == UUO_OP_BWD) {
    hipLaunchKernelGGL(k_bwd_sparse_b, dim3(gx, gy, count), dim3(BWD_NW * 64), 0, s, (const BwdArgs*)d_args);
  } else if (op == UUO_OP_BWD_PART) {  // gy of the record = waves per frame (1 or BWD_NW; one value per batch: same M)
    if (gy == 1)
      hipLaunchKernelGGL(k_bwd_part1_b, dim3(gx, 1, count), dim3(64), 0, s, (const BwdArgs*)d_args);
    else
      hipLaunchKernelGGL(k_bwd_part_b, dim3(gx, 1, count), dim3(BWD_NW * 64), 0, s, (const BwdArgs*)d_args);
  } else if (op == UUO_OP_FIN) {
    hipLaunchKernelGGL(k_finalize_b, dim3(gx, gy, count), dim3(1024), 0, s, (const FinArgs*)d_args);
  } else {
    return 1;
  }
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------
struct StageLayout {
  int n, off_trans, off_z, off_betas, off_pose, off_root;
  int gs_pose = 9, gs_root = 9;  // floats per rotation
};

// `compact`: the SOLVER's packing of the gradient / direction / history vectors (never of the parameters themselves): the
// third row of every optimised rotation is left out.  Under the Gram-Schmidt (6D) normalisation the data term does not
// depend on it, so its gradient is the prior's 2 c (raw - target) -- identically zero when the row starts on its target, as
// it does in every stage of the fit (optim_chamfer starts the pose ON the HMR prior, optim_markers continues from there,
// the final stage re-targets the prior at the normalised pose it starts from) -- and zero gradients mean zero search
// directions: the entry never moves and contributes exact zeros to every dot product.  A third of the L-BFGS history
// (69 of 211 floats per frame) is then dead weight that the two history passes of every iteration stream for nothing.
// uuo_stage_compactable decides per solve (it checks the rows against their targets); uuo_lbfgs_solve then runs L-BFGS on
// n_act = 142 F + 10 (chamfer) / 147 F + 10 (marker: the root's third row has no prior at all) coordinates.
static StageLayout stage_layout(int stage, int F, bool compact = false) {
  StageLayout s;
  s.n = 0;
  s.off_trans = s.off_z = s.off_betas = s.off_pose = s.off_root = -1;
  const int gp = compact ? 6 : 9;
  if (stage == UUO_STAGE_CHAMFER) {
    s.off_trans = 0;
    s.off_z = 3 * F;
    s.off_betas = 4 * F;
    s.off_pose = 4 * F + 10;
    s.n = 4 * F + 10 + 23 * gp * F;
    s.gs_pose = gp;
  } else if (stage == UUO_STAGE_MARKER) {
    s.off_pose = 0;
    s.off_betas = 23 * gp * F;
    s.off_root = 23 * gp * F + 10;
    s.off_trans = 23 * gp * F + 10 + gp * F;
    s.n = 23 * gp * F + 10 + gp * F + 3 * F;
    s.gs_pose = s.gs_root = gp;
  } else {
    s.off_z = 0;
    s.off_trans = 1;
    s.off_betas = 3 * F + 1;
    s.n = 3 * F + 11;
  }
  return s;
}

extern "C" int uuo_problem_num_params(const uuo_problem_t* p) {
  if (!p || p->F <= 0 || p->stage < 0 || p->stage > 2) return -22;
  return stage_layout(p->stage, p->F).n;
}

static int validate_problem(const uuo_fit* fit, const uuo_problem_t* p) {
  UUO_REQUIRE(fit && p, "closure: null fit/problem");
  UUO_REQUIRE(p->stage >= 0 && p->stage <= 2, "closure: unknown stage");
  UUO_REQUIRE(p->F == fit->F && p->M == fit->M, "closure: problem F/M differ from the workspace");
  UUO_REQUIRE(p->d_markers && p->d_o_pose && p->d_o_betas, "closure: markers / o_pose / o_betas required");
  if (p->stage != UUO_STAGE_MARKER) UUO_REQUIRE(p->d_root != nullptr, "closure: fixed root orientation required");
  if (p->stage == UUO_STAGE_MARKER) UUO_REQUIRE(p->d_assign != nullptr, "closure: marker stage needs d_assign");
  UUO_REQUIRE(p->n_corners == 0 || p->n_corners == 1 || (p->n_corners == 3 && p->stage == UUO_STAGE_MARKER && p->d_bary != nullptr),
              "closure: n_corners is 0 / 1 (one-hot placement) or 3 with d_bary (marker stage on a three-corner placement)");
  if (p->stage == UUO_STAGE_PART)
    UUO_REQUIRE(p->d_subset != nullptr && p->n_subset > 0 && p->n_subset <= fit->model->V, "closure: part stage needs a vertex subset");
  UUO_REQUIRE(p->w_soft == 0.f || (p->soft_tau > 0.f && p->stage != UUO_STAGE_MARKER), "closure: w_soft (soft-assignment data term, extension) needs soft_tau > 0 and the chamfer or part stage");
  UUO_REQUIRE(p->w_soft == 0.f || p->stage != UUO_STAGE_PART || (p->pose_cache_id != 0 && p->M <= 16),
              "closure: the part stage's soft-assignment term needs a pose cache id and M <= 16");
  return 0;
}

// get_marker_mask + its sum; recomputed at every public entry (markers may have changed under the same pointer)
int uuo_ensure_mask(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p) {
  int rc = uuo_launch_mask(s, p->F, p->M, p->d_markers, fit->mask, fit->scalars);
  if (rc) return rc;
  UUO_HIP_CHECK(hipMemcpyAsync(&fit->mask_sum, fit->scalars, sizeof(float), hipMemcpyDeviceToHost, s));
  UUO_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

static UuoPoseSrc stage_pose_src(const uuo_problem_t* p, const StageLayout& lay, const float* x) {
  UuoPoseSrc src;
  src.betas = x + lay.off_betas;
  src.betas_stride = 0;
  src.trans = x + lay.off_trans;
  src.z = nullptr;
  if (p->stage == UUO_STAGE_CHAMFER) {
    src.body = x + lay.off_pose;
    src.norm_body = 1;
    src.root = p->d_root;
    src.root_mode = UUO_ROOT_Z_GS;
    src.z = x + lay.off_z;
  } else if (p->stage == UUO_STAGE_MARKER) {
    src.body = x + lay.off_pose;
    src.norm_body = 1;
    src.root = x + lay.off_root;
    src.root_mode = UUO_ROOT_GS;
  } else {
    src.body = p->d_o_pose;
    src.norm_body = 0;
    src.root = p->d_root;
    src.root_mode = UUO_ROOT_ZSHARED;
    src.z = x + lay.off_z;
  }
  return src;
}

#ifdef UUO_DEBUG_HOOKS
// debug flavour (UUO_SKIN_F16_CHECK=1; tools/skin16_stress.py): after k_skin3, the fp32 kernel on the same operands into a scratch
// buffer of the workspace, and a count of the values that differ by more than 1e-5 m / of the boxes that differ at all from the
// fp32 kernel's by more than 1e-5
__device__ unsigned long long g_skin16_check[4];  // launches, vertex values off, box values off, (unused)
__global__ __launch_bounds__(256) void k_skin16_compare(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                        const float* __restrict__ ba, const float* __restrict__ bb, size_t nb) {
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  unsigned bad = 0, badb = 0;
  for (size_t i = i0; i < n; i += stride) bad += (fabsf(a[i] - b[i]) > 1e-5f) ? 1u : 0u;
  for (size_t i = i0; i < nb; i += stride) badb += (fabsf(ba[i] - bb[i]) > 1e-5f) ? 1u : 0u;
  if (bad) atomicAdd(&g_skin16_check[1], (unsigned long long)bad);
  if (badb) atomicAdd(&g_skin16_check[2], (unsigned long long)badb);
  if (i0 == 0) atomicAdd(&g_skin16_check[0], 1ull);
}
static int uuo_debug_skin16_compare(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const UuoPoseSrc& src) {
  const uuo_model* m = fit->model;
  const size_t nv = (size_t)p->F * m->V * 3, nb = (size_t)p->F * ((m->V + 15) / 16) * 6;
  if (!fit->dbg_verts) UUO_HIP_CHECK(hipMalloc((void**)&fit->dbg_verts, (nv + nb) * sizeof(float)));
  int rc = uuo_launch_skin(m, s, p->F, fit->pfaT, fit->A, src.trans, fit->dbg_verts, fit->dbg_verts + nv);
  if (rc) return rc;
  hipLaunchKernelGGL(k_skin16_compare, dim3(512), dim3(256), 0, s, fit->verts, fit->dbg_verts, nv, fit->bbox, fit->dbg_verts + nv, nb);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}
extern "C" int uuo_debug_skin16_check(unsigned long long* h_out /* [3]: launches, vertex values off, box values off */, int reset) {
  UUO_HIP_CHECK(hipDeviceSynchronize());
  unsigned long long h[4] = {0, 0, 0, 0};
  UUO_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_skin16_check), sizeof(h)));
  if (h_out) { h_out[0] = h[0]; h_out[1] = h[1]; h_out[2] = h[2]; }
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    UUO_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_skin16_check), z, sizeof(z)));
  }
  return 0;
}
#endif

// forward half shared by uuo_closure_eval and uuo_time_closure
// `need_verts`: the caller reads fit->verts afterwards (candidate scores); a closure evaluation does not (the backward
// kernel re-skins the winners), which lets the part stage keep its vertices in registers (k_part_fwd)
// `vp_out` (soft chamfer closure): the skinning kernel also leaves v_posed there, for the dense backward; *vp_done says whether it did
static int closure_forward(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const UuoPoseSrc& src, bool need_verts,
                           float* vp_out = nullptr, bool* vp_done = nullptr) {
  if (p->stage == UUO_STAGE_MARKER) return 0;  // gather-LBS: the backward kernel re-skins the M vertices itself
  const uuo_model* m = fit->model;
  int rc = 0;
  static const int no_cache = UUO_ENV_INT("UUO_PART_NOCACHE", 0);  // comparison only
  const bool cached = p->stage == UUO_STAGE_PART && p->pose_cache_id != 0 && !no_cache;
  if (cached && fit->pose_cache_id != p->pose_cache_id) {
    // first evaluation of a solve whose body pose is constant: C = v_t + P . feat through the MFMA kernel with zero
    // shape, identity skinning transforms and no translation
    UUO_REQUIRE(!uuo_recorder, "closure: the pose cache of a lock-step batch is built before its first round");
    if (!fit->pose_cache) UUO_HIP_CHECK(hipMalloc((void**)&fit->pose_cache, (size_t)p->F * m->V * 3 * sizeof(float)));
    UuoPoseSrc c = src;
    c.betas = fit->zeros16;
    c.betas_stride = 0;
    c.trans = nullptr;
    rc = uuo_launch_pose_prep(m, s, p->F, c, fit->pfaT, fit->A, nullptr, nullptr);
    if (rc) return rc;
    rc = uuo_launch_identity_transforms(s, p->F * UUO_NUM_JOINTS, fit->A);
    if (rc) return rc;
    rc = uuo_launch_skin(m, s, p->F, fit->pfaT, fit->A, nullptr, fit->pose_cache, nullptr);
    if (rc) return rc;
    fit->pose_cache_id = p->pose_cache_id;
  }
  const int unfused = UUO_ENV_INT("UUO_PART_UNFUSED", 0);  // debug flavour only: the two-kernel path, for comparison
  const bool fused = cached && !need_verts && !unfused && p->M >= 1 && p->M <= 16;
  if (p->stage == UUO_STAGE_PART && p->w_soft != 0.f && !need_verts) {  // EXTENSION: soft assignment (k_part_soft)
    UUO_REQUIRE(cached && p->M >= 1 && p->M <= 16 && p->soft_tau > 0.f,
                "closure: the fused soft-assignment part closure needs a pose cache id, 1..16 markers and soft_tau > 0");
    rc = uuo_launch_pose_prep(m, s, p->F, src, fit->pfaT, fit->A, nullptr, fit->frames, p->d_subset, p->n_subset, fit->part_sb);
    if (rc) return rc;
    return uuo_launch_part_soft(m, s, p->F, p->M, fit->pose_cache, fit->part_sb, fit->A, src.trans, p->d_subset, p->n_subset,
                                p->d_markers, fit->nn, fit->soft_pre, p->w_data, p->w_soft, p->soft_tau);
  }
  if (fused) {
    rc = uuo_launch_pose_prep(m, s, p->F, src, fit->pfaT, fit->A, nullptr, fit->frames, p->d_subset, p->n_subset, fit->part_sb);
    if (rc) return rc;
    return uuo_launch_part_fwd(m, s, p->F, p->M, fit->pose_cache, fit->part_sb, fit->A, src.trans, p->d_subset, p->n_subset,
                               p->d_markers, fit->nn);
  }
  // The chamfer closure's vertices feed the nearest-vertex SEARCH only (loss and gradient are formed in fp32 on the re-skinned
  // winners): its blend runs on the fp16 matrix pipe with split operands (k_skin3).  Callers that read the vertices (candidate
  // scores; the soft closure, whose soft-min and dense gradient are formed on them) keep the fp32 pipe.
#ifndef UUO_SKIN_F16
#define UUO_SKIN_F16 1  // (0: A/B builds of tools/build_variant.sh)
#endif
  const int skin16_on = UUO_ENV_INT("UUO_SKIN_F16", UUO_SKIN_F16);  // debug flavour only: the fp32 kernel, for comparison
  const bool skin16 = skin16_on && !cached && !need_verts && !vp_out && p->w_soft == 0.f && p->stage == UUO_STAGE_CHAMFER &&
                      m->P16 && p->d_subset == nullptr && (m->VP / 16) <= 512 && p->M <= 512;
  rc = uuo_launch_pose_prep(m, s, p->F, src, fit->pfaT, fit->A, nullptr, fit->frames, nullptr, 0, nullptr, skin16 ? fit->pfa16 : nullptr);
  if (rc) return rc;
  if (cached) {
    // more than 16 markers on a cached pose (hmr_full.yaml: 50 markers, the full skeleton): the candidate's vertices in
    // subset order with a box per 16 candidates, then the box-pruned exact search on that compact cloud -- it reports
    // candidate positions (first position on ties) exactly as the brute-force subset search does, at a third of its time
    const int part_brute = UUO_ENV_INT("UUO_PART_BRUTE", 0);  // debug flavour only: the brute-force search, for comparison
    const int nuc = (p->n_subset + 15) / 16;
    if (!need_verts && !part_brute && nuc <= 512 && p->M <= 512 && p->n_subset >= 16) {
      rc = uuo_launch_skin_cached(m, s, p->F, fit->pose_cache, fit->A, src.betas, src.trans, p->d_subset, p->n_subset,
                                  fit->verts, fit->bbox);
      if (rc) return rc;
      return uuo_launch_nn_cull(s, p->F, p->M, p->n_subset, nuc, p->d_markers, fit->verts, fit->bbox, fit->nn, fit->nn_flags);
    }
    rc = uuo_launch_skin_cached(m, s, p->F, fit->pose_cache, fit->A, src.betas, src.trans, p->d_subset, p->n_subset,
                                fit->verts);
    if (rc) return rc;
    return uuo_launch_nn(s, p->F, p->M, m->V, p->d_markers, fit->verts, p->d_subset, p->n_subset, fit->nn);
  }
  const bool cull = (p->d_subset == nullptr) && (m->VP / 16) <= 512 && p->M <= 512;
  {
    // (v_posed rides along only on the k_skin2 path with unit boxes; otherwise the dense backward skins it itself)
    const bool with_vp = cull && vp_out && !uuo_recorder;
    rc = -22;
    if (skin16) rc = uuo_launch_skin16(m, s, p->F, fit->pfa16, fit->A, src.trans, fit->verts, fit->bbox);
#ifdef UUO_DEBUG_HOOKS
    if (skin16 && rc == 0 && !uuo_recorder && UUO_ENV_INT("UUO_SKIN_F16_CHECK", 0)) {  // stress mode: every launch against the fp32 kernel
      rc = uuo_debug_skin16_compare(fit, s, p, src);
      if (rc) return rc;
    }
#endif
    if (rc == -22)  // (not asked for, or a launch geometry k_skin3 does not cover)
      rc = uuo_launch_skin(m, s, p->F, fit->pfaT, fit->A, src.trans, fit->verts, cull ? fit->bbox : nullptr, with_vp ? vp_out : nullptr);
    if (rc == -22 && with_vp) {  // (the generic skinning kernel was needed: again without the extra output)
      rc = uuo_launch_skin(m, s, p->F, fit->pfaT, fit->A, src.trans, fit->verts, cull ? fit->bbox : nullptr);
    } else if (rc == 0 && vp_done) {
      *vp_done = with_vp;
    }
  }
  if (rc) return rc;
  if (cull)  // exact search pruned by per-unit bounding boxes and the previous closure's assignment (kept in fit->nn)
    return uuo_launch_nn_cull(s, p->F, p->M, m->V, (m->V + 15) / 16, p->d_markers, fit->verts, fit->bbox, fit->nn, fit->nn_flags);
  return uuo_launch_nn(s, p->F, p->M, m->V, p->d_markers, fit->verts, p->d_subset, p->n_subset, fit->nn);
}

int uuo_validate_problem(const uuo_fit* fit, const uuo_problem_t* p) { return validate_problem(fit, p); }

int uuo_closure_forward_at(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x) {
  const StageLayout lay = stage_layout(p->stage, p->F);
  const UuoPoseSrc src = stage_pose_src(p, lay, d_x);
  return closure_forward(fit, s, p, src, true);
}

// Ranking score of part-stage candidates (reference markers/markers_utils.py:575-579: pytorch3d chamfer_distance, both
// directions, unweighted): per frame, the sum over the markers of the squared distance to the nearest subset vertex (read
// from the search result of the forward just run) and the sum over the subset's vertices of the squared distance to the
// nearest marker (distance arithmetic of the CPU loop: ((dx*dx)+(dy*dy))+(dz*dz), separately rounded).  One block per
// (frame, candidate); fp64 sums in a fixed order.
__global__ __launch_bounds__(256) void k_part_score_b(const PartScoreArgs* __restrict__ batch) {
  UUO_BATCH_PICK(PartScoreArgs, batch)
  __shared__ float sm[64 * 3];
  __shared__ double red[2][4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double sx = 0.0, sy = 0.0;
  for (int m = tid; m < a.M; m += 256) sx += (double)__uint_as_float((unsigned)(a.nn[(size_t)f * a.M + m] >> 32));
  // markers of the frame in tiles of 64; each thread keeps the running minimum of its vertices over the tiles
  const float* vf = a.verts + (size_t)f * a.V * 3;
  for (int c0 = 0; c0 < a.ns; c0 += 256) {
    const int c = c0 + tid;
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (c < a.ns) {
      const float* pv = vf + (size_t)a.subset[c] * 3;
      vx = pv[0]; vy = pv[1]; vz = pv[2];
    }
    float best = __builtin_huge_valf();
    for (int m0 = 0; m0 < a.M; m0 += 64) {
      const int mt = min(64, a.M - m0);
      __syncthreads();
      if (tid < mt * 3) sm[tid] = a.markers[((size_t)f * a.M + m0) * 3 + tid];
      __syncthreads();
      for (int m = 0; m < mt; ++m) {
        const float dx = __fsub_rn(vx, sm[m * 3]), dy = __fsub_rn(vy, sm[m * 3 + 1]), dz = __fsub_rn(vz, sm[m * 3 + 2]);
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        best = fminf(best, d);
      }
    }
    if (c < a.ns) sy += (double)best;
  }
  sx = wave_sum_d(sx);
  sy = wave_sum_d(sy);
  if (lane == 0) {
    red[0][wave] = sx;
    red[1][wave] = sy;
  }
  __syncthreads();
  if (tid == 0) {
    a.out[(size_t)f * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    a.out[(size_t)f * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

int uuo_launch_part_scores(hipStream_t s, const void* d_args, int count, int F) {
  hipLaunchKernelGGL(k_part_score_b, dim3(F, 1, count), dim3(256), 0, s, (const PartScoreArgs*)d_args);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// Builds the pose-corrective blend cache of a part-stage problem now (normally the first evaluation does): a lock-step
// batch shares ONE cache among its candidates (same body pose) and must have it before its first recorded round.
int uuo_prepare_pose_cache(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x) {
  if (p->stage != UUO_STAGE_PART || p->pose_cache_id == 0) return 0;
  if (fit->pose_cache_id == p->pose_cache_id) return 0;
  const StageLayout lay = stage_layout(p->stage, p->F);
  const UuoPoseSrc src = stage_pose_src(p, lay, d_x);
  const uuo_model* m = fit->model;
  if (!fit->pose_cache) UUO_HIP_CHECK(hipMalloc((void**)&fit->pose_cache, (size_t)p->F * m->V * 3 * sizeof(float)));
  UuoPoseSrc c = src;
  c.betas = fit->zeros16;
  c.betas_stride = 0;
  c.trans = nullptr;
  int rc = uuo_launch_pose_prep(m, s, p->F, c, fit->pfaT, fit->A, nullptr, nullptr);
  if (rc) return rc;
  rc = uuo_launch_identity_transforms(s, p->F * UUO_NUM_JOINTS, fit->A);
  if (rc) return rc;
  rc = uuo_launch_skin(m, s, p->F, fit->pfaT, fit->A, nullptr, fit->pose_cache, nullptr);
  if (rc) return rc;
  fit->pose_cache_id = p->pose_cache_id;
  return 0;
}

// ---- the solver's compact packing (see stage_layout) --------------------------------------------------------------------
// third rows of the optimised body rotations against their prior targets: flag[0] |= 1 where they differ
__global__ __launch_bounds__(256) void k_third_rows_differ(int count, const float* __restrict__ raw,
                                                            const float* __restrict__ target, int* __restrict__ flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // one rotation each
  if (i >= count) return;
  const float* a = raw + (size_t)i * 9 + 6;
  const float* b = target + (size_t)i * 9 + 6;
  // bit patterns: -0.0 against +0.0 or a NaN anywhere count as different (the packing must be provably exact)
  bool diff = false;
#pragma unroll
  for (int e = 0; e < 3; ++e) diff |= __float_as_uint(a[e]) != __float_as_uint(b[e]);
  if (diff) atomicOr(flag, 1);
}

// May this solve run on the compact packing?  Synchronises `s` (one 4-byte read-back per solve).
int uuo_stage_compactable(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x, bool* compact) {
  *compact = false;
  const int off = UUO_ENV_INT("UUO_NO_COMPACT", 0);  // debug flavour only: the full packing, for comparison
  if (off || p->stage == UUO_STAGE_PART) return 0;
  if (p->w_pose == 0.f) {  // no pose prior: the third rows have no gradient whatever they hold
    *compact = true;
    return 0;
  }
  const StageLayout lay = stage_layout(p->stage, p->F);
  int* flag = reinterpret_cast<int*>(fit->scalars + 16);
  UUO_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), s));
  const int count = p->F * 23;
  hipLaunchKernelGGL(k_third_rows_differ, dim3((count + 255) / 256), dim3(256), 0, s, count, d_x + lay.off_pose, p->d_o_pose, flag);
  UUO_HIP_CHECK(hipGetLastError());
  int h = 1;
  UUO_HIP_CHECK(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s));
  UUO_HIP_CHECK(hipStreamSynchronize(s));
  *compact = (h == 0);
  return 0;
}

// compact index -> index in the reference's packing, as up to four runs (uuo_common.h UuoIndexMap)
UuoIndexMap uuo_stage_index_map(const uuo_problem_t* p, bool compact) {
  UuoIndexMap m;
  std::memset(&m, 0, sizeof(m));
  if (!compact) return m;
  const int F = p->F;
  const StageLayout full = stage_layout(p->stage, F), c = stage_layout(p->stage, F, true);
  auto add = [&](int cb, int fo, int k69) {
    m.cb[m.nseg] = cb;
    m.fo[m.nseg] = fo;
    m.k69[m.nseg] = k69;
    ++m.nseg;
  };
  if (p->stage == UUO_STAGE_CHAMFER) {
    add(0, 0, 0);                          // trans | z | betas
    add(c.off_pose, full.off_pose, 1);     // body rotations: 6 of every 9
  } else if (p->stage == UUO_STAGE_MARKER) {
    add(0, 0, 1);                          // body rotations
    add(c.off_betas, full.off_betas, 0);   // betas
    add(c.off_root, full.off_root, 1);     // root rotations
    add(c.off_trans, full.off_trans, 0);   // trans
  }
  m.cb[m.nseg] = c.n;
  m.n_act = c.n;
  m.n_full = full.n;
  return m;
}

// closure evaluation proper; the marker mask must be current (uuo_ensure_mask)
int uuo_closure_eval_impl(uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x, float* d_loss,
                          float* d_grad, int32_t* d_nn_idx, const float* d_dir, double* d_stats,
                          const UuoEvalReport* report, bool compact) {
  int rc = 0;
  const uuo_model* m = fit->model;
  const int F = p->F, M = p->M;
  const StageLayout lay = stage_layout(p->stage, F);            // the parameters: always the reference's packing
  const StageLayout gl = stage_layout(p->stage, F, compact);    // gradient and direction: the solver's packing
  const UuoPoseSrc src = stage_pose_src(p, lay, d_x);
  const bool soft_chamfer = p->stage == UUO_STAGE_CHAMFER && p->w_soft != 0.f;
  bool have_vp = false;
  if (soft_chamfer) {
    // EXTENSION (below): the workspace of the dense backward, allocated on the first soft evaluation of this fit workspace, each
    // piece on its own (a failed allocation leaves the others usable); the forward's skinning kernel leaves v_posed in it
    UUO_REQUIRE(!uuo_recorder, "closure: the soft-assignment chamfer closure is not available inside a lock-step batch");
    if (!fit->dense) {
      rc = uuo_dense_ws_create(m, s, F, &fit->dense);
      if (rc) return rc;
    }
    if (!fit->soft_gV) UUO_HIP_CHECK(hipMalloc((void**)&fit->soft_gV, (size_t)F * m->V * 3 * sizeof(float)));
    if (!fit->soft_sm) UUO_HIP_CHECK(hipMalloc((void**)&fit->soft_sm, (size_t)4 * F * M * sizeof(float)));
  }
  const int no_vpout = UUO_ENV_INT("UUO_SOFT_NO_VPOUT", 0);  // debug flavour only: v_posed by the dense backward's own skinning launch
  rc = closure_forward(fit, s, p, src, false, (soft_chamfer && !no_vpout) ? fit->dense->vp : nullptr, &have_vp);
  if (rc) return rc;
  if (d_nn_idx && p->stage != UUO_STAGE_MARKER) {
    rc = uuo_launch_nn_unpack(s, F * M, fit->nn, nullptr, d_nn_idx);
    if (rc) return rc;
  }

  double denom;
  if (p->stage == UUO_STAGE_CHAMFER)
    denom = (double)fit->mask_sum;  // pytorch3d: div = weights.sum()
  else
    denom = (double)F * (double)M;  // mean over frames and markers
  const bool soft = p->stage != UUO_STAGE_MARKER && p->w_soft != 0.f;
  // (soft part closure: k_part_soft weights its own terms -- w_data min + w_soft softmin -- so only 1 / (F M) is left here)
  const double data_c = (denom > 0.0) ? (soft ? 1.0 : (double)p->w_data) / denom : 0.0;

  BwdArgs a;
  std::memset(&a, 0, sizeof(a));
  a.PT = m->PT; a.ST = m->ST; a.vt = m->vt; a.Wi = m->Wi; a.Ww = m->Ww; a.tree = m->tree; a.V = m->V;
  a.src = src;
  a.stage = p->stage; a.F = F; a.M = M;
  a.markers = p->d_markers;
  a.mask = fit->mask;
  a.nn = fit->nn;
  a.assign = p->d_assign;
  a.subset = p->d_subset;
  a.raw_pose = (p->stage == UUO_STAGE_PART) ? nullptr : d_x + lay.off_pose;
  a.o_pose = p->d_o_pose;
  a.raw_root = (p->stage == UUO_STAGE_MARKER) ? d_x + lay.off_root : nullptr;
  a.cg = (float)(2.0 * data_c);
  a.cpose = (p->stage == UUO_STAGE_PART) ? 0.f : (float)(2.0 * (double)p->w_pose / ((double)F * 207.0));
  a.d0 = p->marker_distance;
  a.g_pose = (p->stage == UUO_STAGE_PART) ? nullptr : d_grad + gl.off_pose;
  a.g_root = (p->stage == UUO_STAGE_MARKER) ? d_grad + gl.off_root : nullptr;
  a.g_z = (p->stage == UUO_STAGE_CHAMFER) ? d_grad + gl.off_z : nullptr;
  a.g_trans = d_grad + gl.off_trans;
  a.gs_pose = gl.gs_pose;
  a.gs_root = gl.gs_root;
  a.frame_part = fit->frame_part;
  a.frames = (p->stage == UUO_STAGE_MARKER) ? nullptr : fit->frames;
  static const int bwd_stop = UUO_ENV_INT("UUO_BWD_STOP", 0);
  a.stop = bwd_stop;
  a.dir = d_dir;
  a.off_pose = gl.off_pose; a.off_root = gl.off_root; a.off_z = gl.off_z; a.off_trans = gl.off_trans;
  a.h.gx = F;
  a.h.gy = 1;
  FinArgs fa;
  std::memset(&fa, 0, sizeof(fa));
  fa.stage = p->stage; fa.F = F;
  fa.frame_part = fit->frame_part;
  fa.betas = d_x + lay.off_betas;
  fa.o_betas = p->d_o_betas;
  fa.closs = data_c;
  fa.cpose = (p->stage == UUO_STAGE_PART) ? 0.0 : (double)p->w_pose / ((double)F * 207.0);
  fa.cbetas = (double)p->w_betas / 10.0;
  fa.g_betas = d_grad + gl.off_betas;
  fa.g_z = (p->stage == UUO_STAGE_PART) ? d_grad + gl.off_z : nullptr;
  fa.loss = d_loss;
  fa.dir_betas = d_dir ? d_dir + gl.off_betas : nullptr;
  fa.dir_z = (d_dir && p->stage == UUO_STAGE_PART) ? d_dir + gl.off_z : nullptr;
  fa.stats = d_stats;
  fa.rep_host = (report && d_stats) ? report->host : nullptr;
  fa.rep_seq = report ? report->seq : 0ull;
  fa.h.gx = 1;
  fa.h.gy = 1;
  bool fin_fused = false;
  const int part_general = UUO_ENV_INT("UUO_PART_GENERAL_BWD", 0);  // debug flavour only: the general kernel, for comparison
  if (p->stage == UUO_STAGE_PART && p->pose_cache_id != 0 && fit->pose_cache_id == p->pose_cache_id && !part_general) {
    a.C = fit->pose_cache;
    a.pre = soft ? fit->soft_pre : nullptr;
    const int waves = M <= 16 ? 1 : BWD_NW;
    if (!uuo_record(UUO_OP_BWD_PART, F, waves, a)) {
      if (waves == 1)
        hipLaunchKernelGGL(k_bwd_part1, dim3(F), dim3(64), 0, s, a);
      else
        hipLaunchKernelGGL(k_bwd_part, dim3(F), dim3(BWD_NW * 64), 0, s, a);
    }
  } else if (p->stage == UUO_STAGE_MARKER && p->n_corners == 3) {
    // three-corner (barycentric) placement: a forward pass over the 3 M corner vertices forms the virtual markers, the loss and
    // d loss / d corner; the sparse backward runs on those items (k_bwd_items)
    UUO_REQUIRE(!uuo_recorder, "closure: the three-corner marker closure is not available inside a lock-step batch");
    if (!fit->bary_items) UUO_HIP_CHECK(hipMalloc((void**)&fit->bary_items, ((size_t)F * 3 * M * 3 + F) * sizeof(float)));
    float* items = fit->bary_items;
    float* item_loss = fit->bary_items + (size_t)F * 3 * M * 3;
    BaryFwdArgs b;
    std::memset(&b, 0, sizeof(b));
    b.PT = m->PT; b.ST = m->ST; b.vt = m->vt; b.Ww = m->Ww; b.Wi = m->Wi; b.tree = m->tree;
    b.V = m->V; b.F = F; b.M = M;
    b.src = src;
    b.markers = p->d_markers;
    b.mask = fit->mask;
    b.assign3 = p->d_assign;
    b.bary = p->d_bary;
    b.cg = a.cg;
    b.d0 = p->marker_distance;
    b.frames = fit->frames;
    b.up_items = items;
    b.item_loss = item_loss;
    hipLaunchKernelGGL(k_bary_fwd, dim3(F), dim3(BWD_NW * 64), 0, s, b);
    a.M = 3 * M;
    a.up_items = items;
    a.item_loss = item_loss;
    a.frames = fit->frames;
    hipLaunchKernelGGL(k_bwd_items, dim3(F), dim3(BWD_NW * 64), 0, s, a);
  } else if (soft && p->stage == UUO_STAGE_CHAMFER) {
    // EXTENSION: soft-assignment data term of the chamfer stage.  The forward above has skinned the vertices and run the exact
    // search (dmin, the hard assignment); the soft minimum gives EVERY vertex within reach of a marker a gradient, so the
    // backward is the dense one: both blend contractions on the matrix pipe (dense_bwd.hip), then this kernel's kinematic
    // tail (yaw, Gram-Schmidt backward, priors, the solver's statistics) on their sums.
    rc = uuo_launch_soft_chamfer(s, F, M, m->V, p->d_markers, fit->verts, fit->mask, fit->mask_sum, fit->nn, p->w_data, p->w_soft,
                                 p->soft_tau, fit->soft_sm, fit->soft_gV, fit->dense->pre, UUO_PREG,
                                 ((m->VP / 16) <= 512 && M <= 512) ? fit->bbox : nullptr);  // (closure_forward's `cull` condition)
    if (rc) return rc;
    rc = uuo_dense_backward(m, s, F, fit->pfaT, fit->A, fit->soft_gV, fit->dense, have_vp);
    if (rc) return rc;
    a.pre = fit->dense->pre;
    a.dpf_part = fit->dense->part;
    hipLaunchKernelGGL(k_bwd_dense, dim3(F), dim3(BWD_NW * 64), 0, s, a);
  } else {
    UUO_REQUIRE(!soft, "closure: the soft-assignment part closure runs on the cached pose blend only");
    // the general kernel finalizes by itself: its last block to finish sums the per-frame partials and reports (bwd_body)
#ifndef UUO_FIN_FUSED
#define UUO_FIN_FUSED 1  // (0: A/B builds of tools/build_variant.sh)
#endif
    const int fin_unfused = UUO_ENV_INT("UUO_FIN_UNFUSED", !UUO_FIN_FUSED);  // debug flavour only: the separate k_finalize, for comparison
    if (!fin_unfused) {
      a.fin = fa;
      a.fin_counter = reinterpret_cast<unsigned*>(fit->scalars + 32);
      fin_fused = true;
    }
    if (!uuo_record(UUO_OP_BWD, F, 1, a)) hipLaunchKernelGGL(k_bwd_sparse, dim3(F), dim3(BWD_NW * 64), 0, s, a);
  }
  UUO_HIP_CHECK(hipGetLastError());
  if (!fin_fused) {
    if (!uuo_record(UUO_OP_FIN, 1, 1, fa)) hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, s, fa);
    UUO_HIP_CHECK(hipGetLastError());
  }

  return 0;
}

extern "C" int uuo_closure_eval(uuo_fit_t* fit, void* stream, const uuo_problem_t* p, const float* d_x, float* d_loss,
                                float* d_grad, int32_t* d_nn_idx) {
  int rc = validate_problem(fit, p);
  if (rc) return rc;
  UUO_REQUIRE(d_x && d_loss && d_grad, "uuo_closure_eval: null x/loss/grad");
  hipStream_t s = (hipStream_t)stream;
  rc = uuo_ensure_mask(fit, s, p);
  if (rc) return rc;
  return uuo_closure_eval_impl(fit, s, p, d_x, d_loss, d_grad, d_nn_idx, nullptr, nullptr);
}

extern "C" int uuo_time_closure(uuo_fit_t* fit, void* stream, const uuo_problem_t* p, const float* d_x, int iters,
                                int dominant_only, float* ms_per_eval) {
  int rc = validate_problem(fit, p);
  if (rc) return rc;
  UUO_REQUIRE(d_x && ms_per_eval && iters > 0, "uuo_time_closure: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const StageLayout lay = stage_layout(p->stage, p->F);
  float* loss = fit->scalars + 8;
  float* grad = fit->vecs;  // work vector 0
  rc = uuo_ensure_mask(fit, s, p);
  if (rc) return rc;
  const UuoPoseSrc src = stage_pose_src(p, lay, d_x);
  // warm-up
  rc = dominant_only ? closure_forward(fit, s, p, src, false) : uuo_closure_eval_impl(fit, s, p, d_x, loss, grad, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (dominant_only) {
    // the dominant kernel alone, one event pair per launch on the launch stream: the average is the kernel's own
    // duration (what rocprofv3 --kernel-trace reports), without the dispatch gap between back-to-back launches
    // (dominant_only 2: the skinning kernel of the chamfer closure's search, k_skin3 on the fp16 pipe -- the forward above has
    // left its operand; 1: the fp32 kernel k_skin2, which the operators and the other closures run)
    float total = 0.f;
    const bool k3 = dominant_only == 2;
    for (int i = 0; i < 10; ++i) {  // untimed: clocks and caches in the state of a running fit
      rc = k3 ? uuo_launch_skin16(fit->model, s, p->F, fit->pfa16, fit->A, src.trans, fit->verts, fit->bbox)
              : uuo_launch_skin(fit->model, s, p->F, fit->pfaT, fit->A, src.trans, fit->verts, fit->bbox);
      if (rc) return rc;
    }
    for (int i = 0; i < iters; ++i) {
      UUO_HIP_CHECK(hipEventRecord(fit->ev0, s));
      rc = k3 ? uuo_launch_skin16(fit->model, s, p->F, fit->pfa16, fit->A, src.trans, fit->verts, fit->bbox)
              : uuo_launch_skin(fit->model, s, p->F, fit->pfaT, fit->A, src.trans, fit->verts, fit->bbox);
      if (rc) return rc;
      UUO_HIP_CHECK(hipEventRecord(fit->ev1, s));
      UUO_HIP_CHECK(hipEventSynchronize(fit->ev1));
      float one = 0.f;
      UUO_HIP_CHECK(hipEventElapsedTime(&one, fit->ev0, fit->ev1));
      total += one;
    }
    *ms_per_eval = total / (float)iters;
    return 0;
  }
  UUO_HIP_CHECK(hipEventRecord(fit->ev0, s));
  for (int i = 0; i < iters; ++i) {
    rc = uuo_closure_eval_impl(fit, s, p, d_x, loss, grad, nullptr, nullptr, nullptr);
    if (rc) return rc;
  }
  UUO_HIP_CHECK(hipEventRecord(fit->ev1, s));
  UUO_HIP_CHECK(hipEventSynchronize(fit->ev1));
  float ms = 0.f;
  UUO_HIP_CHECK(hipEventElapsedTime(&ms, fit->ev0, fit->ev1));
  *ms_per_eval = ms / (float)iters;
  return 0;
}

#ifdef UUO_DEBUG_HOOKS
// host-only debug hook: the compact -> full index map of a stage at F frames, evaluated for every solver coordinate
// (h_out: n_act ints); returns n_act, or n_full (and fills the identity) for the full packing / the part stage
extern "C" int uuo_debug_index_map(int stage, int F, int compact, int* h_out) {
  UUO_REQUIRE(stage >= 0 && stage <= 2 && F > 0 && h_out, "uuo_debug_index_map: bad arguments");
  uuo_problem_t p;
  std::memset(&p, 0, sizeof(p));
  p.stage = stage;
  p.F = F;
  const UuoIndexMap m = uuo_stage_index_map(&p, compact != 0 && stage != UUO_STAGE_PART);
  const int n = m.nseg ? m.n_act : stage_layout(stage, F).n;
  for (int c = 0; c < n; ++c) h_out[c] = m.full(c);
  return n;
}

// debug hook: shader-clock stamps left by the last k_bwd launch with UUO_BWD_STOP=9 ([4096][BWD_NSTAMP] cycles)
extern "C" int uuo_debug_bwd_stamps(unsigned long long* h_out) {
  UUO_HIP_CHECK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_bwd_stamps), sizeof(unsigned long long) * 4096 * BWD_NSTAMP));
  return 0;
}

// debug/test hook (not in the public header): survivor counts of the last pruned nearest-neighbour search
extern "C" int uuo_debug_nn_flags(uuo_fit_t* fit, int* h_out) {
  UUO_REQUIRE(fit && h_out, "uuo_debug_nn_flags: null argument");
  UUO_HIP_CHECK(hipMemcpy(h_out, fit->nn_flags, (size_t)fit->F * 8 * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

// debug/test hook: v_posed as the last soft chamfer evaluation left it for the dense backward ([F][V][3])
extern "C" int uuo_debug_dense_vp(uuo_fit_t* fit, float* h_out) {
  UUO_REQUIRE(fit && h_out && fit->dense, "uuo_debug_dense_vp: no dense workspace");
  UUO_HIP_CHECK(hipDeviceSynchronize());
  UUO_HIP_CHECK(hipMemcpy(h_out, fit->dense->vp, (size_t)fit->F * fit->model->V * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// debug/test hook (not in the public header): vertices and unit boxes of the last closure evaluation
extern "C" int uuo_debug_fit_buffers(uuo_fit_t* fit, float* h_verts, float* h_bbox) {
  UUO_REQUIRE(fit, "uuo_debug_fit_buffers: null argument");
  if (h_verts)
    UUO_HIP_CHECK(hipMemcpy(h_verts, fit->verts, (size_t)fit->F * fit->model->V * 3 * sizeof(float), hipMemcpyDeviceToHost));
  if (h_bbox)
    UUO_HIP_CHECK(hipMemcpy(h_bbox, fit->bbox, (size_t)fit->F * ((fit->model->V + 15) / 16) * 6 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

#endif  // UUO_DEBUG_HOOKS


// upstream gradient of the vertex-picked joints 24..44 (SMPL.forward's VertexJointSelector) added to their vertices'; the
// picked ids are distinct, one thread each
__global__ __launch_bounds__(64) void k_add_picked_joints(int V, const UuoTree* __restrict__ tree, const float* __restrict__ up_joints,
                                                           float* __restrict__ gV) {
  const int f = blockIdx.x, e = threadIdx.x;
  if (e >= UUO_NUM_EXTRA_JOINTS) return;
  const int v = tree->extra_vids[e];
  for (int k = 0; k < e; ++k)
    if (tree->extra_vids[k] == v) return;  // (a repeated id: its first occurrence adds all of them, below)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int k = e; k < UUO_NUM_EXTRA_JOINTS; ++k)
    if (tree->extra_vids[k] == v) {
      const float* pu = up_joints + ((size_t)f * 45 + UUO_NUM_JOINTS + k) * 3;
      s0 += pu[0]; s1 += pu[1]; s2 += pu[2];
    }
  float* pg = gV + ((size_t)f * V + v) * 3;
  pg[0] += s0; pg[1] += s1; pg[2] += s2;
}

// ----------------------------------------------------------------------------------------------------
// C ABI: backward of SmplInference.forward (reference utils/smpl.py:29-50 is differentiated by torch autograd
// through smplx.lbs).  Same kernel as the fitted closures, in upstream-gradient mode: every vertex is an item.
// ----------------------------------------------------------------------------------------------------
extern "C" int uuo_smpl_backward(uuo_model_t* m, void* stream, int F, const float* d_poses, const float* d_betas,
                                 int betas_rows, const float* d_root, const float* d_trans, const float* d_up_verts,
                                 const float* d_up_joints, float* d_g_poses, float* d_g_betas, float* d_g_root,
                                 float* d_g_trans, float* d_scratch /* [F * UUO_FP] */) {
  UUO_REQUIRE(m && d_poses && d_betas && d_root && d_g_poses && d_g_betas && d_g_root && d_g_trans && d_scratch,
              "uuo_smpl_backward: null argument");
  UUO_REQUIRE(d_up_verts || d_up_joints, "uuo_smpl_backward: no upstream gradient");
  UUO_REQUIRE(F > 0, "uuo_smpl_backward: F must be positive");
  UUO_REQUIRE(betas_rows == 1 || betas_rows == F, "uuo_smpl_backward: betas rows must be 1 or F");
  hipStream_t s = (hipStream_t)stream;
  BwdArgs a;
  std::memset(&a, 0, sizeof(a));
  a.PT = m->PT; a.ST = m->ST; a.vt = m->vt; a.Wi = m->Wi; a.Ww = m->Ww; a.tree = m->tree; a.V = m->V;
  a.src.body = d_poses;
  a.src.norm_body = 0;
  a.src.root = d_root;
  a.src.root_mode = UUO_ROOT_RAW;
  a.src.z = nullptr;
  a.src.betas = d_betas;
  a.src.betas_stride = (betas_rows == 1) ? 0 : 10;
  a.src.trans = d_trans;
  a.stage = UUO_STAGE_UPSTREAM;
  a.F = F;
  a.M = m->V + (d_up_joints ? UUO_NUM_EXTRA_JOINTS : 0);
  a.raw_pose = d_poses;
  a.g_pose = d_g_poses;
  a.g_root = d_g_root;
  a.g_trans = d_g_trans;
  a.g_betas_frame = d_g_betas;
  a.up_verts = d_up_verts;
  a.up_joints = d_up_joints;
  a.frame_part = d_scratch;
  a.off_pose = a.off_root = a.off_z = a.off_trans = -1;
  a.gs_pose = a.gs_root = 9;
  const int gather = UUO_ENV_INT("UUO_SMPL_BWD_GATHER", 0);  // debug flavour only: every vertex an item of the sparse gather
  if (d_up_verts && !gather) {
    // dense route (dense_bwd.hip): both blend contractions on the matrix pipe, the kinematic tail on their sums.  Scratch per
    // stream, under the entry's mutex across reallocation and launches (as uuo_smpl_forward)
    uuo_model::BwdScratch* scp;
    {
      std::lock_guard<std::mutex> lock(m->fwd_mutex);
      scp = &m->bwd[s];
    }
    uuo_model::BwdScratch& sc = *scp;
    std::lock_guard<std::mutex> entry_lock(sc.mu);
    const int nFT = (F + UUO_FT - 1) / UUO_FT;
    if (sc.capF != F) {
      if (sc.pfaT) (void)hipFree(sc.pfaT);
      if (sc.A) (void)hipFree(sc.A);
      if (sc.frames) (void)hipFree(sc.frames);
      if (sc.gcopy) (void)hipFree(sc.gcopy);
      uuo_dense_ws_destroy(sc.ws);
      sc.pfaT = sc.A = sc.frames = sc.gcopy = nullptr;
      sc.ws = nullptr;
      sc.capF = 0;
      UUO_HIP_CHECK(hipMalloc((void**)&sc.pfaT, (size_t)nFT * UUO_KP * UUO_FT * sizeof(float)));
      UUO_HIP_CHECK(hipMalloc((void**)&sc.A, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 12 * sizeof(float)));
      UUO_HIP_CHECK(hipMalloc((void**)&sc.frames, (size_t)F * sizeof(FrameLds)));
      UUO_HIP_CHECK(hipMemsetAsync(sc.pfaT, 0, (size_t)nFT * UUO_KP * UUO_FT * sizeof(float), s));
      UUO_HIP_CHECK(hipMemsetAsync(sc.A, 0, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 12 * sizeof(float), s));
      int rc = uuo_dense_ws_create(m, s, F, &sc.ws);
      if (rc) return rc;
      sc.capF = F;
    }
    int rc = uuo_launch_pose_prep(m, s, F, a.src, sc.pfaT, sc.A, nullptr, sc.frames);
    if (rc) return rc;
    const float* gV = d_up_verts;
    if (d_up_joints) {  // joints 24..44 are vertices picked by id: their upstream gradient joins the vertices'
      if (!sc.gcopy) UUO_HIP_CHECK(hipMalloc((void**)&sc.gcopy, (size_t)F * m->V * 3 * sizeof(float)));
      UUO_HIP_CHECK(hipMemcpyAsync(sc.gcopy, d_up_verts, (size_t)F * m->V * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
      hipLaunchKernelGGL(k_add_picked_joints, dim3(F), dim3(64), 0, s, m->V, m->tree, d_up_joints, sc.gcopy);
      gV = sc.gcopy;
    }
    rc = uuo_dense_backward(m, s, F, sc.pfaT, sc.A, gV, sc.ws);
    if (rc) return rc;
    a.frames = sc.frames;
    a.pre = sc.ws->pre;
    a.dpf_part = sc.ws->part;
    hipLaunchKernelGGL(k_bwd_dense, dim3(F), dim3(BWD_NW * 64), 0, s, a);
    UUO_HIP_CHECK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_bwd_sparse, dim3(F), dim3(BWD_NW * 64), 0, s, a);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}
