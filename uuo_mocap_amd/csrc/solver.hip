// L-BFGS with strong-Wolfe line search on device-resident vectors.
// Mirrors torch.optim.LBFGS.step / _strong_wolfe / _cubic_interpolate (torch 2.10 semantics; the reference
// constructs it at optimization.py:176-183,319-326 and markers/markers_utils.py:428-435): same direction
// update rule (history push iff y.s > 1e-10, H_diag = y.s / y.y), same first-step length, bracket / zoom
// logic, termination tests and their order.  What differs is the arithmetic route, not the algorithm:
//  * the two-loop recursion is evaluated in coefficient space from Gram matrices of the (s, y) history
//    (two passes over the history per iteration instead of 4*k dependent dot/axpy launches),
//  * dot products accumulate in fp64, line-search scalars are fp64 on the host,
//  * one small read-back per closure evaluation is the only host synchronisation.
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>

#include "frame_math.h"

// How a host thread waits for a report word in pinned memory (closure evaluations, Gram rows, lock-step rounds).  Default:
// spin (pause) -- lowest latency, one CPU per waiting thread.  uuo_set_wait_policy(spin_polls, sleep_ns) makes every wait
// sleep `sleep_ns` at a time once it has polled `spin_polls` times: for hosts whose CPU quota is smaller than the number
// of solves in flight (a throttled cgroup stalls ALL threads of the process for the rest of the scheduler period).
#include <sys/prctl.h>
#include <atomic>
static std::atomic<int> g_wait_spin_polls{-1};  // < 0: never sleep
static std::atomic<int> g_wait_sleep_ns{20000};
extern "C" int uuo_set_wait_policy(int spin_polls, int sleep_ns) {
  UUO_REQUIRE(sleep_ns >= 0 && sleep_ns <= 10000000, "uuo_set_wait_policy: sleep_ns must be within [0, 10 ms]");
  g_wait_sleep_ns.store(sleep_ns > 0 ? sleep_ns : 1, std::memory_order_relaxed);
  g_wait_spin_polls.store(spin_polls, std::memory_order_relaxed);
  return 0;
}
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

__global__ void k_lb_init(LbDev* st) {
  st->Hdiag = 1.0;
  st->cg = 0.0;
  st->gg = 0.0;
  st->head = 0;
  st->count = 0;
  st->dmax_bits = 0u;
  for (int i = 0; i < 16; ++i) st->out[i] = 0.0;
}

// ---------------------------------------------------------------------------------------------------- kernels
// first iteration: d = -g and the first trial point x + t d in one pass.  g, d (and the history) live in the SOLVER's index
// space, the iterates x / xt in the reference's parameter packing; `map` takes the former to the latter (the identity unless
// the solve runs on the compact packing of closure.hip's stage_layout, where the never-moving third rows of the rotations have
// no solver coordinate: their entries of x are copied once when the solve starts and not touched again).
__device__ __forceinline__ void lb_neg_body(int n, const float* __restrict__ g, float* __restrict__ d, const float* __restrict__ x,
                         float t, float* __restrict__ xt, const UuoIndexMap& map) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float di = -g[i];
    d[i] = di;
    const int fi = map.full(i);
    xt[fi] = x[fi] + t * di;
  }
}

__device__ __forceinline__ void lb_axpy_body(int n, const float* __restrict__ x, float t, const float* __restrict__ d,
                          float* __restrict__ o, const UuoIndexMap& map) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int fi = map.full(i);
    o[fi] = x[fi] + t * d[i];  // p.add_(d, alpha=t): one multiply, one add (no contraction)
  }
}

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
__global__ void k_lb_neg(LbNegArgs a) { lb_neg_body(a.n, a.g, a.d, a.x, a.t, a.xt, a.map); }
__global__ void k_lb_neg_b(const LbNegArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbNegArgs, batch)
  lb_neg_body(a.n, a.g, a.d, a.x, a.t, a.xt, a.map);
}
struct LbAxpyArgs {
  UuoGridHdr h;
  int n;
  uuo_gptr<const float> x;
  float t;
  uuo_gptr<const float> d;
  uuo_gptr<float> o;
  UuoIndexMap map;
};
__global__ void k_lb_axpy(LbAxpyArgs a) { lb_axpy_body(a.n, a.x, a.t, a.d, a.o, a.map); }
__global__ void k_lb_axpy_b(const LbAxpyArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbAxpyArgs, batch)
  lb_axpy_body(a.n, a.x, a.t, a.d, a.o, a.map);
}

__global__ void k_lb_form(int n, const float* __restrict__ g, const float* __restrict__ gp,
                          const float* __restrict__ d, float t, float* __restrict__ s_new, float* __restrict__ y_new) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    y_new[i] = g[i] - gp[i];
    s_new[i] = d[i] * t;
  }
}

// stats of a gradient against the current direction: partial sums per block
__global__ __launch_bounds__(256) void k_lb_stats(int n, const float* __restrict__ g, const float* __restrict__ d,
                                                   double* __restrict__ part /* [grid][4] */) {
  __shared__ double sh[4][4];
  double dot = 0.0, l1 = 0.0, gg = 0.0;
  float mx = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float gi = g[i];
    const float di = d ? d[i] : 0.f;
    dot += (double)gi * (double)di;
    l1 += (double)fabsf(gi);
    gg += (double)gi * (double)gi;
    mx = fmaxf(mx, fabsf(gi));
  }
  dot = wave_sum_d(dot);
  l1 = wave_sum_d(l1);
  gg = wave_sum_d(gg);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[w][0] = dot;
    sh[w][1] = l1;
    sh[w][2] = gg;
    sh[w][3] = (double)mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 4 + 0] = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
    part[blockIdx.x * 4 + 1] = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
    part[blockIdx.x * 4 + 2] = (sh[0][2] + sh[1][2]) + (sh[2][2] + sh[3][2]);
    part[blockIdx.x * 4 + 3] = fmax(fmax(sh[0][3], sh[1][3]), fmax(sh[2][3], sh[3][3]));
  }
}

__global__ __launch_bounds__(64) void k_lb_stats_final(int nblk, const double* __restrict__ part,
                                                        const float* __restrict__ loss, LbDev* __restrict__ st) {
  const int lane = threadIdx.x;  // nblk <= 64: one partial block per lane, fixed butterfly order
  double dot = 0.0, l1 = 0.0, gg = 0.0, mx = 0.0;
  if (lane < nblk) {
    dot = part[lane * 4];
    l1 = part[lane * 4 + 1];
    gg = part[lane * 4 + 2];
    mx = part[lane * 4 + 3];
  }
  dot = wave_sum_d(dot);
  l1 = wave_sum_d(l1);
  gg = wave_sum_d(gg);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) {
    LbOut* o = reinterpret_cast<LbOut*>(st->out);
    o->loss = (double)loss[0];
    o->gtd_new = dot;
    o->gmax = mx;
    o->g1 = l1;
    o->gg = gg;
    o->dmax = (double)__uint_as_float(st->dmax_bits);
    st->gg = gg;
  }
}

// History layout.  S and Y are stored in column blocks of LB_CW floats: element i of slot j lives at
//   (i / LB_CW) * (capL * LB_CW + LB_CBPAD) + j * LB_CW + i % LB_CW            (capL = allocated slots)
// so that all the slots of one column block are one contiguous region (capL x 2 KB).  Both passes over the history --
// the row-wise dots below and the column-wise combination in k_lb_direction -- then stream contiguous memory:
// a block reads capL consecutive 2-KB pieces instead of 2-KB (or 512-byte) pieces 260 KB apart, which is what
// HBM pages and the MALL like; the first version (row-major history) reached ~2.5 TB/s however many solves ran.
#define LB_CW 512
#define LB_CBPAD 64  // floats of padding after each column block's slots: region stride = odd multiple of 256 B, so
                     // that blocks working on different column blocks at the same slot do not camp on the same
                     // memory channels (101 x 2 KB alone is 808 x 256 B)
#define LB_CBSTRIDE(capL) ((size_t)(capL) * LB_CW + LB_CBPAD)
__device__ __host__ __forceinline__ size_t lb_hist_off(int slot, int i, int capL) {
  return (size_t)(i / LB_CW) * LB_CBSTRIDE(capL) + (size_t)slot * LB_CW + (i % LB_CW);
}

// rows of the history (and g) against {y_new, s_new, g}: skinny GEMM, fp64 accumulation.
// grid = (column groups, LB_DRS row splits).  A block walks the column blocks of its group; per column block every
// wave loads the three right-hand vectors (y_new = g - g_prev and s_new = t d are formed on the fly and stored to
// the candidate slot by split 0) and the 512-column pieces of its rows (row r belongs to wave r mod 4*LB_DRS), all
// loads of a column block in flight together.  Per-lane fp64 accumulators, one wave reduction per row at the end.
#define LB_DRS 16                        // row splits
#define LB_DRW ((LB_ROWS + 4 * LB_DRS - 1) / (4 * LB_DRS))  // rows per wave (4)
__device__ __forceinline__ void lb_dots_body(int n, int cap, int capL, int head, int count, int cand,
                                                  float* __restrict__ S, float* __restrict__ Y,
                                                  const float* __restrict__ g, const float* __restrict__ gp,
                                                  const float* __restrict__ d, float t, int ncb, int gcb,
                                                  double* __restrict__ part /* [groups][LB_ROWS][3] */,
                                                  int skip_lo = 0, int skip_hi = 0) {
  __builtin_amdgcn_s_setprio(1);  // latency-bound kernel: do not queue behind co-resident MFMA waves
  const int grp = blockIdx.x, rs = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wg = rs * 4 + wave;  // wave's index among the 4 * LB_DRS row owners
  // Stored rows: S slots then Y slots of the `count` pairs already in the history; wave wg owns rows wg + 64 q.
  // Rows past the end are clamped to row 0 and their result dropped, so the loop below has no per-row branches
  // and all its loads issue back to back.  The three rows that are not in memory yet (s_new, y_new, g) belong to
  // wave 0 of split 0, which also stores the new pair.
  const int nmem = 2 * count;
  const float* rptr[LB_DRW];
  int rout[LB_DRW];
#pragma unroll
  for (int q = 0; q < LB_DRW; ++q) {
    const int r = wg + q * 4 * LB_DRS;
    const bool ok = r < nmem;
    const int rr = ok ? r : 0;
    const bool isY = rr >= count;
    const int slot = (head + (isY ? rr - count : rr)) % cap;
    rptr[q] = (isY ? Y : S) + (size_t)slot * LB_CW + lane * 4;
    rout[q] = ok ? (isY ? LB_MAXH + slot : slot) : -1;
  }
  const bool special = (rs == 0 && wave == 0);
  double acc[LB_DRW][3], sp[3][3];
#pragma unroll
  for (int q = 0; q < LB_DRW; ++q) acc[q][0] = acc[q][1] = acc[q][2] = 0.0;
#pragma unroll
  for (int a_ = 0; a_ < 3; ++a_) sp[a_][0] = sp[a_][1] = sp[a_][2] = 0.0;

  const int cb0 = grp * gcb, cb1 = min(ncb, cb0 + gcb);
  // software pipeline over the group's column blocks, one 256-column half per stage: the loads of half hb + 1 are
  // issued before half hb is consumed (3 + LB_DRW float4 per lane in flight per stage).  Half a column block per
  // stage rather than a whole one keeps the kernel at 168 registers = 3 waves per SIMD, which is also what lets its
  // blocks start beside a running k_skin2 (2 x 168 of the 512 registers of every SIMD) instead of waiting for it.
  float4 ng, np_, nd, nr[LB_DRW];
  auto issue = [&](int hb) {
    const int cb = hb >> 1, h = hb & 1;
    const int i = cb * LB_CW + h * 256 + lane * 4;  // vectors are padded to a multiple of LB_CW: in-bounds loads
    const size_t cboff = (size_t)cb * LB_CBSTRIDE(capL) + (size_t)h * 256;
    ng = *reinterpret_cast<const float4*>(g + i);
    np_ = *reinterpret_cast<const float4*>(gp + i);
    nd = *reinterpret_cast<const float4*>(d + i);
#pragma unroll
    for (int q = 0; q < LB_DRW; ++q) nr[q] = *reinterpret_cast<const float4*>(rptr[q] + cboff);
  };
  const int hb0 = cb0 * 2, hb1 = cb1 * 2;
  if (hb0 < hb1) issue(hb0);
  for (int hb = hb0; hb < hb1; ++hb) {
    const int cb = hb >> 1, h = hb & 1;
    const int i = cb * LB_CW + h * 256 + lane * 4;
    float4 vg = ng;
    float4 vy = make_float4(ng.x - np_.x, ng.y - np_.y, ng.z - np_.z, ng.w - np_.w);
    float4 vs = make_float4(nd.x * t, nd.y * t, nd.z * t, nd.w * t);
    float4 rv[LB_DRW];
#pragma unroll
    for (int q = 0; q < LB_DRW; ++q) rv[q] = nr[q];
    if (hb + 1 < hb1) issue(hb + 1);
    // Entries past n belong to whatever problem used the work vectors before (a larger one leaves its gradient
    // there) and must reach neither the dot products nor the stored pair.
#define LB_MASK(c, k_) { const bool in_ = i + k_ < n; vg.c = in_ ? vg.c : 0.f; vy.c = in_ ? vy.c : 0.f; vs.c = in_ ? vs.c : 0.f; }
    LB_MASK(x, 0) LB_MASK(y, 1) LB_MASK(z, 2) LB_MASK(w, 3)
#undef LB_MASK
    if (special) {  // wave-uniform: the store of the new pair (before the shared range is masked: the history keeps it)
      const size_t o = (size_t)cb * LB_CBSTRIDE(capL) + (size_t)cand * LB_CW + h * 256 + lane * 4;
      *reinterpret_cast<float4*>(Y + o) = vy;
      *reinterpret_cast<float4*>(S + o) = vs;
    }
    // shared-betas solves (uuo_lbfgs_solve_shared): the replicated shape entries belong to rank 0's partial sums only, so
    // the other ranks drop [skip_lo, skip_hi) from every dot product (kernel-uniform: an empty range everywhere else)
    if (skip_hi > skip_lo && i < skip_hi && i + 4 > skip_lo) {
#define LB_SKIP(c, k_) { const bool out_ = i + k_ >= skip_lo && i + k_ < skip_hi; vg.c = out_ ? 0.f : vg.c; vy.c = out_ ? 0.f : vy.c; vs.c = out_ ? 0.f : vs.c; }
      LB_SKIP(x, 0) LB_SKIP(y, 1) LB_SKIP(z, 2) LB_SKIP(w, 3)
#undef LB_SKIP
    }
#pragma unroll
    for (int q = 0; q < LB_DRW; ++q) {
      const float4 r = rv[q];
#define LB_ACC(c)                                      \
      acc[q][0] += (double)r.c * (double)vy.c;         \
      acc[q][1] += (double)r.c * (double)vs.c;         \
      acc[q][2] += (double)r.c * (double)vg.c;
      LB_ACC(x) LB_ACC(y) LB_ACC(z) LB_ACC(w)
#undef LB_ACC
    }
    if (special) {  // wave-uniform: the new pair's own rows and g
#define LB_SP(c)                                                                                         \
      {                                                                                                  \
        const double y_ = (double)vy.c, s_ = (double)vs.c, g_ = (double)vg.c;                            \
        sp[0][0] += s_ * y_; sp[0][1] += s_ * s_; sp[0][2] += s_ * g_;                                   \
        sp[1][0] += y_ * y_; sp[1][1] += y_ * s_; sp[1][2] += y_ * g_;                                   \
        sp[2][0] += g_ * y_; sp[2][1] += g_ * s_; sp[2][2] += g_ * g_;                                   \
      }
      LB_SP(x) LB_SP(y) LB_SP(z) LB_SP(w)
#undef LB_SP
    }
  }
#pragma unroll
  for (int q = 0; q < LB_DRW; ++q) {
    const double a0 = wave_sum_d_fast(acc[q][0]), a1 = wave_sum_d_fast(acc[q][1]), a2 = wave_sum_d_fast(acc[q][2]);
    if (lane == 0 && rout[q] >= 0) {
      double* o = part + ((size_t)grp * LB_ROWS + rout[q]) * 3;
      o[0] = a0; o[1] = a1; o[2] = a2;
    }
  }
  if (special) {
    const int orow[3] = {cand, LB_MAXH + cand, 2 * LB_MAXH};  // s_new, y_new, g
#pragma unroll
    for (int a_ = 0; a_ < 3; ++a_) {
      const double a0 = wave_sum_d_fast(sp[a_][0]), a1 = wave_sum_d_fast(sp[a_][1]), a2 = wave_sum_d_fast(sp[a_][2]);
      if (lane == 0) {
        double* o = part + ((size_t)grp * LB_ROWS + orow[a_]) * 3;
        o[0] = a0; o[1] = a1; o[2] = a2;
      }
    }
  }
}

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
__global__ __launch_bounds__(256) void k_lb_dots(LbDotsArgs a) {
  lb_dots_body(a.n, a.cap, a.capL, a.head, a.count, a.cand, a.S, a.Y, a.g, a.gp, a.d, a.t, a.ncb, a.gcb, a.part, a.skip_lo,
               a.skip_hi);
}
__global__ __launch_bounds__(256) void k_lb_dots_b(const LbDotsArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbDotsArgs, batch)
  lb_dots_body(a.n, a.cap, a.capL, a.head, a.count, a.cand, a.S, a.Y, a.g, a.gp, a.d, a.t, a.ncb, a.gcb, a.part, a.skip_lo,
               a.skip_hi);
}

// Shared-betas solves: the chunk sums of the new Gram rows of THIS rank (what k_lb_small_inv's first phase computes), written
// to pinned host memory followed by a sequence word; the host gathers the rows of all ranks, adds them in rank order and
// hands the totals back to k_lb_small_inv (rd_in), so every rank computes identical direction coefficients.
__global__ __launch_bounds__(512) void k_lb_rows(int nchunks, int cap, int cand, const double* __restrict__ part,
                                                  const LbDev* __restrict__ st, double* __restrict__ host_rows,
                                                  unsigned long long seq) {
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int head = st->head, count = st->count;
  for (int e = tid; e < LB_ROWS * 3; e += 512) {
    const int row = e / 3;
    const int slot = (row < LB_MAXH) ? row : row - LB_MAXH;
    bool active = (row == 2 * LB_MAXH);
    if (!active && slot < cap) {
      const int rel = (slot - head + cap) % cap;
      active = (rel < count) || (slot == cand);
    }
    double acc = 0.0;
    if (active) {
      double v[LB_MAXCHUNK];
#pragma unroll
      for (int c = 0; c < LB_MAXCHUNK; ++c) v[c] = (c < nchunks) ? part[(size_t)c * LB_ROWS * 3 + e] : 0.0;
#pragma unroll
      for (int c = 0; c < LB_MAXCHUNK; ++c) acc += v[c];
    }
    host_rows[e] = acc;
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_rows + LB_ROWS * 3), seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// One block: reduce the dot partials, update the Gram matrices and the ring, then run the two-loop recursion of
// torch/optim/lbfgs.py:396-441 in coefficient space and write the coefficients of
//   d = cg g + sum_j cy_j y_j + cs_j s_j .
// Both loops are triangular recurrences over U = upper triangle of S.Y^T (logical order, oldest first):
//   loop 1 (i = k-1..0):  al_i = (-s_i.g - sum_{j>i} al_j U_ij) / U_ii
//   loop 2 (i = 0..k-1):  cs_i = al_i - (cg y_i.g + (YY cy)_i + sum_{j<i} cs_j U_ji) / U_ii ,  cy = -Hdiag al
// U (<= 43 KB of fp64) is staged in LDS so each of the 2k dependent steps costs an LDS read + a wave reduction
// instead of an L2 round trip; YY cy has no dependency chain and is a parallel mat-vec over all 256 threads.
#define LB_TRI (LB_MAXH * (LB_MAXH + 1) / 2)
#define LB_RB 8  // Gram rows fetched per wave pass
#define LB_YR ((LB_MAXH + 2) / 3)  // rows of Y.Y^T per helper wave
__device__ __forceinline__ double bcast_lane_d(double v, int src_lane) {  // src_lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int tri_index(int i, int j, int k) {  // j >= i, row-major packed upper triangle of k x k
  return i * k - (i * (i - 1)) / 2 + (j - i);
}

__global__ __launch_bounds__(256) void k_lb_small_ref(int nchunks, int cap, int hist, int cand,
                                                   const double* __restrict__ part, LbDev* __restrict__ st, int stop) {
  __builtin_amdgcn_s_setprio(3);  // latency-bound kernel: do not queue behind co-resident MFMA waves
  __shared__ double U[LB_TRI];
  __shared__ double Sg[LB_MAXH], Yg[LB_MAXH], al[LB_MAXH], cs_s[LB_MAXH], cy_s[LB_MAXH], wv[LB_MAXH];
  __shared__ double rd[LB_ROWS * 3];
  __shared__ int slot_of[LB_MAXH];
  __shared__ double wpart[3][LB_MAXH + 24];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int head = st->head, count = st->count;
  const int nact = count + 1;
  // ---- reduce the partial dots of the active rows (fixed chunk order -> deterministic)
  for (int e = tid; e < LB_ROWS * 3; e += 256) {
    const int row = e / 3;
    const int slot = (row < LB_MAXH) ? row : row - LB_MAXH;
    bool active = (row == 2 * LB_MAXH);
    if (!active && slot < cap) {
      const int rel = (slot - head + cap) % cap;
      active = (rel < count) || (slot == cand);
    }
    double acc = 0.0;
    if (active) {
      double v[LB_MAXCHUNK];
#pragma unroll
      for (int c = 0; c < LB_MAXCHUNK; ++c) v[c] = (c < nchunks) ? part[(size_t)c * LB_ROWS * 3 + e] : 0.0;
#pragma unroll
      for (int c = 0; c < LB_MAXCHUNK; ++c) acc += v[c];
    }
    rd[e] = acc;
  }
  __syncthreads();
  // ---- candidate row / column of the Gram matrices
  const double ys = rd[cand * 3 + 0];              // s_new . y_new
  const double yy = rd[(LB_MAXH + cand) * 3 + 0];  // y_new . y_new
  for (int r = tid; r < nact; r += 256) {
    const int slot = (r < count) ? (head + r) % cap : cand;
    st->SY[slot * LB_MAXH + cand] = rd[slot * 3 + 0];              // s_slot . y_new
    st->SY[cand * LB_MAXH + slot] = rd[(LB_MAXH + slot) * 3 + 1];  // s_new . y_slot
    st->YY[slot * LB_MAXH + cand] = rd[(LB_MAXH + slot) * 3 + 0];
    st->YY[cand * LB_MAXH + slot] = rd[(LB_MAXH + slot) * 3 + 0];
    Sg[slot] = rd[slot * 3 + 2];
    Yg[slot] = rd[(LB_MAXH + slot) * 3 + 2];
  }
  __threadfence_block();
  __syncthreads();
  const bool accept = ys > 1e-10;
  double Hdiag = st->Hdiag;
  if (accept) {
    if (count == hist)
      head = (head + 1) % cap;  // drop the oldest; the candidate slot becomes the newest
    else
      count += 1;
    Hdiag = ys / yy;
  }
  const int k = count;
  for (int j = tid; j < k; j += 256) slot_of[j] = (head + j) % cap;
  __syncthreads();
  if (stop == 1) return;
  // ---- stage U in LDS (logical order)
  for (int i0 = wave * LB_RB; i0 < k; i0 += 4 * LB_RB) {  // LB_RB rows per wave pass, all loads in flight at once
    double v[LB_RB][2];
#pragma unroll
    for (int r = 0; r < LB_RB; ++r) {
      const int i = i0 + r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        v[r][h] = (i < k && j >= i && j < k) ? st->SY[slot_of[i] * LB_MAXH + slot_of[j]] : 0.0;
      }
    }
#pragma unroll
    for (int r = 0; r < LB_RB; ++r) {
      const int i = i0 + r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        if (i < k && j >= i && j < k) U[tri_index(i, j, k)] = v[r][h];
      }
    }
  }
  __syncthreads();
  if (stop == 2) return;
  // Both recurrences are blocked (16 x 16) right-looking triangular solves on wave 0.  Logical index j lives on
  // lane j & 63 (two rows per lane: j0 = lane, j1 = lane + 64).  Inside a diagonal block the owning lane finishes
  // step i and broadcasts the value with v_readlane while the block's entries of U sit in registers; the update of
  // the rows outside the block is a 16-term FMA per lane whose LDS reads are issued together.  So the dependent
  // chain never waits on LDS (one LDS latency per block of 16 steps instead of one per step).
  // Meanwhile waves 1..3 fetch their rows of Y.Y^T into registers for the mat-vec between the two loops.
  double rinv0 = 0.0, rinv1 = 0.0, a0 = 0.0, a1 = 0.0;
  const int j0 = lane, j1 = lane + 64;
  const int nblk = (k + 15) >> 4;
  double yv0[LB_YR], yv1[LB_YR];  // waves 1..3: rows i = (wave-1) + 3 r of YY, columns j0 / j1
  if (wave > 0) {
    const int sl0 = (j0 < k) ? slot_of[j0] : 0, sl1 = (j1 < k) ? slot_of[j1] : 0;
#pragma unroll
    for (int r = 0; r < LB_YR; ++r) {
      const int i = (wave - 1) + 3 * r;
      const int si = (i < k) ? slot_of[i] : 0;
      yv0[r] = (i < k && j0 < k) ? st->YY[si * LB_MAXH + sl0] : 0.0;
      yv1[r] = (i < k && j1 < k) ? st->YY[si * LB_MAXH + sl1] : 0.0;
    }
  } else {
    if (j0 < k) rinv0 = 1.0 / U[tri_index(j0, j0, k)];
    if (j1 < k) rinv1 = 1.0 / U[tri_index(j1, j1, k)];
    const double sg0 = (j0 < k) ? Sg[slot_of[j0]] : 0.0, sg1 = (j1 < k) ? Sg[slot_of[j1]] : 0.0;
    double r0 = 0.0, r1 = 0.0;
    // ---- loop 1 (newest -> oldest):  al_i = (-s_i.g - sum_{m>i} al_m U_im) / U_ii
    for (int b = nblk - 1; b >= 0; --b) {
      const int lo = b << 4, hi = min(lo + 16, k);
      const bool inhi = lo >= 64;
      const int jr = inhi ? j1 : j0;  // this lane's row in the set that contains the block
      double ublk[16], ablk[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = lo + t;
        ublk[t] = (jr >= lo && jr < i && i < hi) ? U[tri_index(jr, i, k)] : 0.0;
      }
#pragma unroll
      for (int t = 15; t >= 0; --t) {
        const int i = lo + t;
        ablk[t] = 0.0;
        if (i < hi) {  // wave-uniform
          const double cand = inhi ? (-sg1 - r1) * rinv1 : (-sg0 - r0) * rinv0;
          const double ai = bcast_lane_d(cand, i & 63);
          ablk[t] = ai;
          if (lane == (i & 63)) {
            if (inhi) a1 = ai; else a0 = ai;
          }
          if (inhi) r1 = fma(ai, ublk[t], r1); else r0 = fma(ai, ublk[t], r0);
        }
      }
      // rows above the block
      if (lo > 0) {
        double un[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) un[t] = (j0 < lo && lo + t < hi) ? U[tri_index(j0, lo + t, k)] : 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t) r0 = fma(ablk[t], un[t], r0);
        if (lo > 64) {
#pragma unroll
          for (int t = 0; t < 16; ++t) un[t] = (j1 < lo && lo + t < hi) ? U[tri_index(j1, lo + t, k)] : 0.0;
#pragma unroll
          for (int t = 0; t < 16; ++t) r1 = fma(ablk[t], un[t], r1);
        }
      }
    }
    if (j0 < k) al[j0] = a0;
    if (j1 < k) al[j1] = a1;
  }
  __syncthreads();
  if (stop == 3) return;
  const double cg = -Hdiag;
  for (int j = tid; j < k; j += 256) cy_s[j] = -Hdiag * al[j];
  __syncthreads();
  // ---- w = YY cy: YY is symmetric, so lane j accumulates sum_i YY[i][j] cy_i over the rows its wave fetched
  if (wave > 0) {
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int r = 0; r < LB_YR; ++r) {
      const int i = (wave - 1) + 3 * r;
      const double c = (i < k) ? cy_s[i] : 0.0;
      acc0 = fma(yv0[r], c, acc0);
      acc1 = fma(yv1[r], c, acc1);
    }
    wpart[wave - 1][j0] = acc0;
    if (j1 < LB_MAXH) wpart[wave - 1][j1] = acc1;
  }
  __syncthreads();
  for (int j = tid; j < k; j += 256) wv[j] = (wpart[0][j] + wpart[1][j]) + wpart[2][j];
  __syncthreads();
  if (stop == 4) return;
  if (wave == 0) {
    const double b0 = (j0 < k) ? cg * Yg[slot_of[j0]] + wv[j0] : 0.0, b1 = (j1 < k) ? cg * Yg[slot_of[j1]] + wv[j1] : 0.0;
    double q0 = 0.0, q1 = 0.0, c0 = 0.0, c1 = 0.0;
    // ---- loop 2 (oldest -> newest):  cs_i = al_i - (cg y_i.g + (YY cy)_i + sum_{m<i} cs_m U_mi) / U_ii
    for (int b = 0; b < nblk; ++b) {
      const int lo = b << 4, hi = min(lo + 16, k);
      const bool inhi = lo >= 64;
      const int jr = inhi ? j1 : j0;
      double ublk[16], cblk[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = lo + t;
        ublk[t] = (jr > i && jr < hi) ? U[tri_index(i, jr, k)] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = lo + t;
        cblk[t] = 0.0;
        if (i < hi) {  // wave-uniform
          const double cand = inhi ? a1 - (b1 + q1) * rinv1 : a0 - (b0 + q0) * rinv0;
          const double ci = bcast_lane_d(cand, i & 63);
          cblk[t] = ci;
          if (lane == (i & 63)) {
            if (inhi) c1 = ci; else c0 = ci;
          }
          if (inhi) q1 = fma(ci, ublk[t], q1); else q0 = fma(ci, ublk[t], q0);
        }
      }
      // rows below the block
      if (hi < k) {
        double un[16];
        if (hi <= 64) {
#pragma unroll
          for (int t = 0; t < 16; ++t) un[t] = (j0 >= hi && j0 < k && lo + t < hi) ? U[tri_index(lo + t, j0, k)] : 0.0;
#pragma unroll
          for (int t = 0; t < 16; ++t) q0 = fma(cblk[t], un[t], q0);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) un[t] = (j1 >= hi && j1 < k && lo + t < hi) ? U[tri_index(lo + t, j1, k)] : 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t) q1 = fma(cblk[t], un[t], q1);
      }
    }
    if (j0 < k) cs_s[j0] = c0;
    if (j1 < k) cs_s[j1] = c1;
  }
  __syncthreads();
  // ---- publish: coefficients by slot, g.d from the Gram data
  double gpart = 0.0;
  for (int j = tid; j < k; j += 256) {
    const int sj = slot_of[j];
    gpart += cy_s[j] * Yg[sj] + cs_s[j] * Sg[sj];
    st->cy[sj] = cy_s[j];
    st->cs[sj] = cs_s[j];
  }
  gpart = wave_sum_d(gpart);
  if (lane == 0) wv[LB_MAXH - 4 + wave] = gpart;  // k <= hist <= LB_MAXH - 4 leaves these free
  __syncthreads();
  if (tid == 0) {
    const double gsum = (wv[LB_MAXH - 4] + wv[LB_MAXH - 3]) + (wv[LB_MAXH - 2] + wv[LB_MAXH - 1]);
    const double gg = rd[(2 * LB_MAXH) * 3 + 2];
    st->cg = cg;
    st->Hdiag = Hdiag;
    st->head = head;
    st->count = count;
    st->dmax_bits = 0u;
    LbOut* o = reinterpret_cast<LbOut*>(st->out);
    o->gtd_dir = cg * gg + gsum;
    o->accepted = accept ? 1.0 : 0.0;
    o->ys = ys;
  }
}

// Same computation as k_lb_small_ref with the dependent chains cut from 2k steps to 2 * ceil(k / 16) block steps:
// the inverses of the 16 x 16 diagonal blocks of U are computed first (every column of every block is an independent
// back-substitution of at most 16 steps: 112 lanes at once), after which a block of the triangular solves is a
// 16 x 16 mat-vec (values fetched with v_readlane inside the block's DPP row) followed by the same right-looking
// update as before.  8 waves: wave 0 runs the two solves, waves 1-2 invert the blocks, waves 3-5 hold Y.Y^T.
#define LB_NB ((LB_MAXH + 15) / 16)
#define LB_US (LB_MAXH + 1)  // padded row stride of the LDS copy of S.Y^T (bank spread)
__global__ __launch_bounds__(512) void k_lb_small(int nchunks, int cap, int hist, int cand,
                                                   const double* __restrict__ part, LbDev* __restrict__ st, int stop) {
  __builtin_amdgcn_s_setprio(3);  // latency-bound kernel: do not queue behind co-resident MFMA waves
  __shared__ double Us[LB_MAXH * LB_US];  // S.Y^T by SLOT (one contiguous copy of the device matrix); U(i,j) logical
                                          // = Us[slot(i)][slot(j)] for i <= j
  __shared__ double Xl[LB_NB][16][16];    // inverses of the diagonal blocks of U (upper triangular, [row][col])
  __shared__ double rinvL[LB_MAXH + 8];
  __shared__ double bl[2][16];  // block values handed from the 16 owning lanes to the whole wave (wave 0 only)
  __shared__ double Sg[LB_MAXH], Yg[LB_MAXH], al[LB_MAXH], cs_s[LB_MAXH], cy_s[LB_MAXH], wv[LB_MAXH];
  __shared__ double rd[LB_ROWS * 3];
  __shared__ int slot_of[LB_MAXH + 24];
  __shared__ double wpart[3][LB_MAXH + 24];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int head = st->head, count = st->count;
  const int nact = count + 1;
  // ---- S.Y^T as it stands before this iteration's column: issued first, consumed after the partial sums
  constexpr int NSY = LB_MAXH * LB_MAXH / 2;               // double2 elements
  constexpr int PSY = (NSY + 511) / 512;                   // per thread (11)
  double2 sycopy[PSY];
  {
    const double2* src = reinterpret_cast<const double2*>(st->SY);
#pragma unroll
    for (int r = 0; r < PSY; ++r) {
      const int e = tid + 512 * r;
      sycopy[r] = (e < NSY) ? src[e] : make_double2(0.0, 0.0);
    }
  }
  // ---- reduce the partial dots of the active rows (fixed chunk order -> deterministic)
  for (int e = tid; e < LB_ROWS * 3; e += 512) {
    const int row = e / 3;
    const int slot = (row < LB_MAXH) ? row : row - LB_MAXH;
    bool active = (row == 2 * LB_MAXH);
    if (!active && slot < cap) {
      const int rel = (slot - head + cap) % cap;
      active = (rel < count) || (slot == cand);
    }
    double acc = 0.0;
    if (active) {
      double v[LB_MAXCHUNK];
#pragma unroll
      for (int c = 0; c < LB_MAXCHUNK; ++c) v[c] = (c < nchunks) ? part[(size_t)c * LB_ROWS * 3 + e] : 0.0;
#pragma unroll
      for (int c = 0; c < LB_MAXCHUNK; ++c) acc += v[c];
    }
    rd[e] = acc;
  }
#pragma unroll
  for (int r = 0; r < PSY; ++r) {
    const int e = tid + 512 * r;
    if (e < NSY) {
      const int row = (2 * e) / LB_MAXH, col = (2 * e) - row * LB_MAXH;  // LB_MAXH is even: pairs never straddle rows
      Us[row * LB_US + col] = sycopy[r].x;
      Us[row * LB_US + col + 1] = sycopy[r].y;
    }
  }
  __syncthreads();
  // ---- candidate row / column of the Gram matrices (device copy for the next iterations, LDS copy for this one)
  const double ys = rd[cand * 3 + 0];              // s_new . y_new
  const double yy = rd[(LB_MAXH + cand) * 3 + 0];  // y_new . y_new
  for (int r = tid; r < nact; r += 512) {
    const int slot = (r < count) ? (head + r) % cap : cand;
    const double sy_col = rd[slot * 3 + 0];              // s_slot . y_new
    const double sy_row = rd[(LB_MAXH + slot) * 3 + 1];  // s_new . y_slot
    st->SY[slot * LB_MAXH + cand] = sy_col;
    st->SY[cand * LB_MAXH + slot] = sy_row;
    Us[slot * LB_US + cand] = sy_col;
    Us[cand * LB_US + slot] = sy_row;
    st->YY[slot * LB_MAXH + cand] = rd[(LB_MAXH + slot) * 3 + 0];
    st->YY[cand * LB_MAXH + slot] = rd[(LB_MAXH + slot) * 3 + 0];
    Sg[slot] = rd[slot * 3 + 2];
    Yg[slot] = rd[(LB_MAXH + slot) * 3 + 2];
  }
  __threadfence_block();
  const bool accept = ys > 1e-10;
  double Hdiag = st->Hdiag;
  if (accept) {
    if (count == hist)
      head = (head + 1) % cap;  // drop the oldest; the candidate slot becomes the newest
    else
      count += 1;
    Hdiag = ys / yy;
  }
  const int k = count;
  // logical index -> slot, arithmetically (head + i < 2 cap): no LDS look-up on the solves' dependent chains
  auto slotf = [&](int i) { const int v = head + i; return (v >= cap) ? v - cap : v; };
  for (int j = tid; j < LB_MAXH + 24; j += 512) slot_of[j] = (j < k) ? slotf(j) : 0;
  __syncthreads();
  if (stop == 1) return;
  // ---- waves 3..5 fetch their rows of Y.Y^T into registers for the mat-vec between the two solves
  const int j0 = lane, j1 = lane + 64;
  const int sl0 = (j0 < k) ? slotf(j0) : 0, sl1 = (j1 < k) ? slotf(j1) : 0;
  double yv0[LB_YR], yv1[LB_YR];  // waves 3..5: rows i = (wave-3) + 3 r of YY, columns j0 / j1
  if (wave >= 3 && wave <= 5) {
#pragma unroll
    for (int r = 0; r < LB_YR; ++r) {
      const int i = (wave - 3) + 3 * r;
      const int si = (i < k) ? slot_of[i] : 0;
      yv0[r] = (i < k && j0 < k) ? st->YY[si * LB_MAXH + sl0] : 0.0;
      yv1[r] = (i < k && j1 < k) ? st->YY[si * LB_MAXH + sl1] : 0.0;
    }
  }
  if (stop == 2) return;
  if (tid < k) rinvL[tid] = 1.0 / Us[slotf(tid) * (LB_US + 1)];
  __syncthreads();
  const int nblk = (k + 15) >> 4;
  // ---- inverses of the diagonal blocks: lane (b, c) solves T x = e_c by back-substitution, T = U[lo:hi, lo:hi].
  // Rows r > c and columns beyond the block contribute zeros (masked products), so the code has no divergent branch.
  if (wave >= 1 && wave <= 2) {
    const int idx = (wave - 1) * 64 + lane;
    const int b = idx >> 4, c = idx & 15;
    if (b < nblk) {
      const int lo = b << 4;
      const int nb = min(16, k - lo);
      const bool colok = c < nb;
      double x[16];
#pragma unroll
      for (int r = 15; r >= 0; --r) {
        // x_r = (delta_rc - sum_{m > r} T_rm x_m) / T_rr, kept only for r <= c
        const int srow = slotf(min(lo + r, k - 1)) * LB_US;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int m = r + 1; m < 16; m += 2) {
          acc0 = fma(Us[srow + slotf(min(lo + m, k - 1))], x[m], acc0);
          if (m + 1 < 16) acc1 = fma(Us[srow + slotf(min(lo + m + 1, k - 1))], x[m + 1], acc1);
        }
        const double rhs = ((r == c) ? 1.0 : 0.0) - (acc0 + acc1);
        x[r] = (colok && r <= c && r < nb) ? rhs * rinvL[lo + r] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) Xl[b][r][c] = x[r];
    }
  }
  __syncthreads();
  if (stop == 3) return;
  double a0 = 0.0, a1 = 0.0;
  if (wave == 0) {
    const double sg0 = (j0 < k) ? Sg[sl0] : 0.0, sg1 = (j1 < k) ? Sg[sl1] : 0.0;
    double r0 = 0.0, r1 = 0.0;
    // ---- loop 1 (newest -> oldest):  U al = -S.g  (upper triangular), blocks from the last to the first
    for (int b = nblk - 1; b >= 0; --b) {
      const int lo = b << 4, hi = min(lo + 16, k);
      const bool inhi = lo >= 64;
      const int base = lo & 63;                 // first lane of the block's DPP row
      const int t = (lane - base) & 15;         // this lane's row inside the block (meaningful for block lanes)
      const bool mine = lane >= base && lane < base + 16;
      double xrow[16], un0[16], un1[16];
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) {
        xrow[s_] = Xl[b][t][s_];                // row t of the block inverse (t is arbitrary but in range off the block)
        const int sc = slotf(min(lo + s_, k - 1));  // wave-uniform
        const double u0 = Us[sl0 * LB_US + sc], u1 = Us[sl1 * LB_US + sc];  // unconditional reads, masked values
        un0[s_] = (j0 < lo && lo + s_ < hi) ? u0 : 0.0;                 // rows above the block
        un1[s_] = (lo > 64 && j1 < lo && lo + s_ < hi) ? u1 : 0.0;
      }
      const double rhs = inhi ? (-sg1 - r1) : (-sg0 - r0);
      // the block's 16 right-hand sides go through LDS (one write, broadcast reads): LDS operations of one wave
      // complete in order, so no barrier is needed, only a compiler fence
      if (mine) bl[0][t] = rhs;
      __builtin_amdgcn_wave_barrier();
      double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;  // four partial sums: short dependent chains
#pragma unroll
      for (int s_ = 0; s_ < 16; s_ += 4) {
        m0 = fma(xrow[s_], bl[0][s_], m0);
        m1 = fma(xrow[s_ + 1], bl[0][s_ + 1], m1);
        m2 = fma(xrow[s_ + 2], bl[0][s_ + 2], m2);
        m3 = fma(xrow[s_ + 3], bl[0][s_ + 3], m3);
      }
      const double mya = (m0 + m1) + (m2 + m3);
      if (mine) {
        if (inhi) a1 = mya; else a0 = mya;
      }
      if (lo > 0) {  // right-looking update of the rows above with the block's al values
        if (mine) bl[1][t] = mya;
        __builtin_amdgcn_wave_barrier();
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < 16; s_ += 2) {
          const double ab0 = bl[1][s_], ab1 = bl[1][s_ + 1];
          p0 = fma(ab0, un0[s_], p0);
          p1 = fma(ab1, un0[s_ + 1], p1);
          p2 = fma(ab0, un1[s_], p2);
          p3 = fma(ab1, un1[s_ + 1], p3);
        }
        r0 += p0 + p1;
        r1 += p2 + p3;
      }
    }
    if (j0 < k) al[j0] = a0;
    if (j1 < k) al[j1] = a1;
  }
  __syncthreads();
  if (stop == 4) return;
  const double cg = -Hdiag;
  for (int j = tid; j < k; j += 512) cy_s[j] = -Hdiag * al[j];
  __syncthreads();
  // ---- w = YY cy: YY is symmetric, so lane j accumulates sum_i YY[i][j] cy_i over the rows its wave fetched
  if (wave >= 3 && wave <= 5) {
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int r = 0; r < LB_YR; ++r) {
      const int i = (wave - 3) + 3 * r;
      const double c = (i < k) ? cy_s[i] : 0.0;
      acc0 = fma(yv0[r], c, acc0);
      acc1 = fma(yv1[r], c, acc1);
    }
    wpart[wave - 3][j0] = acc0;
    if (j1 < LB_MAXH) wpart[wave - 3][j1] = acc1;
  }
  __syncthreads();
  for (int j = tid; j < k; j += 512) wv[j] = (wpart[0][j] + wpart[1][j]) + wpart[2][j];
  __syncthreads();
  if (stop == 5) return;
  if (wave == 0) {
    // ---- loop 2 (oldest -> newest):  U^T cs = D al - (cg Y.g + YY cy)  (lower triangular), blocks first to last
    const double v0 = (j0 < k) ? a0 * Us[sl0 * LB_US + sl0] - (cg * Yg[sl0] + wv[j0]) : 0.0;
    const double v1 = (j1 < k) ? a1 * Us[sl1 * LB_US + sl1] - (cg * Yg[sl1] + wv[j1]) : 0.0;
    double q0 = 0.0, q1 = 0.0, c0 = 0.0, c1 = 0.0;
    for (int b = 0; b < nblk; ++b) {
      const int lo = b << 4, hi = min(lo + 16, k);
      const bool inhi = lo >= 64;
      const int base = lo & 63;
      const int t = (lane - base) & 15;
      const bool mine = lane >= base && lane < base + 16;
      double xcol[16], un0[16], un1[16];
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) {
        xcol[s_] = Xl[b][s_][t];               // column t of the block inverse = row t of its transpose
        const int sr = slotf(min(lo + s_, k - 1)) * LB_US;  // wave-uniform
        const double u0 = Us[sr + sl0], u1 = Us[sr + sl1];
        un0[s_] = (hi <= 64 && j0 >= hi && j0 < k && lo + s_ < hi) ? u0 : 0.0;   // rows below the block
        un1[s_] = (j1 >= hi && j1 < k && lo + s_ < hi) ? u1 : 0.0;
      }
      const double rhs = inhi ? (v1 - q1) : (v0 - q0);
      if (mine) bl[0][t] = rhs;
      __builtin_amdgcn_wave_barrier();
      double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
#pragma unroll
      for (int s_ = 0; s_ < 16; s_ += 4) {
        m0 = fma(xcol[s_], bl[0][s_], m0);
        m1 = fma(xcol[s_ + 1], bl[0][s_ + 1], m1);
        m2 = fma(xcol[s_ + 2], bl[0][s_ + 2], m2);
        m3 = fma(xcol[s_ + 3], bl[0][s_ + 3], m3);
      }
      const double myc = (m0 + m1) + (m2 + m3);
      if (mine) {
        if (inhi) c1 = myc; else c0 = myc;
      }
      if (hi < k) {
        if (mine) bl[1][t] = myc;
        __builtin_amdgcn_wave_barrier();
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < 16; s_ += 2) {
          const double cb0 = bl[1][s_], cb1 = bl[1][s_ + 1];
          p0 = fma(cb0, un0[s_], p0);
          p1 = fma(cb1, un0[s_ + 1], p1);
          p2 = fma(cb0, un1[s_], p2);
          p3 = fma(cb1, un1[s_ + 1], p3);
        }
        q0 += p0 + p1;
        q1 += p2 + p3;
      }
    }
    if (j0 < k) cs_s[j0] = c0;
    if (j1 < k) cs_s[j1] = c1;
  }
  __syncthreads();
  // ---- publish: coefficients by slot, g.d from the Gram data
  double gpart = 0.0;
  for (int j = tid; j < k; j += 512) {
    const int sj = slot_of[j];
    gpart += cy_s[j] * Yg[sj] + cs_s[j] * Sg[sj];
    st->cy[sj] = cy_s[j];
    st->cs[sj] = cs_s[j];
  }
  gpart = wave_sum_d(gpart);
  if (lane == 0) wpart[0][LB_MAXH + wave] = gpart;  // free tail of the scratch rows
  __syncthreads();
  if (tid == 0) {
    double gsum = 0.0;
    for (int w_ = 0; w_ < 8; ++w_) gsum += wpart[0][LB_MAXH + w_];
    const double gg = rd[(2 * LB_MAXH) * 3 + 2];
    st->cg = cg;
    st->Hdiag = Hdiag;
    st->head = head;
    st->count = count;
    st->dmax_bits = 0u;
    LbOut* o = reinterpret_cast<LbOut*>(st->out);
    o->gtd_dir = cg * gg + gsum;
    o->accepted = accept ? 1.0 : 0.0;
    o->ys = ys;
  }
}

// Third formulation of the same step: no triangular SOLVE at all.  The inverse W of U (upper triangle of S.Y^T over
// the history window, logical order oldest first) is kept on the device from one iteration to the next:
//   * accepting the pair (s, y) appends the column u = S_old.y and the diagonal rho = s.y to U; the inverse gains the
//     column -W u / rho and the diagonal 1 / rho (column-by-column inversion of a triangular matrix, Higham,
//     "Accuracy and Stability of Numerical Algorithms", method 2: |W U - I| <= c eps |W| |U|) -- one mat-vec;
//   * dropping the oldest pair removes the first row and column of U, and the inverse of a trailing block of a
//     triangular matrix is the trailing block of its inverse -- nothing to compute, the slot simply leaves the window.
// Both loops of the recursion are then mat-vecs, al = W (-S.g) and cs = W^T (D al - cg Y.g - YY cy): the 2 * 7
// dependent block steps of k_lb_small (~19 us at a full history) become three 104-term mat-vecs spread over 512 lanes
// (4 lanes per row, fixed summation order).  W is indexed by SLOT like S.Y^T; rows are zeroed when a slot is
// (re)inserted, so entries below the logical diagonal are exact zeros and the mat-vecs need no masks.
__device__ __forceinline__ double quad_sum_d(double v) {  // sum over the 4 lanes of a quad, same order on every lane
  const double a = v + __shfl_xor(v, 1, 64);
  return a + __shfl_xor(a, 2, 64);
}
__device__ __forceinline__ void lb_small_inv_body(int nchunks, int cap, int hist, int cand,
                                                       const double* __restrict__ part, LbDev* __restrict__ st,
                                                       int stop, const double* __restrict__ rd_in = nullptr) {
  __builtin_amdgcn_s_setprio(3);  // latency-bound kernel: do not queue behind co-resident MFMA waves
  __shared__ double Ws[LB_MAXH * LB_US];  // W by slot
  __shared__ double Sg[LB_MAXH], Yg[LB_MAXH], al[LB_MAXH], cs_s[LB_MAXH], cy_s[LB_MAXH], wv[LB_MAXH], vv[LB_MAXH];
  __shared__ double ucol[LB_MAXH], udiag[LB_MAXH];
  __shared__ double rd[LB_ROWS * 3];
  __shared__ int slot_of[LB_MAXH + 24];
  __shared__ double wpart[3][LB_MAXH + 24];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int head = st->head, count = st->count;
  const int nact = count + 1;
  // ---- W as it stands before this iteration's pair, and the diagonal of U: issued first, consumed after the sums
  constexpr int NSY = LB_MAXH * LB_MAXH / 2;               // double2 elements
  constexpr int PSY = (NSY + 511) / 512;                   // per thread (11)
  double2 wcopy[PSY];
  {
    const double2* src = reinterpret_cast<const double2*>(st->W);
#pragma unroll
    for (int r = 0; r < PSY; ++r) {
      const int e = tid + 512 * r;
      wcopy[r] = (e < NSY) ? src[e] : make_double2(0.0, 0.0);
    }
  }
  const double dg = (tid < LB_MAXH) ? st->SY[tid * (LB_MAXH + 1)] : 0.0;
  // ---- reduce the partial dots of the active rows (fixed chunk order -> deterministic)
  for (int e = tid; e < LB_ROWS * 3; e += 512) {
    const int row = e / 3;
    const int slot = (row < LB_MAXH) ? row : row - LB_MAXH;
    bool active = (row == 2 * LB_MAXH);
    if (!active && slot < cap) {
      const int rel = (slot - head + cap) % cap;
      active = (rel < count) || (slot == cand);
    }
    double acc = 0.0;
    if (active && rd_in) {  // kernel-uniform: the rows summed over the ranks of a shared-betas solve (k_lb_rows + host)
      acc = rd_in[e];
    } else if (active) {
      double v[LB_MAXCHUNK];
#pragma unroll
      for (int c = 0; c < LB_MAXCHUNK; ++c) v[c] = (c < nchunks) ? part[(size_t)c * LB_ROWS * 3 + e] : 0.0;
#pragma unroll
      for (int c = 0; c < LB_MAXCHUNK; ++c) acc += v[c];
    }
    rd[e] = acc;
  }
#pragma unroll
  for (int r = 0; r < PSY; ++r) {
    const int e = tid + 512 * r;
    if (e < NSY) {
      const int row = (2 * e) / LB_MAXH, col = (2 * e) - row * LB_MAXH;  // LB_MAXH is even: pairs never straddle rows
      Ws[row * LB_US + col] = wcopy[r].x;
      Ws[row * LB_US + col + 1] = wcopy[r].y;
    }
  }
  if (tid < LB_MAXH) udiag[tid] = dg;
  __syncthreads();
  // ---- candidate row / column of the Gram matrices (device copies for the next iterations)
  const double ys = rd[cand * 3 + 0];              // s_new . y_new
  const double yy = rd[(LB_MAXH + cand) * 3 + 0];  // y_new . y_new
  for (int r = tid; r < nact; r += 512) {
    const int slot = (r < count) ? (head + r) % cap : cand;
    const double sy_col = rd[slot * 3 + 0];              // s_slot . y_new
    const double sy_row = rd[(LB_MAXH + slot) * 3 + 1];  // s_new . y_slot
    st->SY[slot * LB_MAXH + cand] = sy_col;
    st->SY[cand * LB_MAXH + slot] = sy_row;
    ucol[slot] = sy_col;
    st->YY[slot * LB_MAXH + cand] = rd[(LB_MAXH + slot) * 3 + 0];
    st->YY[cand * LB_MAXH + slot] = rd[(LB_MAXH + slot) * 3 + 0];
    Sg[slot] = rd[slot * 3 + 2];
    Yg[slot] = rd[(LB_MAXH + slot) * 3 + 2];
  }
  __threadfence_block();
  const bool accept = ys > 1e-10;
  double Hdiag = st->Hdiag;
  if (accept) {
    if (count == hist)
      head = (head + 1) % cap;  // drop the oldest; the candidate slot becomes the newest
    else
      count += 1;
    Hdiag = ys / yy;
  }
  const int k = count;
  auto slotf = [&](int i) { const int v = head + i; return (v >= cap) ? v - cap : v; };
  for (int j = tid; j < LB_MAXH + 24; j += 512) slot_of[j] = (j < k) ? slotf(j) : 0;
  __syncthreads();
  if (stop == 1) return;
  // ---- waves 3..5 fetch their rows of Y.Y^T into registers for the mat-vec between the two products
  const int j0 = lane, j1 = lane + 64;
  const int sl0 = (j0 < k) ? slotf(j0) : 0, sl1 = (j1 < k) ? slotf(j1) : 0;
  double yv0[LB_YR], yv1[LB_YR];  // waves 3..5: rows i = (wave-3) + 3 r of YY, columns j0 / j1
  if (wave >= 3 && wave <= 5) {
#pragma unroll
    for (int r = 0; r < LB_YR; ++r) {
      const int i = (wave - 3) + 3 * r;
      const int si = (i < k) ? slot_of[i] : 0;
      yv0[r] = (i < k && j0 < k) ? st->YY[si * LB_MAXH + sl0] : 0.0;
      yv1[r] = (i < k && j1 < k) ? st->YY[si * LB_MAXH + sl1] : 0.0;
    }
  }
  if (stop == 2) return;
  // quad (4 lanes) per row: row i = tid / 4 (128 >= LB_MAXH rows), lane p of the quad takes columns p, p + 4, ...
  const int qi = tid >> 2, qp = tid & 3;
  const bool qrow = qi < k;
  const int qs = qrow ? slot_of[qi] : 0;
  // ---- the accepted pair's column of W: -W_old u / rho over the rows that stay in the window, 1 / rho on the diagonal
  if (accept) {
    double t = 0.0;
    for (int j = qp; j < k - 1; j += 4) {
      const int sj = slot_of[j];
      t = fma(Ws[qs * LB_US + sj], ucol[sj], t);
    }
    t = quad_sum_d(t);
    __syncthreads();  // every read of the old W is done before the candidate's row and column are rewritten
    const double rinv = 1.0 / ys;
    if (qp == 0 && qi < k - 1) {
      const double wcol = -t * rinv;
      Ws[qs * LB_US + cand] = wcol;
      st->W[qs * LB_MAXH + cand] = wcol;
    }
    for (int c = tid; c < LB_MAXH; c += 512) {  // the candidate's row: zeros below the logical diagonal
      const double wrow = (c == cand) ? rinv : 0.0;
      Ws[cand * LB_US + c] = wrow;
      st->W[cand * LB_MAXH + c] = wrow;
    }
    if (tid == 0) udiag[cand] = ys;
  }
  __syncthreads();
  if (stop == 3) return;
  // ---- loop 1 of the recursion:  al = W (-S.g)
  {
    double t = 0.0;
    for (int j = qp; j < k; j += 4) {
      const int sj = slot_of[j];
      t = fma(Ws[qs * LB_US + sj], -Sg[sj], t);
    }
    t = quad_sum_d(t);
    if (qp == 0 && qrow) al[qi] = t;
  }
  __syncthreads();
  if (stop == 4) return;
  const double cg = -Hdiag;
  for (int j = tid; j < k; j += 512) cy_s[j] = -Hdiag * al[j];
  __syncthreads();
  // ---- w = YY cy: YY is symmetric, so lane j accumulates sum_i YY[i][j] cy_i over the rows its wave fetched
  if (wave >= 3 && wave <= 5) {
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int r = 0; r < LB_YR; ++r) {
      const int i = (wave - 3) + 3 * r;
      const double c = (i < k) ? cy_s[i] : 0.0;
      acc0 = fma(yv0[r], c, acc0);
      acc1 = fma(yv1[r], c, acc1);
    }
    wpart[wave - 3][j0] = acc0;
    if (j1 < LB_MAXH) wpart[wave - 3][j1] = acc1;
  }
  __syncthreads();
  for (int j = tid; j < k; j += 512) {
    const int sj = slot_of[j];
    vv[j] = al[j] * udiag[sj] - (cg * Yg[sj] + ((wpart[0][j] + wpart[1][j]) + wpart[2][j]));
  }
  __syncthreads();
  if (stop == 5) return;
  // ---- loop 2:  cs = W^T (D al - (cg Y.g + YY cy))
  {
    double t = 0.0;
    for (int j = qp; j < k; j += 4) t = fma(Ws[slot_of[j] * LB_US + qs], vv[j], t);
    t = quad_sum_d(t);
    if (qp == 0 && qrow) cs_s[qi] = t;
  }
  __syncthreads();
  // ---- publish: coefficients by slot, g.d from the Gram data
  double gpart = 0.0;
  for (int j = tid; j < k; j += 512) {
    const int sj = slot_of[j];
    gpart += cy_s[j] * Yg[sj] + cs_s[j] * Sg[sj];
    st->cy[sj] = cy_s[j];
    st->cs[sj] = cs_s[j];
  }
  gpart = wave_sum_d(gpart);
  if (lane == 0) wpart[0][LB_MAXH + wave] = gpart;  // free tail of the scratch rows
  __syncthreads();
  if (tid == 0) {
    double gsum = 0.0;
    for (int w_ = 0; w_ < 8; ++w_) gsum += wpart[0][LB_MAXH + w_];
    const double gg = rd[(2 * LB_MAXH) * 3 + 2];
    st->cg = cg;
    st->Hdiag = Hdiag;
    st->head = head;
    st->count = count;
    st->dmax_bits = 0u;
    LbOut* o = reinterpret_cast<LbOut*>(st->out);
    o->gtd_dir = cg * gg + gsum;
    o->accepted = accept ? 1.0 : 0.0;
    o->ys = ys;
  }
}

struct LbSmallArgs {
  UuoGridHdr h;
  int nchunks, cap, hist, cand;
  uuo_gptr<const double> part;
  uuo_gptr<LbDev> st;
  int stop;
  uuo_gptr<const double> rd_in;  // shared-betas solves: the Gram rows already summed over chunks AND ranks; null otherwise
};
__global__ __launch_bounds__(512) void k_lb_small_inv(LbSmallArgs a) {
  lb_small_inv_body(a.nchunks, a.cap, a.hist, a.cand, a.part, a.st, a.stop, a.rd_in);
}
__global__ __launch_bounds__(512) void k_lb_small_inv_b(const LbSmallArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbSmallArgs, batch)
  lb_small_inv_body(a.nchunks, a.cap, a.hist, a.cand, a.part, a.st, a.stop, a.rd_in);
}

#define LB_DQ 4  // slot ranges per 256-column strip of k_lb_direction (one wave each)
__device__ __forceinline__ void lb_direction_body(int n, int cap, int capL, const float* __restrict__ S,
                                                             const float* __restrict__ Y, const float* __restrict__ g,
                                                             LbDev* __restrict__ st, float* __restrict__ d,
                                                             const float* __restrict__ x, float t, float* __restrict__ xt,
                                                             const UuoIndexMap& map) {
  __builtin_amdgcn_s_setprio(1);  // latency-bound kernel: do not queue behind co-resident MFMA waves
  // d = cg g + sum_j cy_j y_j + cs_j s_j, max|d|, and the first line-search trial point xt = x + t d in the same pass.
  // One block per 256-column strip of the history (FOUR columns per lane: 16-byte loads -- the vector-memory pipe costs
  // ~16 cycles per wave instruction whatever its width, and 8-byte loads made this kernel bound by it).  The strip's
  // slots are one contiguous region per column block; they are split into LB_DQ consecutive ranges, one wave each, two
  // batches of 8 slots (16 loads) in flight per lane -- with one wave per strip the kernel had 8 MB of loads in flight
  // on the whole chip and streamed the 53 MB of history at 3.6 TB/s; four waves per strip quadruple that.  The ranges'
  // fp64 partial sums are added in range order by wave 0 (fixed order: deterministic).
  __shared__ double scy[LB_MAXH + 16], scs[LB_MAXH + 16];
  __shared__ int sslot[LB_MAXH + 16];
  __shared__ double spart[LB_DQ][256];
  const int k = st->count, head = st->head;
  for (int j = threadIdx.x; j < LB_MAXH + 16; j += 64 * LB_DQ) {
    const bool on = j < k;
    const int sj = on ? (head + j) % cap : 0;
    sslot[j] = sj;
    scy[j] = on ? st->cy[sj] : 0.0;
    scs[j] = on ? st->cs[sj] : 0.0;
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int cb = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int c = half * 256 + lane * 4;   // four columns inside the column block
  const int i = cb * LB_CW + c;          // work vectors are padded to whole column blocks: 16-byte accesses in bounds
  // this wave's slots: batches of 8, nbq batches per range
  const int nb = (k + 7) >> 3, nbq = (nb + LB_DQ - 1) / LB_DQ;
  const int jlo = wave * nbq * 8, jhi = min(k, jlo + nbq * 8);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  {
    const float* Sb = S + (size_t)cb * LB_CBSTRIDE(capL) + c;
    const float* Yb = Y + (size_t)cb * LB_CBSTRIDE(capL) + c;
    float4 yv[2][8], sv[2][8];
    auto issue = [&](int j0, int buf) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // zero coefficients beyond k; slot 0 is a valid row to read
        const size_t off = (size_t)sslot[j0 + u] * LB_CW;
        yv[buf][u] = *reinterpret_cast<const float4*>(Yb + off);
        sv[buf][u] = *reinterpret_cast<const float4*>(Sb + off);
      }
    };
    auto consume = [&](int j0, int buf) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double cy = scy[j0 + u], cs = scs[j0 + u];
        acc[0] += cy * (double)yv[buf][u].x + cs * (double)sv[buf][u].x;
        acc[1] += cy * (double)yv[buf][u].y + cs * (double)sv[buf][u].y;
        acc[2] += cy * (double)yv[buf][u].z + cs * (double)sv[buf][u].z;
        acc[3] += cy * (double)yv[buf][u].w + cs * (double)sv[buf][u].w;
      }
    };
    if (jlo < jhi) issue(jlo, 0);
    for (int j0 = jlo; j0 < jhi; j0 += 16) {
      if (j0 + 8 < jhi) issue(j0 + 8, 1);
      consume(j0, 0);
      if (j0 + 16 < jhi) issue(j0 + 16, 0);
      if (j0 + 8 < jhi) consume(j0 + 8, 1);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) spart[wave][lane * 4 + e] = acc[e];
  __syncthreads();
  float mx = 0.f;
  if (wave == 0 && i < n) {
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    const double cg = st->cg;
    const double g4[4] = {(double)gv.x, (double)gv.y, (double)gv.z, (double)gv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double v = cg * g4[e];
#pragma unroll
      for (int q = 0; q < LB_DQ; ++q) v += spart[q][lane * 4 + e];
      const float dd = (float)v;
      if (i + e < n) {  // x may be the caller's tensor of exactly n floats: element-wise, guarded
        d[i + e] = dd;
        const int fi = map.full(i + e);  // the iterate lives in the reference's packing (see lb_neg_body)
        xt[fi] = x[fi] + t * dd;
        mx = fmaxf(mx, fabsf(dd));
      }
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) atomicMax(&st->dmax_bits, __float_as_uint(mx));
  }
}

// -------------------------------------------------------------------------------------------------- objectives
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
static int stage_objective_init(StageObjective& obj, uuo_fit* fit, hipStream_t s, const uuo_problem_t* p, const float* d_x) {
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
__global__ __launch_bounds__(64 * LB_DQ) void k_lb_direction(LbDirArgs a) {
  lb_direction_body(a.n, a.cap, a.capL, a.S, a.Y, a.g, a.st, a.d, a.x, a.t, a.xt, a.map);
}
__global__ __launch_bounds__(64 * LB_DQ) void k_lb_direction_b(const LbDirArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbDirArgs, batch)
  lb_direction_body(a.n, a.cap, a.capL, a.S, a.Y, a.g, a.st, a.d, a.x, a.t, a.xt, a.map);
}


// test objectives for the optimiser itself (tests/test_gpu_parity.py::test_lbfgs_*): 0 = convex quadratic with a spread
// spectrum, 1 = chained Rosenbrock.  loss/grad are computed by one block (n is small in the tests).
__global__ __launch_bounds__(256) void k_test_objective(int kind, int n, const float* __restrict__ x,
                                                         float* __restrict__ loss, float* __restrict__ grad) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    if (kind == 0 || kind == 2) {
      // kind 2: well-scaled quadratic (condition 4, |g0|_1 < 1 so the first step length is lr itself): every line-search
      // decision has a healthy margin, so two fp32 implementations must agree evaluation by evaluation
      const float a = (kind == 0) ? 1.0f + 99.0f * (float)i / (float)(n > 1 ? n - 1 : 1)
                                  : 1.0f + 3.0f * (float)i / (float)(n > 1 ? n - 1 : 1);
      const float b = (kind == 0) ? sinf(0.37f * (float)i) : 1e-3f * sinf(0.37f * (float)i);
      const float r = x[i] - b;
      acc += 0.5 * (double)a * (double)r * (double)r;
      grad[i] = a * r;
    } else {
      float gi = 0.f;
      if (i + 1 < n) {
        const float t1 = x[i + 1] - x[i] * x[i];
        const float t2 = 1.f - x[i];
        acc += 100.0 * (double)t1 * (double)t1 + (double)t2 * (double)t2;
        gi += -400.f * x[i] * t1 - 2.f * t2;
      }
      if (i > 0) {
        const float t0 = x[i] - x[i - 1] * x[i - 1];
        gi += 200.f * t0;
      }
      grad[i] = gi;
    }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

struct TestObjective : Objective {
  int kind;
  int eval(hipStream_t s, const float* x, float* loss_dev, float* grad, const float*, double*,
           const UuoEvalReport*) override {
    hipLaunchKernelGGL(k_test_objective, dim3(1), dim3(256), 0, s, kind, n, x, loss_dev, grad);
    UUO_HIP_CHECK(hipGetLastError());
    return 0;
  }
};

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

static int lbws_destroy(LbWs* w) {
  if (!w) return 0;
  if (w->slab) (void)hipFree(w->slab);
  if (w->h_out) (void)hipHostFree(w->h_out);
  if (w->h_rows) (void)hipHostFree(w->h_rows);
  if (w->h_joint) (void)hipHostFree(w->h_joint);
  if (w->h_gb) (void)hipHostFree(w->h_gb);
  if (w->rd_joint) (void)hipFree(w->rd_joint);
  if (w->ev0) (void)hipEventDestroy(w->ev0);
  if (w->ev1) (void)hipEventDestroy(w->ev1);
  delete w;
  return 0;
}

static int lbws_create(int n, int hist, LbWs** out, bool sync = true) {
  UUO_REQUIRE(hist >= 1 && hist <= LB_MAXH - 4, "lbfgs: history_size must be in [1,100]");
  LbWs* w = new LbWs();
  n = (n + LB_CW - 1) / LB_CW * LB_CW;  // whole column blocks: 16-byte loads of the work vectors stay in bounds
  w->n_cap = n;
  w->cap = hist + 1;
  w->nchunks = LB_MAXCHUNK;
  hipError_t e = hipSuccess;
  struct Piece { void** p; size_t bytes; };
  std::vector<Piece> pieces;
  auto A = [&](void** p, size_t bytes) { pieces.push_back({p, (bytes + 255) / 256 * 256}); };
  const size_t hist_floats = (size_t)(n / LB_CW) * LB_CBSTRIDE(w->cap);
  A((void**)&w->S, hist_floats * sizeof(float));
  A((void**)&w->Y, hist_floats * sizeof(float));
  A((void**)&w->vecs, (size_t)LB_NVEC * n * sizeof(float));
  const size_t part_dots = (size_t)w->nchunks * LB_ROWS * 3;
  A((void**)&w->part, (part_dots > 1024 ? part_dots : 1024) * sizeof(double));
  A((void**)&w->st, sizeof(LbDev));
  A((void**)&w->loss_dev, 16 * sizeof(float));
  {  // one allocation, one zero fill
    size_t total = 0;
    for (const Piece& q : pieces) total += q.bytes;
    e = hipMalloc(&w->slab, total);
    if (e == hipSuccess) e = hipMemset(w->slab, 0, total);
    size_t off = 0;
    if (e == hipSuccess)
      for (const Piece& q : pieces) {
        *q.p = (char*)w->slab + off;
        off += q.bytes;
      }
  }
  if (e == hipSuccess) e = hipHostMalloc((void**)&w->h_out, 32 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) std::memset(w->h_out, 0, 32 * sizeof(double));
  if (e == hipSuccess) e = hipEventCreate(&w->ev0);
  if (e == hipSuccess) e = hipEventCreate(&w->ev1);
  // null-stream memsets are not ordered with the (non-blocking) stream the first solve runs on
  if (e == hipSuccess && sync) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    lbws_destroy(w);
    uuo_set_error(std::string("lbfgs workspace: ") + hipGetErrorString(e));
    return -12;
  }
  *out = w;
  return 0;
}

// issue now, or record for the lock-step batch that is stepping this problem (uuo_common.h)
template <class A, class K>
static inline void lb_dispatch(int op, hipStream_t s, dim3 grid, dim3 block, K kernel, A& a) {
  a.h.gx = (int)grid.x;
  a.h.gy = (int)grid.y;
  if (uuo_record(op, (int)grid.x, (int)grid.y, a)) return;
  hipLaunchKernelGGL(kernel, grid, block, 0, s, a);
}
struct LbCopyArgs {  // UUO_OP_COPY: device-to-device copy of n floats
  UuoGridHdr h;
  uuo_gptr<float> dst;
  uuo_gptr<const float> src;
  size_t bytes;
};
static inline int lb_copy(hipStream_t s, float* dst, const float* src, size_t bytes) {
  LbCopyArgs c{{1, 1}, dst, src, bytes};
  if (uuo_record(UUO_OP_COPY, 1, 1, c)) return 0;
  UUO_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ---------------------------------------------------------------------------------------------------- driver
static double cubic_interpolate(double x1, double f1, double g1, double x2, double f2, double g2, bool has_bounds,
                                double lo, double hi) {
  double xmin_bound, xmax_bound;
  if (has_bounds) {
    xmin_bound = lo;
    xmax_bound = hi;
  } else if (x1 <= x2) {
    xmin_bound = x1;
    xmax_bound = x2;
  } else {
    xmin_bound = x2;
    xmax_bound = x1;
  }
  const double d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2);
  const double d2_square = d1 * d1 - g1 * g2;
  if (d2_square >= 0.0) {
    const double d2 = std::sqrt(d2_square);
    double min_pos;
    if (x1 <= x2)
      min_pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2));
    else
      min_pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2));
    return std::fmin(std::fmax(min_pos, xmin_bound), xmax_bound);
  }
  return (xmin_bound + xmax_bound) / 2.0;
}

struct LsPoint {
  double t = 0, f = 0, gtd = 0;
  double gmax = 0;
  int buf = -1;  // index of the work vector holding the gradient
};

struct LbHostOut {  // mirror of the tail of LbDev read back after every closure evaluation
  unsigned dmax_bits;
  int pad;
  LbOut out;
};

// lock-step batches: while a batch steps this problem its launches are recorded (uuo_recorder) and `evaluate` hands control
// back to the batch scheduler instead of polling; the scheduler resumes the solve when the evaluation has reported
static thread_local void (*g_batch_yield)(void) = nullptr;

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
  std::vector<double> all;  // gather target
};

static int shared_gather(SharedCtx* sh, const double* mine, int n) {
  sh->all.resize((size_t)sh->world * n);
  const int rc = sh->gather(sh->user, mine, n, sh->all.data());
  if (rc) {
    uuo_set_error("uuo_lbfgs_solve_shared: the gather hook returned " + std::to_string(rc));
    return rc < 0 ? rc : -rc;
  }
  return 0;
}

static int lbfgs_run(LbWs* w, hipStream_t s, Objective& obj, float* d_x, const uuo_lbfgs_options_t* opt,
                     uuo_lbfgs_stats_t* stats, uuo_eval_callback_t cb, void* cb_user, SharedCtx* sh = nullptr) {
  const bool batched = uuo_recorder != nullptr;
  const int n = obj.n;
  const int n_full = obj.n_full > 0 ? obj.n_full : obj.n;  // floats of the iterate (>= n on the compact packing)
  const UuoIndexMap map = obj.map;
  UUO_REQUIRE(n > 0 && n <= w->n_cap && n_full <= w->n_cap, "lbfgs: parameter count exceeds the workspace");
  const int hist = opt->history_size > 0 ? opt->history_size : 100;
  UUO_REQUIRE(hist + 1 <= w->cap, "lbfgs: history_size exceeds the workspace");
  const int cap = hist + 1;
  const int max_iter = opt->max_iter;
  const int max_eval = opt->max_eval > 0 ? opt->max_eval : (max_iter * 5) / 4;
  const double lr = opt->lr, tol_grad = opt->tolerance_grad, tol_change = opt->tolerance_change;
  const double c1 = 1e-4, c2 = 0.9;
  const size_t stride = (size_t)w->n_cap;
  const int ncb = (n + LB_CW - 1) / LB_CW;                 // column blocks of the history holding this problem
  const int gcb = (ncb + LB_MAXCHUNK - 1) / LB_MAXCHUNK;   // column blocks per dot-kernel group
  const int nchunks = (ncb + gcb - 1) / gcb;               // groups = partial sums per Gram entry (<= LB_MAXCHUNK)
  const int nb = (n + 255) / 256;
  const int nstat = std::min(64, nb);
  auto vec = [&](int i) { return w->vecs + (size_t)i * stride; };
  // work vectors: 0 direction d, 1 spare iterate buffer, 2.. gradient pool.  Iterates and gradients change hands by
  // pointer, never by copy: x lives in d_x or vec(1) (the other one receives the next trial point), the gradient at
  // x and the previous gradient are pool entries.
  float* d = vec(0);
  float* xcur = d_x;
  float* xoth = vec(1);
  bool pool_used[LB_NVEC] = {false};
  auto pool_alloc = [&]() {
    for (int i = 2; i < LB_NVEC; ++i)
      if (!pool_used[i]) {
        pool_used[i] = true;
        return i;
      }
    return -1;
  };
  LbHostOut* hh = reinterpret_cast<LbHostOut*>(w->h_out);
  LbOut* ho = &hh->out;
  double* stats_dev = reinterpret_cast<double*>((char*)w->st + offsetof(LbDev, out));
  int evals_total = 0;

  // evaluate at x_eval into gradient vector gv; statistics against d (or none); read back.
  // Stage closures report through pinned memory: their finalize kernel copies the read-back block into w->h_out and
  // then publishes a sequence number that this thread polls -- no copy command, no stream synchronisation.  A stuck
  // or failed stream is caught by a periodic hipStreamQuery.
  static const int poll_mode = UUO_ENV_INT("UUO_LBFGS_POLL", 1);
  // the slowest evaluation of the path (first closure at F = 3000) is ~10 ms; a minute means the device is gone
  const double eval_timeout_s = (double)UUO_ENV_INT("UUO_LBFGS_EVAL_TIMEOUT_S", 60);
  unsigned long long* rep_words = reinterpret_cast<unsigned long long*>(w->h_out);
  auto host_dmax = [&]() -> double {
    float f;
    std::memcpy(&f, &hh->dmax_bits, sizeof(float));
    return (double)f;
  };
  auto evaluate = [&](const float* x_eval, float* gv, bool with_dir) -> int {
    const float* dir = with_dir ? d : (const float*)nullptr;
    const bool poll = obj.fused_stats && poll_mode != 0;
    UuoEvalReport rep;
    if (poll) {
      rep.host = rep_words;
      rep.seq = ++w->seq;
    }
    int rc = obj.eval(s, x_eval, w->loss_dev, gv, dir, obj.fused_stats ? stats_dev : nullptr, poll ? &rep : nullptr);
    if (rc) return rc;
    if (batched) {  // the batch scheduler issues the recorded launches of all its problems and waits for their reports
      UUO_REQUIRE(poll && g_batch_yield, "lbfgs: a lock-step batch needs the polled report path");
      g_batch_yield();
      return 0;
    }
    if (poll) {
      // Bounded wait: the report word is polled; every ~1M polls the stream is queried (a failed or drained stream that
      // never reported is an error) and the wall clock is checked against eval_timeout_s -- a kernel that never finishes
      // must not pin this host thread for ever.
      UuoWaiter waiter;
      timespec t_start;
      clock_gettime(CLOCK_MONOTONIC, &t_start);
      while (__atomic_load_n(&rep_words[10], __ATOMIC_ACQUIRE) != rep.seq) {
        if (waiter.tick()) {
          const hipError_t q = hipStreamQuery(s);
          if (q != hipErrorNotReady && __atomic_load_n(&rep_words[10], __ATOMIC_ACQUIRE) != rep.seq) {
            uuo_set_error(std::string("lbfgs: closure evaluation did not report: ") + hipGetErrorString(q));
            return -5;
          }
          timespec t_now;
          clock_gettime(CLOCK_MONOTONIC, &t_now);
          const double waited = (double)(t_now.tv_sec - t_start.tv_sec) + 1e-9 * (double)(t_now.tv_nsec - t_start.tv_nsec);
          if (waited > eval_timeout_s) {
            uuo_set_error("lbfgs: closure evaluation " + std::to_string(evals_total) + " did not finish within " +
                          std::to_string((int)eval_timeout_s) + " s (stream still busy); giving up on the solve");
            return -62;  // -ETIME
          }
        }
      }
      if (sh) {
        // joint statistics of this evaluation: one gather of 6 + cnt doubles per rank, reduced here in rank order
        double mine[6 + 16];
        auto word = [&](int i) { double v; std::memcpy(&v, &rep_words[i], sizeof(double)); return v; };
        mine[0] = ho->loss; mine[1] = ho->gtd_new;
        mine[2] = word(11); mine[3] = word(12); mine[4] = word(13);  // own parameters: max|g|, sum|g|, g.g
        mine[5] = host_dmax();
        for (int l = 0; l < sh->cnt; ++l) mine[6 + l] = word(14 + l);
        const int m_ = 6 + sh->cnt;
        const int grc = shared_gather(sh, mine, m_);
        if (grc) return grc;
        double loss_j = 0.0, gtd_j = 0.0, g1_j = 0.0, gg_j = 0.0, gmax_j = 0.0, dmax_j = 0.0, gb[16] = {0.0};
        for (int r = 0; r < sh->world; ++r) {
          const double* a_ = sh->all.data() + (size_t)r * m_;
          loss_j += a_[0]; gtd_j += a_[1];
          gmax_j = std::fmax(gmax_j, a_[2]); g1_j += a_[3]; gg_j += a_[4];
          dmax_j = std::fmax(dmax_j, a_[5]);
          for (int l = 0; l < sh->cnt; ++l) gb[l] += a_[6 + l];
        }
        float* slot = w->h_gb + 16 * (w->gb_slot++ & 3);  // (a slot is reused four reports later: its copy has executed)
        for (int l = 0; l < sh->cnt; ++l) {
          const float gbf = (float)gb[l];
          slot[l] = gbf;
          g1_j += std::fabs((double)gbf);
          gg_j += (double)gbf * (double)gbf;
          gmax_j = std::fmax(gmax_j, std::fabs((double)gbf));
        }
        UUO_HIP_CHECK(hipMemcpyAsync(gv + sh->off, slot, sizeof(float) * sh->cnt, hipMemcpyHostToDevice, s));
        ho->loss = loss_j; ho->gtd_new = gtd_j; ho->gmax = gmax_j; ho->g1 = g1_j; ho->gg = gg_j;
        const float dmf = (float)dmax_j;
        std::memcpy(&hh->dmax_bits, &dmf, sizeof(float));
      }
      return 0;
    }
    if (!obj.fused_stats) {
      hipLaunchKernelGGL(k_lb_stats, dim3(nstat), dim3(256), 0, s, n, gv, dir, w->part);
      hipLaunchKernelGGL(k_lb_stats_final, dim3(1), dim3(64), 0, s, nstat, w->part, w->loss_dev, w->st);
      UUO_HIP_CHECK(hipGetLastError());
    }
    UUO_HIP_CHECK(hipMemcpyAsync(w->h_out, (const char*)w->st + offsetof(LbDev, dmax_bits), sizeof(LbHostOut),
                                 hipMemcpyDeviceToHost, s));
    UUO_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
  };
  auto report = [&](double loss, const float* x_eval) {
    if (cb) cb(cb_user, evals_total, (float)loss, x_eval);
    if (opt->verbose) std::printf("lbfgs eval %d loss %.9g\n", evals_total, loss);
    ++evals_total;
  };

  if (sh) {
    UUO_REQUIRE(!batched && obj.fused_stats && poll_mode != 0, "lbfgs: shared solves need the fused, polled report path");
    UUO_REQUIRE(sh->cnt > 0 && sh->cnt <= 16 && sh->off >= 0 && sh->off + sh->cnt <= n && sh->off_x >= 0 &&
                sh->off_x + sh->cnt <= n_full && sh->world >= 1 &&
                sh->rank >= 0 && sh->rank < sh->world && sh->gather, "lbfgs: bad shared-parameter description");
    if (!w->h_rows) {
      UUO_HIP_CHECK(hipHostMalloc((void**)&w->h_rows, (LB_ROWS * 3 + 1) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
      UUO_HIP_CHECK(hipHostMalloc((void**)&w->h_joint, 2 * LB_ROWS * 3 * sizeof(double), hipHostMallocDefault));
      UUO_HIP_CHECK(hipHostMalloc((void**)&w->h_gb, 4 * 16 * sizeof(float), hipHostMallocDefault));
      UUO_HIP_CHECK(hipMalloc((void**)&w->rd_joint, LB_ROWS * 3 * sizeof(double)));
      std::memset(w->h_rows, 0, (LB_ROWS * 3 + 1) * sizeof(double));
    }
    // the replicas of the shared entries must be bit-identical: every rank takes rank 0's values
    float hb[16];
    double mine[16];
    UUO_HIP_CHECK(hipMemcpyAsync(hb, d_x + sh->off_x, sizeof(float) * sh->cnt, hipMemcpyDeviceToHost, s));
    UUO_HIP_CHECK(hipStreamSynchronize(s));
    for (int l = 0; l < sh->cnt; ++l) mine[l] = (double)hb[l];
    const int grc = shared_gather(sh, mine, sh->cnt);
    if (grc) return grc;
    for (int l = 0; l < sh->cnt; ++l) hb[l] = (float)sh->all[l];
    UUO_HIP_CHECK(hipMemcpyAsync(d_x + sh->off_x, hb, sizeof(float) * sh->cnt, hipMemcpyHostToDevice, s));
    UUO_HIP_CHECK(hipStreamSynchronize(s));  // (hb is a stack buffer)
  }
  if (!batched) UUO_HIP_CHECK(hipEventRecord(w->ev0, s));
  if (map.nseg) {
    // compact packing: the trial points are written at the solver's coordinates only, so the other iterate buffer gets
    // the parameter entries that have no coordinate (and never move) once, here
    const int rc_ = lb_copy(s, xoth, xcur, (size_t)n_full * sizeof(float));
    if (rc_) return rc_;
  }
  hipLaunchKernelGGL(k_lb_init, dim3(1), dim3(1), 0, s, w->st);
  int ig = pool_alloc();  // gradient at the current iterate
  int ipg = -1;           // gradient at the previous iterate
  int rc = evaluate(xcur, vec(ig), false);
  if (rc) return rc;
  double loss = ho->loss;
  double gmax = ho->gmax;
  double g1 = ho->g1;
  const double gg0 = ho->gg;
  report(loss, xcur);
  stats->first_loss = (float)loss;
  int current_evals = 1;
  int n_iter = 0;
  int reason = 0;
  int head = 0, count = 0;
  double t = 0.0, prev_loss = loss;
  if (!(gmax > tol_grad)) {
    reason = 6;
  } else {
    while (n_iter < max_iter) {
      ++n_iter;
      float* g = vec(ig);
      // ---------------------------------------------------------------- step length guess (lbfgs.py:453-456)
      const double t_prev_iter = t;
      if (n_iter == 1)
        t = std::fmin(1.0, 1.0 / g1) * lr;
      else
        t = lr;
      // ---------------------------------------------------------------- direction + first trial point
      if (n_iter == 1) {
        LbNegArgs na{{0, 0}, n, g, d, xcur, (float)t, xoth, map};
        lb_dispatch(UUO_OP_NEG, s, dim3(nb), dim3(256), k_lb_neg, na);
      } else {
        const int cand = (head + count) % cap;
        const int nrows = 2 * (count + 1) + 1;
        LbDotsArgs da{{0, 0}, n, cap, w->cap, head, count, cand, w->S, w->Y, g, vec(ipg), d, (float)t_prev_iter, ncb, gcb, w->part};
        if (sh && sh->rank != 0) {  // the replicated entries are counted once in the joint dot products: on rank 0
          da.skip_lo = sh->off;
          da.skip_hi = sh->off + sh->cnt;
        }
        lb_dispatch(UUO_OP_DOTS, s, dim3(nchunks, LB_DRS), dim3(256), k_lb_dots, da);
        const double* rd_in = nullptr;
        if (sh) {
          const unsigned long long rseq = ++w->row_seq;
          hipLaunchKernelGGL(k_lb_rows, dim3(1), dim3(512), 0, s, nchunks, cap, cand, w->part, w->st, w->h_rows, rseq);
          UUO_HIP_CHECK(hipGetLastError());
          unsigned long long* rw = reinterpret_cast<unsigned long long*>(w->h_rows + LB_ROWS * 3);
          UuoWaiter waiter;
          timespec t_start;
          clock_gettime(CLOCK_MONOTONIC, &t_start);
          while (__atomic_load_n(rw, __ATOMIC_ACQUIRE) != rseq) {
            if (waiter.tick()) {
              const hipError_t q = hipStreamQuery(s);
              timespec t_now;
              clock_gettime(CLOCK_MONOTONIC, &t_now);
              const double waited = (double)(t_now.tv_sec - t_start.tv_sec) + 1e-9 * (double)(t_now.tv_nsec - t_start.tv_nsec);
              if ((q != hipErrorNotReady && __atomic_load_n(rw, __ATOMIC_ACQUIRE) != rseq) || waited > eval_timeout_s) {
                uuo_set_error(std::string("lbfgs: the Gram rows of a shared solve did not arrive: ") + hipGetErrorString(q));
                return -5;
              }
            }
          }
          const int nr = LB_ROWS * 3;
          const int grc = shared_gather(sh, w->h_rows, nr);
          if (grc) return grc;
          double* joint = w->h_joint + (size_t)nr * (w->joint_slot++ & 1);  // (reused two iterations later)
          for (int e = 0; e < nr; ++e) {
            double acc = 0.0;
            for (int r = 0; r < sh->world; ++r) acc += sh->all[(size_t)r * nr + e];
            joint[e] = acc;
          }
          UUO_HIP_CHECK(hipMemcpyAsync(w->rd_joint, joint, sizeof(double) * nr, hipMemcpyHostToDevice, s));
          rd_in = w->rd_joint;
        }
        static const int small_stop = UUO_ENV_INT("UUO_SMALL_STOP", 0);  // ablation only
        static const int small_ref = UUO_ENV_INT("UUO_SMALL_REF", 0);  // comparison only
        static const int small_block = UUO_ENV_INT("UUO_SMALL_BLOCK", 0);  // comparison only
        if (small_ref && !uuo_recorder)
          hipLaunchKernelGGL(k_lb_small_ref, dim3(1), dim3(256), 0, s, nchunks, cap, hist, cand, w->part, w->st, small_stop);
        else if (small_block && !uuo_recorder)
          hipLaunchKernelGGL(k_lb_small, dim3(1), dim3(512), 0, s, nchunks, cap, hist, cand, w->part, w->st, small_stop);
        else {
          LbSmallArgs sa{{0, 0}, nchunks, cap, hist, cand, w->part, w->st, small_stop};
          sa.rd_in = rd_in;
          lb_dispatch(UUO_OP_SMALL, s, dim3(1), dim3(512), k_lb_small_inv, sa);
        }
        LbDirArgs ra{{0, 0}, n, cap, w->cap, w->S, w->Y, g, w->st, d, xcur, (float)t, xoth, map};
        lb_dispatch(UUO_OP_DIR, s, dim3(2 * ncb), dim3(64 * LB_DQ), k_lb_direction, ra);
      }
      UUO_HIP_CHECK(hipGetLastError());
      prev_loss = loss;
      // ---------------------------------------------------------------- first trial (speculative: launched
      // before g.d is known on the host; discarded if the direction test fails)
      LsPoint pnew;
      pnew.buf = pool_alloc();
      UUO_REQUIRE(pnew.buf >= 0, "lbfgs: gradient pool exhausted");
      rc = evaluate(xoth, vec(pnew.buf), true);
      if (rc) return rc;
      double gtd, d_norm;
      if (n_iter == 1) {  // d = -g: g.d = -g.g, max|d| = max|g|
        gtd = -gg0;
        d_norm = gmax;
      } else {
        gtd = ho->gtd_dir;
        d_norm = host_dmax();
        if (ho->accepted != 0.0) {
          if (count == hist)
            head = (head + 1) % cap;
          else
            count += 1;
        }
      }
      if (gtd > -tol_change) {
        pool_used[pnew.buf] = false;
        reason = 5;
        break;
      }
      pnew.t = t;
      pnew.f = ho->loss;
      pnew.gtd = ho->gtd_new;
      pnew.gmax = ho->gmax;
      report(pnew.f, xoth);
      double t_at_xoth = t;  // step whose iterate currently sits in xoth
      // ---------------------------------------------------------------- strong Wolfe (lbfgs.py:40-209)
      const int max_ls = max_eval - current_evals;
      int ls_func_evals = 1;
      LsPoint p0;  // the point at t = 0
      p0.t = 0;
      p0.f = loss;
      p0.gtd = gtd;
      p0.gmax = gmax;
      p0.buf = ig;
      auto release = [&](int buf) {
        if (buf != ig && buf >= 2) pool_used[buf] = false;
      };
      auto trial = [&](LsPoint& pt) -> int {
        pt.buf = pool_alloc();
        UUO_REQUIRE(pt.buf >= 0, "lbfgs: gradient pool exhausted");
        LbAxpyArgs xa{{0, 0}, n, xcur, (float)pt.t, d, xoth, map};
        lb_dispatch(UUO_OP_AXPY, s, dim3(nb), dim3(256), k_lb_axpy, xa);
        int r = evaluate(xoth, vec(pt.buf), true);
        if (r) return r;
        t_at_xoth = pt.t;
        pt.f = ho->loss;
        pt.gtd = ho->gtd_new;
        pt.gmax = ho->gmax;
        report(pt.f, xoth);
        ++ls_func_evals;
        return 0;
      };
      LsPoint pprev = p0;
      LsPoint br[2];
      int nbr = 0;
      bool done = false;
      int ls_iter = 0;
      while (ls_iter < max_ls) {
        if (pnew.f > (loss + c1 * pnew.t * gtd) || (ls_iter > 1 && pnew.f >= pprev.f)) {
          br[0] = pprev;
          br[1] = pnew;
          nbr = 2;
          break;
        }
        if (std::fabs(pnew.gtd) <= -c2 * gtd) {
          br[0] = pnew;
          nbr = 1;
          done = true;
          release(pprev.buf);
          break;
        }
        if (pnew.gtd >= 0) {
          br[0] = pprev;
          br[1] = pnew;
          nbr = 2;
          break;
        }
        const double min_step = pnew.t + 0.01 * (pnew.t - pprev.t);
        const double max_step = pnew.t * 10;
        const double t_next = cubic_interpolate(pprev.t, pprev.f, pprev.gtd, pnew.t, pnew.f, pnew.gtd, true, min_step,
                                                max_step);
        release(pprev.buf);
        pprev = pnew;
        pnew = LsPoint();
        pnew.t = t_next;
        rc = trial(pnew);
        if (rc) return rc;
        ++ls_iter;
      }
      if (nbr == 0) {  // ls_iter == max_ls
        br[0] = p0;
        br[1] = pnew;
        nbr = 2;
        if (pprev.buf != pnew.buf) release(pprev.buf);
      }
      bool insuf_progress = false;
      int low_pos, high_pos;
      if (br[0].f <= br[nbr - 1].f) {
        low_pos = 0;
        high_pos = 1;
      } else {
        low_pos = 1;
        high_pos = 0;
      }
      while (!done && ls_iter < max_ls) {
        // torch 2.10's LBFGS.step does not hand its tolerance_change to _strong_wolfe (lbfgs.py:486-488): the line
        // search always uses that function's default, 1e-9
        if (std::fabs(br[1].t - br[0].t) * d_norm < 1e-9) break;
        double tz = cubic_interpolate(br[0].t, br[0].f, br[0].gtd, br[1].t, br[1].f, br[1].gtd, false, 0, 0);
        const double bmax = std::fmax(br[0].t, br[1].t), bmin = std::fmin(br[0].t, br[1].t);
        const double eps = 0.1 * (bmax - bmin);
        if (std::fmin(bmax - tz, tz - bmin) < eps) {
          if (insuf_progress || tz >= bmax || tz <= bmin) {
            if (std::fabs(tz - bmax) < std::fabs(tz - bmin))
              tz = bmax - eps;
            else
              tz = bmin + eps;
            insuf_progress = false;
          } else {
            insuf_progress = true;
          }
        } else {
          insuf_progress = false;
        }
        LsPoint pz;
        pz.t = tz;
        rc = trial(pz);
        if (rc) return rc;
        ++ls_iter;
        if (pz.f > (loss + c1 * pz.t * gtd) || pz.f >= br[low_pos].f) {
          release(br[high_pos].buf);
          br[high_pos] = pz;
          if (br[0].f <= br[1].f) {
            low_pos = 0;
            high_pos = 1;
          } else {
            low_pos = 1;
            high_pos = 0;
          }
        } else {
          if (std::fabs(pz.gtd) <= -c2 * gtd) {
            done = true;
          } else if (pz.gtd * (br[high_pos].t - br[low_pos].t) >= 0) {
            release(br[high_pos].buf);
            br[high_pos] = br[low_pos];
            br[low_pos] = pz;
            continue;
          }
          // new point becomes new low (the old low is dropped unless it was just moved to high)
          release(br[low_pos].buf);
          br[low_pos] = pz;
        }
      }
      const LsPoint res = (nbr == 1) ? br[0] : br[low_pos];
      // ---------------------------------------------------------------- accept: x <- x + t d by pointer where the
      // accepted point is the trial that already sits in xoth (p.add_(d, alpha=t) rounds exactly like the trial)
      t = res.t;
      loss = res.f;
      gmax = res.gmax;
      if (res.t == 0.0) {
        // line search returned the starting point (bracket low at t = 0): iterate unchanged
      } else {
        if (t_at_xoth != res.t) {
          LbAxpyArgs xa{{0, 0}, n, xcur, (float)t, d, xoth, map};
          lb_dispatch(UUO_OP_AXPY_ACCEPT, s, dim3(nb), dim3(256), k_lb_axpy, xa);
        }
        float* tmp = xcur;
        xcur = xoth;
        xoth = tmp;
      }
      UUO_HIP_CHECK(hipGetLastError());
      {  // gradient hand-over: previous <- current, current <- accepted point's
        const int old_g = ig, old_pg = ipg;
        const int new_g = res.buf;
        for (int i = 2; i < LB_NVEC; ++i) pool_used[i] = false;
        if (new_g == old_g) {
          // accepted point is the starting point: prev gradient must still become a copy of g (y = 0 next time)
          ipg = (old_pg >= 0 && old_pg != old_g) ? old_pg : pool_alloc();
          pool_used[ipg] = true;
          { const int rc_ = lb_copy(s, vec(ipg), vec(old_g), (size_t)n * sizeof(float)); if (rc_) return rc_; }
          ig = old_g;
        } else {
          ipg = old_g;
          ig = new_g;
        }
        pool_used[ig] = true;
        pool_used[ipg] = true;
      }
      current_evals += ls_func_evals;
      // ---------------------------------------------------------------- termination (lbfgs.py:511-526)
      if (n_iter == max_iter) {
        reason = 0;
        break;
      }
      if (current_evals >= max_eval) {
        reason = 1;
        break;
      }
      if (gmax <= tol_grad) {
        reason = 2;
        break;
      }
      if (d_norm * std::fabs(t) <= tol_change) {
        reason = 3;
        break;
      }
      if (std::fabs(loss - prev_loss) < tol_change) {
        reason = 4;
        break;
      }
    }
  }
  if (xcur != d_x) {
    const int rc_ = lb_copy(s, d_x, xcur, (size_t)n_full * sizeof(float));
    if (rc_) return rc_;
  }
  float ms = 0.f;
  if (!batched) {
    UUO_HIP_CHECK(hipEventRecord(w->ev1, s));
    UUO_HIP_CHECK(hipEventSynchronize(w->ev1));
    UUO_HIP_CHECK(hipEventElapsedTime(&ms, w->ev0, w->ev1));
  }
  stats->n_iter = n_iter;
  stats->n_eval = current_evals;
  stats->final_loss = (float)loss;
  stats->stop_reason = reason;
  stats->device_ms = ms;
  return 0;
}

// ---------------------------------------------------------------------------------------------------- fit workspace
static int fit_create_impl(uuo_model_t* model, int F, int M, uuo_fit_t** out, bool sync);
extern "C" int uuo_fit_create(uuo_model_t* model, int F, int M, uuo_fit_t** out) {
  return fit_create_impl(model, F, M, out, true);
}
static int fit_create_impl(uuo_model_t* model, int F, int M, uuo_fit_t** out, bool sync) {
  UUO_REQUIRE(model && out, "uuo_fit_create: null argument");
  UUO_REQUIRE(F > 0 && M > 0, "uuo_fit_create: F and M must be positive");
  uuo_fit* fit = new uuo_fit();
  fit->model = model;
  fit->F = F;
  fit->M = M;
  fit->nFT = (F + UUO_FT - 1) / UUO_FT;
  fit->n_max = 219 * F + 10;
  const int nFT = fit->nFT;
  // one device allocation and one zero fill for the whole workspace (a lock-step batch creates hundreds of these)
  hipError_t e = hipSuccess;
  struct Piece { void** p; size_t bytes; };
  std::vector<Piece> pieces;
  auto A = [&](void** p, size_t bytes) { pieces.push_back({p, (bytes + 255) / 256 * 256}); };
  A((void**)&fit->pfaT, (size_t)nFT * UUO_KP * UUO_FT * sizeof(float));
  A((void**)&fit->A, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 12 * sizeof(float));
  A((void**)&fit->verts, (size_t)F * model->V * 3 * sizeof(float));
  A((void**)&fit->nn_flags, (size_t)F * 8 * sizeof(int));
  A((void**)&fit->part_sb, (size_t)model->V * 8 * sizeof(float));
  A((void**)&fit->bbox, (size_t)F * ((model->V + 15) / 16) * 6 * sizeof(float));
  A((void**)&fit->nn, (size_t)F * M * sizeof(unsigned long long));
  A((void**)&fit->frame_part, (size_t)F * UUO_FP * sizeof(float));
  A((void**)&fit->frames, (size_t)F * sizeof(FrameLds));
  A((void**)&fit->mask, (size_t)F * M * sizeof(float));
  A((void**)&fit->scalars, 64 * sizeof(float));
  A((void**)&fit->zeros16, 16 * sizeof(float));
  A((void**)&fit->vecs, (size_t)fit->n_max * sizeof(float));
  {
    size_t total = 0;
    for (const Piece& q : pieces) total += q.bytes;
    e = hipMalloc(&fit->slab, total);
    if (e == hipSuccess) e = hipMemset(fit->slab, 0, total);
    size_t off = 0;
    if (e == hipSuccess)
      for (const Piece& q : pieces) {
        *q.p = (char*)fit->slab + off;
        off += q.bytes;
      }
  }
  if (e == hipSuccess) e = hipEventCreate(&fit->ev0);
  if (e == hipSuccess) e = hipEventCreate(&fit->ev1);
  if (e != hipSuccess) {
    uuo_set_error(std::string("uuo_fit_create: ") + hipGetErrorString(e));
    uuo_fit_destroy(fit);
    return -12;
  }
  if (sync) UUO_HIP_CHECK(hipDeviceSynchronize());  // the zero fills above ran on the null stream
  *out = fit;
  return 0;
}

extern "C" int uuo_fit_destroy(uuo_fit_t* fit) {
  if (!fit) return 0;
  if (fit->slab) (void)hipFree(fit->slab);
  if (fit->pose_cache && !fit->shared_pose_cache) (void)hipFree(fit->pose_cache);
  if (fit->ev0) (void)hipEventDestroy(fit->ev0);
  if (fit->ev1) (void)hipEventDestroy(fit->ev1);
  if (fit->lbws) lbws_destroy((LbWs*)fit->lbws);
  delete fit;
  return 0;
}

extern "C" int uuo_lbfgs_solve(uuo_fit_t* fit, void* stream, const uuo_problem_t* p, float* d_x,
                               const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats, uuo_eval_callback_t cb,
                               void* cb_user) {
  int rc = uuo_validate_problem(fit, p);
  if (rc) return rc;
  UUO_REQUIRE(d_x && opt && stats, "uuo_lbfgs_solve: null argument");
  UUO_REQUIRE(opt->max_iter > 0, "uuo_lbfgs_solve: max_iter must be positive");
  hipStream_t s = (hipStream_t)stream;
  const int hist = opt->history_size > 0 ? opt->history_size : 100;
  // The optimiser's workspace (history S, Y: 2 x (hist+1) x n floats, 53 MB at n = 65 710) is sized for THIS problem,
  // not for the largest stage of the sequence: a workspace that only ever solves the part stage (n = 3F + 11) stays
  // ~70x smaller.  It is re-created only when the parameter count or the history grows.
  const int n_params = uuo_problem_num_params(p);
  LbWs* w = (LbWs*)fit->lbws;
  if (!w || w->cap < hist + 1 || w->n_cap < n_params) {
    const int keep_hist = w ? std::max(hist, w->cap - 1) : hist;
    if (w) {
      UUO_HIP_CHECK(hipStreamSynchronize(s));
      lbws_destroy(w);
    }
    fit->lbws = nullptr;
    rc = lbws_create(n_params, keep_hist, &w);
    if (rc) return rc;
    fit->lbws = w;
  }
  rc = uuo_ensure_mask(fit, s, p);
  if (rc) return rc;
  StageObjective obj;
  rc = stage_objective_init(obj, fit, s, p, d_x);
  if (rc) return rc;
  std::memset(stats, 0, sizeof(*stats));
  return lbfgs_run(w, s, obj, d_x, opt, stats, cb, cb_user);
}

// EXTENSION (BASELINE configs[3]; not reference behaviour): uuo_lbfgs_solve where the `world` ranks that call
// it together -- one stage problem each, same stage -- share the shape vector.  See SharedCtx above.
extern "C" int uuo_lbfgs_solve_shared(uuo_fit_t* fit, void* stream, const uuo_problem_t* p, float* d_x,
                                      const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats, const uuo_shared_t* shared,
                                      uuo_eval_callback_t cb, void* cb_user) {
  UUO_REQUIRE(shared && shared->gather && shared->world >= 1 && shared->rank >= 0 && shared->rank < shared->world,
              "uuo_lbfgs_solve_shared: bad rank description");
  int rc = uuo_validate_problem(fit, p);
  if (rc) return rc;
  UUO_REQUIRE(d_x && opt && stats, "uuo_lbfgs_solve_shared: null argument");
  UUO_REQUIRE(opt->max_iter > 0, "uuo_lbfgs_solve_shared: max_iter must be positive");
  UUO_REQUIRE(uuo_recorder == nullptr, "uuo_lbfgs_solve_shared: not inside a lock-step batch");
  hipStream_t s = (hipStream_t)stream;
  const int hist = opt->history_size > 0 ? opt->history_size : 100;
  const int n_params = uuo_problem_num_params(p);
  LbWs* w = (LbWs*)fit->lbws;
  if (!w || w->cap < hist + 1 || w->n_cap < n_params) {
    const int keep_hist = w ? std::max(hist, w->cap - 1) : hist;
    if (w) {
      UUO_HIP_CHECK(hipStreamSynchronize(s));
      lbws_destroy(w);
    }
    fit->lbws = nullptr;
    rc = lbws_create(n_params, keep_hist, &w);
    if (rc) return rc;
    fit->lbws = w;
  }
  rc = uuo_ensure_mask(fit, s, p);
  if (rc) return rc;
  StageObjective obj;
  rc = stage_objective_init(obj, fit, s, p, d_x);
  if (rc) return rc;
  SharedCtx sh;
  sh.gather = shared->gather;
  sh.user = shared->user;
  sh.rank = shared->rank;
  sh.world = shared->world;
  sh.cnt = UUO_NUM_BETAS;
  const int F = p->F;  // offset of the betas in the parameter vector and in the solver's packing (closure.hip stage_layout)
  sh.off_x = (p->stage == UUO_STAGE_CHAMFER) ? 4 * F : (p->stage == UUO_STAGE_MARKER) ? 207 * F : 3 * F + 1;
  sh.off = (p->stage == UUO_STAGE_MARKER && obj.compact) ? 138 * F : sh.off_x;
  {  // every rank must run the same packing: a rank whose third rows differ from their targets makes all of them run full
    double mine = obj.compact ? 1.0 : 0.0;
    rc = shared_gather(&sh, &mine, 1);
    if (rc) return rc;
    bool all_compact = true;
    for (int r = 0; r < sh.world; ++r) all_compact = all_compact && sh.all[r] != 0.0;
    if (obj.compact && !all_compact) {
      obj.compact = false;
      obj.map = uuo_stage_index_map(p, false);
      obj.n = obj.n_full;
      sh.off = sh.off_x;
    }
  }
  std::memset(stats, 0, sizeof(*stats));
  return lbfgs_run(w, s, obj, d_x, opt, stats, cb, cb_user, &sh);
}

// ---------------------------------------------------------------------------------------------------- lock-step batches
// B independent L-BFGS problems of one stage and one (F, M) -- the candidate body parts of find_best_part_fits
// (reference markers/markers_utils.py:416-610 solves them one after the other) or the yaw hypotheses of
// multimodal_video_mocap (multimodal.py:462-574) -- stepped together: one ROUND = one closure evaluation of every live
// problem, every kernel of the round launched once for all of them (grid z = problem).  Each problem runs the unchanged
// lbfgs_run (same decisions, same arithmetic, same kernels' bodies: bit-identical to solving it alone) as a coroutine on
// its own stack; where the single-problem driver would poll for its evaluation's report it yields to the scheduler, which
// merges the launches the live problems recorded, stages their argument structs with one host-to-device copy, issues them
// in the canonical order of uuo_common.h and waits for every report.  A problem that converged simply stops taking part.
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
    switch (k) {
      case UUO_OP_AXPY_ACCEPT:
      case UUO_OP_AXPY:
        hipLaunchKernelGGL(k_lb_axpy_b, dim3(gx[k], gy[k], n_), dim3(256), 0, s, (const LbAxpyArgs*)da);
        rc = 0;
        break;
      case UUO_OP_NEG:
        hipLaunchKernelGGL(k_lb_neg_b, dim3(gx[k], gy[k], n_), dim3(256), 0, s, (const LbNegArgs*)da);
        rc = 0;
        break;
      case UUO_OP_DOTS:
        hipLaunchKernelGGL(k_lb_dots_b, dim3(gx[k], gy[k], n_), dim3(256), 0, s, (const LbDotsArgs*)da);
        rc = 0;
        break;
      case UUO_OP_SMALL:
        hipLaunchKernelGGL(k_lb_small_inv_b, dim3(gx[k], gy[k], n_), dim3(512), 0, s, (const LbSmallArgs*)da);
        rc = 0;
        break;
      case UUO_OP_DIR:
        hipLaunchKernelGGL(k_lb_direction_b, dim3(gx[k], gy[k], n_), dim3(64 * LB_DQ), 0, s, (const LbDirArgs*)da);
        rc = 0;
        break;
      default:
        rc = uuo_batched_launch_smpl(k, s, da, n_, gx[k], gy[k]);
        if (rc == 1) rc = uuo_batched_launch_nn(k, s, da, n_, gx[k], gy[k]);
        if (rc == 1) rc = uuo_batched_launch_closure(k, s, da, n_, gx[k], gy[k]);
        UUO_REQUIRE(rc != 1, "batch: no batched kernel for a recorded launch kind");
        break;
    }
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

// host copy of a device vector, ordered on `stream` and complete on return (the iter_fn adapter of the Python mirror
// uses it inside the evaluation callback, where it only has the raw pointer)
// ---------------------------------------------------------------------------------------------------- host-composed closures
struct CallbackObjective : Objective {
  uuo_closure_fn fn = nullptr;
  void* user = nullptr;
  int eval(hipStream_t s, const float* x, float* loss_dev, float* grad, const float*, double*, const UuoEvalReport*) override {
    const int rc = fn(user, (void*)s, x, loss_dev, grad);
    if (rc) {
      uuo_set_error("uuo_lbfgs_minimize: the closure returned " + std::to_string(rc));
      return rc < 0 ? rc : -rc;
    }
    return 0;
  }
};

extern "C" int uuo_lbfgs_minimize(void* stream, int n, float* d_x, const uuo_lbfgs_options_t* opt, uuo_lbfgs_stats_t* stats,
                                  uuo_closure_fn closure, void* user, uuo_eval_callback_t cb, void* cb_user) {
  UUO_REQUIRE(d_x && opt && stats && closure && n > 0, "uuo_lbfgs_minimize: bad arguments");
  UUO_REQUIRE(opt->max_iter > 0, "uuo_lbfgs_minimize: max_iter must be positive");
  UUO_REQUIRE(uuo_recorder == nullptr, "uuo_lbfgs_minimize: not inside a lock-step batch");
  const int hist = opt->history_size > 0 ? opt->history_size : 100;
  LbWs* w = nullptr;
  int rc = lbws_create(n, hist, &w);
  if (rc) return rc;
  CallbackObjective obj;
  obj.fn = closure;
  obj.user = user;
  obj.n = n;
  std::memset(stats, 0, sizeof(*stats));
  rc = lbfgs_run(w, (hipStream_t)stream, obj, d_x, opt, stats, cb, cb_user);
  (void)hipStreamSynchronize((hipStream_t)stream);
  lbws_destroy(w);
  return rc;
}

// the 2D-prior fit (reprojection.hip) under the same driver
struct ReprojObjective : Objective {
  uuo_reprojection* h = nullptr;
  float* x_last = nullptr;
  float* kp_last = nullptr;
  int eval(hipStream_t s, const float* x, float* loss_dev, float* grad, const float*, double*, const UuoEvalReport*) override {
    if (x_last) UUO_HIP_CHECK(hipMemcpyAsync(x_last, x, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return uuo_reprojection_eval_impl(h, s, x, loss_dev, grad, kp_last, nullptr);
  }
};

extern "C" int uuo_reprojection_solve(uuo_reprojection_t* h, void* stream, float* d_x, const uuo_lbfgs_options_t* opt,
                                      uuo_lbfgs_stats_t* stats, float* d_x_last, float* d_kp_last, uuo_eval_callback_t cb,
                                      void* cb_user) {
  UUO_REQUIRE(h && d_x && opt && stats, "uuo_reprojection_solve: null argument");
  UUO_REQUIRE(opt->max_iter > 0, "uuo_reprojection_solve: max_iter must be positive");
  UUO_REQUIRE(uuo_recorder == nullptr, "uuo_reprojection_solve: not inside a lock-step batch");
  const int n = uuo_reprojection_num_params(&h->p);
  const int hist = opt->history_size > 0 ? opt->history_size : 100;
  LbWs* w = nullptr;
  int rc = lbws_create(n, hist, &w);
  if (rc) return rc;
  ReprojObjective obj;
  obj.h = h;
  obj.n = n;
  obj.x_last = d_x_last;
  obj.kp_last = d_kp_last;
  std::memset(stats, 0, sizeof(*stats));
  rc = lbfgs_run(w, (hipStream_t)stream, obj, d_x, opt, stats, cb, cb_user);
  (void)hipStreamSynchronize((hipStream_t)stream);
  lbws_destroy(w);
  return rc;
}

extern "C" int uuo_copy_device(void* stream, void* d_dst, const void* d_src, size_t bytes) {
  UUO_REQUIRE(d_dst && d_src, "uuo_copy_device: null argument");
  if (bytes) UUO_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

extern "C" int uuo_copy_to_host(void* stream, const float* d_src, float* h_dst, int n) {
  UUO_REQUIRE(d_src && h_dst && n >= 0, "uuo_copy_to_host: bad arguments");
  UUO_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  UUO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

#ifdef UUO_DEBUG_HOOKS
// ---- everything below exists only in libuuo_hip_debug.so (tests/ and tools/) ------------------------------------------
// host-only: runs a script of staging operations against a fresh UuoStaging (no device call: usable without a GPU).
// ops[3 i ..] = {code, region, bytes}; code 0 begin_flush -> out = {needs a synchronise first, 0}; 1 append -> {fits, offset};
// 2 report_arrived -> {0, 0}; 3 synchronized -> {0, 0}.  After every op out[4 i + 2 ..] = pending[0] | pending[1] << 1, used[region].
extern "C" int uuo_debug_staging_script(const long long* ops, int n_ops, long long region_cap, long long* out) {
  UUO_REQUIRE(ops && out && n_ops >= 0 && region_cap > 0, "uuo_debug_staging_script: bad arguments");
  UuoStaging st;
  st.region_cap = (size_t)region_cap;
  for (int i = 0; i < n_ops; ++i) {
    const int code = (int)ops[3 * i], r = (int)ops[3 * i + 1];
    const size_t bytes = (size_t)ops[3 * i + 2];
    UUO_REQUIRE(r == 0 || r == 1, "uuo_debug_staging_script: region must be 0 or 1");
    long long a = 0, bval = 0;
    if (code == 0) {
      a = st.begin_flush(r, bytes) ? 1 : 0;
    } else if (code == 1) {
      size_t off = 0;
      a = st.append(r, bytes, &off) ? 1 : 0;
      bval = (long long)off;
    } else if (code == 2) {
      st.report_arrived(r);
    } else if (code == 3) {
      st.synchronized();
    } else {
      UUO_REQUIRE(false, "uuo_debug_staging_script: unknown op");
    }
    out[4 * i] = a;
    out[4 * i + 1] = bval;
    out[4 * i + 2] = (st.pending[0] ? 1 : 0) | (st.pending[1] ? 2 : 0);
    out[4 * i + 3] = (long long)st.used[r];
  }
  return 0;
}

// optimiser self-test on analytic objectives (tests/test_gpu_parity.py::test_lbfgs_* compare with torch.optim.LBFGS on
// the CPU, evaluation by evaluation through `cb`)
extern "C" int uuo_lbfgs_selftest(void* stream, int kind, int n, float* d_x, const uuo_lbfgs_options_t* opt,
                                  uuo_lbfgs_stats_t* stats, uuo_eval_callback_t cb, void* cb_user) {
  UUO_REQUIRE(d_x && opt && stats && n > 0 && kind >= 0 && kind <= 2, "uuo_lbfgs_selftest: bad arguments");
  const int hist = opt->history_size > 0 ? opt->history_size : 100;
  LbWs* w = nullptr;
  int rc = lbws_create(n, hist, &w);
  if (rc) return rc;
  TestObjective obj;
  obj.kind = kind;
  obj.n = n;
  std::memset(stats, 0, sizeof(*stats));
  rc = lbfgs_run(w, (hipStream_t)stream, obj, d_x, opt, stats, cb, cb_user);
  lbws_destroy(w);
  return rc;
}

// debug hook (not in the public header): device time of k_lb_small at a fixed history size k, optionally cut
// short after a phase (stop = 1..4) -- used to attribute its latency (tools/, not on the product path)
extern "C" int uuo_debug_time_small(int k, int iters, int stop, float* ms_out) {
  UUO_REQUIRE(k >= 1 && k <= LB_MAXH - 4 && iters > 0 && ms_out, "uuo_debug_time_small: bad arguments");
  LbWs* w = nullptr;
  int rc = lbws_create(4096, LB_MAXH - 4, &w);
  if (rc) return rc;
  std::vector<double> SY((size_t)LB_MAXH * LB_MAXH, 0.0), YY((size_t)LB_MAXH * LB_MAXH, 0.0);
  for (int i = 0; i < LB_MAXH; ++i)
    for (int j = 0; j < LB_MAXH; ++j) {
      SY[(size_t)i * LB_MAXH + j] = (i == j) ? 2.0 : 0.01 / (1 + std::abs(i - j));
      YY[(size_t)i * LB_MAXH + j] = (i == j) ? 3.0 : 0.02 / (1 + std::abs(i - j));
    }
  UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, SY), SY.data(), SY.size() * sizeof(double), hipMemcpyHostToDevice));
  UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, YY), YY.data(), YY.size() * sizeof(double), hipMemcpyHostToDevice));
  std::vector<double> part((size_t)LB_MAXCHUNK * LB_ROWS * 3, 1e-3);
  UUO_HIP_CHECK(hipMemcpy(w->part, part.data(), part.size() * sizeof(double), hipMemcpyHostToDevice));
  {
    std::vector<double> W((size_t)LB_MAXH * LB_MAXH, 0.0);
    for (int i = 0; i < LB_MAXH; ++i) W[(size_t)i * LB_MAXH + i] = 0.5;
    UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, W), W.data(), W.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  const int cap = LB_MAXH - 3, hist = LB_MAXH - 4;
  float total = 0.f;
  for (int it = 0; it < iters + 1; ++it) {
    const int head = 0, count = k - 1;  // the kernel accepts the candidate -> k pairs
    UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, head), &head, sizeof(int), hipMemcpyHostToDevice));
    UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, count), &count, sizeof(int), hipMemcpyHostToDevice));
    const double one = 1.0;
    UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, Hdiag), &one, sizeof(double), hipMemcpyHostToDevice));
    UUO_HIP_CHECK(hipEventRecord(w->ev0, nullptr));
    if (stop >= 200)
      { LbSmallArgs sa_{{1, 1}, LB_MAXCHUNK, cap, hist, k - 1, w->part, w->st, stop - 200}; hipLaunchKernelGGL(k_lb_small_inv, dim3(1), dim3(512), 0, nullptr, sa_); }
    else if (stop >= 100)
      hipLaunchKernelGGL(k_lb_small_ref, dim3(1), dim3(256), 0, nullptr, LB_MAXCHUNK, cap, hist, k - 1, w->part, w->st, stop - 100);
    else
      hipLaunchKernelGGL(k_lb_small, dim3(1), dim3(512), 0, nullptr, LB_MAXCHUNK, cap, hist, k - 1, w->part, w->st, stop);
    UUO_HIP_CHECK(hipEventRecord(w->ev1, nullptr));
    UUO_HIP_CHECK(hipEventSynchronize(w->ev1));
    float ms = 0.f;
    UUO_HIP_CHECK(hipEventElapsedTime(&ms, w->ev0, w->ev1));
    if (it > 0) total += ms;
  }
  *ms_out = total / iters;
  lbws_destroy(w);
  return 0;
}

// debug hook (not in the public header): host-side cost of `count` launches of a one-thread kernel on `stream`
// followed by a stream synchronisation; returns microseconds of host time spent enqueueing and in total
extern "C" int uuo_debug_launch_rate(void* stream, int count, double* us_enqueue, double* us_total) {
  UUO_REQUIRE(count > 0 && us_enqueue && us_total, "uuo_debug_launch_rate: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  LbDev* st = nullptr;
  UUO_HIP_CHECK(hipMalloc((void**)&st, sizeof(LbDev)));
  hipLaunchKernelGGL(k_lb_init, dim3(1), dim3(1), 0, s, st);
  UUO_HIP_CHECK(hipStreamSynchronize(s));
  timespec t0, t1, t2;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < count; ++i) hipLaunchKernelGGL(k_lb_init, dim3(1), dim3(1), 0, s, st);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  UUO_HIP_CHECK(hipStreamSynchronize(s));
  clock_gettime(CLOCK_MONOTONIC, &t2);
  *us_enqueue = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
  *us_total = (t2.tv_sec - t0.tv_sec) * 1e6 + (t2.tv_nsec - t0.tv_nsec) * 1e-3;
  (void)hipFree(st);
  return 0;
}

// debug/test hook (not in the public header): direction coefficients of one k_lb_small call on a synthetic history
// of k pairs (deterministic pseudo-random Gram data, moderately conditioned), with the reference (serial) or the
// block-inverse kernel; out = [cs(LB_MAXH) | cy(LB_MAXH) | g.d]
extern "C" int uuo_debug_small_coeffs(int k, int use_ref, int seed, double* out) {
  UUO_REQUIRE(k >= 1 && k <= LB_MAXH - 4 && out, "uuo_debug_small_coeffs: bad arguments");
  LbWs* w = nullptr;
  int rc = lbws_create(4096, LB_MAXH - 4, &w);
  if (rc) return rc;
  auto rnd = [&](unsigned a, unsigned b) {
    unsigned long long z = (unsigned long long)(a * 1315423911u + b * 2654435761u + (unsigned)seed * 97u) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0 - 0.5;
  };
  std::vector<double> SY((size_t)LB_MAXH * LB_MAXH, 0.0), YY((size_t)LB_MAXH * LB_MAXH, 0.0);
  for (int i = 0; i < LB_MAXH; ++i)
    for (int j = 0; j < LB_MAXH; ++j) {
      SY[(size_t)i * LB_MAXH + j] = (i == j) ? 1.0 + 0.5 * rnd(i, i) + 0.02 * i : 0.3 * rnd(i, j) / (1.0 + 0.2 * std::abs(i - j));
      const double yy = (i == j) ? 3.0 + rnd(i + 500, i) : 0.4 * rnd(std::min(i, j) + 900, std::max(i, j)) / (1.0 + 0.1 * std::abs(i - j));
      YY[(size_t)i * LB_MAXH + j] = yy;
    }
  UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, SY), SY.data(), SY.size() * sizeof(double), hipMemcpyHostToDevice));
  UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, YY), YY.data(), YY.size() * sizeof(double), hipMemcpyHostToDevice));
  std::vector<double> part((size_t)LB_MAXCHUNK * LB_ROWS * 3, 0.0);
  for (int r = 0; r < LB_ROWS; ++r)
    for (int c = 0; c < 3; ++c) part[(size_t)r * 3 + c] = (r == k - 1 && c == 0) ? 1.3 : 0.7 * rnd(r + 2000, c);  // chunk 0 only
  part[(size_t)(LB_MAXH + k - 1) * 3 + 0] = 2.9;  // y_new . y_new
  part[(size_t)(LB_MAXH + k - 1) * 3 + 1] = 1.3;  // y_new . s_new: the same number as s_new . y_new, as in a real run
  UUO_HIP_CHECK(hipMemcpy(w->part, part.data(), part.size() * sizeof(double), hipMemcpyHostToDevice));
  const int cap = LB_MAXH - 3, hist = LB_MAXH - 4;
  const int head = 0, count = k - 1;
  UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, head), &head, sizeof(int), hipMemcpyHostToDevice));
  UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, count), &count, sizeof(int), hipMemcpyHostToDevice));
  const double one = 1.0;
  UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, Hdiag), &one, sizeof(double), hipMemcpyHostToDevice));
  if (use_ref == 2) {
    // the state k_lb_small_inv expects: W = inverse of the upper triangle of S.Y^T over the k - 1 pairs already in the
    // window (slots 0..k-2), by back-substitution on the host; everything else in W is poisoned to catch stray reads
    const int m = k - 1;
    std::vector<double> W((size_t)LB_MAXH * LB_MAXH, std::nan(""));
    for (int c = 0; c < m; ++c) {
      std::vector<double> x(m, 0.0);
      for (int r = c; r >= 0; --r) {
        double acc = (r == c) ? 1.0 : 0.0;
        for (int q = r + 1; q <= c; ++q) acc -= SY[(size_t)r * LB_MAXH + q] * x[q];
        x[r] = acc / SY[(size_t)r * LB_MAXH + r];
      }
      for (int r = 0; r < m; ++r) W[(size_t)r * LB_MAXH + c] = x[r];
    }
    UUO_HIP_CHECK(hipMemcpy((char*)w->st + offsetof(LbDev, W), W.data(), W.size() * sizeof(double), hipMemcpyHostToDevice));
    { LbSmallArgs sa_{{1, 1}, LB_MAXCHUNK, cap, hist, k - 1, w->part, w->st, 0}; hipLaunchKernelGGL(k_lb_small_inv, dim3(1), dim3(512), 0, nullptr, sa_); }
  } else if (use_ref)
    hipLaunchKernelGGL(k_lb_small_ref, dim3(1), dim3(256), 0, nullptr, LB_MAXCHUNK, cap, hist, k - 1, w->part, w->st, 0);
  else
    hipLaunchKernelGGL(k_lb_small, dim3(1), dim3(512), 0, nullptr, LB_MAXCHUNK, cap, hist, k - 1, w->part, w->st, 0);
  UUO_HIP_CHECK(hipDeviceSynchronize());
  UUO_HIP_CHECK(hipMemcpy(out, (char*)w->st + offsetof(LbDev, cs), LB_MAXH * sizeof(double), hipMemcpyDeviceToHost));
  UUO_HIP_CHECK(hipMemcpy(out + LB_MAXH, (char*)w->st + offsetof(LbDev, cy), LB_MAXH * sizeof(double), hipMemcpyDeviceToHost));
  LbOut o;
  UUO_HIP_CHECK(hipMemcpy(&o, (char*)w->st + offsetof(LbDev, out), sizeof(LbOut), hipMemcpyDeviceToHost));
  out[2 * LB_MAXH] = o.gtd_dir;
  lbws_destroy(w);
  return 0;
}
#endif  // UUO_DEBUG_HOOKS
