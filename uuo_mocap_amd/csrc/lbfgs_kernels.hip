// L-BFGS on device-resident vectors: the kernels (see lbfgs_driver.hip for the algorithm they serve).
//  * the two-loop recursion is evaluated in coefficient space from Gram matrices of the (s, y) history
//    (two passes over the history per iteration instead of 4*k dependent dot/axpy launches),
//  * dot products accumulate in fp64,
//  * every kernel has two launch forms over one device body: k_X(XArgs) and k_X_b(const XArgs* batch) (uuo_common.h).
#include "lbfgs.h"

// a 16-byte read of the history: NON-TEMPORAL -- a pass streams its 36 MB exactly once, and with twelve solves in flight what
// those bytes displace from the L2s are the skinning kernel's basis slices and the backward's posedirs rows.  Same values,
// same arithmetic (bit-identical fits); alternating builds in one call, three rounds of 12 fits each: 5.305 / 5.442 / 5.354
// -> 5.429 / 5.477 / 5.382 M frame-evaluations/s, and with k_skin2's vertex stores non-temporal as well 5.437 / 5.322 / 5.365
// -> 5.441 / 5.466 / 5.487 (those stores alone: no change, left as they were) -- +1-2 % (profiles/r4_ab_nontemporal.log).
// -DLB_NT=0 restores plain loads.
#ifndef LB_NT
#define LB_NT 1
#endif
typedef float lb_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lb_ld_hist(const float* p) {
#if LB_NT
  const lb_f4 v = __builtin_nontemporal_load(reinterpret_cast<const lb_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
#include "frame_math.h"

#ifndef LB_PASS_PRIO
#define LB_PASS_PRIO 1  // wave priority of the two history passes (0..3)
#endif
#ifndef LB_ACC32
#define LB_ACC32 false  // true: the history passes on packed fp32 partial sums (see lb_dots_body: measured in round 4, not adopted)
#endif

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

__global__ void k_lb_neg(LbNegArgs a) { lb_neg_body(a.n, a.g, a.d, a.x, a.t, a.xt, a.map); }
__global__ void k_lb_neg_b(const LbNegArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbNegArgs, batch)
  lb_neg_body(a.n, a.g, a.d, a.x, a.t, a.xt, a.map);
}
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

// rows of the history (and g) against {y_new, s_new, g}: skinny GEMM, fp64 accumulation.
// grid = (column groups, LB_DRS row splits).  A block walks the column blocks of its group; per column block every
// wave loads the three right-hand vectors (y_new = g - g_prev and s_new = t d are formed on the fly and stored to
// the candidate slot by split 0) and the 512-column pieces of its rows (row r belongs to wave r mod 4*LB_DRS), all
// loads of a column block in flight together.  Per-lane fp64 accumulators, one wave reduction per row at the end.
#define LB_DRW ((LB_ROWS + 4 * LB_DRS - 1) / (4 * LB_DRS))  // rows per wave (4)
typedef float lbf2 __attribute__((ext_vector_type(2)));
// ACC32 (round 4; NOT the default): the per-lane partial sums are fp32 pairs updated by packed FMAs
// (v_pk_fma_f32: two history elements per instruction); a lane adds <= 4 * gcb products per accumulator component before the
// sums continue in fp64 (lane pair, wave, column groups) -- torch.optim.LBFGS's own dots are fp32 throughout
// (lbfgs.py:396-441).  Measured (tools/lb_pass_bench.hip, profiles/r4_lb_pass_bench.log): 12.6 -> 10.8 us with twelve
// histories cycling through HBM, 10.3 -> 7.7 us out of the Infinity Cache, against 7.5 / 3.8 us for a plain read of the same
// bytes; in the fit (alternating runs, 3 x 9 sequences each) 5.32 vs 5.31 M frame-evaluations/s -- nothing.  And fp32 sums in
// another lane order are another trajectory: the solves of the small reference fixtures ended outside the bands the
// end-to-end tests hold (5 of 125 GPU tests).  ACC32 = false is the fp64 accumulation of rounds 1-3.
template <bool ACC32>
__device__ __forceinline__ void lb_dots_body(int n, int cap, int capL, int head, int count, int cand,
                                                  float* __restrict__ S, float* __restrict__ Y,
                                                  const float* __restrict__ g, const float* __restrict__ gp,
                                                  const float* __restrict__ d, float t, int ncb, int gcb,
                                                  double* __restrict__ part /* [groups][LB_ROWS][3] */,
                                                  int skip_lo = 0, int skip_hi = 0) {
  __builtin_amdgcn_s_setprio(LB_PASS_PRIO);  // do not queue behind co-resident MFMA waves
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
  lbf2 acc2[LB_DRW][3];
#pragma unroll
  for (int q = 0; q < LB_DRW; ++q) {
    acc[q][0] = acc[q][1] = acc[q][2] = 0.0;
    acc2[q][0] = acc2[q][1] = acc2[q][2] = lbf2{0.f, 0.f};
  }
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
    for (int q = 0; q < LB_DRW; ++q) nr[q] = lb_ld_hist(rptr[q] + cboff);
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
    if constexpr (ACC32) {
      const lbf2 ylo{vy.x, vy.y}, yhi{vy.z, vy.w}, slo{vs.x, vs.y}, shi{vs.z, vs.w}, glo{vg.x, vg.y}, ghi{vg.z, vg.w};
#pragma unroll
      for (int q = 0; q < LB_DRW; ++q) {
        const lbf2 rlo{rv[q].x, rv[q].y}, rhi{rv[q].z, rv[q].w};
        acc2[q][0] = __builtin_elementwise_fma(rhi, yhi, __builtin_elementwise_fma(rlo, ylo, acc2[q][0]));
        acc2[q][1] = __builtin_elementwise_fma(rhi, shi, __builtin_elementwise_fma(rlo, slo, acc2[q][1]));
        acc2[q][2] = __builtin_elementwise_fma(rhi, ghi, __builtin_elementwise_fma(rlo, glo, acc2[q][2]));
      }
    } else {
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
    if constexpr (ACC32) {
#pragma unroll
      for (int c_ = 0; c_ < 3; ++c_) acc[q][c_] = (double)acc2[q][c_].x + (double)acc2[q][c_].y;
    }
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

__global__ __launch_bounds__(256) void k_lb_dots(LbDotsArgs a) {
  lb_dots_body<LB_ACC32>(a.n, a.cap, a.capL, a.head, a.count, a.cand, a.S, a.Y, a.g, a.gp, a.d, a.t, a.ncb, a.gcb, a.part, a.skip_lo,
               a.skip_hi);
}
__global__ __launch_bounds__(256) void k_lb_dots_b(const LbDotsArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbDotsArgs, batch)
  lb_dots_body<LB_ACC32>(a.n, a.cap, a.capL, a.head, a.count, a.cand, a.S, a.Y, a.g, a.gp, a.d, a.t, a.ncb, a.gcb, a.part, a.skip_lo,
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

#define LB_YR ((LB_MAXH + 2) / 3)  // rows of Y.Y^T per helper wave
#define LB_US (LB_MAXH + 1)  // padded row stride of the LDS copy of the window's inverse factor (bank spread)
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

__global__ __launch_bounds__(512) void k_lb_small_inv(LbSmallArgs a) {
  lb_small_inv_body(a.nchunks, a.cap, a.hist, a.cand, a.part, a.st, a.stop, a.rd_in);
}
__global__ __launch_bounds__(512) void k_lb_small_inv_b(const LbSmallArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbSmallArgs, batch)
  lb_small_inv_body(a.nchunks, a.cap, a.hist, a.cand, a.part, a.st, a.stop, a.rd_in);
}
template <bool ACC32>
__device__ __forceinline__ void lb_direction_body(int n, int cap, int capL, const float* __restrict__ S,
                                                             const float* __restrict__ Y, const float* __restrict__ g,
                                                             LbDev* __restrict__ st, float* __restrict__ d,
                                                             const float* __restrict__ x, float t, float* __restrict__ xt,
                                                             const UuoIndexMap& map) {
  __builtin_amdgcn_s_setprio(LB_PASS_PRIO);  // do not queue behind co-resident MFMA waves
  // d = cg g + sum_j cy_j y_j + cs_j s_j, max|d|, and the first line-search trial point xt = x + t d in the same pass.
  // One block per 256-column strip of the history (FOUR columns per lane: 16-byte loads -- the vector-memory pipe costs
  // ~16 cycles per wave instruction whatever its width, and 8-byte loads made this kernel bound by it).  The strip's
  // slots are one contiguous region per column block; they are split into LB_DQ consecutive ranges, one wave each, two
  // batches of 8 slots (16 loads) in flight per lane -- with one wave per strip the kernel had 8 MB of loads in flight
  // on the whole chip and streamed the 53 MB of history at 3.6 TB/s; four waves per strip quadruple that.  The ranges'
  // fp64 partial sums are added in range order by wave 0 (fixed order: deterministic).
  // ACC32: coefficients rounded to fp32 and the combination accumulated by packed fp32 FMAs (two columns per instruction) --
  // torch forms d by 2 k fp32 axpys (lbfgs.py:396-441); the LB_DQ range sums and cg g are added in fp64, in range order.
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
        yv[buf][u] = lb_ld_hist(Yb + off);
        sv[buf][u] = lb_ld_hist(Sb + off);
      }
    };
    lbf2 alo{0.f, 0.f}, ahi{0.f, 0.f};
    auto consume = [&](int j0, int buf) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (ACC32) {
          const float cyf = (float)scy[j0 + u], csf = (float)scs[j0 + u];
          const lbf2 cy2{cyf, cyf}, cs2{csf, csf};
          alo = __builtin_elementwise_fma(cy2, lbf2{yv[buf][u].x, yv[buf][u].y}, alo);
          ahi = __builtin_elementwise_fma(cy2, lbf2{yv[buf][u].z, yv[buf][u].w}, ahi);
          alo = __builtin_elementwise_fma(cs2, lbf2{sv[buf][u].x, sv[buf][u].y}, alo);
          ahi = __builtin_elementwise_fma(cs2, lbf2{sv[buf][u].z, sv[buf][u].w}, ahi);
        } else {
          const double cy = scy[j0 + u], cs = scs[j0 + u];
          acc[0] += cy * (double)yv[buf][u].x + cs * (double)sv[buf][u].x;
          acc[1] += cy * (double)yv[buf][u].y + cs * (double)sv[buf][u].y;
          acc[2] += cy * (double)yv[buf][u].z + cs * (double)sv[buf][u].z;
          acc[3] += cy * (double)yv[buf][u].w + cs * (double)sv[buf][u].w;
        }
      }
    };
    if (jlo < jhi) issue(jlo, 0);
    for (int j0 = jlo; j0 < jhi; j0 += 16) {
      if (j0 + 8 < jhi) issue(j0 + 8, 1);
      consume(j0, 0);
      if (j0 + 16 < jhi) issue(j0 + 16, 0);
      if (j0 + 8 < jhi) consume(j0 + 8, 1);
    }
    if constexpr (ACC32) {
      acc[0] = (double)alo.x; acc[1] = (double)alo.y; acc[2] = (double)ahi.x; acc[3] = (double)ahi.y;
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

__global__ __launch_bounds__(64 * LB_DQ) void k_lb_direction(LbDirArgs a) {
  lb_direction_body<LB_ACC32>(a.n, a.cap, a.capL, a.S, a.Y, a.g, a.st, a.d, a.x, a.t, a.xt, a.map);
}
__global__ __launch_bounds__(64 * LB_DQ) void k_lb_direction_b(const LbDirArgs* __restrict__ batch) {
  UUO_BATCH_PICK(LbDirArgs, batch)
  lb_direction_body<LB_ACC32>(a.n, a.cap, a.capL, a.S, a.Y, a.g, a.st, a.d, a.x, a.t, a.xt, a.map);
}


// ---------------------------------------------------------------------------------------------------- launchers
void uuo_lb_launch_init(hipStream_t s, LbDev* st) { hipLaunchKernelGGL(k_lb_init, dim3(1), dim3(1), 0, s, st); }
void uuo_lb_launch_neg(hipStream_t s, dim3 grid, const LbNegArgs& a) { hipLaunchKernelGGL(k_lb_neg, grid, dim3(256), 0, s, a); }
void uuo_lb_launch_axpy(hipStream_t s, dim3 grid, const LbAxpyArgs& a) { hipLaunchKernelGGL(k_lb_axpy, grid, dim3(256), 0, s, a); }
void uuo_lb_launch_dots(hipStream_t s, dim3 grid, const LbDotsArgs& a) { hipLaunchKernelGGL(k_lb_dots, grid, dim3(256), 0, s, a); }
void uuo_lb_launch_small(hipStream_t s, const LbSmallArgs& a) { hipLaunchKernelGGL(k_lb_small_inv, dim3(1), dim3(512), 0, s, a); }
void uuo_lb_launch_direction(hipStream_t s, dim3 grid, const LbDirArgs& a) {
  hipLaunchKernelGGL(k_lb_direction, grid, dim3(64 * LB_DQ), 0, s, a);
}
void uuo_lb_launch_rows(hipStream_t s, int nchunks, int cap, int cand, const double* part, const LbDev* st, double* host_rows,
                        unsigned long long seq) {
  hipLaunchKernelGGL(k_lb_rows, dim3(1), dim3(512), 0, s, nchunks, cap, cand, part, st, host_rows, seq);
}
void uuo_lb_launch_stats(hipStream_t s, int nstat, int n, const float* g, const float* d, double* part, const float* loss,
                         LbDev* st) {
  hipLaunchKernelGGL(k_lb_stats, dim3(nstat), dim3(256), 0, s, n, g, d, part);
  hipLaunchKernelGGL(k_lb_stats_final, dim3(1), dim3(64), 0, s, nstat, part, loss, st);
}
int uuo_batched_launch_lbfgs(int op, hipStream_t s, const void* da, int n_, int gx, int gy) {
  switch (op) {
    case UUO_OP_AXPY_ACCEPT:
    case UUO_OP_AXPY:
      hipLaunchKernelGGL(k_lb_axpy_b, dim3(gx, gy, n_), dim3(256), 0, s, (const LbAxpyArgs*)da);
      return 0;
    case UUO_OP_NEG:
      hipLaunchKernelGGL(k_lb_neg_b, dim3(gx, gy, n_), dim3(256), 0, s, (const LbNegArgs*)da);
      return 0;
    case UUO_OP_DOTS:
      hipLaunchKernelGGL(k_lb_dots_b, dim3(gx, gy, n_), dim3(256), 0, s, (const LbDotsArgs*)da);
      return 0;
    case UUO_OP_SMALL:
      hipLaunchKernelGGL(k_lb_small_inv_b, dim3(gx, gy, n_), dim3(512), 0, s, (const LbSmallArgs*)da);
      return 0;
    case UUO_OP_DIR:
      hipLaunchKernelGGL(k_lb_direction_b, dim3(gx, gy, n_), dim3(64 * LB_DQ), 0, s, (const LbDirArgs*)da);
      return 0;
    default:
      return 1;
  }
}
