// Debug flavour only (libuuo_hip_debug.so, -DUUO_DEBUG_HOOKS; tests/ and tools/): cross-check kernels of the coefficient
// step, the optimiser's self-test objectives and timing hooks.  Nothing of this file is in the product library.
#ifdef UUO_DEBUG_HOOKS
#include "lbfgs.h"
#include "frame_math.h"

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


void uuo_debug_launch_small(int kind, hipStream_t s, int nchunks, int cap, int hist, int cand, const double* part, LbDev* st,
                            int stop) {
  if (kind == 1)
    hipLaunchKernelGGL(k_lb_small_ref, dim3(1), dim3(256), 0, s, nchunks, cap, hist, cand, part, st, stop);
  else
    hipLaunchKernelGGL(k_lb_small, dim3(1), dim3(512), 0, s, nchunks, cap, hist, cand, part, st, stop);
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
      { LbSmallArgs sa_{{1, 1}, LB_MAXCHUNK, cap, hist, k - 1, w->part, w->st, stop - 200}; uuo_lb_launch_small(nullptr, sa_); }
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
  uuo_lb_launch_init(s, st);
  UUO_HIP_CHECK(hipStreamSynchronize(s));
  timespec t0, t1, t2;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < count; ++i) uuo_lb_launch_init(s, st);
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
    { LbSmallArgs sa_{{1, 1}, LB_MAXCHUNK, cap, hist, k - 1, w->part, w->st, 0}; uuo_lb_launch_small(nullptr, sa_); }
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
