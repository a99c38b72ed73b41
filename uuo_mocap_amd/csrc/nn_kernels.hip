// Nearest-neighbour assignment kernels: K=1 search (chamfer data term) and marker placement.
#include <cstdlib>

#include "frame_math.h"

#define UUO_INF __builtin_huge_valf()

// IEEE square root and division, correctly rounded.  HIP's __fsqrt_rn is NOT that: without OCML_BASIC_ROUNDED_OPERATIONS it
// is __ocml_native_sqrt_f32 (v_sqrt_f32, 1 ulp) -- found in round 3 when the rigidity matrix differed from numpy's in a third
// of its entries by one ulp.  The plain operators are correctly rounded in HIP device code
// (-fhip-fp32-correctly-rounded-divide-sqrt is the compiler's default and this library never turns it off).
__device__ __forceinline__ float uuo_sqrt_rn(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float uuo_div_rn(float x, float y) { return x / y; }

__device__ __forceinline__ unsigned long long pack_key(float d, unsigned idx) {
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)idx;
}

// squared distance exactly as pytorch3d's CPU loop: ((0 + dx*dx) + dy*dy) + dz*dz, every op rounded separately
__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = __fsub_rn(qx, px), dy = __fsub_rn(qy, py), dz = __fsub_rn(qz, pz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ----------------------------------------------------------------------------------------------------
// K=1 nearest neighbour.  One wave per (cloud n, 64-query group, candidate split).  Lane = query; the
// candidate tile is staged in LDS (coalesced load, broadcast reads), the running (dist, index) minimum
// lives in registers, candidates are visited in ascending order with a strict '<' so the first index wins
// inside a split; splits are merged with a 64-bit atomicMin on (dist bits << 32 | index), which is the
// same lexicographic order and is exact and order-independent.
// Replaces pytorch3d knn_points(K=1) behind chamfer_distance (reference losses/chamfer_distance.py:15-20).
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void nn_body(const float* __restrict__ x, const float* __restrict__ y,
                                        const int* __restrict__ ysub, int P1, int P2, int nc, int S,
                                        unsigned long long* __restrict__ out, int n, int qg, int s) {
  __shared__ float4 sc[64];
  const int lane = threadIdx.x;
  const int q = qg * 64 + lane;
  const bool valid = q < P1;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (valid) {
    const float* px = x + ((size_t)n * P1 + q) * 3;
    qx = px[0];
    qy = px[1];
    qz = px[2];
  }
  const int per = (nc + S - 1) / S;
  const int c0 = s * per;
  const int c1 = min(nc, c0 + per);
  float best = UUO_INF;
  unsigned besti = 0xFFFFFFFFu;
  for (int base = c0; base < c1; base += 64) {
    const int c = base + lane;
    float4 p = make_float4(UUO_INF, UUO_INF, UUO_INF, 0.f);
    if (c < c1) {
      const int vi = ysub ? ysub[c] : c;
      const float* py = y + ((size_t)n * P2 + vi) * 3;
      p = make_float4(py[0], py[1], py[2], 0.f);
    }
    __syncthreads();
    sc[lane] = p;
    __syncthreads();
#pragma unroll 8
    for (int t = 0; t < 64; ++t) {
      const float4 cnd = sc[t];
      const float d = sqdist(qx, qy, qz, cnd.x, cnd.y, cnd.z);
      if (d < best) {
        best = d;
        besti = (unsigned)(base + t);
      }
    }
  }
  if (valid && besti != 0xFFFFFFFFu) atomicMin(&out[(size_t)n * P1 + q], pack_key(best, besti));
}

struct NnArgs {
  UuoGridHdr h;  // batched form: gy = query groups x splits
  uuo_gptr<const float> x;
  uuo_gptr<const float> y;
  uuo_gptr<const int> ysub;
  int P1, P2, nc, S;
  uuo_gptr<unsigned long long> out;
};
__global__ __launch_bounds__(64) void k_nn(NnArgs a) {
  nn_body(a.x, a.y, a.ysub, a.P1, a.P2, a.nc, a.S, a.out, blockIdx.x, blockIdx.y, blockIdx.z);
}
__global__ __launch_bounds__(64) void k_nn_b(const NnArgs* __restrict__ batch) {
  UUO_BATCH_PICK(NnArgs, batch)
  nn_body(a.x, a.y, a.ysub, a.P1, a.P2, a.nc, a.S, a.out, blockIdx.x, (int)blockIdx.y / a.S, (int)blockIdx.y % a.S);
}
struct FillArgs {  // UUO_OP_FILL: all-ones fill of the packed (distance, index) keys before a split search
  UuoGridHdr h;
  uuo_gptr<unsigned long long> p;
  int count;
};
__global__ void k_fill_keys_b(const FillArgs* __restrict__ batch) {
  UUO_BATCH_PICK(FillArgs, batch)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.count) a.p[i] = ~0ull;
}

// Few queries per cloud (partial marker sets: P1 <= 16): lane = query would leave most of the wave idle, so here
// lane = candidate.  Every lane keeps the running (dist, index) key of each query over its candidates
// (c = lane, lane + 64, ... in ascending order, strict '<' on the packed key = first index on ties), the wave merges
// with 64-bit minima and one atomicMin per query and split.  Same packed-key order as k_nn: bit-identical results.
#define NNQ_MAX 16
__device__ __forceinline__ void nn_fewq_body(const float* __restrict__ x, const float* __restrict__ y,
                                             const int* __restrict__ ysub, int P1, int P2, int nc, int S,
                                             unsigned long long* __restrict__ out) {
  __shared__ float sq[NNQ_MAX * 3];
  __shared__ unsigned long long sk[NNQ_MAX];  // the block's minima: with one split they ARE the result (no zero-fill
                                              // of the output, no global atomic)
  const int n = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  if (tid < P1 * 3) sq[tid] = x[(size_t)n * P1 * 3 + tid];
  if (tid < NNQ_MAX) sk[tid] = ~0ull;
  __syncthreads();
  const int per = (nc + S - 1) / S;
  const int c0 = s * per, c1 = min(nc, c0 + per);
  unsigned long long best[NNQ_MAX];
#pragma unroll
  for (int q = 0; q < NNQ_MAX; ++q) best[q] = ~0ull;
  for (int c = c0 + tid; c < c1; c += 256) {
    const int vi = ysub ? ysub[c] : c;
    const float* py = y + ((size_t)n * P2 + vi) * 3;
    const float px = py[0], pyv = py[1], pz = py[2];
#pragma unroll
    for (int q = 0; q < NNQ_MAX; ++q) {
      if (q < P1) {  // block-uniform
        const unsigned long long key = pack_key(sqdist(sq[q * 3], sq[q * 3 + 1], sq[q * 3 + 2], px, pyv, pz), (unsigned)c);
        best[q] = key < best[q] ? key : best[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NNQ_MAX; ++q) {
    if (q < P1) {
      unsigned long long k = best[q];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off, 64);
        k = o < k ? o : k;
      }
      if ((tid & 63) == 0 && k != ~0ull) atomicMin(&sk[q], k);
    }
  }
  __syncthreads();
  if (tid < P1) {
    if (S == 1)
      out[(size_t)n * P1 + tid] = sk[tid];
    else if (sk[tid] != ~0ull)
      atomicMin(&out[(size_t)n * P1 + tid], sk[tid]);
  }
}

__global__ __launch_bounds__(256) void k_nn_fewq(NnArgs a) { nn_fewq_body(a.x, a.y, a.ysub, a.P1, a.P2, a.nc, a.S, a.out); }
__global__ __launch_bounds__(256) void k_nn_fewq_b(const NnArgs* __restrict__ batch) {
  UUO_BATCH_PICK(NnArgs, batch)
  nn_fewq_body(a.x, a.y, a.ysub, a.P1, a.P2, a.nc, a.S, a.out);
}

// ----------------------------------------------------------------------------------------------------
// Part stage (constant body pose, P1 <= 16 markers per frame): skinning of the candidate's vertices fused with the
// nearest-vertex search.  The vertex positions only exist in registers:
//   v = T_f(v) (C[f][v] + SB[v]) + trans_f,   T_f(v) = sum_n w_n A_f[j_n]        (skin_cached_body's arithmetic)
// with C the cached template + pose-corrective blend of the frame; S[v] . beta, the joints and the weights of the vertex
// come packed in subset order from k_pose_prep's tail (coalesced loads instead of three gathers per vertex); each
// lane keeps the (distance, candidate) minimum of every marker over its vertices, the wave merges them into the packed
// keys k_nn_fewq would write (same order, same tie rule).  The backward kernel re-skins the winners itself, so nothing but the P1 keys
// per frame is written: 12 B of C per vertex and frame instead of 12 B read + 12 B written + 12 B read again.
// One wave per (candidate, frame) - the kernel is bound by instruction issue, and the per-frame work that does not scale
// with the vertices (staging, 2 * P1 cross-lane minima) is paid once per wave; consecutive blocks of an XCD walk the candidates of ONE frame, so the frame's slice
// of C (82 KB, shared by all candidates of a lock-step batch) is fetched into that XCD's L2 once.
// ----------------------------------------------------------------------------------------------------
struct PartFwdArgs {
  UuoGridHdr h;  // gx = F
  int F, V, ns, P1;
  uuo_gptr<const int32_t> subset;
  uuo_gptr<const float> C;
  uuo_gptr<const float> SB;  // [ns][8]: shape offset | packed joints | weights (k_pose_prep's tail)
  uuo_gptr<const float> A;
  uuo_gptr<const float> trans;
  uuo_gptr<const float> x;
  uuo_gptr<unsigned long long> out;
};
// minimum over the wave's 64 lanes, uniform result: 4 DPP steps inside each row of 16 lanes, then the 4 row results
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_u32<0x141>(v));  // row_half_mirror
  v = min(v, dpp_u32<0x140>(v));  // row_mirror: every lane of a row holds the row's minimum
  const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return min(min(r0, r1), min(r2, r3));
}
typedef float pf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pfw_uniform(float v) {  // a wave-uniform value, kept in a scalar register
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
#define PFW_U 4  // candidates per lane in flight (their loads are issued together)
#define PFW_T 64  // one wave per (candidate, frame): one set of cross-lane minima per frame, no block-level merge
// QB = the markers per frame (the running minima live in 2 * QB registers; one instantiation per count, no per-marker branch)
template <int QB>
__device__ __forceinline__ void part_fwd_body(const PartFwdArgs& a, int f) {
  __shared__ __align__(16) float sA[UUO_NUM_JOINTS * 12];
  const int tid = threadIdx.x;
  constexpr int P1 = QB;
  // the lane's first candidates: their vertex ids are on their way while the skinning matrices are staged
  // (past the end of the subset a lane repeats the last candidate: the same (distance, index) pair again changes no
  // minimum, and the loop needs no validity mask)
  const int last = a.ns - 1;
  int cc[PFW_U], vv[PFW_U];
#pragma unroll
  for (int u = 0; u < PFW_U; ++u) {
    cc[u] = min(tid + PFW_T * u, last);
    vv[u] = a.subset[cc[u]];
  }
  {
    const float4* Af = reinterpret_cast<const float4*>(a.A + (size_t)f * UUO_NUM_JOINTS * 12);
    for (int i = tid; i < UUO_NUM_JOINTS * 3; i += PFW_T) reinterpret_cast<float4*>(sA)[i] = Af[i];
  }
  // the frame's markers and translation are block-uniform: they live in SGPRs, the markers as pairs (the distance
  // arithmetic runs two markers per instruction on the packed-fp32 pipe: same IEEE operations, element by element)
  constexpr int QP = (QB + 1) / 2;  // an odd count repeats its last marker in the spare half
  pf2 qx[QP], qy[QP], qz[QP];
  float sTr[3];
  const float* xq = a.x + (size_t)f * P1 * 3;
#pragma unroll
  for (int k = 0; k < QP; ++k) {
    const int q0 = 2 * k, q1 = (2 * k + 1 < QB) ? 2 * k + 1 : QB - 1;
    qx[k] = pf2{pfw_uniform(xq[q0 * 3]), pfw_uniform(xq[q1 * 3])};
    qy[k] = pf2{pfw_uniform(xq[q0 * 3 + 1]), pfw_uniform(xq[q1 * 3 + 1])};
    qz[k] = pf2{pfw_uniform(xq[q0 * 3 + 2]), pfw_uniform(xq[q1 * 3 + 2])};
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) sTr[i] = a.trans ? pfw_uniform(a.trans[(size_t)f * 3 + i]) : 0.f;
  // the distance bits of a (non-negative) squared distance order as unsigned exactly as the packed key does; a lane sees
  // its candidates in ascending order, so strict '<' keeps the first index on ties, as the packed comparison would
  unsigned bd[QB], bi[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) bd[q] = bi[q] = 0xFFFFFFFFu;
  const float* Cf = a.C + (size_t)f * a.V * 3;
  const float4* SB4 = reinterpret_cast<const float4*>(a.SB.get());
  __syncthreads();
  for (int c0 = tid; c0 < a.ns; c0 += PFW_T * PFW_U) {
    float p[PFW_U][3];
    unsigned pj[PFW_U];
    float4 ww[PFW_U];
#pragma unroll
    for (int u = 0; u < PFW_U; ++u) {
      const unsigned v = min((unsigned)vv[u], (unsigned)(a.V - 1)), c = (unsigned)cc[u];  // (memory safety only)
      const float* pc = Cf + v * 3u;
      const float4 k0 = SB4[c * 2u];
      ww[u] = SB4[c * 2u + 1u];
      p[u][0] = pc[0] + k0.x;
      p[u][1] = pc[1] + k0.y;
      p[u][2] = pc[2] + k0.z;
      pj[u] = __float_as_uint(k0.w);
    }
    int cn[PFW_U], vn[PFW_U];  // the next round's vertex ids
#pragma unroll
    for (int u = 0; u < PFW_U; ++u) {
      cn[u] = min(c0 + PFW_T * (PFW_U + u), last);
      vn[u] = a.subset[cn[u]];
    }
#pragma unroll
    for (int u = 0; u < PFW_U; ++u) {
      const float wv[4] = {ww[u].x, ww[u].y, ww[u].z, ww[u].w};
      pf2 T2[6];  // the blended 3x4 transform, two entries per register pair
#pragma unroll
      for (int e = 0; e < 6; ++e) T2[e] = pf2{0.f, 0.f};
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        // byte n of pj = 3 * joint: the row's float4 index (an unused slot is joint 0 with weight 0: adds exact zeros)
        const float4* pa = reinterpret_cast<const float4*>(sA) + ((pj[u] >> (8 * n)) & 0xFFu);
        const float4 r0 = pa[0], r1 = pa[1], r2 = pa[2];
        const pf2 ar[6] = {pf2{r0.x, r0.y}, pf2{r0.z, r0.w}, pf2{r1.x, r1.y}, pf2{r1.z, r1.w}, pf2{r2.x, r2.y}, pf2{r2.z, r2.w}};
        const pf2 w2 = pf2{wv[n], wv[n]};
#pragma unroll
        for (int e = 0; e < 6; ++e) T2[e] = __builtin_elementwise_fma(w2, ar[e], T2[e]);
      }
      const float px = p[u][0], py = p[u][1], pz = p[u][2];
      const float ox = fmaf(T2[1].x, pz, fmaf(T2[0].y, py, T2[0].x * px)) + T2[1].y + sTr[0];
      const float oy = fmaf(T2[3].x, pz, fmaf(T2[2].y, py, T2[2].x * px)) + T2[3].y + sTr[1];
      const float oz = fmaf(T2[5].x, pz, fmaf(T2[4].y, py, T2[4].x * px)) + T2[5].y + sTr[2];
      const pf2 ox2 = pf2{ox, ox}, oy2 = pf2{oy, oy}, oz2 = pf2{oz, oz};
#pragma unroll
      for (int k = 0; k < QP; ++k) {
        // sqdist() for two markers: ((dx*dx) + (dy*dy)) + (dz*dz), every operation rounded separately
        const pf2 dx = qx[k] - ox2, dy = qy[k] - oy2, dz = qz[k] - oz2;
        const pf2 d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = 2 * k + h;
          if (q < QB) {
            const unsigned d = __float_as_uint(h ? d2.y : d2.x);
            const bool better = d < bd[q];
            bd[q] = better ? d : bd[q];
            bi[q] = better ? (unsigned)cc[u] : bi[q];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PFW_U; ++u) {
      cc[u] = cn[u];
      vv[u] = vn[u];
    }
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const unsigned dmin = wave_min_u32(bd[q]);
    const unsigned imin = wave_min_u32(bd[q] == dmin ? bi[q] : 0xFFFFFFFFu);
    if (tid == 0) a.out[(size_t)f * P1 + q] = ((unsigned long long)dmin << 32) | (unsigned long long)imin;
  }
}
// block id -> (candidate, frame): the 8 XCDs take blocks round-robin; block b of XCD (b & 7) works on frame
// 8 * ((b >> 3) / count) + (b & 7) of candidate (b >> 3) % count
template <int QB>
__global__ __launch_bounds__(PFW_T) void k_part_fwd(PartFwdArgs a) {
  const int f = (int)(blockIdx.x >> 3) * 8 + (int)(blockIdx.x & 7);
  if (f >= a.F) return;
  part_fwd_body<QB>(a, f);
}
template <int QB>
__global__ __launch_bounds__(PFW_T) void k_part_fwd_b(const PartFwdArgs* __restrict__ batch, int count) {
  const int r = (int)(blockIdx.x >> 3);
  const PartFwdArgs a = batch[r % count];
  const int f = (r / count) * 8 + (int)(blockIdx.x & 7);
  if (f >= a.F) return;
  part_fwd_body<QB>(a, f);
}

int uuo_launch_part_fwd(const uuo_model* m, hipStream_t s, int F, int P1, const float* cache, const float* sb, const float* A,
                        const float* trans, const int32_t* subset, int n_subset, const float* markers,
                        unsigned long long* packed) {
  UUO_REQUIRE(P1 >= 1 && P1 <= NNQ_MAX && subset && n_subset > 0 && F > 0, "uuo_launch_part_fwd: bad arguments");
  PartFwdArgs a{{F, P1}, F, m->V, n_subset, P1, subset, cache, sb, A, trans, markers, packed};
  if (uuo_record(UUO_OP_PART_FWD, F, P1, a)) return 0;  // P1 rides in the record's gy
  const dim3 grid(8 * ((F + 7) / 8));
  switch (P1) {
#define PFW_CASE(Q) case Q: hipLaunchKernelGGL(k_part_fwd<Q>, grid, dim3(PFW_T), 0, s, a); break;
    PFW_CASE(1) PFW_CASE(2) PFW_CASE(3) PFW_CASE(4) PFW_CASE(5) PFW_CASE(6) PFW_CASE(7) PFW_CASE(8)
    PFW_CASE(9) PFW_CASE(10) PFW_CASE(11) PFW_CASE(12) PFW_CASE(13) PFW_CASE(14) PFW_CASE(15) PFW_CASE(16)
#undef PFW_CASE
  }
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// EXTENSION (not reference behaviour; BASELINE configs[2] "hmr_part.yaml, soft-assignment path"): the part stage's data term
// with a SOFT assignment of every marker to the candidate's vertices, fused like k_part_fwd (vertices only in registers):
//   loss_data = (1 / (F M)) sum_{f,m} [ w_hard min_v d2_fmv  +  w_soft ( -tau log sum_v exp(-d2_fmv / tau) ) ]
// A soft assignment makes the backward DENSE -- every vertex of the candidate receives gradient
//   g_v = - sum_m (c_soft p_mv + c_hard [v = argmin_m]) (x_m - v),   p_mv = exp((dmin_m - d2_mv)/tau) / S_m
// -- so it cannot be the <= 16-item gather of k_bwd_part.  But the part stage optimises a yaw about the root joint, a
// translation per frame and the shape only (markers/markers_utils.py:416-434), and for those the dense backward collapses
// to per-frame sums over the vertices that this kernel forms while it holds the vertex:
//   d trans_f = sum_v g_v;   torque_z = sum_v (v - trans_f) x g_v  (the yaw: d v / d z = e_z x (v - root joint));
//   d beta (blend-shape path) = sum_v S_v^T T_v^R^T g_v;   joint forces F_j = sum_v w_vj g_v  (= d A_j^t: the shape's joint path).
// They go to a [F][UUO_PRE] record; k_bwd_part in `pre` mode skips its item loop and runs its kinematic tail on them.
// One wave per (candidate, frame), two passes over the candidate's vertices: pass 1 keeps per lane and marker the running
// (minimum, first index, sum of exp relative to that minimum) with ONE exponential per pair (online soft-min), merged over
// the wave in a fixed order; pass 2 re-skins, forms p_mv and accumulates.  No float atomics: the joint forces are
// accumulated in per-lane LDS columns and reduced in a fixed order, everything else by DPP wave sums: bit-reproducible.
// ----------------------------------------------------------------------------------------------------
struct PartSoftArgs {
  UuoGridHdr h;  // gx = F
  int F, V, ns, P1;
  uuo_gptr<const int32_t> subset;
  uuo_gptr<const float> C;
  uuo_gptr<const float> SB;   // [ns][8] (k_pose_prep's tail, as for k_part_fwd)
  uuo_gptr<const float> A;
  uuo_gptr<const float> trans;
  uuo_gptr<const float> x;
  uuo_gptr<const float> ST;   // [V][3][10] shape basis
  uuo_gptr<unsigned long long> out;  // [F][P1] packed (min distance bits << 32 | first candidate)
  uuo_gptr<float> pre;        // [F][UUO_PRE]
  float kexp;                 // log2(e) / tau
  float tau_ln2;              // tau * ln 2
  float w_hard, w_soft;       // loss weights
  float c_hard, c_soft;       // 2 w / (F M)
};
#define PSO_U 2  // vertices per lane in flight
#define PSO_EXP(x_) __builtin_amdgcn_exp2f(x_)  // (ablation of the round: without the exponentials the kernel is 1 % shorter, without
                                                // pass 2's per-vertex sums 25 %: it is bound by vector issue, not by the transcendental unit)
// the candidate's vertex c (clamped id cl) of frame f: position o, blended transform T2 (3x4 row-major, as pairs)
__device__ __forceinline__ void pso_skin(const float* __restrict__ Cf, const float4* __restrict__ SB4, const float* sA,
                                         const float* sTr, unsigned v, unsigned cl, float* o, pf2* T2, float4& ww, unsigned& pj) {
  const float* pc = Cf + v * 3u;
  const float4 k0 = SB4[cl * 2u];
  ww = SB4[cl * 2u + 1u];
  const float px = pc[0] + k0.x, py = pc[1] + k0.y, pz = pc[2] + k0.z;
  pj = __float_as_uint(k0.w);
  const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
  for (int e = 0; e < 6; ++e) T2[e] = pf2{0.f, 0.f};
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float4* pa = reinterpret_cast<const float4*>(sA) + ((pj >> (8 * n)) & 0xFFu);
    const float4 r0 = pa[0], r1 = pa[1], r2 = pa[2];
    const pf2 ar[6] = {pf2{r0.x, r0.y}, pf2{r0.z, r0.w}, pf2{r1.x, r1.y}, pf2{r1.z, r1.w}, pf2{r2.x, r2.y}, pf2{r2.z, r2.w}};
    const pf2 w2 = pf2{wv[n], wv[n]};
#pragma unroll
    for (int e = 0; e < 6; ++e) T2[e] = __builtin_elementwise_fma(w2, ar[e], T2[e]);
  }
  o[0] = fmaf(T2[1].x, pz, fmaf(T2[0].y, py, T2[0].x * px)) + T2[1].y + sTr[0];
  o[1] = fmaf(T2[3].x, pz, fmaf(T2[2].y, py, T2[2].x * px)) + T2[3].y + sTr[1];
  o[2] = fmaf(T2[5].x, pz, fmaf(T2[4].y, py, T2[4].x * px)) + T2[5].y + sTr[2];
}
template <int QB>
__device__ __forceinline__ void part_soft_body(const PartSoftArgs& a, int f) {
  __shared__ __align__(16) float sA[UUO_NUM_JOINTS * 12];
  __shared__ float sF[UUO_NUM_JOINTS * 3 * 64];  // joint forces: entry (3 j + e) of lane l at [(3 j + e) * 64 + l]
  const int tid = threadIdx.x;
  constexpr int P1 = QB;
  constexpr int QP = (QB + 1) / 2;
  {
    const float4* Af = reinterpret_cast<const float4*>(a.A + (size_t)f * UUO_NUM_JOINTS * 12);
    for (int i = tid; i < UUO_NUM_JOINTS * 3; i += 64) reinterpret_cast<float4*>(sA)[i] = Af[i];
#pragma unroll
    for (int i = 0; i < UUO_NUM_JOINTS * 3; ++i) sF[i * 64 + tid] = 0.f;
  }
  pf2 qx[QP], qy[QP], qz[QP];
  float sTr[3];
  const float* xq = a.x + (size_t)f * P1 * 3;
#pragma unroll
  for (int k = 0; k < QP; ++k) {
    const int q0 = 2 * k, q1 = (2 * k + 1 < QB) ? 2 * k + 1 : QB - 1;
    qx[k] = pf2{pfw_uniform(xq[q0 * 3]), pfw_uniform(xq[q1 * 3])};
    qy[k] = pf2{pfw_uniform(xq[q0 * 3 + 1]), pfw_uniform(xq[q1 * 3 + 1])};
    qz[k] = pf2{pfw_uniform(xq[q0 * 3 + 2]), pfw_uniform(xq[q1 * 3 + 2])};
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) sTr[i] = a.trans ? pfw_uniform(a.trans[(size_t)f * 3 + i]) : 0.f;
  const float* Cf = a.C + (size_t)f * a.V * 3;
  const float4* SB4 = reinterpret_cast<const float4*>(a.SB.get());
  const int last = a.ns - 1;
  const float kexp = a.kexp;
  __syncthreads();

  // ---- pass 1: per lane and marker the running minimum, its first candidate, and sum exp((min - d2) / tau)
  float pm[QB], ps[QB];
  unsigned pi[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    pm[q] = 3.0e38f;  // (finite: the first candidate's exp(-(3e38 - d2)/tau) underflows to an exact 0, no inf - inf anywhere)
    ps[q] = 0.f;
    pi[q] = 0xFFFFFFFFu;
  }
  // reach mask: bit `it` = the 64 * PSO_U vertices of iteration `it` may carry a non-zero weight.  A weight exp2((M - d) k) is an
  // exact zero once (d - M) k > 150; M <= every lane's running minimum, so an iteration in which (d - running minimum) k > 160 for
  // every lane and marker holds exact zeros only and pass 2 skips it whole (conservative, the sums are unchanged)
  unsigned long long reach = 0ull;
  const float cutd = 160.f / kexp;
  for (int base = 0; base < a.ns; base += 64 * PSO_U) {
    float o[PSO_U][3];
    unsigned cid[PSO_U];
    float mdiff = 3.0e38f;
#pragma unroll
    for (int u = 0; u < PSO_U; ++u) {
      const int c = base + tid + 64 * u;
      const bool valid = c < a.ns;
      const unsigned cl = (unsigned)min(c, last);
      const unsigned v = min((unsigned)a.subset[cl], (unsigned)(a.V - 1));
      pf2 T2[6];
      float4 ww;
      unsigned pj;
      pso_skin(Cf, SB4, sA, sTr, v, cl, o[u], T2, ww, pj);
      if (!valid) o[u][0] = o[u][1] = o[u][2] = 1.0e18f;  // a lane past the end: d2 ~ 3e36, every weight an exact 0
      cid[u] = valid ? (unsigned)c : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < PSO_U; ++u) {
      const pf2 ox2 = pf2{o[u][0], o[u][0]}, oy2 = pf2{o[u][1], o[u][1]}, oz2 = pf2{o[u][2], o[u][2]};
#pragma unroll
      for (int k = 0; k < QP; ++k) {
        const pf2 dx = qx[k] - ox2, dy = qy[k] - oy2, dz = qz[k] - oz2;
        const pf2 d2 = ((dx * dx) + (dy * dy)) + (dz * dz);  // sqdist(): the K=1 search's arithmetic (same minimum, same ties)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = 2 * k + h;
          if (q < QB) {
            const float d = h ? d2.y : d2.x;
            const float diff = d - pm[q];
            mdiff = fminf(mdiff, diff);  // (a new minimum makes it negative)
            const float e = PSO_EXP(-fabsf(diff) * kexp);
            const bool lt = diff < 0.f;  // strict: ascending candidates per lane -> the first index is kept on ties
            ps[q] = lt ? fmaf(ps[q], e, 1.f) : ps[q] + e;
            pm[q] = lt ? d : pm[q];
            pi[q] = lt ? cid[u] : pi[q];
          }
        }
      }
    }
    const int it = base / (64 * PSO_U);
    if (it < 64 && __any(mdiff <= cutd)) reach |= 1ull << it;
  }
  // ---- merge over the wave (fixed order): minimum and first candidate as k_part_fwd, the sums rescaled to the wave's minimum
  float uM[QB], uW[QB];
  unsigned uI[QB];
  float loss = 0.f;
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const unsigned mb = __float_as_uint(pm[q]);
    const unsigned dmin = wave_min_u32(mb);
    const unsigned imin = wave_min_u32(mb == dmin ? pi[q] : 0xFFFFFFFFu);
    const float Mf = __uint_as_float(dmin);
    const float S = wave_sum_fast(ps[q] * __builtin_amdgcn_exp2f((Mf - pm[q]) * kexp));  // >= 1
    const float soft = Mf - a.tau_ln2 * __builtin_amdgcn_logf(S);                          // v_log_f32 = log2
    loss += a.w_hard * Mf + a.w_soft * soft;
    uM[q] = pfw_uniform(Mf);
    uW[q] = pfw_uniform(a.c_soft / S);
    uI[q] = (unsigned)__builtin_amdgcn_readfirstlane((int)imin);
    if (tid == 0) a.out[(size_t)f * P1 + q] = ((unsigned long long)dmin << 32) | (unsigned long long)imin;
  }

  // ---- pass 2: weights, vertex gradients, the per-frame sums
  float gs[3] = {0.f, 0.f, 0.f}, tq = 0.f, db[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) db[k] = 0.f;
  const float c_hard = a.c_hard;
  for (int base = 0; base < a.ns; base += 64 * PSO_U) {
    {
      const int it = base / (64 * PSO_U);
      if (it < 64 && !((reach >> it) & 1ull)) continue;  // wave-uniform: nothing but exact zeros in this iteration
    }
#pragma unroll
    for (int u = 0; u < PSO_U; ++u) {
      const int c = base + tid + 64 * u;
      const bool valid = c < a.ns;
      const unsigned cl = (unsigned)min(c, last);
      const unsigned v = min((unsigned)a.subset[cl], (unsigned)(a.V - 1));
      const float2* st2 = reinterpret_cast<const float2*>(a.ST + (size_t)v * 30);
      float2 sv[15];
#pragma unroll
      for (int i = 0; i < 15; ++i) sv[i] = st2[i];
      pf2 T2[6];
      float4 ww;
      unsigned pj;
      float o[3];
      pso_skin(Cf, SB4, sA, sTr, v, cl, o, T2, ww, pj);
      const float rx = o[0] - sTr[0], ry = o[1] - sTr[1];
      if (!valid) o[0] = o[1] = o[2] = 1.0e18f;
      const unsigned cidv = valid ? (unsigned)c : 0xFFFFFFFFu;
      const pf2 ox2 = pf2{o[0], o[0]}, oy2 = pf2{o[1], o[1]}, oz2 = pf2{o[2], o[2]};
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int k = 0; k < QP; ++k) {
        const pf2 dx = qx[k] - ox2, dy = qy[k] - oy2, dz = qz[k] - oz2;
        const pf2 d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = 2 * k + h;
          if (q < QB) {
            const float d = h ? d2.y : d2.x;
            float p = PSO_EXP((uM[q] - d) * kexp) * uW[q];
            p += (cidv == uI[q]) ? c_hard : 0.f;
            ax = fmaf(p, h ? dx.y : dx.x, ax);
            ay = fmaf(p, h ? dy.y : dy.x, ay);
            az = fmaf(p, h ? dz.y : dz.x, az);
          }
        }
      }
      const float g0 = -ax, g1 = -ay, g2 = -az;  // d loss / d vertex (exact zeros for a lane past the end)
      // vertices out of every marker's reach carry an exact zero (their weights underflow): when that holds for the whole wave,
      // everything below would add zeros -- skipped (wave-uniform branch; the sums are unchanged)
      if (!__any((ax != 0.f) | (ay != 0.f) | (az != 0.f))) continue;
      gs[0] += g0;
      gs[1] += g1;
      gs[2] += g2;
      tq += rx * g1 - ry * g0;
      // d v_posed = T^R^T g, then the blend-shape path of the shape gradient
      const float dv0 = fmaf(T2[4].x, g2, fmaf(T2[2].x, g1, T2[0].x * g0));
      const float dv1 = fmaf(T2[4].y, g2, fmaf(T2[2].y, g1, T2[0].y * g0));
      const float dv2 = fmaf(T2[5].x, g2, fmaf(T2[3].x, g1, T2[1].x * g0));
      const float* svf = reinterpret_cast<const float*>(sv);
#pragma unroll
      for (int k = 0; k < 10; ++k) db[k] += fmaf(svf[20 + k], dv2, fmaf(svf[10 + k], dv1, svf[k] * dv0));
      // joint forces into this lane's LDS column (an unused weight slot is joint 0 with weight 0: adds exact zeros)
      const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float* col = sF + ((pj >> (8 * n)) & 0xFFu) * 64u + tid;
        col[0] += wv[n] * g0;
        col[64] += wv[n] * g1;
        col[128] += wv[n] * g2;
      }
    }
  }
  // ---- the frame's record
  float* pre = a.pre + (size_t)f * UUO_PRE;
  const float r1 = wave_sum_fast(gs[0]), r2 = wave_sum_fast(gs[1]), r3 = wave_sum_fast(gs[2]), r14 = wave_sum_fast(tq);
  float rb[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) rb[k] = wave_sum_fast(db[k]);
  if (tid == 0) {
    pre[0] = loss;
    pre[1] = r1;
    pre[2] = r2;
    pre[3] = r3;
#pragma unroll
    for (int k = 0; k < 10; ++k) pre[4 + k] = rb[k];
    pre[14] = r14;
    pre[15] = 0.f;
  }
  __syncthreads();
  for (int t = tid; t < UUO_NUM_JOINTS * 3; t += 64) {  // lane t sums entry t over the lanes' columns, rotated: no bank conflict
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) acc += sF[t * 64 + ((i + t) & 63)];
    pre[16 + t] = acc;
  }
}
template <int QB>
__global__ __launch_bounds__(64) void k_part_soft(PartSoftArgs a) {
  const int f = (int)(blockIdx.x >> 3) * 8 + (int)(blockIdx.x & 7);
  if (f >= a.F) return;
  part_soft_body<QB>(a, f);
}
template <int QB>
__global__ __launch_bounds__(64) void k_part_soft_b(const PartSoftArgs* __restrict__ batch, int count) {
  const int r = (int)(blockIdx.x >> 3);
  const PartSoftArgs a = batch[r % count];
  const int f = (r / count) * 8 + (int)(blockIdx.x & 7);
  if (f >= a.F) return;
  part_soft_body<QB>(a, f);
}

int uuo_launch_part_soft(const uuo_model* m, hipStream_t s, int F, int P1, const float* cache, const float* sb, const float* A,
                         const float* trans, const int32_t* subset, int n_subset, const float* markers,
                         unsigned long long* packed, float* pre, float w_hard, float w_soft, float tau) {
  UUO_REQUIRE(P1 >= 1 && P1 <= NNQ_MAX && subset && n_subset > 0 && F > 0 && pre, "uuo_launch_part_soft: bad arguments");
  UUO_REQUIRE(tau > 0.f && w_soft != 0.f, "uuo_launch_part_soft: needs a positive temperature and a soft weight");
  const double fm = (double)F * (double)P1;
  PartSoftArgs a{{F, P1}, F, m->V, n_subset, P1, subset, cache, sb, A, trans, markers, m->ST, packed, pre,
                 (float)(1.4426950408889634 / (double)tau), (float)((double)tau * 0.6931471805599453), w_hard, w_soft,
                 (float)(2.0 * (double)w_hard / fm), (float)(2.0 * (double)w_soft / fm)};
  if (uuo_record(UUO_OP_PART_SOFT, F, P1, a)) return 0;
  const dim3 grid(8 * ((F + 7) / 8));
  switch (P1) {
#define PSO_CASE(Q) case Q: hipLaunchKernelGGL(k_part_soft<Q>, grid, dim3(64), 0, s, a); break;
    PSO_CASE(1) PSO_CASE(2) PSO_CASE(3) PSO_CASE(4) PSO_CASE(5) PSO_CASE(6) PSO_CASE(7) PSO_CASE(8)
    PSO_CASE(9) PSO_CASE(10) PSO_CASE(11) PSO_CASE(12) PSO_CASE(13) PSO_CASE(14) PSO_CASE(15) PSO_CASE(16)
#undef PSO_CASE
  }
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

static int fill_keys(hipStream_t s, unsigned long long* packed, size_t count) {
  FillArgs f{{(int)((count + 255) / 256), 1}, packed, (int)count};
  if (uuo_record(UUO_OP_FILL, f.h.gx, 1, f)) return 0;
  UUO_HIP_CHECK(hipMemsetAsync(packed, 0xFF, count * sizeof(unsigned long long), s));
  return 0;
}

int uuo_launch_nn(hipStream_t s, int N, int P1, int P2, const float* x, const float* y, const int32_t* ysub, int P2s,
                  unsigned long long* packed) {
  const int nc = ysub ? P2s : P2;
  if (nc > 0 && N > 0 && P1 > 0 && P1 <= NNQ_MAX) {
    // splits so that N * S blocks of 256 threads cover the chip a few times, each lane seeing >= 4 candidates (the
    // result is the exact first-index minimum whatever the split, so the choice never shows in the numbers); a
    // lock-step batch fills the chip with its other problems: one split, no zero-fill, no global atomics
    int S = 1;
    if (N < 1024) S = (1024 + N - 1) / N;
    const int maxS = (nc + 1023) / 1024;
    if (S > maxS) S = maxS;
    if (S < 1 || uuo_recorder) S = 1;
    if (S > 1) {
      const int rc = fill_keys(s, packed, (size_t)N * P1);
      if (rc) return rc;
    }
    NnArgs a{{N, S}, x, y, ysub, P1, P2, nc, S, packed};
    if (uuo_record(UUO_OP_NN_FEWQ, N, S, a)) return 0;
    hipLaunchKernelGGL(k_nn_fewq, dim3(N, S), dim3(256), 0, s, a);
    UUO_HIP_CHECK(hipGetLastError());
    return 0;
  }
  {
    const int rc = fill_keys(s, packed, (size_t)N * P1);
    if (rc) return rc;
  }
  if (nc <= 0 || N <= 0 || P1 <= 0) return 0;
  const int qgroups = (P1 + 63) / 64;
  // enough waves to fill the chip: target >= 4096 waves, at least 256 candidates per split
  int S = 1;
  const long waves = (long)N * qgroups;
  if (waves < 4096) S = (int)((4096 + waves - 1) / waves);
  const int maxS = (nc + 255) / 256;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  NnArgs a{{N, qgroups * S}, x, y, ysub, P1, P2, nc, S, packed};
  if (uuo_record(UUO_OP_NN, N, qgroups * S, a)) return 0;
  hipLaunchKernelGGL(k_nn, dim3(N, qgroups, S), dim3(64), 0, s, a);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// Exact K=1 search with pruning (chamfer closure: markers -> all vertices of the frame).
// One block per frame.  k_skin's epilogue leaves the bounding box of every 16-vertex unit of the frame.  The previous closure's assignment gives, per marker, an upper
// bound d_ub = |x - v[prev]|^2 and a starting key (d_ub, prev).  Phase A tests every (marker, unit) pair:
// lb = ((dx*dx)+(dy*dy))+(dz*dz) with dx = max(lo-x, x-hi, 0).  Every fp32 operation involved is monotone and the
// summation order is that of the vertex distance, so lb <= the computed distance of every vertex in the box; a pair
// with lb > d_ub can neither improve nor tie the key and is dropped without an epsilon.  Survivors go to an LDS list;
// phase B evaluates their 16 vertices (16 lanes per entry), reduces (dist bits << 32 | vertex) inside the DPP row
// and merges with a 64-bit LDS atomicMin: the result equals the brute-force first-index minimum bit for bit.
// A frame is split over several blocks by marker groups so that several waves share each SIMD.  If a block has too
// many survivors for its LDS list (poor bounds: first call, markers far from the body) it enumerates all pairs.
// ----------------------------------------------------------------------------------------------------
#define CULL_CAP 3072
#define CULL_CAP2 1024  // surviving (marker, super-box) pairs a block lists; more = poor bounds: enumerate everything
#define CULL_MG 64
#define CULL_MAXU 512
#define CULL_UB 4
#define CULL_MAXG 8
#ifndef CULL_SUPER
#define CULL_SUPER 8  // units per super-box of the two-level box test (0: one level, rounds 1-3)
#endif
#ifndef CULL_T
#define CULL_T 256  // threads per block (128: 7 us slower alone and 2 % slower fits; 512: no faster)
#endif
__device__ __forceinline__ void nn_cull_body(int M, int V, int nunits, int mper, const float* __restrict__ x,
                                             const float* __restrict__ verts, const float* __restrict__ bbox,
                                             unsigned long long* __restrict__ packed, int* __restrict__ stats) {
  __builtin_amdgcn_s_setprio(1);  // latency-bound kernel: do not queue behind co-resident MFMA waves
  __shared__ float sbox[CULL_MAXU * 6];
  __shared__ float smx[CULL_MG * 3];
  __shared__ float sub[CULL_MG];
  __shared__ unsigned long long skey[CULL_MG];
  __shared__ unsigned slist[CULL_CAP];
  __shared__ unsigned scount, scount2;
  const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.y * mper;
  const int mg = min(mper, M - m0);
  if (mg <= 0) return;
  const float* vf = verts + (size_t)f * V * 3;
  {  // unit bounding boxes of this frame (written by k_skin's epilogue) -> LDS
    const float* bf = bbox + (size_t)f * nunits * 6;
    for (int i = tid; i < nunits * 6; i += CULL_T) sbox[i] = bf[i];
  }
  if (tid < mg) {
    const float* px = x + ((size_t)f * M + m0 + tid) * 3;
    const float qx = px[0], qy = px[1], qz = px[2];
    smx[tid * 3] = qx;
    smx[tid * 3 + 1] = qy;
    smx[tid * 3 + 2] = qz;
    unsigned prev = (unsigned)(packed[(size_t)f * M + m0 + tid] & 0xFFFFFFFFull);  // previous closure's assignment
    if (prev >= (unsigned)V) prev = 0;
    const float* pv = vf + (size_t)prev * 3;
    const unsigned long long key = pack_key(sqdist(qx, qy, qz, pv[0], pv[1], pv[2]), prev);
    skey[tid] = key;
    sub[tid] = __uint_as_float((unsigned)(key >> 32));
  }
  if (tid == 0) scount = 0u;
  if (tid == 1) scount2 = 0u;
  __syncthreads();
#if CULL_SUPER
  // ---- phase A, two levels (round 4).  A marker needs a unit only if the unit's box can hold a vertex at most d_ub away;
  // 2-3 % of the 21 550 (marker, unit) pairs of a frame pass that test, and testing them all was most of this kernel's
  // vector work (as many VALU cycles per launch as k_skin2's whole epilogue, profiles/r2_pmc_sq_summary.json).  The boxes of
  // CULL_SUPER consecutive units are merged into a super-box first; a marker is tested against the 54 super-boxes, and
  // against the units of the survivors only.  The lower bound of a super-box is <= that of each of its units (every
  // operation of the bound is monotone), so exactly the same (marker, unit) pairs survive as before: bit-identical result.
  constexpr int NSUP_MAX = (CULL_MAXU + CULL_SUPER - 1) / CULL_SUPER;
  __shared__ float ssup[NSUP_MAX * 6];
  __shared__ unsigned slist2[CULL_CAP2];
  const int nsup = (nunits + CULL_SUPER - 1) / CULL_SUPER;
  for (int sidx = tid; sidx < nsup * 6; sidx += CULL_T) {
    const int sp = sidx / 6, c = sidx - sp * 6;
    const int u0 = sp * CULL_SUPER, u1 = min(nunits, u0 + CULL_SUPER);
    float v = sbox[u0 * 6 + c];
    for (int u = u0 + 1; u < u1; ++u) v = (c < 3) ? fminf(v, sbox[u * 6 + c]) : fmaxf(v, sbox[u * 6 + c]);
    ssup[sidx] = v;
  }
  __syncthreads();
  for (int e = tid; e < nsup * mg; e += CULL_T) {  // (super-box, marker) pairs
    const int sp = e / mg, m = e - sp * mg;
    const float* b = ssup + sp * 6;
    const float qx = smx[m * 3], qy = smx[m * 3 + 1], qz = smx[m * 3 + 2];
    const float dx = fmaxf(fmaxf(__fsub_rn(b[0], qx), __fsub_rn(qx, b[3])), 0.f);
    const float dy = fmaxf(fmaxf(__fsub_rn(b[1], qy), __fsub_rn(qy, b[4])), 0.f);
    const float dz = fmaxf(fmaxf(__fsub_rn(b[2], qz), __fsub_rn(qz, b[5])), 0.f);
    const float lb = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    if (lb <= sub[m]) {
      const unsigned pos = atomicAdd(&scount2, 1u);
      if (pos < CULL_CAP2) slist2[pos] = ((unsigned)m << 16) | (unsigned)sp;
    }
  }
  __syncthreads();
  const int found2 = (int)scount2;
  if (found2 > CULL_CAP2 && tid == 0) scount = CULL_CAP + 1u;  // (the overflow path below enumerates all pairs)
  for (int e = tid; e < min(found2, CULL_CAP2 + 0) * CULL_SUPER && found2 <= CULL_CAP2; e += CULL_T) {  // the units of the surviving super-boxes
    const unsigned ent = slist2[e / CULL_SUPER];
    const int m = (int)(ent >> 16), u = (int)(ent & 0xFFFFu) * CULL_SUPER + (e % CULL_SUPER);
    if (u < nunits) {
      const float* b = sbox + u * 6;
      const float qx = smx[m * 3], qy = smx[m * 3 + 1], qz = smx[m * 3 + 2];
      const float dx = fmaxf(fmaxf(__fsub_rn(b[0], qx), __fsub_rn(qx, b[3])), 0.f);
      const float dy = fmaxf(fmaxf(__fsub_rn(b[1], qy), __fsub_rn(qy, b[4])), 0.f);
      const float dz = fmaxf(fmaxf(__fsub_rn(b[2], qz), __fsub_rn(qz, b[5])), 0.f);
      const float lb = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (lb <= sub[m]) {
        const unsigned pos = atomicAdd(&scount, 1u);
        if (pos < CULL_CAP) slist[pos] = ((unsigned)m << 16) | (unsigned)u;
      }
    }
  }
#else
  // ---- phase A: thread = unit (box in registers); the marker loop only sets bits of a survivor mask (no branch,
  // no atomic on the loop path), then one LDS atomicAdd per thread reserves the list slots
  for (int u = tid; u < nunits; u += CULL_T) {
    const float* b = sbox + u * 6;
    const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b5 = b[5];
    unsigned long long mask = 0ull;
#pragma unroll 4
    for (int m = 0; m < mg; ++m) {
      const float qx = smx[m * 3], qy = smx[m * 3 + 1], qz = smx[m * 3 + 2];
      const float dx = fmaxf(fmaxf(__fsub_rn(b0, qx), __fsub_rn(qx, b3)), 0.f);
      const float dy = fmaxf(fmaxf(__fsub_rn(b1, qy), __fsub_rn(qy, b4)), 0.f);
      const float dz = fmaxf(fmaxf(__fsub_rn(b2, qz), __fsub_rn(qz, b5)), 0.f);
      const float lb = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      mask |= (lb <= sub[m]) ? (1ull << m) : 0ull;
    }
    const int cnt = __popcll(mask);
    if (cnt) {
      unsigned pos = atomicAdd(&scount, (unsigned)cnt);
      while (mask) {
        const int m = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        if (pos < CULL_CAP) slist[pos] = ((unsigned)m << 16) | (unsigned)u;
        ++pos;
      }
    }
  }
#endif
  __syncthreads();
  const int found = (int)scount;
  const bool overflow = found > CULL_CAP;  // poor bounds (first call, markers far from the body): enumerate all pairs
  const int nent = overflow ? nunits * mg : found;
  // ---- phase B: 16 lanes per (marker, unit) pair, 4 pairs per wave pass, CULL_UB passes in flight
  // (the vertex gathers are L2 round trips: issue them for several passes before the first is consumed)
  for (int e0 = wave * 4 * CULL_UB; e0 < nent; e0 += (CULL_T / 64) * 4 * CULL_UB) {
    unsigned long long key[CULL_UB];
    int mm[CULL_UB];
    float vx[CULL_UB], vy[CULL_UB], vz[CULL_UB];
    int vid[CULL_UB];
#pragma unroll
    for (int r = 0; r < CULL_UB; ++r) {
      const int e = e0 + 4 * r + (lane >> 4);
      mm[r] = 0;
      vid[r] = -1;
      vx[r] = vy[r] = vz[r] = 0.f;
      if (e < nent) {
        int u;
        if (overflow) {
          u = e / mg;
          mm[r] = e - u * mg;
        } else {
          const unsigned ent = slist[e];
          mm[r] = (int)(ent >> 16);
          u = (int)(ent & 0xFFFFu);
        }
        const int vtx = u * 16 + (lane & 15);
        if (vtx < V) {
          const float* pv = vf + (size_t)vtx * 3;
          vx[r] = pv[0];
          vy[r] = pv[1];
          vz[r] = pv[2];
          vid[r] = vtx;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < CULL_UB; ++r) {
      const int m = mm[r];
      key[r] = (vid[r] >= 0)
                   ? pack_key(sqdist(smx[m * 3], smx[m * 3 + 1], smx[m * 3 + 2], vx[r], vy[r], vz[r]), (unsigned)vid[r])
                   : ~0ull;
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(key[r], off, 16);
        key[r] = o < key[r] ? o : key[r];
      }
      if ((lane & 15) == 0 && key[r] != ~0ull) atomicMin(&skey[m], key[r]);
    }
  }
  __syncthreads();
  if (tid < mg) packed[(size_t)f * M + m0 + tid] = skey[tid];
  if (tid == 0 && stats) stats[f * CULL_MAXG + blockIdx.y] = overflow ? -found : found;
}

struct NnCullArgs {
  UuoGridHdr h;
  int M, V, nunits, mper;
  uuo_gptr<const float> x;
  uuo_gptr<const float> verts;
  uuo_gptr<const float> bbox;
  uuo_gptr<unsigned long long> packed;
  uuo_gptr<int> stats;
};
__global__ __launch_bounds__(CULL_T) void k_nn_cull(NnCullArgs a) {
  nn_cull_body(a.M, a.V, a.nunits, a.mper, a.x, a.verts, a.bbox, a.packed, a.stats);
}
__global__ __launch_bounds__(CULL_T) void k_nn_cull_b(const NnCullArgs* __restrict__ batch) {
  UUO_BATCH_PICK(NnCullArgs, batch)
  nn_cull_body(a.M, a.V, a.nunits, a.mper, a.x, a.verts, a.bbox, a.packed, a.stats);
}

int uuo_launch_nn_cull(hipStream_t s, int F, int M, int V, int nunits, const float* markers, const float* verts,
                       const float* bbox, unsigned long long* packed, int* stats) {
  if (F <= 0 || M <= 0) return 0;
  UUO_REQUIRE(nunits <= CULL_MAXU, "uuo_launch_nn_cull: too many vertex units for the LDS box table");
  // marker groups per frame: enough blocks (>= ~1024) that several waves share a SIMD, at most 64 markers per group
  int G = (M + CULL_MG - 1) / CULL_MG;
  while (G < CULL_MAXG && (long)F * G < 1024 && (M + G) / (G + 1) >= 8) ++G;
  const int mper = (M + G - 1) / G;
  UUO_REQUIRE(mper <= CULL_MG && G <= CULL_MAXG, "uuo_launch_nn_cull: too many markers per frame for the pruned search");
  NnCullArgs a{{F, G}, M, V, nunits, mper, markers, verts, bbox, packed, stats};
  if (uuo_record(UUO_OP_NN_CULL, F, G, a)) return 0;
  hipLaunchKernelGGL(k_nn_cull, dim3(F, G), dim3(CULL_T), 0, s, a);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// batched launches of this file's kernels (uuo_common.h): 0 = launched, 1 = not one of mine, < 0 = error
int uuo_batched_launch_nn(int op, hipStream_t s, const void* d_args, int count, int gx, int gy) {
  if (op == UUO_OP_FILL) {
    hipLaunchKernelGGL(k_fill_keys_b, dim3(gx, gy, count), dim3(256), 0, s, (const FillArgs*)d_args);
  } else if (op == UUO_OP_NN) {
    hipLaunchKernelGGL(k_nn_b, dim3(gx, gy, count), dim3(64), 0, s, (const NnArgs*)d_args);
  } else if (op == UUO_OP_NN_FEWQ) {
    hipLaunchKernelGGL(k_nn_fewq_b, dim3(gx, gy, count), dim3(256), 0, s, (const NnArgs*)d_args);
  } else if (op == UUO_OP_NN_CULL) {
    hipLaunchKernelGGL(k_nn_cull_b, dim3(gx, gy, count), dim3(CULL_T), 0, s, (const NnCullArgs*)d_args);
  } else if (op == UUO_OP_PART_FWD) {  // gx = frames of the longest problem, gy = markers per frame (one value per batch)
    const dim3 grid(8 * ((gx + 7) / 8) * count);
    const PartFwdArgs* pa = (const PartFwdArgs*)d_args;
    switch (gy) {
#define PFW_CASE(Q) case Q: hipLaunchKernelGGL(k_part_fwd_b<Q>, grid, dim3(PFW_T), 0, s, pa, count); break;
      PFW_CASE(1) PFW_CASE(2) PFW_CASE(3) PFW_CASE(4) PFW_CASE(5) PFW_CASE(6) PFW_CASE(7) PFW_CASE(8)
      PFW_CASE(9) PFW_CASE(10) PFW_CASE(11) PFW_CASE(12) PFW_CASE(13) PFW_CASE(14) PFW_CASE(15) PFW_CASE(16)
#undef PFW_CASE
      default: UUO_REQUIRE(false, "batched k_part_fwd: markers per frame out of range");
    }
  } else if (op == UUO_OP_PART_SOFT) {  // as UUO_OP_PART_FWD
    const dim3 grid(8 * ((gx + 7) / 8) * count);
    const PartSoftArgs* pa = (const PartSoftArgs*)d_args;
    switch (gy) {
#define PSO_CASE(Q) case Q: hipLaunchKernelGGL(k_part_soft_b<Q>, grid, dim3(64), 0, s, pa, count); break;
      PSO_CASE(1) PSO_CASE(2) PSO_CASE(3) PSO_CASE(4) PSO_CASE(5) PSO_CASE(6) PSO_CASE(7) PSO_CASE(8)
      PSO_CASE(9) PSO_CASE(10) PSO_CASE(11) PSO_CASE(12) PSO_CASE(13) PSO_CASE(14) PSO_CASE(15) PSO_CASE(16)
#undef PSO_CASE
      default: UUO_REQUIRE(false, "batched k_part_soft: markers per frame out of range");
    }
  } else {
    return 1;
  }
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void k_nn_unpack(int count, const unsigned long long* __restrict__ packed, float* __restrict__ dist,
                            int32_t* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const unsigned long long k = packed[i];
  if (dist) dist[i] = __uint_as_float((unsigned)(k >> 32));
  if (idx) idx[i] = (int32_t)(unsigned)(k & 0xFFFFFFFFull);
}

int uuo_launch_nn_unpack(hipStream_t s, int count, const unsigned long long* packed, float* dist, int32_t* idx) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_nn_unpack, dim3((count + 255) / 256), dim3(256), 0, s, count, packed, dist, idx);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

extern "C" int uuo_nn_argmin(void* stream, int N, int P1, int P2, const float* d_x, const float* d_y,
                             const int32_t* d_y_subset, int P2s, float* d_dist, int32_t* d_idx, void* d_ws) {
  UUO_REQUIRE(d_x && d_y && d_ws, "uuo_nn_argmin: null argument (workspace of N*P1 uint64 is required)");
  UUO_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0, "uuo_nn_argmin: negative size");
  hipStream_t s = (hipStream_t)stream;
  int rc = uuo_launch_nn(s, N, P1, P2, d_x, d_y, d_y_subset, P2s, (unsigned long long*)d_ws);
  if (rc) return rc;
  return uuo_launch_nn_unpack(s, N * P1, (const unsigned long long*)d_ws, d_dist, d_idx);
}

// ----------------------------------------------------------------------------------------------------
// Marker placement (reference optimization.py:479-486,595-603): idx[m] = argmin_v mean_f ||v_fv - x_fm||
// with numpy's fp32 semantics: norm = sqrt((dx*dx + dy*dy) + dz*dz), sum over valid frames sequentially
// in f, then divide by the count; first index on ties.  Thread = vertex, marker group of 8 in registers
// so each vertex row is read once per 8 markers.
// ----------------------------------------------------------------------------------------------------
#define ASSIGN_MG 4    // markers per thread (a vertex row is read once per group; 13 groups x 27 vertex blocks fill the chip)
#define ASSIGN_FC 32   // frames whose markers are staged in LDS at a time
#define ASSIGN_U 4     // frames whose vertex loads are in flight together
// The sum over the frames is SEQUENTIAL in f by definition (numpy's mean over axis 0 of the [F, M, V] matrix: K-E of
// SURVEY.md section 4), one dependent add per frame and (marker, vertex); everything else of a frame -- the vertex load, the
// distance, the square root -- is independent of it, so the loop stages the markers of 32 frames per barrier pair and keeps
// the vertex loads of four frames in flight (round 2's loop had two barriers and one exposed L2 round trip per frame:
// 320 us at 300 x 50; this one ~4x less).
__global__ __launch_bounds__(256) void k_assign(int F, int M, int V, const float* __restrict__ verts,
                                                 const float* __restrict__ markers,
                                                 const unsigned char* __restrict__ valid, int count,
                                                 unsigned long long* __restrict__ out) {
  __shared__ float sm[ASSIGN_FC][ASSIGN_MG * 3];
  __shared__ unsigned char sv[ASSIGN_FC];
  const int v = blockIdx.x * 256 + threadIdx.x;
  const int vc = v < V ? v : V - 1;  // lanes past the end repeat the last vertex (their result is dropped)
  const int m0 = blockIdx.y * ASSIGN_MG;
  float acc[ASSIGN_MG];
#pragma unroll
  for (int g = 0; g < ASSIGN_MG; ++g) acc[g] = 0.f;
  for (int f0 = 0; f0 < F; f0 += ASSIGN_FC) {
    const int nf = min(ASSIGN_FC, F - f0);
    __syncthreads();
    for (int i = threadIdx.x; i < nf * ASSIGN_MG * 3; i += 256) {
      const int ff = i / (ASSIGN_MG * 3), r = i - ff * (ASSIGN_MG * 3), g = r / 3, c = r - g * 3;
      sm[ff][r] = (m0 + g < M) ? markers[((size_t)(f0 + ff) * M + m0 + g) * 3 + c] : 0.f;
    }
    if (threadIdx.x < nf) sv[threadIdx.x] = valid[f0 + threadIdx.x];
    __syncthreads();
    for (int q0 = 0; q0 < nf; q0 += ASSIGN_U) {
      float px[ASSIGN_U], py[ASSIGN_U], pz[ASSIGN_U];
#pragma unroll
      for (int u = 0; u < ASSIGN_U; ++u) {
        const int ff = min(q0 + u, nf - 1);
        const float* pv = verts + ((size_t)(f0 + ff) * V + vc) * 3;
        px[u] = pv[0]; py[u] = pv[1]; pz[u] = pv[2];
      }
#pragma unroll
      for (int u = 0; u < ASSIGN_U; ++u) {
        const int ff = q0 + u;
        if (ff < nf && sv[ff]) {  // uniform across the block
#pragma unroll
          for (int g = 0; g < ASSIGN_MG; ++g) {
            const float d2 = sqdist(px[u], py[u], pz[u], sm[ff][g * 3], sm[ff][g * 3 + 1], sm[ff][g * 3 + 2]);
            acc[g] = __fadd_rn(acc[g], uuo_sqrt_rn(d2));
          }
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < ASSIGN_MG; ++g) {
    if (m0 + g < M) {  // uniform across the block
      unsigned long long key = ~0ull;
      if (v < V) key = pack_key(uuo_div_rn(acc[g], (float)count), (unsigned)v);
      // wave-level lexicographic min (all 64 lanes participate) before the global atomic
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, 64);
        key = o < key ? o : key;
      }
      if ((threadIdx.x & 63) == 0 && key != ~0ull) atomicMin(&out[m0 + g], key);
    }
  }
}

int uuo_launch_assign(hipStream_t s, int F, int M, int V, const float* verts, const float* markers,
                      const uint8_t* valid, int32_t* idx, unsigned long long* packed) {
  // count of valid frames is needed on the host for the divide; valid is tiny
  unsigned char* h = (unsigned char*)malloc(F);
  if (!h) return -12;
  hipError_t e = hipMemcpyAsync(h, valid, F, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    free(h);
    uuo_set_error(std::string("uuo_assign_mean_argmin: ") + hipGetErrorString(e));
    return -5;
  }
  int count = 0;
  for (int f = 0; f < F; ++f) count += h[f] ? 1 : 0;
  free(h);
  UUO_REQUIRE(count > 0, "uuo_assign_mean_argmin: no valid frame");
  UUO_HIP_CHECK(hipMemsetAsync(packed, 0xFF, (size_t)M * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_assign, dim3((V + 255) / 256, (M + ASSIGN_MG - 1) / ASSIGN_MG), dim3(256), 0, s, F, M, V, verts,
                     markers, valid, count, packed);
  UUO_HIP_CHECK(hipGetLastError());
  return uuo_launch_nn_unpack(s, M, packed, nullptr, idx);
}

extern "C" int uuo_assign_mean_argmin(void* stream, int F, int M, int V, const float* d_verts, const float* d_markers,
                                      const uint8_t* d_valid, int32_t* d_idx, void* d_ws) {
  UUO_REQUIRE(d_verts && d_markers && d_valid && d_idx && d_ws, "uuo_assign_mean_argmin: null argument");
  UUO_REQUIRE(F > 0 && M > 0 && V > 0, "uuo_assign_mean_argmin: sizes must be positive");
  return uuo_launch_assign((hipStream_t)stream, F, M, V, d_verts, d_markers, d_valid, d_idx, (unsigned long long*)d_ws);
}

// ----------------------------------------------------------------------------------------------------
// EXTENSION (not reference behaviour; BASELINE's north star names a soft-assignment chamfer): soft-min nearest
// neighbour.  softmin_i = -tau * log sum_j exp(-d2_ij / tau) = dmin_i - tau * log sum_j exp((dmin_i - d2_ij) / tau),
// with dmin from the exact K=1 search so that every exponent is <= 0.  Forward: one wave per (cloud, query), lanes
// stride the candidates, fixed-order wave sum.  Backward: softmax weights p_ij = exp((dmin_i - d2_ij)/tau) / S_i;
//   d softmin_i / d y_j = -2 p_ij (x_i - y_j),  d softmin_i / d x_i = 2 sum_j p_ij (x_i - y_j)
// gy: thread = candidate, loop over the queries (no atomics); gx: wave = query, loop over the candidates.
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_soft_fwd(int P1, int P2, const float* __restrict__ x,
                                                 const float* __restrict__ y, const float* __restrict__ dmin,
                                                 float inv_tau, float tau, float* __restrict__ softmin,
                                                 float* __restrict__ sumexp) {
  const int n = blockIdx.y, q = blockIdx.x, lane = threadIdx.x;
  const float* px = x + ((size_t)n * P1 + q) * 3;
  const float qx = px[0], qy = px[1], qz = px[2];
  const float dm = dmin[(size_t)n * P1 + q];
  const float* yn = y + (size_t)n * P2 * 3;
  float s = 0.f;
  for (int j = lane; j < P2; j += 64) {
    const float d2 = sqdist(qx, qy, qz, yn[j * 3], yn[j * 3 + 1], yn[j * 3 + 2]);
    s += __expf((dm - d2) * inv_tau);
  }
  s = wave_sum(s);
  if (lane == 0) {
    sumexp[(size_t)n * P1 + q] = s;
    softmin[(size_t)n * P1 + q] = dm - tau * __logf(s);
  }
}

#define SOFT_QT 64  // queries staged per LDS tile of the candidate-gradient kernel
__global__ __launch_bounds__(256) void k_soft_bwd_y(int P1, int P2, const float* __restrict__ x,
                                                    const float* __restrict__ y, const float* __restrict__ dmin,
                                                    const float* __restrict__ sumexp, const float* __restrict__ gq,
                                                    float inv_tau, float* __restrict__ gy) {
  __shared__ float sq[SOFT_QT * 5];  // x, y, z, dmin, g / S
  const int n = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  const float* yn = y + (size_t)n * P2 * 3;
  const bool on = j < P2;
  const float cx = on ? yn[j * 3] : 0.f, cy = on ? yn[j * 3 + 1] : 0.f, cz = on ? yn[j * 3 + 2] : 0.f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int q0 = 0; q0 < P1; q0 += SOFT_QT) {
    __syncthreads();
    if (threadIdx.x < SOFT_QT) {
      const int q = q0 + threadIdx.x;
      const bool qon = q < P1;
      const size_t o = (size_t)n * P1 + (qon ? q : 0);
      sq[threadIdx.x * 5] = qon ? x[o * 3] : 0.f;
      sq[threadIdx.x * 5 + 1] = qon ? x[o * 3 + 1] : 0.f;
      sq[threadIdx.x * 5 + 2] = qon ? x[o * 3 + 2] : 0.f;
      sq[threadIdx.x * 5 + 3] = qon ? dmin[o] : 0.f;
      sq[threadIdx.x * 5 + 4] = qon ? gq[o] / sumexp[o] : 0.f;  // 0 switches the padding queries off
    }
    __syncthreads();
    const int nq = min(SOFT_QT, P1 - q0);
    for (int t = 0; t < nq; ++t) {
      const float qx = sq[t * 5], qy = sq[t * 5 + 1], qz = sq[t * 5 + 2];
      const float d2 = sqdist(qx, qy, qz, cx, cy, cz);
      const float w = sq[t * 5 + 4] * __expf((sq[t * 5 + 3] - d2) * inv_tau);  // g_i p_ij
      a0 = fmaf(w, qx - cx, a0);
      a1 = fmaf(w, qy - cy, a1);
      a2 = fmaf(w, qz - cz, a2);
    }
  }
  if (on) {
    float* o = gy + ((size_t)n * P2 + j) * 3;
    o[0] = -2.f * a0; o[1] = -2.f * a1; o[2] = -2.f * a2;
  }
}

__global__ __launch_bounds__(64) void k_soft_bwd_x(int P1, int P2, const float* __restrict__ x,
                                                   const float* __restrict__ y, const float* __restrict__ dmin,
                                                   const float* __restrict__ sumexp, const float* __restrict__ gq,
                                                   float inv_tau, float* __restrict__ gx) {
  const int n = blockIdx.y, q = blockIdx.x, lane = threadIdx.x;
  const size_t o = (size_t)n * P1 + q;
  const float qx = x[o * 3], qy = x[o * 3 + 1], qz = x[o * 3 + 2];
  const float dm = dmin[o], sc = gq[o] / sumexp[o];
  const float* yn = y + (size_t)n * P2 * 3;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int j = lane; j < P2; j += 64) {
    const float cx = yn[j * 3], cy = yn[j * 3 + 1], cz = yn[j * 3 + 2];
    const float w = __expf((dm - sqdist(qx, qy, qz, cx, cy, cz)) * inv_tau);
    a0 = fmaf(w, qx - cx, a0);
    a1 = fmaf(w, qy - cy, a1);
    a2 = fmaf(w, qz - cz, a2);
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
  if (lane == 0) {
    gx[o * 3] = 2.f * sc * a0; gx[o * 3 + 1] = 2.f * sc * a1; gx[o * 3 + 2] = 2.f * sc * a2;
  }
}

extern "C" int uuo_soft_nn_forward(void* stream, int N, int P1, int P2, const float* d_x, const float* d_y, float tau,
                                   float* d_softmin, float* d_dmin, float* d_sumexp, void* d_ws) {
  UUO_REQUIRE(d_x && d_y && d_softmin && d_dmin && d_sumexp && d_ws, "uuo_soft_nn_forward: null argument");
  UUO_REQUIRE(N >= 0 && P1 >= 0 && P2 > 0 && tau > 0.f, "uuo_soft_nn_forward: bad sizes / temperature");
  if (N == 0 || P1 == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  int rc = uuo_launch_nn(s, N, P1, P2, d_x, d_y, nullptr, 0, (unsigned long long*)d_ws);
  if (rc) return rc;
  rc = uuo_launch_nn_unpack(s, N * P1, (const unsigned long long*)d_ws, d_dmin, nullptr);
  if (rc) return rc;
  hipLaunchKernelGGL(k_soft_fwd, dim3(P1, N), dim3(64), 0, s, P1, P2, d_x, d_y, d_dmin, 1.f / tau, tau, d_softmin, d_sumexp);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

extern "C" int uuo_soft_nn_backward(void* stream, int N, int P1, int P2, const float* d_x, const float* d_y, float tau,
                                    const float* d_dmin, const float* d_sumexp, const float* d_grad_softmin,
                                    float* d_gx, float* d_gy) {
  UUO_REQUIRE(d_x && d_y && d_dmin && d_sumexp && d_grad_softmin, "uuo_soft_nn_backward: null argument");
  UUO_REQUIRE(N >= 0 && P1 >= 0 && P2 > 0 && tau > 0.f, "uuo_soft_nn_backward: bad sizes / temperature");
  if (N == 0 || P1 == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (d_gy)
    hipLaunchKernelGGL(k_soft_bwd_y, dim3((P2 + 255) / 256, N), dim3(256), 0, s, P1, P2, d_x, d_y, d_dmin, d_sumexp,
                       d_grad_softmin, 1.f / tau, d_gy);
  if (d_gx)
    hipLaunchKernelGGL(k_soft_bwd_x, dim3(P1, N), dim3(64), 0, s, P1, P2, d_x, d_y, d_dmin, d_sumexp, d_grad_softmin,
                       1.f / tau, d_gx);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- the chamfer stage's data term with a soft assignment (EXTENSION), on the fused closure's buffers -------------------------
//   loss_data = (1 / W) sum_{f,m} mask_fm (w_hard dmin_fm + w_soft softmin_fm),   W = sum mask   (weighted_chamfer_distance's
// normalisation, losses/chamfer_distance.py:5-21).  dmin and the hard assignment come from the closure's own exact search
// (keys); the vertices' gradient -- every vertex within reach of a marker gets one -- goes to gV [F][V][3] for the dense
// backward (dense_bwd.hip), the frame's weighted loss sum to entry 0 of its record.
__global__ __launch_bounds__(256) void k_softc_prepare(int count, const unsigned long long* __restrict__ keys,
                                                       const float* __restrict__ mask, float scale, float* __restrict__ dmin,
                                                       float* __restrict__ gsm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  dmin[i] = __uint_as_float((unsigned)(keys[i] >> 32));
  gsm[i] = scale * mask[i];  // d loss / d softmin_fm
}
__global__ __launch_bounds__(64) void k_softc_finish(int M, int V, const unsigned long long* __restrict__ keys,
                                                     const float* __restrict__ mask, const float* __restrict__ softmin,
                                                     const float* __restrict__ x, const float* __restrict__ verts,
                                                     float w_hard, float w_soft, float c_hard, float* __restrict__ gV,
                                                     float* __restrict__ pre, int pre_stride) {
  const int f = blockIdx.x, lane = threadIdx.x;
  float acc = 0.f;
  for (int m = lane; m < M; m += 64) {
    const size_t o = (size_t)f * M + m;
    acc += mask[o] * (w_hard * __uint_as_float((unsigned)(keys[o] >> 32)) + w_soft * softmin[o]);
  }
  acc = wave_sum_fast(acc);
  if (lane == 0) {
    pre[(size_t)f * pre_stride] = acc;
    if (c_hard != 0.f) {  // the hard term's gradient joins the winners' (markers one after the other: several may share a vertex)
      for (int m = 0; m < M; ++m) {
        const size_t o = (size_t)f * M + m;
        const unsigned idx = (unsigned)(keys[o] & 0xFFFFFFFFull);
        if (idx >= (unsigned)V) continue;
        const float sc = -c_hard * mask[o];
        const float* pv = verts + ((size_t)f * V + idx) * 3;
        float* pg = gV + ((size_t)f * V + idx) * 3;
        pg[0] += sc * (x[o * 3] - pv[0]);
        pg[1] += sc * (x[o * 3 + 1] - pv[1]);
        pg[2] += sc * (x[o * 3 + 2] - pv[2]);
      }
    }
  }
}
// The same two soft-min kernels pruned by the unit boxes k_skin2 leaves (16 vertices each): a (marker, unit) pair whose box lies
// farther than dmin + cut from the marker holds only vertices with exp((dmin - d2)/tau) < exp(-cut/tau) = 2^-36 -- six thousand of
// them move the normaliser S >= 1 by less than 1e-7 -- and is skipped.  The box bound is k_nn_cull's (every operation monotone,
// the distance's own summation order), so the unit of the nearest vertex always survives.
__device__ __forceinline__ float softc_box_lb(const float* __restrict__ b, float x, float y, float z) {
  const float dx = fmaxf(fmaxf(b[0] - x, x - b[3]), 0.f), dy = fmaxf(fmaxf(b[1] - y, y - b[4]), 0.f),
              dz = fmaxf(fmaxf(b[2] - z, z - b[5]), 0.f);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
// forward: one wave per (frame, marker); surviving units listed in ascending order, four units per pass (16 lanes each)
__global__ __launch_bounds__(256) void k_softc_fwd(int M, int V, int nur, const float* __restrict__ x, const float* __restrict__ verts,
                                                   const float* __restrict__ bbox, const float* __restrict__ dmin, float inv_tau,
                                                   float tau, float cut, float* __restrict__ softmin, float* __restrict__ sumexp) {
  __shared__ unsigned short list[4][512];
  const int f = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  const bool on = m < M;
  const size_t o = (size_t)f * M + (on ? m : 0);
  const float qx = x[o * 3], qy = x[o * 3 + 1], qz = x[o * 3 + 2], dm = dmin[o];
  const float* bf = bbox + (size_t)f * nur * 6;
  int n = 0;
  for (int u0 = 0; u0 < nur; u0 += 64) {
    const int u = u0 + lane;
    const bool keep = on && u < nur && (softc_box_lb(bf + (size_t)u * 6, qx, qy, qz) - dm) <= cut;
    const unsigned long long b = __ballot(keep);
    if (keep) list[wave][n + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)u;
    n += __popcll(b);
  }
  __syncthreads();
  const float* vf = verts + (size_t)f * V * 3;
  float s = 0.f;
  for (int k = 0; k < n; k += 4) {
    const int idx = k + (lane >> 4);
    if (idx < n) {
      const int v = (int)list[wave][idx] * 16 + (lane & 15);
      if (v < V) s += __expf((dm - sqdist(qx, qy, qz, vf[v * 3], vf[v * 3 + 1], vf[v * 3 + 2])) * inv_tau);
    }
  }
  s = wave_sum_fast(s);
  if (on && lane == 0) {
    sumexp[o] = s;
    softmin[o] = dm - tau * __logf(s);
  }
}
// backward: thread = vertex, block = 16 units; wave 0 lists the markers within reach of the block's boxes (ascending), all
// threads then visit those only
__global__ __launch_bounds__(256) void k_softc_bwd(int M, int V, int nur, const float* __restrict__ x, const float* __restrict__ verts,
                                                   const float* __restrict__ bbox, const float* __restrict__ dmin,
                                                   const float* __restrict__ sumexp, const float* __restrict__ gq, float inv_tau,
                                                   float cut, float* __restrict__ gV) {
  __shared__ float sq[64 * 5];
  __shared__ float sb[16 * 6];
  __shared__ int s_n;
  const int f = blockIdx.y, tid = threadIdx.x, ub = blockIdx.x * 16;
  const int j = ub * 16 + tid;
  const float* vf = verts + (size_t)f * V * 3;
  const bool on = j < V;
  const float cx = on ? vf[j * 3] : 0.f, cy = on ? vf[j * 3 + 1] : 0.f, cz = on ? vf[j * 3 + 2] : 0.f;
  if (tid < 16 * 6) {
    const int u = ub + tid / 6;
    sb[tid] = (u < nur) ? bbox[((size_t)f * nur + u) * 6 + tid % 6] : ((tid % 6 < 3) ? 3.0e38f : -3.0e38f);  // (an empty box: infinitely far)
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int q0 = 0; q0 < M; q0 += 64) {
    __syncthreads();
    if (tid < 64) {
      const int q = q0 + tid;
      const bool qon = q < M;
      const size_t o = (size_t)f * M + (qon ? q : 0);
      const float qx = x[o * 3], qy = x[o * 3 + 1], qz = x[o * 3 + 2], dm = dmin[o], g = qon ? gq[o] : 0.f;
      float lb = 3.0e38f;
#pragma unroll
      for (int u = 0; u < 16; ++u) lb = fminf(lb, softc_box_lb(sb + u * 6, qx, qy, qz));
      const bool keep = qon && g != 0.f && (lb - dm) <= cut;
      const unsigned long long b = __ballot(keep);
      if (keep) {
        const int pos = __popcll(b & ((1ull << tid) - 1ull));
        sq[pos * 5] = qx; sq[pos * 5 + 1] = qy; sq[pos * 5 + 2] = qz; sq[pos * 5 + 3] = dm;
        sq[pos * 5 + 4] = g / sumexp[o];
      }
      if (tid == 0) s_n = __popcll(b);
    }
    __syncthreads();
    const int nq = s_n;
    for (int t = 0; t < nq; ++t) {
      const float qx = sq[t * 5], qy = sq[t * 5 + 1], qz = sq[t * 5 + 2];
      const float d2 = sqdist(qx, qy, qz, cx, cy, cz);
      const float w = sq[t * 5 + 4] * __expf((sq[t * 5 + 3] - d2) * inv_tau);
      a0 = fmaf(w, qx - cx, a0);
      a1 = fmaf(w, qy - cy, a1);
      a2 = fmaf(w, qz - cz, a2);
    }
  }
  if (on) {
    float* o = gV + ((size_t)f * V + j) * 3;
    o[0] = -2.f * a0; o[1] = -2.f * a1; o[2] = -2.f * a2;
  }
}
#define SOFTC_CUT_TAUS 24.953298500f  // 36 ln 2

// sm: 4 * F * M floats of scratch (dmin | d loss / d softmin | softmin | sum of exp); bbox: k_skin2's unit boxes [F][ceil(V/16)][6]
// of THESE vertices, or null (brute force over all vertices)
int uuo_launch_soft_chamfer(hipStream_t s, int F, int M, int V, const float* markers, const float* verts, const float* mask,
                            float mask_sum, const unsigned long long* keys, float w_hard, float w_soft, float tau, float* sm,
                            float* gV, float* pre, int pre_stride, const float* bbox) {
  UUO_REQUIRE(markers && verts && mask && keys && sm && gV && pre && tau > 0.f, "uuo_launch_soft_chamfer: bad arguments");
  UUO_REQUIRE(!uuo_recorder, "the soft-assignment chamfer closure is not available inside a lock-step batch");
  const int n = F * M;
  float *dmin = sm, *gsm = sm + n, *softmin = sm + 2 * (size_t)n, *sumexp = sm + 3 * (size_t)n;
  const double inv_w = mask_sum > 0.f ? 1.0 / (double)mask_sum : 0.0;
  hipLaunchKernelGGL(k_softc_prepare, dim3((n + 255) / 256), dim3(256), 0, s, n, keys, mask, (float)((double)w_soft * inv_w), dmin, gsm);
  const int nur = (V + 15) / 16;
  if (bbox && nur <= 512) {
    const float cut = SOFTC_CUT_TAUS * tau;
    hipLaunchKernelGGL(k_softc_fwd, dim3((M + 3) / 4, F), dim3(256), 0, s, M, V, nur, markers, verts, bbox, dmin, 1.f / tau, tau, cut,
                       softmin, sumexp);
    hipLaunchKernelGGL(k_softc_bwd, dim3((nur + 15) / 16, F), dim3(256), 0, s, M, V, nur, markers, verts, bbox, dmin, sumexp, gsm,
                       1.f / tau, cut, gV);
  } else {
    hipLaunchKernelGGL(k_soft_fwd, dim3(M, F), dim3(64), 0, s, M, V, markers, verts, dmin, 1.f / tau, tau, softmin, sumexp);
    hipLaunchKernelGGL(k_soft_bwd_y, dim3((V + 255) / 256, F), dim3(256), 0, s, M, V, markers, verts, dmin, sumexp, gsm, 1.f / tau, gV);
  }
  hipLaunchKernelGGL(k_softc_finish, dim3(F), dim3(64), 0, s, M, V, keys, mask, softmin, markers, verts, w_hard, w_soft,
                     (float)(2.0 * (double)w_hard * inv_w), gV, pre, pre_stride);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// Closest point on the body surface (the reference's barycentric placement: igl.signed_distance followed by
// trimesh.triangles.points_to_barycentric, optimization.py:494-500,519-523).  Brute force over the faces with the
// closest-point-on-triangle region test of Ericson, Real-Time Collision Detection 5.1.5 (what libigl's
// point_simplex_squared_distance implements).  Thread = face (its three corners stay in registers), the block's
// markers are broadcast from LDS; faces are merged with a 64-bit min on (d^2 bits << 32 | face), so ties go to the
// lowest face index independent of the schedule.
// ----------------------------------------------------------------------------------------------------
struct TriHit {
  float cx, cy, cz, d2;
};
__device__ __forceinline__ TriHit closest_on_triangle(float px, float py, float pz, float ax, float ay, float az,
                                                      float bx, float by, float bz, float cx, float cy, float cz) {
  const float abx = bx - ax, aby = by - ay, abz = bz - az;
  const float acx = cx - ax, acy = cy - ay, acz = cz - az;
  const float apx = px - ax, apy = py - ay, apz = pz - az;
  const float d1 = abx * apx + aby * apy + abz * apz, d2 = acx * apx + acy * apy + acz * apz;
  const float bpx = px - bx, bpy = py - by, bpz = pz - bz;
  const float d3 = abx * bpx + aby * bpy + abz * bpz, d4 = acx * bpx + acy * bpy + acz * bpz;
  const float cpx = px - cx, cpy = py - cy, cpz = pz - cz;
  const float d5 = abx * cpx + aby * cpy + abz * cpz, d6 = acx * cpx + acy * cpy + acz * cpz;
  const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  float v, w;  // closest point = a + v*ab + w*ac
  if (d1 <= 0.f && d2 <= 0.f) {  // vertex region a
    v = 0.f; w = 0.f;
  } else if (d3 >= 0.f && d4 <= d3) {  // vertex region b
    v = 1.f; w = 0.f;
  } else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {  // edge ab
    v = d1 / (d1 - d3); w = 0.f;
  } else if (d6 >= 0.f && d5 <= d6) {  // vertex region c
    v = 0.f; w = 1.f;
  } else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {  // edge ac
    v = 0.f; w = d2 / (d2 - d6);
  } else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {  // edge bc
    w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); v = 1.f - w;
  } else {  // interior
    const float denom = 1.f / (va + vb + vc);
    v = vb * denom; w = vc * denom;
  }
  TriHit h;
  h.cx = ax + abx * v + acx * w;
  h.cy = ay + aby * v + acy * w;
  h.cz = az + abz * v + acz * w;
  const float ex = px - h.cx, ey = py - h.cy, ez = pz - h.cz;
  h.d2 = ex * ex + ey * ey + ez * ez;
  if (!(h.d2 == h.d2)) h.d2 = UUO_INF;  // degenerate triangle (0/0): never the winner
  return h;
}

#define MESH_MB 16  // query points per block
__global__ __launch_bounds__(256) void k_mesh_closest(int M, int V, int NF, const float* __restrict__ verts,
                                                       const int32_t* __restrict__ faces,
                                                       const float* __restrict__ points, float* __restrict__ dist,
                                                       int32_t* __restrict__ face_out, float* __restrict__ closest,
                                                       float* __restrict__ bary) {
  __shared__ float sp[MESH_MB * 3];
  __shared__ unsigned long long skey[MESH_MB];
  const int f = blockIdx.x, m0 = blockIdx.y * MESH_MB, tid = threadIdx.x;
  const int mg = min(MESH_MB, M - m0);
  const float* vf = verts + (size_t)f * V * 3;
  if (tid < MESH_MB * 3) sp[tid] = (tid < mg * 3) ? points[((size_t)f * M + m0) * 3 + tid] : 0.f;
  if (tid < MESH_MB) skey[tid] = ~0ull;
  __syncthreads();
  float best[MESH_MB];
  int bestf[MESH_MB];
#pragma unroll
  for (int q = 0; q < MESH_MB; ++q) {
    best[q] = UUO_INF;
    bestf[q] = -1;
  }
  for (int t = tid; t < NF; t += 256) {
    const int i0 = faces[(size_t)t * 3], i1 = faces[(size_t)t * 3 + 1], i2 = faces[(size_t)t * 3 + 2];
    if ((unsigned)i0 >= (unsigned)V || (unsigned)i1 >= (unsigned)V || (unsigned)i2 >= (unsigned)V) continue;
    const float ax = vf[i0 * 3], ay = vf[i0 * 3 + 1], az = vf[i0 * 3 + 2];
    const float bx = vf[i1 * 3], by = vf[i1 * 3 + 1], bz = vf[i1 * 3 + 2];
    const float cx = vf[i2 * 3], cy = vf[i2 * 3 + 1], cz = vf[i2 * 3 + 2];
#pragma unroll
    for (int q = 0; q < MESH_MB; ++q) {
      const TriHit h = closest_on_triangle(sp[q * 3], sp[q * 3 + 1], sp[q * 3 + 2], ax, ay, az, bx, by, bz, cx, cy, cz);
      if (h.d2 < best[q]) {  // ascending faces per thread + strict '<': the thread's first minimum
        best[q] = h.d2;
        bestf[q] = t;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < MESH_MB; ++q) {
    unsigned long long key = (bestf[q] >= 0) ? pack_key(best[q], (unsigned)bestf[q]) : ~0ull;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(key, off, 64);
      key = o < key ? o : key;
    }
    if ((tid & 63) == 0 && key != ~0ull) atomicMin(&skey[q], key);
  }
  __syncthreads();
  if (tid < mg) {
    const unsigned long long key = skey[tid];
    const size_t o = (size_t)f * M + m0 + tid;
    if (key == ~0ull) {  // no usable face
      dist[o] = UUO_INF;
      face_out[o] = -1;
      for (int c = 0; c < 3; ++c) closest[o * 3 + c] = bary[o * 3 + c] = 0.f;
      return;
    }
    const int t = (int)(unsigned)(key & 0xFFFFFFFFull);
    const int i0 = faces[(size_t)t * 3], i1 = faces[(size_t)t * 3 + 1], i2 = faces[(size_t)t * 3 + 2];
    const float ax = vf[i0 * 3], ay = vf[i0 * 3 + 1], az = vf[i0 * 3 + 2];
    const float bx = vf[i1 * 3], by = vf[i1 * 3 + 1], bz = vf[i1 * 3 + 2];
    const float cx = vf[i2 * 3], cy = vf[i2 * 3 + 1], cz = vf[i2 * 3 + 2];
    const TriHit h = closest_on_triangle(sp[tid * 3], sp[tid * 3 + 1], sp[tid * 3 + 2], ax, ay, az, bx, by, bz, cx, cy, cz);
    dist[o] = sqrtf(h.d2);
    face_out[o] = t;
    closest[o * 3] = h.cx;
    closest[o * 3 + 1] = h.cy;
    closest[o * 3 + 2] = h.cz;
    // trimesh.triangles.points_to_barycentric(method="cramer") of the closest point
    const float e0x = bx - ax, e0y = by - ay, e0z = bz - az, e1x = cx - ax, e1y = cy - ay, e1z = cz - az;
    const float wx = h.cx - ax, wy = h.cy - ay, wz = h.cz - az;
    const float dot00 = e0x * e0x + e0y * e0y + e0z * e0z, dot01 = e0x * e1x + e0y * e1y + e0z * e1z;
    const float dot02 = e0x * wx + e0y * wy + e0z * wz, dot11 = e1x * e1x + e1y * e1y + e1z * e1z;
    const float dot12 = e1x * wx + e1y * wy + e1z * wz;
    const float inv = 1.f / (dot00 * dot11 - dot01 * dot01);
    const float b2 = (dot00 * dot12 - dot01 * dot02) * inv, b1 = (dot11 * dot02 - dot01 * dot12) * inv;
    bary[o * 3] = 1.f - b1 - b2;
    bary[o * 3 + 1] = b1;
    bary[o * 3 + 2] = b2;
  }
}

extern "C" int uuo_mesh_closest_points(void* stream, int F, int M, int V, int NF, const float* d_verts,
                                       const int32_t* d_faces, const float* d_points, float* d_dist,
                                       int32_t* d_face, float* d_closest, float* d_bary) {
  UUO_REQUIRE(d_verts && d_faces && d_points && d_dist && d_face && d_closest && d_bary,
              "uuo_mesh_closest_points: null argument");
  UUO_REQUIRE(F >= 0 && M >= 0 && V > 0 && NF > 0, "uuo_mesh_closest_points: sizes must be positive");
  if (F == 0 || M == 0) return 0;
  UUO_REQUIRE((long)F <= 2147483647L && (M + MESH_MB - 1) / MESH_MB <= 65535, "uuo_mesh_closest_points: grid too large");
  hipLaunchKernelGGL(k_mesh_closest, dim3(F, (M + MESH_MB - 1) / MESH_MB), dim3(256), 0, (hipStream_t)stream, M, V, NF,
                     d_verts, d_faces, d_points, d_dist, d_face, d_closest, d_bary);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// get_marker_mask (reference optimization.py:703-715): sum(|xyz|) != 0, plus the count of set entries
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mask(int count, const float* __restrict__ markers, float* __restrict__ mask,
                                               float* __restrict__ mask_sum) {
  __shared__ float red[4];
  float local = 0.f;
  for (int i = threadIdx.x; i < count; i += 256) {
    const float* p = markers + (size_t)i * 3;
    const float s = fabsf(p[0]) + fabsf(p[1]) + fabsf(p[2]);
    const float w = (s != 0.0f) ? 1.f : 0.f;
    mask[i] = w;
    local += w;
  }
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) mask_sum[0] = red[0] + red[1] + red[2] + red[3];
}

int uuo_launch_mask(hipStream_t s, int F, int M, const float* markers, float* mask, float* mask_sum_dev) {
  hipLaunchKernelGGL(k_mask, dim3(1), dim3(256), 0, s, F * M, markers, mask, mask_sum_dev);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// Rigidity matrix of segment_rigid (reference markers/markers_utils.py:254-259):
//   mat[i][j] = np.std(np.linalg.norm(points[:, i] - points[:, j], axis=-1))        points [F, M, 3] float32
// The average-linkage clustering cuts this matrix at 5 mm, so the values are reproduced BIT FOR BIT, which means numpy's
// float32 arithmetic in numpy's order: norm = sqrt((dx*dx + dy*dy) + dz*dz); std = sqrt(sum((d - mean)^2) / F) with
// mean = sum(d) / F, and both sums are numpy's PAIRWISE summation of a contiguous float32 vector (umath loops_utils.h,
// @TYPE@_pairwise_sum): blocks of <= 128 elements summed on 8 interleaved accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail, longer vectors split at floor(n/2) rounded down to a multiple
// of 8, and the reduction itself chunked by the ufunc buffer (8192 elements), the chunks' sums added in order.  One thread
// per marker pair walks that tree with an explicit stack; distances are recomputed in the second pass instead of being
// stored (F floats per pair would not fit in registers).  2 500 pairs x 300 frames: a few microseconds, instead of 2-3 ms
// of numpy on the host per fit.
// ----------------------------------------------------------------------------------------------------
#define NPY_PW_BLOCK 128
#define NPY_BUFSIZE 8192
template <class Elem>
__device__ __forceinline__ float npy_pairwise_leaf(const Elem& elem, int lo, int n) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res = __fadd_rn(res, elem(lo + i));
    return res;
  }
  float r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = elem(lo + k);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = __fadd_rn(r[k], elem(lo + i + k));
  }
  float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                        __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
  for (; i < n; ++i) res = __fadd_rn(res, elem(lo + i));
  return res;
}
template <class Elem>
__device__ __forceinline__ float npy_pairwise_sum(const Elem& elem, int lo0, int n0) {  // n0 <= NPY_BUFSIZE
  // post-order walk of numpy's recursion: state 0 = entered, 1 = left half done, 2 = right half done
  int lo_s[8], n_s[8], st_s[8];
  float left_s[8];
  int sp = 0;
  lo_s[0] = lo0; n_s[0] = n0; st_s[0] = 0;
  float ret = 0.f;
  while (sp >= 0) {
    const int lo = lo_s[sp], n = n_s[sp];
    if (st_s[sp] == 0) {
      if (n <= NPY_PW_BLOCK) {
        ret = npy_pairwise_leaf(elem, lo, n);
        --sp;
        continue;
      }
      int n2 = n / 2;
      n2 -= n2 % 8;
      st_s[sp] = 1;
      ++sp;
      lo_s[sp] = lo; n_s[sp] = n2; st_s[sp] = 0;
    } else if (st_s[sp] == 1) {
      int n2 = n / 2;
      n2 -= n2 % 8;
      left_s[sp] = ret;
      st_s[sp] = 2;
      ++sp;
      lo_s[sp] = lo + n2; n_s[sp] = n - n2; st_s[sp] = 0;
    } else {
      ret = __fadd_rn(left_s[sp], ret);
      --sp;
    }
  }
  return ret;
}
template <class Elem>
__device__ __forceinline__ float npy_sum_f32(const Elem& elem, int n) {  // np.add.reduce of a contiguous float32 vector
  float res = 0.f;
  for (int c = 0; c < n; c += NPY_BUFSIZE) res = __fadd_rn(res, npy_pairwise_sum(elem, c, min(NPY_BUFSIZE, n - c)));
  return res;
}

__global__ __launch_bounds__(64) void k_rigid_std(int F, int M, const float* __restrict__ pts, float* __restrict__ out) {
  const int pair = blockIdx.x * 64 + threadIdx.x;
  if (pair >= M * M) return;
  const int i = pair / M, j = pair - i * M;
  const float* pi = pts + (size_t)i * 3;
  const float* pj = pts + (size_t)j * 3;
  const size_t fs = (size_t)M * 3;
  auto dist = [&](int f) -> float {
    const float dx = __fsub_rn(pi[f * fs], pj[f * fs]), dy = __fsub_rn(pi[f * fs + 1], pj[f * fs + 1]),
                dz = __fsub_rn(pi[f * fs + 2], pj[f * fs + 2]);
    return uuo_sqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  };
  const float nf = (float)F;
  const float mean = uuo_div_rn(npy_sum_f32(dist, F), nf);
  auto sq = [&](int f) -> float {
    const float x = __fsub_rn(dist(f), mean);
    return __fmul_rn(x, x);
  };
  out[pair] = uuo_sqrt_rn(uuo_div_rn(npy_sum_f32(sq, F), nf));
}

extern "C" int uuo_rigid_distance_std(void* stream, int F, int M, const float* d_points, float* d_std) {
  UUO_REQUIRE(d_points && d_std, "uuo_rigid_distance_std: null argument");
  UUO_REQUIRE(F > 0 && M > 0 && (long)M * M < (1L << 30), "uuo_rigid_distance_std: bad sizes");
  hipLaunchKernelGGL(k_rigid_std, dim3((M * M + 63) / 64), dim3(64), 0, (hipStream_t)stream, F, M, d_points, d_std);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}
