// Dense backward of the skinning: dL/dvertices given for EVERY vertex of every frame.
//
// Where it is needed: a caller differentiating SmplInference.forward itself (reference utils/smpl.py:29-50 under torch
// autograd: what the operator-composed closures of the reference's optional objectives do), and the soft-assignment data
// terms (extension), whose gradient reaches every vertex within reach of a marker.  The sparse backward (closure.hip,
// k_bwd_sparse) gathers the <= M touched vertices of a frame, 2.5 KB of posedirs rows each; with 6 890 items per frame that
// gather IS the 207 x 20 670 blend contraction, recomputed on the vector pipe and then transposed on it: 0.72 ms at 300 frames.
// Here the two contractions run where the forward's does:
//   v_posed           = k_skin2 with identity skinning matrices (the forward kernel itself, one more launch)
//   d v_posed         = T^R^T g                  k_dvp   (elementwise, written in MFMA A-operand order)
//   d [pose-feat|beta]= d v_posed . Baug^T       k_dpf   (v_mfma_f32_16x16x4_f32; K = the 20 670 vertex coordinates)
//   d A_j             = sum_v w_vj g_v [v_posed;1]^T   k_dA (per (frame, joint) over the joint's own vertex list)
// and k_bwd_sparse's kinematic tail (reverse sweep, Gram-Schmidt backward, priors, solver statistics) consumes the sums
// (BwdArgs.pre / dpf_part).  Every reduction has a fixed order: bit-reproducible.
#include "frame_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- d v_posed = T^R^T g, in the A-operand order of k_dpf ----------------------------------------------------------------
// element (frame f, vertex v, coordinate c): frame tile ft = f / 16, row i = f % 16, unit u = v / 16, K index
// vc = 3 (v % 16) + c inside the unit: group vc / 16, lane (vc % 4) * 16 + i, slot (vc / 4) % 4   (cf. pose_prep's `put`).
// Block = (frame tile, 4 units): the 16 x 64 (frame, vertex) pairs are computed four per thread into an LDS image of the
// block's 12 KB of tiles, which then leaves as 768 coalesced 16-byte stores (element-wise 4-byte stores from one block per
// frame, the first version, left every 64-byte line to be assembled from 16 blocks: 27 us; frames past F write zeros).
#define DVP_UNITS 4
__global__ __launch_bounds__(256) void k_dvp(int F, int V, int VP, const float* __restrict__ A, const int* __restrict__ Wi,
                                             const float* __restrict__ Ww, const float* __restrict__ gV,
                                             float* __restrict__ dvpT) {
  __shared__ __align__(16) float sA[UUO_FT * UUO_NUM_JOINTS * 12];
  __shared__ __align__(16) float sT[DVP_UNITS * 3 * 256];
  const int ft = blockIdx.y, ub = blockIdx.x * DVP_UNITS, tid = threadIdx.x;
  {
    const float4* src = reinterpret_cast<const float4*>(A + (size_t)ft * UUO_FT * UUO_NUM_JOINTS * 12);  // (A holds whole tiles)
    for (int i = tid; i < UUO_FT * UUO_NUM_JOINTS * 3; i += 256) reinterpret_cast<float4*>(sA)[i] = src[i];
  }
  const int vl64 = tid & 63, v = ub * 16 + vl64;  // < VP: the grid covers whole units
  const int4 wi = *reinterpret_cast<const int4*>(Wi + (size_t)v * 4);
  const float4 ww = *reinterpret_cast<const float4*>(Ww + (size_t)v * 4);
  const int wj[4] = {wi.x, wi.y, wi.z, wi.w};
  const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
  float g[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int f = ft * UUO_FT + (tid >> 6) + 4 * k;
    const bool on = f < F && v < V;
    const float* pg = gV + ((size_t)(on ? f : 0) * V + (on ? v : 0)) * 3;
    g[k][0] = on ? pg[0] : 0.f;
    g[k][1] = on ? pg[1] : 0.f;
    g[k][2] = on ? pg[2] : 0.f;
  }
  __syncthreads();
  const int ul = vl64 >> 4, vl = vl64 & 15;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = (tid >> 6) + 4 * k;
    float T[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) T[e] = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const float* pa = sA + (i * UUO_NUM_JOINTS + wj[n]) * 12;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[r * 3 + c] = fmaf(wv[n], pa[r * 4 + c], T[r * 3 + c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = fmaf(T[6 + c], g[k][2], fmaf(T[3 + c], g[k][1], T[c] * g[k][0]));
      const int vc = 3 * vl + c;
      sT[ul * 768 + (vc >> 4) * 256 + (((vc & 3) * 16 + i) << 2) + ((vc >> 2) & 3)] = d;
    }
  }
  __syncthreads();
  const int nunits = VP / 16;
  float4* dst = reinterpret_cast<float4*>(dvpT + ((size_t)ft * nunits + ub) * 3 * 256);
  for (int i = tid; i < DVP_UNITS * 3 * 64; i += 256) dst[i] = reinterpret_cast<const float4*>(sT)[i];
}

// ---- d A_j = sum over the joint's vertices of w g [v_posed;1]^T; block 24 of a frame: d trans = sum_v g ------------------------
// four waves per (frame, joint): threads stride the joint's list, DPP wave sums, the waves' sums added in wave order.  Block
// sizes measured at 300 frames: 64 threads 43 us, 128: 35, 256: 29, 512: 33, 1024: 50 (the twelve wave sums and the LDS pass are
// per-wave overhead; the gathers of 12-byte items run at the L2's transaction rate)
#define DA_T 256
__global__ __launch_bounds__(DA_T) void k_dA(int F, int V, const int* __restrict__ JLoff, const int* __restrict__ JLv,
                                             const float* __restrict__ JLw, const float* __restrict__ gV,
                                             const float* __restrict__ vp, float* __restrict__ pre) {
  __shared__ float sw[DA_T / 64][12];
  const int j = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* gf = gV + (size_t)f * V * 3;
  float* out = pre + (size_t)f * UUO_PREG;
  float acc[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) acc[e] = 0.f;
  if (j == UUO_NUM_JOINTS) {
    for (int v = tid; v < V; v += DA_T) {
      acc[0] += gf[v * 3];
      acc[1] += gf[v * 3 + 1];
      acc[2] += gf[v * 3 + 2];
    }
  } else {
    const float* pf = vp + (size_t)f * V * 3;
    const int e0 = JLoff[j], e1 = JLoff[j + 1];
#pragma unroll 4
    for (int e = e0 + tid; e < e1; e += DA_T) {
      const int v = JLv[e];
      const float w = JLw[e];
      const float g[3] = {w * gf[v * 3], w * gf[v * 3 + 1], w * gf[v * 3 + 2]};
      const float p[3] = {pf[v * 3], pf[v * 3 + 1], pf[v * 3 + 2]};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        acc[r * 4 + 0] = fmaf(g[r], p[0], acc[r * 4 + 0]);
        acc[r * 4 + 1] = fmaf(g[r], p[1], acc[r * 4 + 1]);
        acc[r * 4 + 2] = fmaf(g[r], p[2], acc[r * 4 + 2]);
        acc[r * 4 + 3] += g[r];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    const float t = wave_sum_fast(acc[e]);
    if (lane == 0) sw[wave][e] = t;
  }
  __syncthreads();
  if (tid < 12) {
    float t = sw[0][tid];
#pragma unroll
    for (int w = 1; w < DA_T / 64; ++w) t += sw[w][tid];
    if (j == UUO_NUM_JOINTS) {
      if (tid < 3) out[1 + tid] = t;
    } else {
      out[16 + j * 12 + tid] = t;
    }
  }
}

// ---- d [pose-feature | beta] = d v_posed . Baug^T on the matrix pipe -------------------------------------------------------------
// D[16 frames x 16 features] += A[16 frames x 4 coordinates] . B[4 coordinates x 16 features]; a unit of 16 vertices is 12
// K-steps, 14 feature tiles: 168 MFMAs, as many as the forward spends on it.  Block = (frame tile, one of 27 vertex chunks of
// 16 units), its 4 waves take the chunk's units round-robin (4 each) and keep 14 accumulator tiles; the waves' tiles are summed
// through LDS in wave order and written as the chunk's partial.  57 KB of LDS: two blocks per CU, so the 513 blocks of a
// 300-frame launch are resident at once (the first version -- 8 waves, 114 KB, 266 blocks on 256 CUs -- ran ten CUs twice:
// 58 us).  XCD x owns chunks x, x + 8, x + 16, x + 24: all frame tiles of a chunk share one L2 slice of the basis (0.7 MB).
#define DPF_WAVES 4
__global__ __launch_bounds__(DPF_WAVES * 64) void k_dpf(const float4* __restrict__ PB, const float4* __restrict__ dvpT,
                                                         float* __restrict__ part, int nFT, int nunits, int F) {
  __shared__ float red[DPF_WAVES][14 * 256];
  const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
  const int cb = xcd + 8 * (pos & 3), ft = pos >> 2;
  if (cb >= UUO_DPF_NCB || ft >= nFT) return;  // block-uniform
  const int u0 = (cb * nunits) / UUO_DPF_NCB, u1 = ((cb + 1) * nunits) / UUO_DPF_NCB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  f32x4 acc[14];
#pragma unroll
  for (int jt = 0; jt < 14; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The basis streams through a ring of 7 stages (one stage = the 3 K-groups of one feature tile), refilled DPF_LEAD tiles
  // ahead and across the unit boundary; the next unit's d v_posed tiles are requested at tile 8 of the current one.
#ifndef DPF_LEAD_N
#define DPF_LEAD_N 4
#endif
  constexpr int DPF_RING = 7, DPF_LEAD = DPF_LEAD_N;
  const int nu = (u1 - (u0 + wave) + DPF_WAVES - 1) / DPF_WAVES;  // units of this wave (<= 0: none)
  if (nu > 0) {
    float4 ring[DPF_RING][3];
    auto pbase = [&](int ui) { return PB + (size_t)(u0 + wave + ui * DPF_WAVES) * 14 * 3 * 64 + lane; };
    auto abase = [&](int ui) { return dvpT + ((size_t)ft * nunits + (u0 + wave + ui * DPF_WAVES)) * 3 * 64 + lane; };
    {
      const float4* pb = pbase(0);
#pragma unroll
      for (int st = 0; st < DPF_LEAD; ++st) {
        ring[st][0] = pb[(st * 3 + 0) * 64];
        ring[st][1] = pb[(st * 3 + 1) * 64];
        ring[st][2] = pb[(st * 3 + 2) * 64];
      }
    }
    float4 an0, an1, an2;
    {
      const float4* pa = abase(0);
      an0 = pa[0];
      an1 = pa[64];
      an2 = pa[128];
    }
    for (int ui = 0; ui < nu; ++ui) {
      const float4 a0 = an0, a1 = an1, a2 = an2;
      const float4* pb = pbase(ui);
      const bool more = ui + 1 < nu;
      const float4* pbn = pbase(more ? ui + 1 : ui);
#pragma unroll
      for (int jt = 0; jt < 14; ++jt) {
        {  // refill the stage that tile jt + DPF_LEAD will read (this unit's, or the next unit's first tiles).  No branch
          // anywhere in the unit's body: past the wave's last unit the refills re-read its own first tiles, so the compiler's
          // vmcnt bookkeeping stays exact and the loads stay DPF_LEAD tiles ahead of the MFMAs that use them
          const int jn = jt + DPF_LEAD;
          const float4* src = (jn < 14) ? pb + (size_t)(jn * 3) * 64 : pbn + (size_t)((jn - 14) * 3) * 64;
          ring[jn % DPF_RING][0] = src[0];
          ring[jn % DPF_RING][1] = src[64];
          ring[jn % DPF_RING][2] = src[128];
        }
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise gathers a unit's loads at its top and drains them before the back-edge)
        if (jt == 8) {
          const float4* pa = abase(more ? ui + 1 : ui);
          an0 = pa[0];
          an1 = pa[64];
          an2 = pa[128];
        }
        // one accumulator chain per tile: back-to-back dependent MFMAs forward their accumulator (two alternating chains,
        // dependent distance 2, measured 47 us against 40; three -- the forward's pattern -- in groups of 3, 3, 3, 3, 2 tiles: 53)
        f32x4 d = acc[jt];
        const float4 b0 = ring[jt % DPF_RING][0], b1 = ring[jt % DPF_RING][1], b2 = ring[jt % DPF_RING][2];
#define DPF_STEP(av, bv)                                                  \
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, d, 0, 0, 0);       \
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, d, 0, 0, 0);       \
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, d, 0, 0, 0);       \
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, d, 0, 0, 0);
        DPF_STEP(a0, b0)
        DPF_STEP(a1, b1)
        DPF_STEP(a2, b2)
#undef DPF_STEP
        acc[jt] = d;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int jt = 0; jt < 14; ++jt) *reinterpret_cast<f32x4*>(&red[wave][jt * 256 + lane * 4]) = acc[jt];
  __syncthreads();
  // D layout: lane l, register r of tile jt = row i = 4 (l >> 4) + r (frame), column j = l & 15 (feature 16 jt + j)
  for (int idx = tid; idx < 14 * 256; idx += DPF_WAVES * 64) {
    float t = red[0][idx];
#pragma unroll
    for (int w = 1; w < DPF_WAVES; ++w) t += red[w][idx];
    const int jt = idx >> 8, l = (idx >> 2) & 63, r = idx & 3;
    const int f = ft * 16 + 4 * (l >> 4) + r;
    if (f < F) part[((size_t)cb * F + f) * UUO_KP + 16 * jt + (l & 15)] = t;
  }
}

// ---- workspace ----------------------------------------------------------------------------------------------------------------------
int uuo_dense_ws_create(const uuo_model* m, hipStream_t s, int F, UuoDenseWs** out) {
  UUO_REQUIRE(m && out && F > 0, "uuo_dense_ws_create: bad arguments");
  UuoDenseWs* ws = new UuoDenseWs();
  ws->F = F;
  ws->nFT = (F + UUO_FT - 1) / UUO_FT;
  const size_t nA = (size_t)ws->nFT * UUO_FT * UUO_NUM_JOINTS * 12;
  const size_t nT = (size_t)ws->nFT * (m->VP / 16) * 3 * 256;
  hipError_t e = hipMalloc((void**)&ws->A_id, nA * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&ws->vp, (size_t)F * m->V * 3 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&ws->dvpT, nT * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&ws->part, (size_t)UUO_DPF_NCB * F * UUO_KP * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&ws->pre, (size_t)F * UUO_PREG * sizeof(float));
  // on the caller's stream (a null-stream fill is not ordered with non-blocking streams)
  if (e == hipSuccess) e = hipMemsetAsync(ws->dvpT, 0, nT * sizeof(float), s);  // rows of frames past F stay zero
  if (e == hipSuccess) e = hipMemsetAsync(ws->pre, 0, (size_t)F * UUO_PREG * sizeof(float), s);
  if (e == hipSuccess) {
    std::vector<float> id(nA, 0.f);
    for (size_t k = 0; k < nA / 12; ++k) id[k * 12] = id[k * 12 + 5] = id[k * 12 + 10] = 1.f;
    e = hipMemcpyAsync(ws->A_id, id.data(), nA * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // `id` leaves scope
  }
  if (e != hipSuccess) {
    uuo_set_error(std::string("uuo_dense_ws_create: ") + hipGetErrorString(e));
    uuo_dense_ws_destroy(ws);
    return -12;
  }
  *out = ws;
  return 0;
}

void uuo_dense_ws_destroy(UuoDenseWs* ws) {
  if (!ws) return;
  void* ptrs[] = {ws->A_id, ws->vp, ws->dvpT, ws->part, ws->pre};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete ws;
}

int uuo_dense_backward(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A, const float* gV, UuoDenseWs* ws,
                       bool have_vp) {
  UUO_REQUIRE(m && pfaT && A && gV && ws && ws->F == F, "uuo_dense_backward: bad arguments / workspace of another size");
  UUO_REQUIRE(!uuo_recorder, "uuo_dense_backward: not available inside a lock-step batch");
  // v_posed: the forward's own contraction with identity skinning matrices and no translation
  if (!have_vp) {
    const int rc = uuo_launch_skin(m, s, F, pfaT, ws->A_id, nullptr, ws->vp, nullptr);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_dvp, dim3((m->VP / 16) / DVP_UNITS, ws->nFT), dim3(256), 0, s, F, m->V, m->VP, A, m->Wi, m->Ww, gV, ws->dvpT);
  hipLaunchKernelGGL(k_dA, dim3(UUO_NUM_JOINTS + 1, F), dim3(DA_T), 0, s, F, m->V, m->JLoff, m->JLv, m->JLw, gV, ws->vp, ws->pre);
  hipLaunchKernelGGL(k_dpf, dim3(8 * 4 * ws->nFT), dim3(DPF_WAVES * 64), 0, s, reinterpret_cast<const float4*>(m->PB),
                     reinterpret_cast<const float4*>(ws->dvpT), ws->part, ws->nFT, m->VP / 16, F);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}
