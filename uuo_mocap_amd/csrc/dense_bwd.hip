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
// vc = 3 (v % 16) + c inside the unit: group vc / 16, lane (vc % 4) * 16 + i, slot (vc / 4) % 4   (cf. pose_prep's `put`)
__global__ __launch_bounds__(256) void k_dvp(int F, int V, int VP, const float* __restrict__ A, const int* __restrict__ Wi,
                                             const float* __restrict__ Ww, const float* __restrict__ gV,
                                             float* __restrict__ dvpT) {
  __shared__ float sA[UUO_NUM_JOINTS * 12];
  const int f = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < UUO_NUM_JOINTS * 12; i += 256) sA[i] = A[(size_t)f * UUO_NUM_JOINTS * 12 + i];
  __syncthreads();
  const int v = blockIdx.x * 256 + tid;
  if (v >= VP) return;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (v < V) {
    const float* pg = gV + ((size_t)f * V + v) * 3;
    g0 = pg[0]; g1 = pg[1]; g2 = pg[2];
  }
  const int4 wi = *reinterpret_cast<const int4*>(Wi + (size_t)v * 4);
  const float4 ww = *reinterpret_cast<const float4*>(Ww + (size_t)v * 4);
  const int wj[4] = {wi.x, wi.y, wi.z, wi.w};
  const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
  float T[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) T[e] = 0.f;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float* pa = sA + wj[n] * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) T[r * 3 + c] = fmaf(wv[n], pa[r * 4 + c], T[r * 3 + c]);
  }
  const int nunits = VP / 16;
  const int ft = f >> 4, i = f & 15, u = v >> 4, vl = v & 15;
  float* tile = dvpT + ((size_t)ft * nunits + u) * 3 * 256;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float d = fmaf(T[6 + c], g2, fmaf(T[3 + c], g1, T[c] * g0));
    const int vc = 3 * vl + c;
    tile[(vc >> 4) * 256 + (((vc & 3) * 16 + i) << 2) + ((vc >> 2) & 3)] = d;
  }
}

// ---- d A_j = sum over the joint's vertices of w g [v_posed;1]^T; block 24 of a frame: d trans = sum_v g ------------------------
__global__ __launch_bounds__(64) void k_dA(int F, int V, const int* __restrict__ JLoff, const int* __restrict__ JLv,
                                           const float* __restrict__ JLw, const float* __restrict__ gV,
                                           const float* __restrict__ vp, float* __restrict__ pre) {
  const int j = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
  const float* gf = gV + (size_t)f * V * 3;
  float* out = pre + (size_t)f * UUO_PREG;
  if (j == UUO_NUM_JOINTS) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int v = lane; v < V; v += 64) {
      s0 += gf[v * 3];
      s1 += gf[v * 3 + 1];
      s2 += gf[v * 3 + 2];
    }
    s0 = wave_sum_fast(s0);
    s1 = wave_sum_fast(s1);
    s2 = wave_sum_fast(s2);
    if (lane == 0) {
      out[1] = s0;
      out[2] = s1;
      out[3] = s2;
    }
    return;
  }
  const float* pf = vp + (size_t)f * V * 3;
  float acc[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) acc[e] = 0.f;
  const int e0 = JLoff[j], e1 = JLoff[j + 1];
  for (int e = e0 + lane; e < e1; e += 64) {
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
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    const float t = wave_sum_fast(acc[e]);
    if (lane == 0) out[16 + j * 12 + e] = t;
  }
}

// ---- d [pose-feature | beta] = d v_posed . Baug^T on the matrix pipe -------------------------------------------------------------
// D[16 frames x 16 features] += A[16 frames x 4 coordinates] . B[4 coordinates x 16 features]; a unit of 16 vertices is 12
// K-steps, 14 feature tiles: 168 MFMAs, as many as the forward spends on it.  Block = (frame tile, one of 14 vertex chunks),
// its 8 waves take the chunk's units round-robin and keep 14 accumulator tiles; the waves' tiles are summed through LDS in
// wave order and written as the chunk's partial.  XCD x owns chunks x and x + 8: all frame tiles of a chunk share one L2
// slice of the basis (1.3 MB).
#define DPF_WAVES 8
__global__ __launch_bounds__(DPF_WAVES * 64) void k_dpf(const float4* __restrict__ PB, const float4* __restrict__ dvpT,
                                                         float* __restrict__ part, int nFT, int nunits, int F) {
  __shared__ float red[DPF_WAVES][14 * 256];
  const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
  const int cb = xcd + 8 * (pos & 1), ft = pos >> 1;
  if (cb >= UUO_DPF_NCB || ft >= nFT) return;  // block-uniform
  const int u0 = (cb * nunits) / UUO_DPF_NCB, u1 = ((cb + 1) * nunits) / UUO_DPF_NCB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  f32x4 acc[14];
#pragma unroll
  for (int jt = 0; jt < 14; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int u = u0 + wave; u < u1; u += DPF_WAVES) {
    const float4* pa = dvpT + ((size_t)ft * nunits + u) * 3 * 64 + lane;
    const float4 a0 = pa[0], a1 = pa[64], a2 = pa[128];
    const float4* pb = PB + (size_t)u * 14 * 3 * 64 + lane;
    float4 b[2][3];
    b[0][0] = pb[0];
    b[0][1] = pb[64];
    b[0][2] = pb[128];
#pragma unroll
    for (int jt = 0; jt < 14; ++jt) {
      const int cur = jt & 1;
      if (jt + 1 < 14) {
        b[cur ^ 1][0] = pb[((jt + 1) * 3 + 0) * 64];
        b[cur ^ 1][1] = pb[((jt + 1) * 3 + 1) * 64];
        b[cur ^ 1][2] = pb[((jt + 1) * 3 + 2) * 64];
      }
      f32x4 d = acc[jt];
#define DPF_STEP(av, bv)                                                  \
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, d, 0, 0, 0);       \
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, d, 0, 0, 0);       \
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, d, 0, 0, 0);       \
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, d, 0, 0, 0);
      DPF_STEP(a0, b[cur][0])
      DPF_STEP(a1, b[cur][1])
      DPF_STEP(a2, b[cur][2])
#undef DPF_STEP
      acc[jt] = d;
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

int uuo_dense_backward(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A, const float* gV, UuoDenseWs* ws) {
  UUO_REQUIRE(m && pfaT && A && gV && ws && ws->F == F, "uuo_dense_backward: bad arguments / workspace of another size");
  UUO_REQUIRE(!uuo_recorder, "uuo_dense_backward: not available inside a lock-step batch");
  // v_posed: the forward's own contraction with identity skinning matrices and no translation
  int rc = uuo_launch_skin(m, s, F, pfaT, ws->A_id, nullptr, ws->vp, nullptr);
  if (rc) return rc;
  hipLaunchKernelGGL(k_dvp, dim3((m->VP + 255) / 256, F), dim3(256), 0, s, F, m->V, m->VP, A, m->Wi, m->Ww, gV, ws->dvpT);
  hipLaunchKernelGGL(k_dA, dim3(UUO_NUM_JOINTS + 1, F), dim3(64), 0, s, F, m->V, m->JLoff, m->JLv, m->JLw, gV, ws->vp, ws->pre);
  hipLaunchKernelGGL(k_dpf, dim3(8 * 2 * ws->nFT), dim3(DPF_WAVES * 64), 0, s, reinterpret_cast<const float4*>(m->PB),
                     reinterpret_cast<const float4*>(ws->dvpT), ws->part, ws->nFT, m->VP / 16, F);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}
