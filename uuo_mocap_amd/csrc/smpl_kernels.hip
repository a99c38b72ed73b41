// SMPL forward on gfx950: per-frame pose preparation, MFMA blend + skinning, 45-joint gather.
// Replaces smplx.lbs.lbs / SMPL.forward as called by SmplInference.forward
// (reference src/video_mocap/utils/smpl.py:29-50).
#include "frame_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ----------------------------------------------------------------------------------------------------
// K_A  pose_prep: one wave per frame.  Writes the A operand of the blend GEMM (pose features | betas) in
// the frame-tile-major layout pfaT[ft][k][32], the 24 skinning matrices A[f][j][3x4] and posed joints.
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pose_prep(UuoPoseSrc src, const UuoTree* __restrict__ tree, int F,
                                                   float* __restrict__ pfaT, float* __restrict__ A,
                                                   float* __restrict__ jposed) {
  __shared__ FrameLds L;
  const int f = blockIdx.x;
  const int l = threadIdx.x;
  frame_forward(src, tree, f, L);
  const int ft = f >> 5, i = f & 31;
  float* tile = pfaT + (size_t)ft * UUO_KP * 32;
  if (l >= 1 && l < UUO_NUM_JOINTS) {
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      float v = L.R[l][e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
      tile[((l - 1) * 9 + e) * 32 + i] = v;
    }
  }
  if (l < 10) tile[(UUO_NUM_POSE_FEATS + l) * 32 + i] = L.beta[l];
  if (l >= 10 && l < 13) tile[(UUO_NUM_POSE_FEATS + l) * 32 + i] = 0.f;
  if (l < UUO_NUM_JOINTS) {
    float a12[12];
    frame_skin_matrix(L, l, a12);
    float* pa = A + ((size_t)f * UUO_NUM_JOINTS + l) * 12;
#pragma unroll
    for (int e = 0; e < 12; ++e) pa[e] = a12[e];
    if (jposed) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        jposed[((size_t)f * UUO_NUM_JOINTS + l) * 3 + c] = L.Gt[l][c] + (src.trans ? src.trans[(size_t)f * 3 + c] : 0.f);
    }
  }
}

int uuo_launch_pose_prep(const uuo_model* m, hipStream_t s, int F, const UuoPoseSrc& src, float* pfaT, float* A,
                         float* jposed) {
  hipLaunchKernelGGL(k_pose_prep, dim3(F), dim3(64), 0, s, src, m->tree, F, pfaT, A, jposed);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// K_B  skin: v_posed = v_template + [pose_feature | beta] . [posedirs ; shapedirs]   (exact-fp32 MFMA)
//            verts   = (sum_j W_vj A_fj) . [v_posed ; 1] + transl
// Block = 4 waves, tile = 32 frames x 128 vertices; each wave owns 32 vertices x 32 frames x 3 coords
// (three 32x32 accumulators).  MFMA 32x32x2 f32: lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; D row i = (reg&3) + 8*(reg>>2) + 4*(l>>5), column j = l&31, so after the K loop
// a lane holds x,y,z of ONE vertex for 16 frames and can skin them with no cross-lane traffic.
// B rows stream from HBM/L2 as 128-B segments (coordinate-planar table); the A tile and the 32x24
// skinning matrices sit in LDS.
// ----------------------------------------------------------------------------------------------------
#define SKIN_CHUNK 10                      // K-steps (of 2) per register buffer
#define SKIN_NCHUNK (UUO_KP / 2 / SKIN_CHUNK)  // 11

template <bool SPARSE>
__global__ __launch_bounds__(256) void k_skin(const float* __restrict__ P3, const float* __restrict__ vt3,
                                               const int* __restrict__ Wi, const float* __restrict__ Ww,
                                               const float* __restrict__ Wd, const float* __restrict__ pfaT,
                                               const float* __restrict__ A, const float* __restrict__ trans,
                                               float* __restrict__ verts, int F, int V, int VP, int nFT,
                                               int nblocks) {
  __shared__ float sA[UUO_KP * 32];            // [k][i]
  __shared__ float sT[32 * UUO_NUM_JOINTS * 12];  // [i][j][12]
  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous
  // range of logical ids so the frame tiles of one vertex tile reuse the same P3 rows from one L2.
  const int b = blockIdx.x;
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = b & 7, pos = b >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  const int vtile = logical / nFT, ft = logical - vtile * nFT;

  const int tid = threadIdx.x;
  {
    const float4* srcA = reinterpret_cast<const float4*>(pfaT + (size_t)ft * UUO_KP * 32);
    float4* dstA = reinterpret_cast<float4*>(sA);
    for (int i = tid; i < UUO_KP * 32 / 4; i += 256) dstA[i] = srcA[i];
    const float4* srcT = reinterpret_cast<const float4*>(A + (size_t)ft * 32 * UUO_NUM_JOINTS * 12);
    float4* dstT = reinterpret_cast<float4*>(sT);
    for (int i = tid; i < 32 * UUO_NUM_JOINTS * 12 / 4; i += 256) dstT[i] = srcT[i];
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kk = lane >> 5;
  const int v = vtile * 128 + wave * 32 + j;  // < VP by construction

  f32x16 acc0, acc1, acc2;
  {
    const float t0 = vt3[v], t1 = vt3[VP + v], t2 = vt3[2 * VP + v];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      acc0[e] = t0;
      acc1[e] = t1;
      acc2[e] = t2;
    }
  }
  const float* pb = P3 + (size_t)kk * VP + v;
  const size_t plane = (size_t)UUO_KP * VP;
  const float* pa = sA + kk * 32 + j;

  float a0[SKIN_CHUNK], b0[3][SKIN_CHUNK], a1[SKIN_CHUNK], b1[3][SKIN_CHUNK];
#define SKIN_LOAD(abuf, bbuf, chunk)                                        \
  _Pragma("unroll") for (int s_ = 0; s_ < SKIN_CHUNK; ++s_) {               \
    const int k2_ = ((chunk)*SKIN_CHUNK + s_) * 2;                          \
    abuf[s_] = pa[k2_ * 32];                                                \
    bbuf[0][s_] = pb[(size_t)k2_ * VP];                                     \
    bbuf[1][s_] = pb[plane + (size_t)k2_ * VP];                             \
    bbuf[2][s_] = pb[2 * plane + (size_t)k2_ * VP];                         \
  }
#define SKIN_COMPUTE(abuf, bbuf)                                                          \
  _Pragma("unroll") for (int s_ = 0; s_ < SKIN_CHUNK; ++s_) {                             \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(abuf[s_], bbuf[0][s_], acc0, 0, 0, 0);    \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(abuf[s_], bbuf[1][s_], acc1, 0, 0, 0);    \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(abuf[s_], bbuf[2][s_], acc2, 0, 0, 0);    \
  }
  SKIN_LOAD(a0, b0, 0);
#pragma unroll 1
  for (int it = 0; it < SKIN_NCHUNK - 1; it += 2) {
    SKIN_LOAD(a1, b1, it + 1);
    SKIN_COMPUTE(a0, b0);
    SKIN_LOAD(a0, b0, it + 2);
    SKIN_COMPUTE(a1, b1);
  }
  SKIN_COMPUTE(a0, b0);
#undef SKIN_LOAD
#undef SKIN_COMPUTE

  // ---- skinning epilogue: lane = vertex, 16 frames
  int wj[4];
  float ww[4];
  if (SPARSE) {
    const int4 wi4 = *reinterpret_cast<const int4*>(Wi + (size_t)v * 4);
    const float4 ww4 = *reinterpret_cast<const float4*>(Ww + (size_t)v * 4);
    wj[0] = wi4.x; wj[1] = wi4.y; wj[2] = wi4.z; wj[3] = wi4.w;
    ww[0] = ww4.x; ww[1] = ww4.y; ww[2] = ww4.z; ww[3] = ww4.w;
  }
  const bool vok = v < V;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = (e & 3) + 8 * (e >> 2) + 4 * kk;
    const int f = ft * 32 + i;
    float T[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) T[c] = 0.f;
    if (SPARSE) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float4* pt = reinterpret_cast<const float4*>(sT + (i * UUO_NUM_JOINTS + wj[n]) * 12);
        const float4 r0 = pt[0], r1 = pt[1], r2 = pt[2];
        const float w = ww[n];
        T[0] = fmaf(w, r0.x, T[0]); T[1] = fmaf(w, r0.y, T[1]); T[2] = fmaf(w, r0.z, T[2]); T[3] = fmaf(w, r0.w, T[3]);
        T[4] = fmaf(w, r1.x, T[4]); T[5] = fmaf(w, r1.y, T[5]); T[6] = fmaf(w, r1.z, T[6]); T[7] = fmaf(w, r1.w, T[7]);
        T[8] = fmaf(w, r2.x, T[8]); T[9] = fmaf(w, r2.y, T[9]); T[10] = fmaf(w, r2.z, T[10]); T[11] = fmaf(w, r2.w, T[11]);
      }
    } else {
      for (int jn = 0; jn < UUO_NUM_JOINTS; ++jn) {
        const float w = vok ? Wd[(size_t)v * UUO_NUM_JOINTS + jn] : 0.f;
        const float* pt = sT + (i * UUO_NUM_JOINTS + jn) * 12;
#pragma unroll
        for (int c = 0; c < 12; ++c) T[c] = fmaf(w, pt[c], T[c]);
      }
    }
    const float px = acc0[e], py = acc1[e], pz = acc2[e];
    float ox = fmaf(T[2], pz, fmaf(T[1], py, T[0] * px)) + T[3];
    float oy = fmaf(T[6], pz, fmaf(T[5], py, T[4] * px)) + T[7];
    float oz = fmaf(T[10], pz, fmaf(T[9], py, T[8] * px)) + T[11];
    if (vok && f < F) {
      if (trans) {
        ox += trans[(size_t)f * 3 + 0];
        oy += trans[(size_t)f * 3 + 1];
        oz += trans[(size_t)f * 3 + 2];
      }
      float* po = verts + ((size_t)f * V + v) * 3;
      po[0] = ox;
      po[1] = oy;
      po[2] = oz;
    }
  }
}

int uuo_launch_skin(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A, const float* trans,
                    float* verts) {
  const int nFT = (F + 31) / 32;
  const int nblocks = nFT * (m->VP / 128);
  if (m->nnz <= 4)
    hipLaunchKernelGGL(k_skin<true>, dim3(nblocks), dim3(256), 0, s, m->P3, m->vt3, m->Wi, m->Ww, m->W, pfaT, A, trans,
                       verts, F, m->V, m->VP, nFT, nblocks);
  else
    hipLaunchKernelGGL(k_skin<false>, dim3(nblocks), dim3(256), 0, s, m->P3, m->vt3, m->Wi, m->Ww, m->W, pfaT, A,
                       trans, verts, F, m->V, m->VP, nFT, nblocks);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// 45 joints of smplx SMPL.forward: 24 posed joints + 21 vertex-picked joints (VertexJointSelector)
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_joints45(const UuoTree* __restrict__ tree, int F, int V,
                                                  const float* __restrict__ jposed, const float* __restrict__ verts,
                                                  float* __restrict__ out) {
  const int f = blockIdx.x, l = threadIdx.x;
  if (l < UUO_NUM_JOINTS) {
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((size_t)f * 45 + l) * 3 + c] = jposed[((size_t)f * UUO_NUM_JOINTS + l) * 3 + c];
  } else if (l < 45) {
    const int vid = tree->extra_vids[l - UUO_NUM_JOINTS];
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((size_t)f * 45 + l) * 3 + c] = verts[((size_t)f * V + vid) * 3 + c];
  }
}

int uuo_launch_joints45(const uuo_model* m, hipStream_t s, int F, const float* jposed, const float* verts,
                        float* out) {
  hipLaunchKernelGGL(k_joints45, dim3(F), dim3(64), 0, s, m->tree, F, m->V, jposed, verts, out);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// C ABI: SmplInference.forward
// ----------------------------------------------------------------------------------------------------
extern "C" int uuo_smpl_forward(uuo_model_t* m, void* stream, int F, const float* d_poses, const float* d_betas,
                                int betas_rows, const float* d_root, const float* d_trans, float* d_verts,
                                float* d_joints) {
  UUO_REQUIRE(m && d_poses && d_betas && d_root, "uuo_smpl_forward: null argument");
  UUO_REQUIRE(F > 0, "uuo_smpl_forward: F must be positive");
  UUO_REQUIRE(betas_rows == 1 || betas_rows == F, "uuo_smpl_forward: betas rows must be 1 or F");
  UUO_REQUIRE(d_verts != nullptr, "uuo_smpl_forward: d_verts is required (joints 24..44 are picked from it)");
  hipStream_t s = (hipStream_t)stream;
  const int nFT = (F + 31) / 32;
  uuo_model::FwdScratch sc;
  {
    std::lock_guard<std::mutex> lock(m->fwd_mutex);
    uuo_model::FwdScratch& ref = m->fwd[s];
    if (ref.cap < nFT) {
      if (ref.pfaT) (void)hipFree(ref.pfaT);
      if (ref.A) (void)hipFree(ref.A);
      if (ref.jp) (void)hipFree(ref.jp);
      ref = uuo_model::FwdScratch();
      UUO_HIP_CHECK(hipMalloc((void**)&ref.pfaT, (size_t)nFT * UUO_KP * 32 * sizeof(float)));
      UUO_HIP_CHECK(hipMalloc((void**)&ref.A, (size_t)nFT * 32 * UUO_NUM_JOINTS * 12 * sizeof(float)));
      UUO_HIP_CHECK(hipMalloc((void**)&ref.jp, (size_t)nFT * 32 * UUO_NUM_JOINTS * 3 * sizeof(float)));
      UUO_HIP_CHECK(hipMemset(ref.pfaT, 0, (size_t)nFT * UUO_KP * 32 * sizeof(float)));
      UUO_HIP_CHECK(hipMemset(ref.A, 0, (size_t)nFT * 32 * UUO_NUM_JOINTS * 12 * sizeof(float)));
      ref.cap = nFT;
    }
    sc = ref;
  }
  UuoPoseSrc src;
  src.body = d_poses;
  src.norm_body = 0;
  src.root = d_root;
  src.root_mode = UUO_ROOT_RAW;
  src.z = nullptr;
  src.betas = d_betas;
  src.betas_stride = (betas_rows == 1) ? 0 : 10;
  src.trans = d_trans;
  int rc = uuo_launch_pose_prep(m, s, F, src, sc.pfaT, sc.A, sc.jp);
  if (rc) return rc;
  rc = uuo_launch_skin(m, s, F, sc.pfaT, sc.A, d_trans, d_verts);
  if (rc) return rc;
  if (d_joints) rc = uuo_launch_joints45(m, s, F, sc.jp, d_verts, d_joints);
  return rc;
}
