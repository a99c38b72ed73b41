// SMPL forward on gfx950: per-frame pose preparation, MFMA blend + skinning, 45-joint gather.
// Replaces smplx.lbs.lbs / SMPL.forward as called by SmplInference.forward
// (reference src/video_mocap/utils/smpl.py:29-50).
#include <cstdlib>

#include "frame_math.h"


// ----------------------------------------------------------------------------------------------------
// K_A  pose_prep: one wave per frame.  Writes the A operand of the blend GEMM (pose features | betas) in
// MFMA-operand order pfaT[ft][14][64][4], the 24 skinning matrices A[f][j][3x4] and posed joints.
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pose_prep(UuoPoseSrc src, const UuoTree* __restrict__ tree, int F,
                                                   float* __restrict__ pfaT, float* __restrict__ A,
                                                   float* __restrict__ jposed) {
  __shared__ FrameLds L;
  const int f = blockIdx.x;
  const int l = threadIdx.x;
  frame_forward(src, tree, f, L);
  // A operand in MFMA order: lane (k&3)*16 + i of group (k>>4) holds A[i][k] at slot (k>>2)&3  (see model.hip)
  const int ft = f / UUO_FT, i = f % UUO_FT;
  float* tile = pfaT + (size_t)ft * UUO_KP * UUO_FT;
  auto put = [&](int k, float v) { tile[(((k >> 4) * 64 + ((k & 3) * 16 + i)) << 2) + ((k >> 2) & 3)] = v; };
  if (l >= 1 && l < UUO_NUM_JOINTS) {
#pragma unroll
    for (int e = 0; e < 9; ++e) put((l - 1) * 9 + e, L.R[l][e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f));
  }
  if (l < 10) put(UUO_NUM_POSE_FEATS + l, L.beta[l]);
  if (l >= 10 && l < 17) put(UUO_NUM_POSE_FEATS + l, 0.f);
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
// Work unit = 16 frames x 16 vertices x 3 coordinates on `v_mfma_f32_16x16x4_f32`: lane l supplies
// A[i = l&15][k = l>>4] (frame i of the tile) and B[k = l>>4][j = l&15] (vertex j); D has column j = l&15 and
// rows i = 4*(l>>4) + reg, so after the K loop a lane holds x,y,z of ONE vertex for 4 frames and skins them
// with no cross-lane traffic.  A block is 8 waves on one 16-frame tile (A tile 14 KB + 16x24 skinning matrices
// 18 KB in LDS, shared); its waves take vertex units round-robin.  The grid is sized to one resident round
// (<= 256 blocks): F=300 -> 19 frame tiles x 13 vertex ranges = 247 blocks, 432 units per frame tile, so the
// 8 208 units spread over 1 976 waves (2 per SIMD: one wave's MFMA phase overlaps the other's VALU epilogue).
// The B table is stored in MFMA-operand order (model.hip), so one global_load_dwordx4 per wave brings 4 K-steps
// of B as a fully coalesced 1-KB block (the vector-memory pipe costs ~16 cycles per wave instruction whatever its
// width: dword loads made the kernel TA-bound); the A tile uses the same order in LDS (ds_read_b128).  Block ids are remapped so that all frame tiles of one vertex range share an XCD's L2.
// ----------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define UUO_BIG 3.0e38f
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {  // lanes without a source keep their own value
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_min(float v) {  // row_ror:8,4,2,1 -> reduction over the 16 lanes of a row
  v = fminf(v, dpp_row<0x128>(v));
  v = fminf(v, dpp_row<0x124>(v));
  v = fminf(v, dpp_row<0x122>(v));
  return fminf(v, dpp_row<0x121>(v));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_row<0x128>(v));
  v = fmaxf(v, dpp_row<0x124>(v));
  v = fmaxf(v, dpp_row<0x122>(v));
  return fmaxf(v, dpp_row<0x121>(v));
}
#define UUO_BIG 3.0e38f
#define SKIN_WAVES 8
#define SKIN_GROUPS (UUO_KP / 16)  // 14 groups of 4 K-steps (one dwordx4 per lane per coordinate each)
#define SKIN_CG 2                  // groups per register buffer  -> 7 chunks, 24 MFMAs each
#define SKIN_NCHUNK (SKIN_GROUPS / SKIN_CG)

template <bool SPARSE, int VAR>
__global__ __launch_bounds__(SKIN_WAVES * 64) void k_skin(const float* __restrict__ P3, const float* __restrict__ vt3,
                                                           const int* __restrict__ Wi, const float* __restrict__ Ww,
                                                           const float* __restrict__ Wd, const float* __restrict__ pfaT,
                                                           const float* __restrict__ A, const float* __restrict__ trans,
                                                           float* __restrict__ verts, float* __restrict__ bbox, int F,
                                                           int V, int VP, int nFT, int nVB, int nblocks) {
  __shared__ float sA[UUO_KP * UUO_FT];                    // [k][i]
  __shared__ float sT[UUO_FT * UUO_NUM_JOINTS * 12];       // [i][j][12]
  const int b = blockIdx.x;
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = b & 7, pos = b >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  const int vb = logical / nFT, ft = logical - vb * nFT;

  const int tid = threadIdx.x;
  {
    const float4* srcA = reinterpret_cast<const float4*>(pfaT + (size_t)ft * UUO_KP * UUO_FT);
    float4* dstA = reinterpret_cast<float4*>(sA);
    for (int i = tid; i < UUO_KP * UUO_FT / 4; i += SKIN_WAVES * 64) dstA[i] = srcA[i];
    const float4* srcT = reinterpret_cast<const float4*>(A + (size_t)ft * UUO_FT * UUO_NUM_JOINTS * 12);
    float4* dstT = reinterpret_cast<float4*>(sT);
    for (int i = tid; i < UUO_FT * UUO_NUM_JOINTS * 12 / 4; i += SKIN_WAVES * 64) dstT[i] = srcT[i];
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, kq = lane >> 4;
  // Waves w and w+4 share a SIMD and run the same program: delay the second half by about half a unit's MFMA time
  // so one wave's VALU/LDS epilogue falls under its partner's MFMA phase instead of both phases colliding.
  if (VAR != 4 && __builtin_amdgcn_readfirstlane(wave) >= SKIN_WAVES / 2) __builtin_amdgcn_s_sleep(40);
  const int nunits = VP / 16;
  const int u_begin = (int)(((long)vb * nunits) / nVB), u_end = (int)(((long)(vb + 1) * nunits) / nVB);
  const size_t cplane = (size_t)nunits * SKIN_GROUPS * 64;  // float4 elements per coordinate plane
  const float4* pa = reinterpret_cast<const float4*>(sA) + lane;
  const float4* P3v = reinterpret_cast<const float4*>(P3);

  // Software pipeline across units: the first K-chunk, the template values and the skin weights of the NEXT unit are
  // requested before the current unit's epilogue, so neither their latency nor the epilogue's stores (vmcnt is
  // in-order) sit in front of the next unit's first MFMA.
#define SKIN_LOAD(abuf, bbuf, pbase, chunk)                                                   \
  _Pragma("unroll") for (int g_ = 0; g_ < SKIN_CG; ++g_) {                                    \
    const int gi_ = (chunk)*SKIN_CG + g_;                                                     \
    abuf[g_] = pa[gi_ * 64];                                                                  \
    if (VAR == 1) {                                                                           \
      bbuf[0][g_] = bbuf[1][g_] = bbuf[2][g_] = make_float4(1e-3f * lane, 2e-3f, 3e-3f, 4e-3f); \
    } else {                                                                                  \
      bbuf[0][g_] = (pbase)[gi_ * 64];                                                        \
      bbuf[1][g_] = (pbase)[cplane + gi_ * 64];                                               \
      bbuf[2][g_] = (pbase)[2 * cplane + gi_ * 64];                                           \
    }                                                                                         \
  }
#define SKIN_MFMA3(av, bv0, bv1, bv2)                                          \
  if (VAR == 2) {                                                              \
    asm volatile("" ::"v"(av), "v"(bv0), "v"(bv1), "v"(bv2));                  \
  } else {                                                                     \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv0, acc0, 0, 0, 0);       \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv1, acc1, 0, 0, 0);       \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv2, acc2, 0, 0, 0);       \
  }
#define SKIN_COMPUTE(abuf, bbuf)                                               \
  _Pragma("unroll") for (int g_ = 0; g_ < SKIN_CG; ++g_) {                     \
    SKIN_MFMA3(abuf[g_].x, bbuf[0][g_].x, bbuf[1][g_].x, bbuf[2][g_].x)        \
    SKIN_MFMA3(abuf[g_].y, bbuf[0][g_].y, bbuf[1][g_].y, bbuf[2][g_].y)        \
    SKIN_MFMA3(abuf[g_].z, bbuf[0][g_].z, bbuf[1][g_].z, bbuf[2][g_].z)        \
    SKIN_MFMA3(abuf[g_].w, bbuf[0][g_].w, bbuf[1][g_].w, bbuf[2][g_].w)        \
  }
  float4 a0[SKIN_CG], b0[3][SKIN_CG], a1[SKIN_CG], b1[3][SKIN_CG];
  float tn0 = 0.f, tn1 = 0.f, tn2 = 0.f;
  int4 wi_n = make_int4(0, 0, 0, 0);
  float4 ww_n = make_float4(0.f, 0.f, 0.f, 0.f);
  int u = u_begin + wave;
  if (u < u_end) {
    const int v = u * 16 + j;
    const float4* pb = P3v + (size_t)u * SKIN_GROUPS * 64 + lane;
    SKIN_LOAD(a0, b0, pb, 0);
    tn0 = vt3[v];
    tn1 = vt3[VP + v];
    tn2 = vt3[2 * VP + v];
    if (SPARSE) {
      wi_n = *reinterpret_cast<const int4*>(Wi + (size_t)v * 4);
      ww_n = *reinterpret_cast<const float4*>(Ww + (size_t)v * 4);
    }
  }
  for (; u < u_end; u += SKIN_WAVES) {
    const int v = u * 16 + j;  // < VP
    const float4* pb = P3v + (size_t)u * SKIN_GROUPS * 64 + lane;
    f32x4 acc0, acc1, acc2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc0[e] = tn0;
      acc1[e] = tn1;
      acc2[e] = tn2;
    }
    const int4 wi4 = wi_n;
    const float4 ww4 = ww_n;
#pragma unroll 1
    for (int it = 0; it < SKIN_NCHUNK - 1; it += 2) {
      SKIN_LOAD(a1, b1, pb, it + 1);
      SKIN_COMPUTE(a0, b0);
      SKIN_LOAD(a0, b0, pb, it + 2);
      SKIN_COMPUTE(a1, b1);
    }
    SKIN_COMPUTE(a0, b0);
    {
      const int un = u + SKIN_WAVES;
      if (un < u_end) {  // wave-uniform
        const int vn = un * 16 + j;
        const float4* pbn = P3v + (size_t)un * SKIN_GROUPS * 64 + lane;
        SKIN_LOAD(a0, b0, pbn, 0);
        tn0 = vt3[vn];
        tn1 = vt3[VP + vn];
        tn2 = vt3[2 * VP + vn];
        if (SPARSE) {
          wi_n = *reinterpret_cast<const int4*>(Wi + (size_t)vn * 4);
          ww_n = *reinterpret_cast<const float4*>(Ww + (size_t)vn * 4);
        }
      }
    }

    // ---- skinning epilogue: lane = vertex j, frames 4*kq .. 4*kq+3
    int wj[4];
    float ww[4];
    if (SPARSE) {
      wj[0] = wi4.x; wj[1] = wi4.y; wj[2] = wi4.z; wj[3] = wi4.w;
      ww[0] = ww4.x; ww[1] = ww4.y; ww[2] = ww4.z; ww[3] = ww4.w;
    }
    const bool vok = v < V;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * kq + e;
      const int f = ft * UUO_FT + i;
      float T[12];
#pragma unroll
      for (int c = 0; c < 12; ++c) T[c] = 0.f;
      if (VAR == 5) {
        T[0] = T[5] = T[10] = ww[0] + 1.f;
      } else if (SPARSE) {
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
      if (trans && f < F) {
        ox += trans[(size_t)f * 3 + 0];
        oy += trans[(size_t)f * 3 + 1];
        oz += trans[(size_t)f * 3 + 2];
      }
      if (VAR == 3) {
        asm volatile("" ::"v"(ox), "v"(oy), "v"(oz));
      } else if (vok && f < F) {
        float* po = verts + ((size_t)f * V + v) * 3;
        po[0] = ox;
        po[1] = oy;
        po[2] = oz;
      }
      if (bbox) {
        // bounding box of the unit's 16 vertices in frame f: they sit on the 16 lanes of one DPP row, so four
        // row-rotate steps (v_min/v_max with a DPP operand, no LDS traffic) leave the box on every lane
        const float lx = row16_min(vok ? ox : UUO_BIG), ly = row16_min(vok ? oy : UUO_BIG),
                    lz = row16_min(vok ? oz : UUO_BIG);
        const float hx = row16_max(vok ? ox : -UUO_BIG), hy = row16_max(vok ? oy : -UUO_BIG),
                    hz = row16_max(vok ? oz : -UUO_BIG);
        if (j == 0 && f < F) {
          float* pbx = bbox + ((size_t)f * nunits + u) * 6;
          pbx[0] = lx; pbx[1] = ly; pbx[2] = lz;
          pbx[3] = hx; pbx[4] = hy; pbx[5] = hz;
        }
      }
    }
  }
#undef SKIN_LOAD
#undef SKIN_COMPUTE
#undef SKIN_MFMA3
}

int uuo_launch_skin(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A, const float* trans,
                    float* verts, float* bbox) {
  const int nFT = (F + UUO_FT - 1) / UUO_FT;
  const int nunits = m->VP / 16;
  // vertex ranges per frame tile: one resident round of <= 256 blocks, every wave gets at least one unit
  static const int slots = getenv("UUO_SKIN_SLOTS") ? atoi(getenv("UUO_SKIN_SLOTS")) : 256;
  int nVB = slots / nFT;
  const int maxVB = (nunits + SKIN_WAVES - 1) / SKIN_WAVES;
  if (nVB > maxVB) nVB = maxVB;
  if (nVB < 1) nVB = 1;
  const int nblocks = nFT * nVB;
  static const int variant = getenv("UUO_SKIN_VARIANT") ? atoi(getenv("UUO_SKIN_VARIANT")) : 0;  // ablation only
#define SKIN_LAUNCH(SP, VAR)                                                                                     \
  hipLaunchKernelGGL((k_skin<SP, VAR>), dim3(nblocks), dim3(SKIN_WAVES * 64), 0, s, m->P3, m->vt3, m->Wi, m->Ww, \
                     m->W, pfaT, A, trans, verts, bbox, F, m->V, m->VP, nFT, nVB, nblocks)
  if (m->nnz > 4) SKIN_LAUNCH(false, 0);
  else if (variant == 1) SKIN_LAUNCH(true, 1);
  else if (variant == 2) SKIN_LAUNCH(true, 2);
  else if (variant == 3) SKIN_LAUNCH(true, 3);
  else if (variant == 4) SKIN_LAUNCH(true, 4);
  else if (variant == 5) SKIN_LAUNCH(true, 5);
  else SKIN_LAUNCH(true, 0);
#undef SKIN_LAUNCH
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
  const int nFT = (F + UUO_FT - 1) / UUO_FT;
  uuo_model::FwdScratch sc;
  {
    std::lock_guard<std::mutex> lock(m->fwd_mutex);
    uuo_model::FwdScratch& ref = m->fwd[s];
    if (ref.cap < nFT) {
      if (ref.pfaT) (void)hipFree(ref.pfaT);
      if (ref.A) (void)hipFree(ref.A);
      if (ref.jp) (void)hipFree(ref.jp);
      ref = uuo_model::FwdScratch();
      UUO_HIP_CHECK(hipMalloc((void**)&ref.pfaT, (size_t)nFT * UUO_KP * UUO_FT * sizeof(float)));
      UUO_HIP_CHECK(hipMalloc((void**)&ref.A, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 12 * sizeof(float)));
      UUO_HIP_CHECK(hipMalloc((void**)&ref.jp, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 3 * sizeof(float)));
      UUO_HIP_CHECK(hipMemset(ref.pfaT, 0, (size_t)nFT * UUO_KP * UUO_FT * sizeof(float)));
      UUO_HIP_CHECK(hipMemset(ref.A, 0, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 12 * sizeof(float)));
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
  rc = uuo_launch_skin(m, s, F, sc.pfaT, sc.A, d_trans, d_verts, nullptr);
  if (rc) return rc;
  if (d_joints) rc = uuo_launch_joints45(m, s, F, sc.jp, d_verts, d_joints);
  return rc;
}
