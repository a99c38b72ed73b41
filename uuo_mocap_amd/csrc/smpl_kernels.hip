// SMPL forward on gfx950: per-frame pose preparation, MFMA blend + skinning, 45-joint gather.
// Replaces smplx.lbs.lbs / SMPL.forward as called by SmplInference.forward
// (reference src/video_mocap/utils/smpl.py:29-50).
#include <cstdlib>

#include "frame_math.h"


// ----------------------------------------------------------------------------------------------------
// K_A  pose_prep: one wave per frame.  Writes the A operand of the blend GEMM (pose features | betas) in
// MFMA-operand order pfaT[ft][14][64][4], the 24 skinning matrices A[f][j][3x4] and posed joints.
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pose_prep_body(const UuoPoseSrc& src, const UuoTree* __restrict__ tree, int F,
                                               float* __restrict__ pfaT, float* __restrict__ A,
                                               float* __restrict__ jposed, float* __restrict__ frames,
                                               const float* __restrict__ ST = nullptr, const int* __restrict__ Wi = nullptr,
                                               const float* __restrict__ Ww = nullptr, int V = 0,
                                               const int32_t* __restrict__ subset = nullptr, int ns = 0,
                                               float* __restrict__ sb_out = nullptr, _Float16* __restrict__ pfa16 = nullptr) {
  __builtin_amdgcn_s_setprio(3);  // latency-bound kernel: do not queue behind co-resident MFMA waves
  __shared__ FrameLds L;
  const int f = blockIdx.x;
  const int l = threadIdx.x;
  frame_forward(src, tree, f, L);
  if (frames) {  // the backward kernel of this closure starts from these instead of redoing the kinematic chain
    constexpr int NW = sizeof(FrameLds) / 4;
    const float* src_l = reinterpret_cast<const float*>(&L);
    for (int i = l; i < NW; i += 64) frames[(size_t)f * NW + i] = src_l[i];
  }
  // A operand in MFMA order: lane (k&3)*16 + i of group (k>>4) holds A[i][k] at slot (k>>2)&3  (see model.hip)
  const int ft = f / UUO_FT, i = f % UUO_FT;
  float* tile = pfaT + (size_t)ft * UUO_KP * UUO_FT;
  // the same operand for k_skin3 (when asked for): 128 x the value as hi = fp16(x), lo = fp16(x - hi) in v_mfma_f32_16x16x32_f16
  // order pfa16[ft][step k >> 5][plane][lane][slot 4 ((k >> 4) & 1) + ((k >> 2) & 3)]  (see model.hip, P16)
  _Float16* tile16 = pfa16 ? pfa16 + (size_t)ft * UUO_KP * UUO_FT * 2 : nullptr;
  auto put = [&](int k, float v) {
    tile[(((k >> 4) * 64 + ((k & 3) * 16 + i)) << 2) + ((k >> 2) & 3)] = v;
    if (tile16) {
      const float x = v * UUO_SK16_ASCALE;
      const _Float16 hi = (_Float16)x;
      const int o = (((k >> 5) * 2) * 64 + ((k & 3) * 16 + i)) * 8 + (((k >> 4) & 1) * 4 + ((k >> 2) & 3));
      tile16[o] = hi;
      tile16[o + 512] = (_Float16)(x - (float)hi);
    }
  };
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
  if (sb_out) {
    // part stage: per vertex of the candidate, what k_part_fwd needs at every frame, packed in subset order so that its
    // loads are coalesced: [S[v] . beta (3 floats, one beta for all frames; accumulation order of skin_cached_body) |
    // the 4 skinning joints, 3 * joint per byte] [the 4 skin weights].  A slice of the subset per frame block.
    const int per = (ns + F - 1) / F, c0 = f * per;
    for (int i = l; i < per; i += 64) {
      const int c = c0 + i;
      if (c >= ns) break;
      const int v = subset[c];
      float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)v < (unsigned)V) {
        float sb[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int comp = 0; comp < 3; ++comp)
#pragma unroll
          for (int k = 0; k < 10; ++k) sb[comp] = fmaf(ST[(size_t)v * 30 + comp * 10 + k], src.betas[k], sb[comp]);
        const int4 wi = *reinterpret_cast<const int4*>(Wi + (size_t)v * 4);
        o1 = *reinterpret_cast<const float4*>(Ww + (size_t)v * 4);
        // byte n = 3 * joint n (the float4 index of its skinning matrix); an unused slot becomes joint 0 with weight 0,
        // which adds exact zeros to the blend
        const int wj[4] = {wi.x, wi.y, wi.z, wi.w};
        float wv[4] = {o1.x, o1.y, o1.z, o1.w};
        unsigned pj = 0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if (wj[n] < 0) wv[n] = 0.f;
          else pj |= (unsigned)(wj[n] * 3) << (8 * n);
        }
        o1 = make_float4(wv[0], wv[1], wv[2], wv[3]);
        o0 = make_float4(sb[0], sb[1], sb[2], __uint_as_float(pj));
      }
      reinterpret_cast<float4*>(sb_out)[(size_t)c * 2] = o0;
      reinterpret_cast<float4*>(sb_out)[(size_t)c * 2 + 1] = o1;
    }
  }
}

struct PosePrepArgs {
  UuoGridHdr h;
  UuoPoseSrc src;
  uuo_gptr<const UuoTree> tree;
  int F;
  uuo_gptr<float> pfaT;
  uuo_gptr<float> A;
  uuo_gptr<float> jposed;
  uuo_gptr<float> frames;
  uuo_gptr<const float> ST;        // part stage (k_part_fwd follows): shape blend and skin weight tables, the candidate's vertices,
  uuo_gptr<const int> Wi;          // their count and the [ns][8] buffer that takes their per-vertex constants; null / 0 otherwise
  uuo_gptr<const float> Ww;
  int V;
  uuo_gptr<const int32_t> subset;
  int ns;
  uuo_gptr<float> sb_out;
  uuo_gptr<_Float16> pfa16;        // k_skin3 follows: the fp16-split copy of the A operand; null otherwise
};
__global__ __launch_bounds__(64) void k_pose_prep(PosePrepArgs a) {
  pose_prep_body(a.src, a.tree, a.F, a.pfaT, a.A, a.jposed, a.frames, a.ST, a.Wi, a.Ww, a.V, a.subset, a.ns, a.sb_out, a.pfa16);
}
__global__ __launch_bounds__(64) void k_pose_prep_b(const PosePrepArgs* __restrict__ batch) {
  UUO_BATCH_PICK(PosePrepArgs, batch)
  pose_prep_body(a.src, a.tree, a.F, a.pfaT, a.A, a.jposed, a.frames, a.ST, a.Wi, a.Ww, a.V, a.subset, a.ns, a.sb_out, a.pfa16);
}

int uuo_launch_pose_prep(const uuo_model* m, hipStream_t s, int F, const UuoPoseSrc& src, float* pfaT, float* A,
                         float* jposed, float* frames, const int32_t* sb_subset, int sb_ns, float* sb_out, void* pfa16) {
  PosePrepArgs a{{F, 1}, src, m->tree, F, pfaT, A, jposed, frames, sb_out ? m->ST : nullptr, m->Wi, m->Ww, m->V, sb_subset, sb_ns, sb_out,
                 (_Float16*)pfa16};
  if (uuo_record(UUO_OP_POSE_PREP, F, 1, a)) return 0;
  hipLaunchKernelGGL(k_pose_prep, dim3(F), dim3(64), 0, s, a);
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
// cache policy of k_skin2's vertex stores (buffer instruction aux bits; 2 = nt: 25 MB written once per launch and read back
// only in part by the pruned search need not displace the basis slices from the L2s) -- A/B builds
#ifndef SK2_ST_AUX
#define SK2_ST_AUX 0
#endif
#ifndef SKIN_WAVES
#define SKIN_WAVES 8  // waves per block of the skinning kernels (4: A/B builds only, one wave per SIMD)
#endif
#define SKIN_GROUPS (UUO_KP / 16)  // 14 groups of 4 K-steps (one dwordx4 per lane per coordinate each)
#define SKIN_CG 2                  // groups per register buffer  -> 7 chunks, 24 MFMAs each
#define SKIN_NCHUNK (SKIN_GROUPS / SKIN_CG)

template <int VAR>
__global__ __launch_bounds__(SKIN_WAVES * 64) void k_skin(const float* __restrict__ P3, const float* __restrict__ vt3,
                                                           const int* __restrict__ Wi, const float* __restrict__ Ww,
                                                           const float* __restrict__ pfaT,
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
    wi_n = *reinterpret_cast<const int4*>(Wi + (size_t)v * 4);
    ww_n = *reinterpret_cast<const float4*>(Ww + (size_t)v * 4);
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
        wi_n = *reinterpret_cast<const int4*>(Wi + (size_t)vn * 4);
        ww_n = *reinterpret_cast<const float4*>(Ww + (size_t)vn * 4);
      }
    }

    // ---- skinning epilogue: lane = vertex j, frames 4*kq .. 4*kq+3
    const int wj[4] = {wi4.x, wi4.y, wi4.z, wi4.w};
    const float ww[4] = {ww4.x, ww4.y, ww4.z, ww4.w};
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
      } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const float4* pt = reinterpret_cast<const float4*>(sT + (i * UUO_NUM_JOINTS + wj[n]) * 12);
          const float4 r0 = pt[0], r1 = pt[1], r2 = pt[2];
          const float w = ww[n];
          T[0] = fmaf(w, r0.x, T[0]); T[1] = fmaf(w, r0.y, T[1]); T[2] = fmaf(w, r0.z, T[2]); T[3] = fmaf(w, r0.w, T[3]);
          T[4] = fmaf(w, r1.x, T[4]); T[5] = fmaf(w, r1.y, T[5]); T[6] = fmaf(w, r1.z, T[6]); T[7] = fmaf(w, r1.w, T[7]);
          T[8] = fmaf(w, r2.x, T[8]); T[9] = fmaf(w, r2.y, T[9]); T[10] = fmaf(w, r2.z, T[10]); T[11] = fmaf(w, r2.w, T[11]);
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
        if (j == 0 && f < F && u * 16 < V) {
          float* pbx = bbox + ((size_t)f * ((V + 15) / 16) + u) * 6;
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

// ----------------------------------------------------------------------------------------------------
// K_B2  skin, self-overlapped variant (sparse weights, the default path).
// Same work unit as k_skin (16 frames x 16 vertices x 3 coordinates, 168 x v_mfma_f32_16x16x4_f32), but a wave
// never leaves the matrix pipe idle between units:
//   * every wave owns a CONTIGUOUS run of (frame tile, unit) tasks of one XCD's vertex range (balanced split:
//     F=300 -> 1026 tasks per XCD over 256 waves = 4 or 5 each); a block's run touches at most two frame tiles,
//     whose A operand tiles, skinning matrices and translations are staged in two LDS slots;
//   * the skinning epilogue (and stores, bounding boxes) of unit t-1 is cut into 12 pieces that are placed between
//     the MFMA groups of unit t, so its VALU / LDS / store work issues in the shadow of the 32-cycle MFMAs;
//   * B operands stream through a 7-stage register ring (one stage = 4 K-steps x 3 coordinates = 12 MFMAs), refilled
//     7 groups (~2 700 cycles) ahead and across the unit boundary, so L2 latency and the in-order vmcnt behind the
//     epilogue's stores stay off the critical path;
//   * stores go through raw buffer instructions (out-of-range lanes are dropped by the bounds check, no exec
//     juggling inside the MFMA stream): one 12-byte store per vertex and frame.
// XCD x (blockIdx & 7) works on units [x*nunits/8, (x+1)*nunits/8) for all frame tiles: 2.3 MB of the blend basis
// per XCD L2.
// ----------------------------------------------------------------------------------------------------
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
#define SK2_RING 7
#ifndef UUO_SK2_PRIO
#define UUO_SK2_PRIO 0  // raising the MFMA waves above the latency-bound kernels of other hypotheses cost 2 % of fit throughput
#endif
#define SK2_NPOS_LOG2 5  // 32 blocks per XCD: one 8-wave block per CU
#define SK2_MAX_FT 31  // frame tiles per launch (a block's task run must stay within two frame tiles)

struct Sk2Unit {   // one task (wave-uniform)
  int u;           // unit
  int i0;          // first frame (within the launch) of the tile; >= F marks "nothing to store"
  int slot;        // LDS slot of the tile
};

// min / max over the 16 lanes of each DPP row for six values at once: v_min/v_max with a row-rotated DPP operand,
// one rotate distance per call (8, 4, 2, 1 -> the box is on every lane).  The six chains are interleaved so that no
// instruction reads a register written less than five instructions earlier (a DPP read needs two wait states after
// a VALU write; the leading s_nop covers the producers of the first one).
#define SK2_BOX_STEP(R, l0, l1, l2, h0, h1, h2)                           \
  asm("s_nop 1\n"                                                         \
      "v_min_f32_dpp %0, %0, %0 " R " row_mask:0xf bank_mask:0xf\n"       \
      "v_min_f32_dpp %1, %1, %1 " R " row_mask:0xf bank_mask:0xf\n"       \
      "v_min_f32_dpp %2, %2, %2 " R " row_mask:0xf bank_mask:0xf\n"       \
      "v_max_f32_dpp %3, %3, %3 " R " row_mask:0xf bank_mask:0xf\n"       \
      "v_max_f32_dpp %4, %4, %4 " R " row_mask:0xf bank_mask:0xf\n"       \
      "v_max_f32_dpp %5, %5, %5 " R " row_mask:0xf bank_mask:0xf\n"       \
      : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(h0), "+v"(h1), "+v"(h2))

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Sk2Epi {      // registers of the epilogue that runs in the shadow of the next unit's MFMAs
  f32x4 R[3];        // skinning-matrix rows of the joint being blended
  f32x2 T[6];        // blended 3x4 transform of the frame in flight, row r = (T[2r], T[2r+1])
  float4 tr;         // translation of that frame
  float o[3];        // skinned vertex
  float bx[6];       // box candidates (min xyz, max xyz)
  // per-unit addresses (formed once when the unit's skin weights arrive): LDS byte address of the lane's four
  // joints' matrices for frame 4*kq, of its translation, and the byte offsets of its first frame's vertex / box
  unsigned wa[4], tra, vo, bo;
};

// One twelfth of one epilogue piece.  Piece = (frame register e = piece / 3, part = piece % 3): parts 0 and 1 blend the
// skinning matrices of joints {0,1} / {2,3}, part 2 applies the transform, stores the vertex and the unit's box.
// `k` (0..11) is the MFMA of the group after which the slice is issued.  FP32 MFMAs and VALU instructions share the
// SIMD's FP32 datapath on gfx950 (they do not overlap, measured), so the slices are written for the fewest VALU
// instructions: LDS addresses are per-unit bases plus immediate offsets, matrix rows are blended as register pairs
// (v_pk_fma_f32 on adjacent registers), frames past F fall outside the buffers' ranges by construction and padding
// vertices duplicate the last real one (model.hip), so neither needs a select.
// VPOUT: also store v_posed (the blend, before skinning) in the vertices' layout -- the dense backward needs it (dense_bwd.hip)
// SCALAR_APPLY (k_skin3): the 3x4 transform is applied with scalar FMAs on opaque copies of its entries instead of whatever
// the compiler packs.  Same fma chain, same bits.  The unexplained failure of k_skin3's two-waves-per-SIMD version (see k_skin3)
// was traced to the compiler's packed form of this expression for frame register 1 -- v_mov into one half of a register pair,
// v_pk_fma reading the pair a few instructions later -- going wrong in lanes 48-63 when another wave's MFMAs share the SIMD:
// with this form, 0 values off in 300 launches even at two waves per SIMD (2 363 in 100 without).
template <bool BBOX, bool VPOUT = false, bool SCALAR_APPLY = false>
__device__ __forceinline__ void sk2_slice(const int piece, const int k, Sk2Epi& E, const float4& ww, const f32x4& p0,
                                          const f32x4& p1, const f32x4& p2, const char* sTb, const char* sTrb,
                                          const unsigned v12, const unsigned n24, __amdgpu_buffer_rsrc_t rv,
                                          __amdgpu_buffer_rsrc_t rb, __amdgpu_buffer_rsrc_t rvp) {
  const int e = piece / 3, part = piece - 3 * e;
  if (part < 2) {
    if (k == 0 || k == 5) {
      const int n = 2 * part + (k == 5 ? 1 : 0);
      const f32x4* pt = reinterpret_cast<const f32x4*>(sTb + E.wa[n] + e * (UUO_NUM_JOINTS * 48));
      E.R[0] = pt[0];
      E.R[1] = pt[1];
      E.R[2] = pt[2];
    } else if ((k >= 2 && k <= 4) || (k >= 7 && k <= 9)) {
      const int h = (k >= 7) ? 1 : 0, row = (k >= 7) ? k - 7 : k - 2;  // joint of the pair, matrix row
      const int n = 2 * part + h;                        // joint: weight = component n of ww
      // v_pk_fma_f32 on register pairs: the weight is one half of an aligned pair of the float4 it was loaded into
      // and is broadcast to both result lanes by op_sel / op_sel_hi; the matrix row halves are pairs of the ds_read
      const f32x2 wp = (n < 2) ? f32x2{ww.x, ww.y} : f32x2{ww.z, ww.w};
      const f32x4 r = E.R[row];
      const f32x2 rlo = __builtin_shufflevector(r, r, 0, 1), rhi = __builtin_shufflevector(r, r, 2, 3);
      if (part == 0 && h == 0) {
        if (n & 1) {
          asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(E.T[2 * row]) : "v"(wp), "v"(rlo));
          asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(E.T[2 * row + 1]) : "v"(wp), "v"(rhi));
        } else {
          asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(E.T[2 * row]) : "v"(wp), "v"(rlo));
          asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(E.T[2 * row + 1]) : "v"(wp), "v"(rhi));
        }
      } else {
        if (n & 1) {
          asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(E.T[2 * row]) : "v"(wp), "v"(rlo));
          asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(E.T[2 * row + 1]) : "v"(wp), "v"(rhi));
        } else {
          asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(E.T[2 * row]) : "v"(wp), "v"(rlo));
          asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(E.T[2 * row + 1]) : "v"(wp), "v"(rhi));
        }
      }
    }
  } else {
    if (k == 0) {
      E.tr = *reinterpret_cast<const float4*>(sTrb + E.tra + e * 16);
    } else if (k >= 1 && k <= 3) {
      const int c = k - 1;
      const float px = p0[e], py = p1[e], pz = p2[e];
      if constexpr (SCALAR_APPLY) {
        float t0 = E.T[2 * c][0], t1 = E.T[2 * c][1], t2 = E.T[2 * c + 1][0], t3 = E.T[2 * c + 1][1];
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
        float oc = fmaf(t2, pz, fmaf(t1, py, t0 * px)) + t3;
        asm volatile("" : "+v"(oc));
        E.o[c] = oc;
      } else {
        E.o[c] = fmaf(E.T[2 * c + 1][0], pz, fmaf(E.T[2 * c][1], py, E.T[2 * c][0] * px)) + E.T[2 * c + 1][1];
      }
    } else if (k == 4) {
      E.o[0] += E.tr.x; E.o[1] += E.tr.y; E.o[2] += E.tr.z;
      u32x3 o = {__builtin_bit_cast(unsigned, E.o[0]), __builtin_bit_cast(unsigned, E.o[1]),
                 __builtin_bit_cast(unsigned, E.o[2])};
      __builtin_amdgcn_raw_buffer_store_b96(o, rv, E.vo + e * v12, 0, SK2_ST_AUX);
      if constexpr (VPOUT) {  // v_posed of the same (frame, vertex), same offset in its own buffer
        // (through float temporaries: __builtin_bit_cast applied to the vector ELEMENT p0[e] itself reads element 0 whatever e is)
        const float vx = p0[e], vy = p1[e], vz = p2[e];
        u32x3 pv = {__builtin_bit_cast(unsigned, vx), __builtin_bit_cast(unsigned, vy), __builtin_bit_cast(unsigned, vz)};
        __builtin_amdgcn_raw_buffer_store_b96(pv, rvp, E.vo + e * v12, 0, 0);
      }
    } else if (BBOX && k == 5) {
#pragma unroll
      for (int c = 0; c < 3; ++c) E.bx[c] = E.bx[3 + c] = E.o[c];
    } else if (BBOX && k == 6) {
      SK2_BOX_STEP("row_ror:8", E.bx[0], E.bx[1], E.bx[2], E.bx[3], E.bx[4], E.bx[5]);
    } else if (BBOX && k == 7) {
      SK2_BOX_STEP("row_ror:4", E.bx[0], E.bx[1], E.bx[2], E.bx[3], E.bx[4], E.bx[5]);
    } else if (BBOX && k == 8) {
      SK2_BOX_STEP("row_ror:2", E.bx[0], E.bx[1], E.bx[2], E.bx[3], E.bx[4], E.bx[5]);
    } else if (BBOX && k == 9) {
      SK2_BOX_STEP("row_ror:1", E.bx[0], E.bx[1], E.bx[2], E.bx[3], E.bx[4], E.bx[5]);
    } else if (BBOX && k == 10) {
      u32x3 lo = {__builtin_bit_cast(unsigned, E.bx[0]), __builtin_bit_cast(unsigned, E.bx[1]),
                  __builtin_bit_cast(unsigned, E.bx[2])};
      u32x3 hi = {__builtin_bit_cast(unsigned, E.bx[3]), __builtin_bit_cast(unsigned, E.bx[4]),
                  __builtin_bit_cast(unsigned, E.bx[5])};
      const unsigned boff = E.bo + e * n24;
      __builtin_amdgcn_raw_buffer_store_b96(lo, rb, boff, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b96(hi, rb, boff + 12u, 0, 0);
    }
  }
}

// per-unit addresses of the epilogue (see Sk2Epi): 24-bit multiplies (full rate), everything else folds into immediates
__device__ __forceinline__ void sk2_unit_addresses(Sk2Epi& E, const Sk2Unit& P, const int4& wi, const int kq, const int j,
                                                   const int V, const unsigned v12, const unsigned n24) {
  const unsigned rowbase = (unsigned)(P.slot * UUO_FT + 4 * kq) * (UUO_NUM_JOINTS * 48);
  E.wa[0] = __umul24((unsigned)wi.x, 48u) + rowbase;
  E.wa[1] = __umul24((unsigned)wi.y, 48u) + rowbase;
  E.wa[2] = __umul24((unsigned)wi.z, 48u) + rowbase;
  E.wa[3] = __umul24((unsigned)wi.w, 48u) + rowbase;
  E.tra = (unsigned)(P.slot * UUO_FT + 4 * kq) * 16u;
  const unsigned f0 = (unsigned)(P.i0 + 4 * kq);
  const int v = P.u * 16 + j;
  E.vo = (__umul24(f0, v12) + (unsigned)v * 12u) | (v < V ? 0u : 0x80000000u);
  E.bo = (__umul24(f0, n24) + (unsigned)P.u * 24u) | (j == 0 ? 0u : 0x80000000u);
}

#ifdef UUO_DEBUG_HOOKS
__device__ unsigned long long g_sk2_stamps[2048 * 16];  // debug flavour (UUO_SK2_VAR=9): per-wave shader-clock stamps
#else
__device__ unsigned long long g_sk2_stamps[16];  // product: only the VAR = 0 instantiations exist, nothing ever indexes this
#endif

template <bool BBOX, int VAR, bool VPOUT = false>
__global__ __launch_bounds__(SKIN_WAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_skin2(const float4* __restrict__ P3v, const float* __restrict__ vt3,
                                                            const int* __restrict__ Wi, const float* __restrict__ Ww,
                                                            const float* __restrict__ pfaT, const float* __restrict__ A,
                                                            const float* __restrict__ trans, float* __restrict__ verts,
                                                            float* __restrict__ bbox, int F, int V, int VP, int nFT,
                                                            float* __restrict__ vp_out) {
  __shared__ float4 sA[2 * UUO_KP * UUO_FT / 4];               // [slot][14][64]
  __shared__ f32x4 sT[2 * UUO_FT * UUO_NUM_JOINTS * 3];        // [slot][i][j][3]
  __shared__ float4 sTr[2 * UUO_FT];                            // [slot][i] translation
  __shared__ int sQ[64];                                        // sQ[0] = next unclaimed task of the block
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (UUO_SK2_PRIO) __builtin_amdgcn_s_setprio(UUO_SK2_PRIO);
  const unsigned long long stamp0 = (VAR & 8) ? __builtin_readcyclecounter() : 0ull;
  const int j = lane & 15, kq = lane >> 4;
  const int nunits = VP / 16;      // units of the blend-basis table (padded)
  const int nur = (V + 15) / 16;   // units that hold vertices: the box table's stride, the task space
  const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;  // grid = 8 XCDs x SK2_NPOS blocks
  // Task space: slice x = units [x*nur/8, (x+1)*nur/8), tasks of a slice ordered (frame tile, unit), slices back to
  // back.  XCD x (= blockIdx & 7) takes the x-th eighth of that list -- its own slice give or take a few tasks of
  // the neighbour, so every XCD's L2 holds one slice of the basis and no XCD has more than ceil(total / 8) tasks
  // (F = 300: 8 189 tasks, 1 023 or 1 024 per XCD, 31 or 32 per block, exactly 8 per SIMD).  The block's waves claim
  // its tasks one at a time from a counter in LDS: the two waves of a SIMD do not share it evenly (the older wave
  // wins the issue slot), a static split would leave the younger one with most of its units when its partner retires.
  const int ntot = nFT * nur;
  const int xlo = (int)(((unsigned)ntot * (unsigned)xcd) >> 3), xhi = (int)(((unsigned)ntot * (unsigned)(xcd + 1)) >> 3);
  const int tb0 = xlo + (((xhi - xlo) * pos) >> SK2_NPOS_LOG2), tb1 = xlo + (((xhi - xlo) * (pos + 1)) >> SK2_NPOS_LOG2);
  if (tb1 <= tb0) return;  // block-uniform
  // The block's tasks lie in at most two of three segments (checked on the host, sk2_fits): (slice a, tile ftA),
  // (slice a, tile ftA + 1), (slice a + 1, tile 0).  Segment 1 uses LDS slot 0, the other one slot 1.
  int xa = 0;
  while (xa < 7 && nFT * (((xa + 1) * nur) >> 3) <= tb0) ++xa;
  const int ua0 = (xa * nur) >> 3, ub0 = ((xa + 1) * nur) >> 3;
  const int nua = ub0 - ua0;
  const int Sa = nFT * ua0, Sb = nFT * ub0;          // first task of slice a / slice a + 1
  const int ftA = (tb0 - Sa) / nua;
  const int g1 = Sa + ftA * nua;                     // first task of (slice a, ftA)
  const int g2 = (g1 + nua < Sb) ? g1 + nua : Sb;    // end of segment 1
  const int ftB = (tb1 - 1 < Sb) ? ftA + 1 : 0;      // frame tile of slot 1
  const int nslots = (tb1 > g2) ? 2 : 1;

  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(verts, 0, F * V * 12, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(bbox, 0, BBOX ? F * nur * 24 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvp = VPOUT ? __builtin_amdgcn_make_buffer_rsrc(vp_out, 0, F * V * 12, 0x00020000) : rv;
  // the blend basis, template and skin weights are addressed through buffer resources: one address VGPR each,
  // everything else scalar
  const unsigned cplane = (unsigned)nunits * SKIN_GROUPS * 1024u;  // bytes per coordinate plane
  const __amdgpu_buffer_rsrc_t rp =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(P3v), 0, (int)(3u * cplane), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwi = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(Wi), 0, VP * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rww = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ww), 0, VP * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vt3), 0, VP * 12, 0x00020000);
  const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(g_sk2_stamps, 0, (VAR & 8) ? 2048 * 16 * 8 : 0, 0x00020000);
  const unsigned lane16 = lane * 16, j16 = j * 16, j4 = j * 4;
#define SK2_LDB(c, ubase, g) \
  __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, lane16, (ubase) + (c)*cplane + (g)*1024u, 0))
#define SK2_DECODE(U, t_)                                                        \
  {                                                                              \
    const bool s1_ = (t_) < g2;                                                  \
    const int base_ = s1_ ? g1 : ((t_) < Sb ? g2 : Sb);                          \
    (U).u = (((t_) < Sb) ? ua0 : ub0) + (t_)-base_;                              \
    (U).i0 = (s1_ ? ftA : ftB) * UUO_FT;                                         \
    (U).slot = s1_ ? 0 : 1;                                                      \
  }

  const unsigned long long stampS = (VAR & 8) ? __builtin_readcyclecounter() : 0ull;
  // stage the block's (one or two) frame tiles; every load is issued before the first LDS store (one round trip),
  // and before the wave's B stream so that the tiles are not queued behind 21 KB of basis rows per wave
  constexpr int NA = UUO_KP * UUO_FT / 4, NT = UUO_FT * UUO_NUM_JOINTS * 3;  // float4 per slot: 896 + 1152
  constexpr int PER = (2 * (NA + NT) + SKIN_WAVES * 64 - 1) / (SKIN_WAVES * 64);  // 8 per thread
  float4 tmp[PER];
  float4 trv = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const float4* gA0 = reinterpret_cast<const float4*>(pfaT + (size_t)ftA * UUO_KP * UUO_FT);
    const float4* gA1 = reinterpret_cast<const float4*>(pfaT + (size_t)ftB * UUO_KP * UUO_FT);
    const float4* gT0 = reinterpret_cast<const float4*>(A + (size_t)ftA * UUO_FT * UUO_NUM_JOINTS * 12);
    const float4* gT1 = reinterpret_cast<const float4*>(A + (size_t)ftB * UUO_FT * UUO_NUM_JOINTS * 12);
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = tid + r * SKIN_WAVES * 64;  // [0, 2 NA): A tiles of slot 0, 1; then the skinning matrices
      tmp[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NA) tmp[r] = gA0[i];
      else if (i < 2 * NA) { if (nslots > 1) tmp[r] = gA1[i - NA]; }
      else if (i < 2 * NA + NT) tmp[r] = gT0[i - 2 * NA];
      else if (i < 2 * NA + 2 * NT) { if (nslots > 1) tmp[r] = gT1[i - 2 * NA - NT]; }
    }
    if (tid < 2 * UUO_FT) {
      const int f = ((tid < UUO_FT) ? ftA : ftB) * UUO_FT + (tid & (UUO_FT - 1));
      if (trans && f < F && (tid < UUO_FT || nslots > 1))
        trv = make_float4(trans[(size_t)f * 3], trans[(size_t)f * 3 + 1], trans[(size_t)f * 3 + 2], 0.f);
    }
  }

  // first task of the wave: its B stream (L2 / HBM latency) is started before the LDS tiles are written
  int t_cur = tb0 + wave;
  const bool active = t_cur < tb1;  // wave-uniform
  if (!active) t_cur = tb0;
  Sk2Unit cur, prv, nxt;
  SK2_DECODE(cur, t_cur);
  prv.u = cur.u; prv.i0 = F; prv.slot = 0;  // nothing to store for the first unit's "previous" epilogue
  nxt = cur;
  unsigned pb = (unsigned)cur.u * (SKIN_GROUPS * 1024u);
  float tn0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, cur.u * 64, 0));
  float tn1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, VP * 4 + cur.u * 64, 0));
  float tn2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, VP * 8 + cur.u * 64, 0));
  f32x4 rbuf[3][SK2_RING];
#pragma unroll
  for (int s = 0; s < SK2_RING; ++s) {
    rbuf[0][s] = SK2_LDB(0, pb, s);
    rbuf[1][s] = SK2_LDB(1, pb, s);
    rbuf[2][s] = SK2_LDB(2, pb, s);
  }

  if (tid < 64) sQ[tid] = (tid == 0) ? tb0 + SKIN_WAVES : 0;
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    const int i = tid + r * SKIN_WAVES * 64;
    if (i < 2 * NA) sA[i] = tmp[r];
    else if (i < 2 * (NA + NT)) sT[i - 2 * NA] = __builtin_bit_cast(f32x4, tmp[r]);
  }
  if (tid < 2 * UUO_FT) sTr[tid] = trv;
  __syncthreads();
  const unsigned long long stamp1 = (VAR & 8) ? __builtin_readcyclecounter() : 0ull;
  if (!active) return;  // wave-uniform

  int4 wi = make_int4(0, 0, 0, 0);                // skin weights of the unit whose epilogue is running
  float4 ww = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0;  // accumulators of the previous unit
  Sk2Epi E;
  const char* sTb = reinterpret_cast<const char*>(sT);
  const char* sTrb = reinterpret_cast<const char*>(sTr);
  const unsigned v12 = (unsigned)V * 12u, n24 = (unsigned)nur * 24u;  // bytes per frame of vertices / boxes
  sk2_unit_addresses(E, prv, wi, kq, j, V, v12, n24);
  int nunits_done = 0;
  bool more = true;

  while (more) {
    // claim the next task: lane 0 adds 1 to the block's counter, the other lanes add 0 to private words (no exec
    // juggling, no bank conflict); the result is consumed a few groups later
    const int claimed = __hip_atomic_fetch_add(&sQ[lane], lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned pbn = pb;
    int t_nxt = t_cur;
    const float4* pa = sA + cur.slot * (UUO_KP * UUO_FT / 4) + lane;
    f32x4 acc0 = {tn0, tn0, tn0, tn0}, acc1 = {tn1, tn1, tn1, tn1}, acc2 = {tn2, tn2, tn2, tn2};
    float4 ra[2];
    ra[0] = pa[0];
    // The unit is written as 14 groups x 12 (MFMA, slice) pairs with a scheduling fence after every pair, so the
    // previous unit's epilogue is spread evenly through the MFMA stream (loads and LDS reads get their latency
    // covered, the refills stay 7 groups ahead).
#pragma unroll
    for (int g = 0; g < SKIN_GROUPS; ++g) {
      const int st = g % SK2_RING;
      const float4 av = ra[g & 1];
      const int part = g % 3;
      const int k_aread = (g < 12 && part == 2) ? 11 : 10;
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int ks = k / 3, c = k - 3 * ks;
        const float a = (ks == 0) ? av.x : (ks == 1) ? av.y : (ks == 2) ? av.z : av.w;
        if (VAR & 4) {
          asm volatile("" ::"v"(a), "v"(rbuf[c][st][ks]));
        } else if (c == 0) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rbuf[0][st][ks], acc0, 0, 0, 0);
        } else if (c == 1) {
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rbuf[1][st][ks], acc1, 0, 0, 0);
        } else {
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rbuf[2][st][ks], acc2, 0, 0, 0);
        }
        if (!(VAR & 1) && g < 12) sk2_slice<BBOX, VPOUT>(g, k, E, ww, q0, q1, q2, sTb, sTrb, v12, n24, rv, rb, rvp);
        if (k == k_aread && g + 1 < SKIN_GROUPS) ra[(g + 1) & 1] = pa[(g + 1) * 64];
        if (g == 2 && k == 11) {  // the claim has landed: decode the next task (or keep re-reading this one)
          const int t_n = __builtin_amdgcn_readfirstlane(claimed);
          more = t_n < tb1;
          t_nxt = more ? t_n : t_cur;
          SK2_DECODE(nxt, t_nxt);
          pbn = (unsigned)nxt.u * (SKIN_GROUPS * 1024u);
        }
        // Loads that the next unit needs at its very first group: this unit's skin weights (its epilogue runs during
        // the next unit; the previous epilogue read the old ones for the last time in piece 10) and the next
        // template values.  vmcnt retires in order, so waiting for them at the unit boundary also waits for the ring
        // refills issued before them -- those are the next unit's first groups and are due then anyway.
        if (g == 11 && k == 10) {
          wi = __builtin_bit_cast(int4, __builtin_amdgcn_raw_buffer_load_b128(rwi, j16, cur.u * 256, 0));
          ww = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rww, j16, cur.u * 256, 0));
          tn0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, nxt.u * 64, 0));
          tn1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, VP * 4 + nxt.u * 64, 0));
          tn2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, VP * 8 + nxt.u * 64, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(VAR & 2)) {  // refill the ring stage this group has just consumed: 7 groups (~2 700 cycles) of lead
        const unsigned ub = (g + SK2_RING < SKIN_GROUPS) ? pb : pbn;
        const int gg = (g + SK2_RING) % SKIN_GROUPS;
        rbuf[0][st] = SK2_LDB(0, ub, gg);
        rbuf[1][st] = SK2_LDB(1, ub, gg);
        rbuf[2][st] = SK2_LDB(2, ub, gg);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (VAR & 8) {  // stamp of this unit's end, written straight to memory by lane 0 (other lanes out of range)
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const unsigned long long c = __builtin_readcyclecounter();
      const u32x2 cv = {(unsigned)c, (unsigned)(c >> 32)};
      const unsigned so = (unsigned)(((blockIdx.x * SKIN_WAVES + wave) * 16 + 2 + (nunits_done < 8 ? nunits_done : 7)) * 8);
      __builtin_amdgcn_raw_buffer_store_b64(cv, rst, so | (lane == 0 ? 0u : 0x80000000u), 0, 0);
    }
    ++nunits_done;
    q0 = acc0; q1 = acc1; q2 = acc2;
    prv = cur; cur = nxt; pb = pbn; t_cur = t_nxt;
    sk2_unit_addresses(E, prv, wi, kq, j, V, v12, n24);
  }
#undef SK2_LDB
#undef SK2_DECODE
  // epilogue of the wave's last unit
#pragma unroll
  for (int g = 0; g < 12; ++g) {
#pragma unroll
    for (int k = 0; k < 12; ++k) sk2_slice<BBOX, VPOUT>(g, k, E, ww, q0, q1, q2, sTb, sTrb, v12, n24, rv, rb, rvp);
  }
  if ((VAR & 8) && lane == 0) {
    unsigned long long* o = g_sk2_stamps + (size_t)(blockIdx.x * SKIN_WAVES + wave) * 16;
    o[0] = stamp0; o[1] = stamp1;
    o[10] = __builtin_readcyclecounter();
    o[11] = (unsigned long long)nunits_done;
    o[12] = stampS;
  }
}

// ----------------------------------------------------------------------------------------------------
// K_B3  skin on the fp16 matrix pipe: the blend of a closure's SEARCH (chamfer stage: the vertices feed the nearest-vertex search
// only -- loss and gradient are formed in fp32 on the re-skinned winners by the backward kernel).
// v_mfma_f32_16x16x32_f16 runs at 16 x the FLOP rate of v_mfma_f32_16x16x4_f32.  Both operands are split into two fp16 planes of a
// power-of-two multiple, x = hi + lo + O(2^-22 |x|) (hi = fp16(x), lo = fp16(x - hi)), and three of the four products are kept:
//   sum a b  ~=  sum a_hi b_hi  +  (sum a_hi b_lo + sum a_lo b_hi)          (fp32 accumulation; a_lo b_lo < 2^-22 |a b| is dropped)
// The large and the small sums have their own accumulators, and the template is added LAST (k_skin2 accumulates onto it, rounding
// at the template's magnitude 56 times), so the blend is at least as close to the exact one as the fp32 pipe's
// (test_skin16_is_as_close_to_float64_as_the_fp32_kernel).  63 MFMAs of 16 cycles per task instead of 168 of 32.
// Everything else -- tasks, claiming, LDS tiles, the 7-stage operand ring (same bytes: two fp16 planes = one fp32), the epilogue
// slices of the previous unit between the MFMAs -- is k_skin2's.
// ----------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define SK3_STEPS (SKIN_GROUPS / 2)        // 7 K-steps of 32
#define SK3_NMFMA (SK3_STEPS * 9)          // 63 MFMA slots per task
#define SK3_NSLICE 144                     // 12 pieces x 12 slices of the previous unit's epilogue

// ONE wave per SIMD, and no other matrix-pipe wave beside it.  With two waves of this kernel per SIMD (an 8-wave block) the x
// coordinate of one frame of a unit -- always frame 13 of a tile -- came out wrong once or twice per launch (y and z exact; only
// with the MFMAs in, whatever the register allocation, the order of the epilogue pieces or the padding after an MFMA; never with
// four waves: 0 of 2.5e8 values in 40 launches).  It was traced to the compiler's packed form of the 3x4 apply (sk2_slice,
// SCALAR_APPLY), which k_skin3 no longer uses -- that took the failure from about one unit per launch to ONE unit in 60 000
// launches of fits in flight when two k_skin3 blocks may share a CU (no LDS pad), so it is not the whole story, and the
// conditions stay avoided by construction: four waves per block, and SK3_LDS_PAD bytes of dynamic LDS on top of the 66 304
// static ones, so that no block of a kernel that issues MFMAs fits beside it on a CU (160 KB): k_skin2 / k_skin3 66 304 B and
// k_dpf 57 344 by LDS, k_skin (32 768 B, 2 x 170 registers per SIMD) by registers.  The other chains' search, backward, L-BFGS
// passes, k_finalize and k_pose_prep (<= 43 KB) still fit -- with a pad that kept k_bwd_sparse out the fit in flight was 2.4 %
// slower.  With this pad: 184 009 launches of fits in flight checked against the fp32 kernel, 0 values off
// (tools/skin16_stress.py, debug flavour).
#ifndef SK3_WAVES
#define SK3_WAVES 4
#endif
#ifndef SK3_LDS_PAD
#define SK3_LDS_PAD 44288  // 66 304 + 44 288 = 110 592; + 57 344 (k_dpf) > 163 840
#endif
template <bool BBOX>
__global__ __launch_bounds__(SK3_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_skin3(
    const float4* __restrict__ P16v, const float* __restrict__ vt3, const int* __restrict__ Wi, const float* __restrict__ Ww,
    const float4* __restrict__ pfa16, const float* __restrict__ A, const float* __restrict__ trans, float* __restrict__ verts,
    float* __restrict__ bbox, int F, int V, int VP, int nFT, float inv_scale) {
  __shared__ float4 sA[2 * UUO_KP * UUO_FT / 4];               // [slot][7][2][64] x 16 B (8 halfs)
  __shared__ f32x4 sT[2 * UUO_FT * UUO_NUM_JOINTS * 3];        // [slot][i][j][3]
  __shared__ float4 sTr[2 * UUO_FT];                            // [slot][i] translation
  __shared__ int sQ[64];                                        // sQ[0] = next unclaimed task of the block
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int nunits = VP / 16;
  const int nur = (V + 15) / 16;
  const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
  // (task space, segments and LDS slots: see k_skin2)
  const int ntot = nFT * nur;
  const int xlo = (int)(((unsigned)ntot * (unsigned)xcd) >> 3), xhi = (int)(((unsigned)ntot * (unsigned)(xcd + 1)) >> 3);
  const int tb0 = xlo + (((xhi - xlo) * pos) >> SK2_NPOS_LOG2), tb1 = xlo + (((xhi - xlo) * (pos + 1)) >> SK2_NPOS_LOG2);
  if (tb1 <= tb0) return;  // block-uniform
  int xa = 0;
  while (xa < 7 && nFT * (((xa + 1) * nur) >> 3) <= tb0) ++xa;
  const int ua0 = (xa * nur) >> 3, ub0 = ((xa + 1) * nur) >> 3;
  const int nua = ub0 - ua0;
  const int Sa = nFT * ua0, Sb = nFT * ub0;
  const int ftA = (tb0 - Sa) / nua;
  const int g1 = Sa + ftA * nua;
  const int g2 = (g1 + nua < Sb) ? g1 + nua : Sb;
  const int ftB = (tb1 - 1 < Sb) ? ftA + 1 : 0;
  const int nslots = (tb1 > g2) ? 2 : 1;

  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(verts, 0, F * V * 12, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(bbox, 0, BBOX ? F * nur * 24 : 0, 0x00020000);
  const unsigned cplane = (unsigned)nunits * SKIN_GROUPS * 1024u;  // bytes per coordinate plane (as the fp32 table)
  const __amdgpu_buffer_rsrc_t rp =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(P16v), 0, (int)(3u * cplane), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwi = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(Wi), 0, VP * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rww = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ww), 0, VP * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vt3), 0, VP * 12, 0x00020000);
  const unsigned lane16 = lane * 16, j16 = j * 16, j4 = j * 4;
  // block g of a unit's 14 1-KB blocks per coordinate: (K-step g >> 1, plane g & 1)
#define SK3_LDB(c, ubase, g) \
  __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, lane16, (ubase) + (c)*cplane + (g)*1024u, 0))
#define SK3_DECODE(U, t_)                                                        \
  {                                                                              \
    const bool s1_ = (t_) < g2;                                                  \
    const int base_ = s1_ ? g1 : ((t_) < Sb ? g2 : Sb);                          \
    (U).u = (((t_) < Sb) ? ua0 : ub0) + (t_)-base_;                              \
    (U).i0 = (s1_ ? ftA : ftB) * UUO_FT;                                         \
    (U).slot = s1_ ? 0 : 1;                                                      \
  }

  constexpr int NA = UUO_KP * UUO_FT / 4, NT = UUO_FT * UUO_NUM_JOINTS * 3;  // float4 per slot: 896 + 1152
  constexpr int PER = (2 * (NA + NT) + SK3_WAVES * 64 - 1) / (SK3_WAVES * 64);
  float4 tmp[PER];
  float4 trv = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const float4* gA0 = pfa16 + (size_t)ftA * NA;
    const float4* gA1 = pfa16 + (size_t)ftB * NA;
    const float4* gT0 = reinterpret_cast<const float4*>(A + (size_t)ftA * UUO_FT * UUO_NUM_JOINTS * 12);
    const float4* gT1 = reinterpret_cast<const float4*>(A + (size_t)ftB * UUO_FT * UUO_NUM_JOINTS * 12);
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = tid + r * SK3_WAVES * 64;
      tmp[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NA) tmp[r] = gA0[i];
      else if (i < 2 * NA) { if (nslots > 1) tmp[r] = gA1[i - NA]; }
      else if (i < 2 * NA + NT) tmp[r] = gT0[i - 2 * NA];
      else if (i < 2 * NA + 2 * NT) { if (nslots > 1) tmp[r] = gT1[i - 2 * NA - NT]; }
    }
    if (tid < 2 * UUO_FT) {
      const int f = ((tid < UUO_FT) ? ftA : ftB) * UUO_FT + (tid & (UUO_FT - 1));
      if (trans && f < F && (tid < UUO_FT || nslots > 1))
        trv = make_float4(trans[(size_t)f * 3], trans[(size_t)f * 3 + 1], trans[(size_t)f * 3 + 2], 0.f);
    }
  }

  int t_cur = tb0 + wave;
  const bool active = t_cur < tb1;  // wave-uniform
  if (!active) t_cur = tb0;
  Sk2Unit cur, prv, nxt;
  SK3_DECODE(cur, t_cur);
  prv.u = cur.u; prv.i0 = F; prv.slot = 0;  // nothing to store for the first unit's "previous" epilogue
  nxt = cur;
  unsigned pb = (unsigned)cur.u * (SKIN_GROUPS * 1024u);
  f32x4 rbuf[3][SK2_RING];
#pragma unroll
  for (int s = 0; s < SK2_RING; ++s) {
    rbuf[0][s] = SK3_LDB(0, pb, s);
    rbuf[1][s] = SK3_LDB(1, pb, s);
    rbuf[2][s] = SK3_LDB(2, pb, s);
  }

  if (tid < 64) sQ[tid] = (tid == 0) ? tb0 + SK3_WAVES : 0;
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    const int i = tid + r * SK3_WAVES * 64;
    if (i < 2 * NA) sA[i] = tmp[r];
    else if (i < 2 * (NA + NT)) sT[i - 2 * NA] = __builtin_bit_cast(f32x4, tmp[r]);
  }
  if (tid < 2 * UUO_FT) sTr[tid] = trv;
  __syncthreads();
  if (!active) return;  // wave-uniform

  int4 wi = make_int4(0, 0, 0, 0);                // skin weights of the unit whose epilogue is running
  float4 ww = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0;  // blend of the previous unit
  Sk2Epi E;
  const char* sTb = reinterpret_cast<const char*>(sT);
  const char* sTrb = reinterpret_cast<const char*>(sTr);
  const unsigned v12 = (unsigned)V * 12u, n24 = (unsigned)nur * 24u;
  sk2_unit_addresses(E, prv, wi, kq, j, V, v12, n24);
  bool more = true;

  while (more) {
    const int claimed = __hip_atomic_fetch_add(&sQ[lane], lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned pbn = pb;
    int t_nxt = t_cur;
    // this unit's template and skin weights: asked for now, consumed at the unit's end (template) / by its epilogue, which runs
    // during the next unit (weights: the running epilogue still reads the previous unit's)
    const float tn0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, cur.u * 64, 0));
    const float tn1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, VP * 4 + cur.u * 64, 0));
    const float tn2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvt, j4, VP * 8 + cur.u * 64, 0));
    const int4 wi_n = __builtin_bit_cast(int4, __builtin_amdgcn_raw_buffer_load_b128(rwi, j16, cur.u * 256, 0));
    const float4 ww_n = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rww, j16, cur.u * 256, 0));
    const float4* pa = sA + cur.slot * (UUO_KP * UUO_FT / 4) + lane;
    f32x4 am0 = {0.f, 0.f, 0.f, 0.f}, am1 = am0, am2 = am0;  // sum a_hi b_hi
    f32x4 as0 = am0, as1 = am0, as2 = am0;                    // sum a_hi b_lo + a_lo b_hi
    float4 ah[2], al[2];
    ah[0] = pa[0];
    al[0] = pa[64];
    // The MFMAs accumulate IN PLACE through inline assembly ("+v": destination = accumulator input, the same six register quads
    // for the whole unit; with the compiler's intrinsic the allocator lets an accumulator hop into whatever quad is free, e.g. the
    // data registers of a vertex store issued a few instructions earlier).  What the compiler no longer sees is kept safe by
    // construction: the nine MFMAs of a step go large sums first, then the two small terms, so an accumulator is never touched by
    // two of three consecutive MFMAs; its first MFMA comes long after the VALU zero fill; the blend is read twelve wait states
    // after the last MFMA (below).
#ifdef SK3_NO_MFMA  // (experiment: the kernel without its matrix instructions -- the failure described above needs them)
#define SK3_MFMA(acc, a_, b_) asm volatile("" : "+v"(acc) : "v"(a_), "v"(b_))
#else
#define SK3_MFMA(acc, a_, b_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a_), "v"(b_))
#endif
    // one K-step of 32 (literal step number: the slice schedule and the ring indices are constants of each copy):
    // 9 MFMAs, behind each of them 2 or 3 of the 144 slices of the previous unit's epilogue; then the refill of the two ring
    // stages the step has consumed -- 7 blocks (3.5 steps) of lead, across the unit boundary
#define SK3_STEP(ST)                                                         \
    {                                                                        \
      constexpr int st = ST;                                                 \
      constexpr int rh = (2 * st) % SK2_RING, rl = (2 * st + 1) % SK2_RING; \
      const f16x8 a_hi = __builtin_bit_cast(f16x8, ah[st & 1]), a_lo = __builtin_bit_cast(f16x8, al[st & 1]); \
_Pragma("unroll") \
      for (int q = 0; q < 9; ++q) { \
        const int term = q / 3, c = q - 3 * term; \
        const f16x8 b_hi = __builtin_bit_cast(f16x8, rbuf[c][rh]), b_lo = __builtin_bit_cast(f16x8, rbuf[c][rl]); \
        const f16x8 ax = (term == 2) ? a_lo : a_hi, bx = (term == 1) ? b_lo : b_hi; \
        if (term == 0) { \
          if (c == 0) SK3_MFMA(am0, ax, bx); \
          else if (c == 1) SK3_MFMA(am1, ax, bx); \
          else SK3_MFMA(am2, ax, bx); \
        } else { \
          if (c == 0) SK3_MFMA(as0, ax, bx); \
          else if (c == 1) SK3_MFMA(as1, ax, bx); \
          else SK3_MFMA(as2, ax, bx); \
        } \
        const int mslot = st * 9 + q; \
        const int n0 = (mslot * SK3_NSLICE) / SK3_NMFMA, n1 = ((mslot + 1) * SK3_NSLICE) / SK3_NMFMA; \
_Pragma("unroll") \
        for (int i = 0; i < 3; ++i) { \
          const int n = n0 + i; \
          if (n < n1) sk2_slice<BBOX, false, true>(n / 12, n % 12, E, ww, q0, q1, q2, sTb, sTrb, v12, n24, rv, rb, rv); \
        } \
        if (q == 5 && st + 1 < SK3_STEPS) { \
          ah[(st + 1) & 1] = pa[(2 * (st + 1)) * 64]; \
          al[(st + 1) & 1] = pa[(2 * (st + 1) + 1) * 64]; \
        } \
        if (st == 2 && q == 4) { \
          const int t_n = __builtin_amdgcn_readfirstlane(claimed); \
          more = t_n < tb1; \
          t_nxt = more ? t_n : t_cur; \
          SK3_DECODE(nxt, t_nxt); \
          pbn = (unsigned)nxt.u * (SKIN_GROUPS * 1024u); \
        } \
        __builtin_amdgcn_sched_barrier(0); \
      } \
      { \
        constexpr int gh = 2 * st + SK2_RING, gl = 2 * st + 1 + SK2_RING; \
        const unsigned ubh = (gh < SKIN_GROUPS) ? pb : pbn, ubl = (gl < SKIN_GROUPS) ? pb : pbn; \
        rbuf[0][rh] = SK3_LDB(0, ubh, gh % SKIN_GROUPS); \
        rbuf[1][rh] = SK3_LDB(1, ubh, gh % SKIN_GROUPS); \
        rbuf[2][rh] = SK3_LDB(2, ubh, gh % SKIN_GROUPS); \
        rbuf[0][rl] = SK3_LDB(0, ubl, gl % SKIN_GROUPS); \
        rbuf[1][rl] = SK3_LDB(1, ubl, gl % SKIN_GROUPS); \
        rbuf[2][rl] = SK3_LDB(2, ubl, gl % SKIN_GROUPS); \
      } \
      __builtin_amdgcn_sched_barrier(0); \
    }
    SK3_STEP(0) SK3_STEP(1) SK3_STEP(2) SK3_STEP(3) SK3_STEP(4) SK3_STEP(5) SK3_STEP(6)
#undef SK3_STEP
#undef SK3_MFMA
    asm volatile("s_nop 7\n\ts_nop 3");  // MFMA result -> VALU read (the compiler does not know the asm statements are MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    // v_posed = template + 2^-k (large + small)
    q0 = (am0 + as0) * inv_scale + tn0;
    q1 = (am1 + as1) * inv_scale + tn1;
    q2 = (am2 + as2) * inv_scale + tn2;
    wi = wi_n;
    ww = ww_n;
    prv = cur; cur = nxt; pb = pbn; t_cur = t_nxt;
    sk2_unit_addresses(E, prv, wi, kq, j, V, v12, n24);
  }
#undef SK3_LDB
#undef SK3_DECODE
  // epilogue of the wave's last unit
#pragma unroll
  for (int g = 0; g < 12; ++g) {
#pragma unroll
    for (int k = 0; k < 12; ++k) sk2_slice<BBOX, false, true>(g, k, E, ww, q0, q1, q2, sTb, sTrb, v12, n24, rv, rb, rv);
  }
}

// true when every block's tasks of k_skin2 lie in at most two of the three segments it can address (see the kernel)
static bool sk2_fits(int nFT, int nur, int npos) {
  if (nur < 8) return false;
  const int ntot = nFT * nur;
  for (int x = 0; x < 8; ++x) {
    const int xlo = (int)(((long)ntot * x) >> 3), xhi = (int)(((long)ntot * (x + 1)) >> 3);
    for (int pos = 0; pos < npos; ++pos) {
      const int tb0 = xlo + (((xhi - xlo) * pos) >> SK2_NPOS_LOG2), tb1 = xlo + (((xhi - xlo) * (pos + 1)) >> SK2_NPOS_LOG2);
      if (tb1 <= tb0) continue;
      int xa = 0;
      while (xa < 7 && nFT * (((xa + 1) * nur) >> 3) <= tb0) ++xa;
      const int ua0 = (xa * nur) >> 3, ub0 = ((xa + 1) * nur) >> 3, nua = ub0 - ua0;
      const int Sa = nFT * ua0, Sb = nFT * ub0;
      const int ftA = (tb0 - Sa) / nua;
      const int g1 = Sa + ftA * nua;
      const int g2 = (g1 + nua < Sb) ? g1 + nua : Sb;
      const bool seg2 = tb1 > g2 && g2 < Sb;   // tasks of (slice a, ftA + 1)
      const bool seg3 = tb1 > Sb;              // tasks of (slice a + 1, tile 0)
      if (seg2 && seg3) return false;
      if (seg2 && tb1 > g2 + nua) return false;                              // would reach tile ftA + 2
      if (seg3 && (xa >= 7 || tb1 - Sb > (((xa + 2) * nur) >> 3) - ub0)) return false;  // past tile 0 of slice a + 1
    }
  }
  return true;
}

static int uuo_launch_skin_v1(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A,
                              const float* trans, float* verts, float* bbox);

struct SkinCallArgs {  // UUO_OP_SKIN: a whole-GPU kernel; a lock-step batch replays these calls one problem after the other
  UuoGridHdr h;
  const uuo_model* m;
  int F;
  const float* pfaT;
  const float* A;
  const float* trans;
  float* verts;
  float* bbox;
  const void* pfa16;  // non-null: the call is k_skin3's (uuo_launch_skin16), pfaT is unused
};

int uuo_launch_skin(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A, const float* trans,
                    float* verts, float* bbox, float* vp_out) {
  UUO_REQUIRE(!vp_out || (bbox && !uuo_recorder), "uuo_launch_skin: v_posed output goes with the unit boxes, outside lock-step batches");
  {
    SkinCallArgs c{{1, 1}, m, F, pfaT, A, trans, verts, bbox, nullptr};
    if (uuo_record(UUO_OP_SKIN, 1, 1, c)) return 0;
  }
  static const int force_v1 = UUO_ENV_INT("UUO_SKIN_V1", 0);  // ablation / comparison only
  const int nur = (m->V + 15) / 16;  // units with vertices = stride of the box table
  const int npos = 1 << SK2_NPOS_LOG2;  // 8 XCDs x 32 CUs, one 8-wave block per CU
  if (force_v1 || (size_t)SK2_MAX_FT * UUO_FT * m->V * 12 >= 0x7FFFFFF0u) {
    UUO_REQUIRE(!vp_out, "uuo_launch_skin: v_posed output needs the k_skin2 path");
    return uuo_launch_skin_v1(m, s, F, pfaT, A, trans, verts, bbox);
  }
  const int nFT_all = (F + UUO_FT - 1) / UUO_FT;
  for (int ft0 = 0; ft0 < nFT_all; ft0 += SK2_MAX_FT) {
    const int nFT = (nFT_all - ft0 < SK2_MAX_FT) ? nFT_all - ft0 : SK2_MAX_FT;
    const int f0 = ft0 * UUO_FT;
    const int Fl = (F - f0 < nFT * UUO_FT) ? F - f0 : nFT * UUO_FT;
    if (!sk2_fits(nFT, nur, npos)) {
      UUO_REQUIRE(!vp_out, "uuo_launch_skin: v_posed output needs the k_skin2 path");
      return uuo_launch_skin_v1(m, s, F, pfaT, A, trans, verts, bbox);
    }
    const float* pf = pfaT + (size_t)ft0 * UUO_KP * UUO_FT;
    const float* pA = A + (size_t)f0 * UUO_NUM_JOINTS * 12;
    const float* pt = trans ? trans + (size_t)f0 * 3 : nullptr;
    float* pv = verts + (size_t)f0 * m->V * 3;
    static const int var2 = UUO_ENV_INT("UUO_SK2_VAR", 0);  // ablation (timing) only
#define SK2_LAUNCH(BB, VAR)                                                                             \
  hipLaunchKernelGGL((k_skin2<BB, VAR>), dim3(8 * npos), dim3(SKIN_WAVES * 64), 0, s,                   \
                     reinterpret_cast<const float4*>(m->P3), m->vt3, m->Wi, m->Ww, pf, pA, pt, pv,      \
                     (BB) ? bbox + (size_t)f0 * nur * 6 : (float*)nullptr, Fl, m->V, m->VP, nFT, (float*)nullptr)
    if (!bbox) SK2_LAUNCH(false, 0);
    else if (vp_out)
      hipLaunchKernelGGL((k_skin2<true, 0, true>), dim3(8 * npos), dim3(SKIN_WAVES * 64), 0, s,
                         reinterpret_cast<const float4*>(m->P3), m->vt3, m->Wi, m->Ww, pf, pA, pt, pv,
                         bbox + (size_t)f0 * nur * 6, Fl, m->V, m->VP, nFT, vp_out + (size_t)f0 * m->V * 3);
#ifdef UUO_DEBUG_HOOKS  // timing ablations (MFMAs off, epilogue off, refills off, cycle stamps): never in the product binary
    else if (var2 == 1) SK2_LAUNCH(true, 1);
    else if (var2 == 2) SK2_LAUNCH(true, 2);
    else if (var2 == 4) SK2_LAUNCH(true, 4);
    else if (var2 == 3) SK2_LAUNCH(true, 3);   // MFMAs only
    else if (var2 == 5) SK2_LAUNCH(true, 5);   // B stream + LDS operand reads only
    else if (var2 == 6) SK2_LAUNCH(true, 6);   // epilogue only
    else if (var2 == 7) SK2_LAUNCH(true, 7);   // staging + task loop only
    else if (var2 == 16) SK2_LAUNCH(false, 0);
    else if (var2 == 8) SK2_LAUNCH(true, 8);
    else if (var2 == 9) SK2_LAUNCH(true, 9);
    else if (var2 == 10) SK2_LAUNCH(true, 10);
    else if (var2 == 11) SK2_LAUNCH(true, 11);
    else if (var2 == 12) SK2_LAUNCH(true, 12);
    else if (var2 == 24) SK2_LAUNCH(false, 8);
#endif
    else SK2_LAUNCH(true, 0);
    (void)var2;
#undef SK2_LAUNCH
  }
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

#ifdef UUO_DEBUG_HOOKS
extern "C" int uuo_debug_skin_stamps(unsigned long long* h_out) {  // 2048 waves x 8 stamps (UUO_SK2_VAR=9)
  UUO_HIP_CHECK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_sk2_stamps), sizeof(unsigned long long) * 2048 * 16));
  return 0;
}
#endif  // UUO_DEBUG_HOOKS

int uuo_launch_skin16(const uuo_model* m, hipStream_t s, int F, const void* pfa16, const float* A, const float* trans, float* verts,
                      float* bbox) {
  UUO_REQUIRE(m->P16 && pfa16 && bbox, "uuo_launch_skin16: needs the fp16 tables and the unit boxes");
  static const hipError_t lds_attr =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_skin3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SK3_LDS_PAD);
  UUO_HIP_CHECK(lds_attr);
  const int nur = (m->V + 15) / 16;
  const int npos = 1 << SK2_NPOS_LOG2;
  if ((size_t)SK2_MAX_FT * UUO_FT * m->V * 12 >= 0x7FFFFFF0u) return -22;
  const int nFT_all = (F + UUO_FT - 1) / UUO_FT;
  for (int ft0 = 0; ft0 < nFT_all; ft0 += SK2_MAX_FT) {
    const int nFT = (nFT_all - ft0 < SK2_MAX_FT) ? nFT_all - ft0 : SK2_MAX_FT;
    if (!sk2_fits(nFT, nur, npos)) return -22;
  }
  {  // (a lock-step batch replays the call for one problem after the other, like the fp32 kernel's)
    SkinCallArgs c{{1, 1}, m, F, nullptr, A, trans, verts, bbox, pfa16};
    if (uuo_record(UUO_OP_SKIN, 1, 1, c)) return 0;
  }
  for (int ft0 = 0; ft0 < nFT_all; ft0 += SK2_MAX_FT) {
    const int nFT = (nFT_all - ft0 < SK2_MAX_FT) ? nFT_all - ft0 : SK2_MAX_FT;
    const int f0 = ft0 * UUO_FT;
    const int Fl = (F - f0 < nFT * UUO_FT) ? F - f0 : nFT * UUO_FT;
    hipLaunchKernelGGL((k_skin3<true>), dim3(8 * npos), dim3(SK3_WAVES * 64), SK3_LDS_PAD, s, reinterpret_cast<const float4*>(m->P16), m->vt3,
                       m->Wi, m->Ww, reinterpret_cast<const float4*>(pfa16) + (size_t)ft0 * (UUO_KP * UUO_FT / 4),
                       A + (size_t)f0 * UUO_NUM_JOINTS * 12, trans ? trans + (size_t)f0 * 3 : (const float*)nullptr,
                       verts + (size_t)f0 * m->V * 3, bbox + (size_t)f0 * nur * 6, Fl, m->V, m->VP, nFT, m->skin16_inv);
  }
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

static int uuo_launch_skin_v1(const uuo_model* m, hipStream_t s, int F, const float* pfaT, const float* A,
                              const float* trans, float* verts, float* bbox) {
  const int nFT = (F + UUO_FT - 1) / UUO_FT;
  const int nunits = m->VP / 16;
  // vertex ranges per frame tile: one resident round of <= 256 blocks, every wave gets at least one unit
  static const int slots = UUO_ENV_INT("UUO_SKIN_SLOTS", 256);
  int nVB = slots / nFT;
  const int maxVB = (nunits + SKIN_WAVES - 1) / SKIN_WAVES;
  if (nVB > maxVB) nVB = maxVB;
  if (nVB < 1) nVB = 1;
  const int nblocks = nFT * nVB;
  static const int variant = UUO_ENV_INT("UUO_SKIN_VARIANT", 0);  // ablation only
#define SKIN_LAUNCH(VAR)                                                                                     \
  hipLaunchKernelGGL((k_skin<VAR>), dim3(nblocks), dim3(SKIN_WAVES * 64), 0, s, m->P3, m->vt3, m->Wi, m->Ww, \
                     pfaT, A, trans, verts, bbox, F, m->V, m->VP, nFT, nVB, nblocks)
#ifdef UUO_DEBUG_HOOKS  // timing ablations: never in the product binary
  if (variant == 1) SKIN_LAUNCH(1);
  else if (variant == 2) SKIN_LAUNCH(2);
  else if (variant == 3) SKIN_LAUNCH(3);
  else if (variant == 4) SKIN_LAUNCH(4);
  else if (variant == 5) SKIN_LAUNCH(5);
  else
#endif
    SKIN_LAUNCH(0);
  (void)variant;
#undef SKIN_LAUNCH
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// 45 joints of smplx SMPL.forward: 24 posed joints + 21 vertex-picked joints (VertexJointSelector)
// ----------------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------------
// Part stage with a constant body pose (find_best_part_fits optimises yaw, translation and shape only): the template
// plus pose-corrective offsets of every frame, C[f][v] = v_t + P . feat_f, do not change between the closure
// evaluations of one solve.  They are produced once by the MFMA kernel (identity skinning transforms, zero shape) and
// each evaluation then only adds the shape blend and skins the vertices the candidate body part owns:
//   v = T_f(v) (C[f][v] + S[v] . beta) + trans_f,   T_f(v) = sum_n w_n A_f[j_n]
// Thread = vertex (shape offset and skin weights in registers), loop over a group of frames whose skinning matrices sit
// in LDS: 12 B read + 12 B written per vertex and frame instead of the 207-term contraction.
// ----------------------------------------------------------------------------------------------------
#define SKC_FB 10  // frames per block
// `bbox` != null selects the COMPACT form: vertex of candidate i is written at verts[f][i] (subset order, stride ns) and
// every 16 consecutive candidates get their bounding box bbox[f][i / 16][6] -- the layout the box-pruned search
// (k_nn_cull) works on, which then reports candidate positions directly, ties and all, as the subset search must.
__device__ __forceinline__ void skin_cached_body(int F, int V, int ns, const int32_t* __restrict__ subset,
                                                 const float* __restrict__ C, const float* __restrict__ ST,
                                                 const int* __restrict__ Wi, const float* __restrict__ Ww,
                                                 const float* __restrict__ A, const float* __restrict__ betas,
                                                 const float* __restrict__ trans, float* __restrict__ verts,
                                                 float* __restrict__ bbox) {
  __shared__ float sA[SKC_FB * UUO_NUM_JOINTS * 12];
  __shared__ float sTr[SKC_FB * 3];
  const int f0 = blockIdx.y * SKC_FB, nf = min(SKC_FB, F - f0);
  for (int i = threadIdx.x; i < nf * UUO_NUM_JOINTS * 12; i += 256) sA[i] = A[(size_t)f0 * UUO_NUM_JOINTS * 12 + i];
  if (threadIdx.x < nf * 3) sTr[threadIdx.x] = trans ? trans[(size_t)f0 * 3 + threadIdx.x] : 0.f;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool compact = bbox != nullptr;  // block-uniform
  if (!compact && i >= ns) return;
  if (compact && (i & ~15) >= ns) return;  // whole DPP rows past the end leave together; a partly filled one stays whole
  const bool mine = i < ns;
  const int ic = mine ? i : ns - 1;  // lanes past the end of a partly filled unit repeat the last candidate: same box
  const int v = subset ? subset[ic] : ic;
  if ((unsigned)v >= (unsigned)V) {
    if (!compact) return;
  }
  const int vs = ((unsigned)v < (unsigned)V) ? v : 0;  // (memory safety only: subsets hold valid vertex ids)
  float sb[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int l = 0; l < 10; ++l) sb[c] = fmaf(ST[(size_t)vs * 30 + c * 10 + l], betas[l], sb[c]);
  const int4 wi = *reinterpret_cast<const int4*>(Wi + (size_t)vs * 4);
  const float4 ww = *reinterpret_cast<const float4*>(Ww + (size_t)vs * 4);
  const int wj[4] = {wi.x, wi.y, wi.z, wi.w};
  const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
  const int nuc = (ns + 15) >> 4;  // units of 16 candidates (compact form)
  // the cached blend values of all the block's frames are requested before the first one is used (one L2 / HBM round trip
  // per vertex instead of one per frame: the kernel is nothing but latency)
  float cx[SKC_FB], cy[SKC_FB], cz[SKC_FB];
#pragma unroll
  for (int q = 0; q < SKC_FB; ++q) {
    const float* pc = C + ((size_t)(f0 + (q < nf ? q : nf - 1)) * V + vs) * 3;
    cx[q] = pc[0]; cy[q] = pc[1]; cz[q] = pc[2];
  }
#pragma unroll
  for (int q = 0; q < SKC_FB; ++q) {
    if (q >= nf) continue;  // block-uniform (the last block of frames may be short)
    const float px = cx[q] + sb[0], py = cy[q] + sb[1], pz = cz[q] + sb[2];
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const float* a = sA + (q * UUO_NUM_JOINTS + (wj[n] < 0 ? 0 : wj[n])) * 12;
      const float w = (wj[n] < 0) ? 0.f : wv[n];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = fmaf(w, a[e], T[e]);
    }
    const float ox = fmaf(T[2], pz, fmaf(T[1], py, T[0] * px)) + T[3] + sTr[q * 3];
    const float oy = fmaf(T[6], pz, fmaf(T[5], py, T[4] * px)) + T[7] + sTr[q * 3 + 1];
    const float oz = fmaf(T[10], pz, fmaf(T[9], py, T[8] * px)) + T[11] + sTr[q * 3 + 2];
    if (!compact) {
      float* o = verts + ((size_t)(f0 + q) * V + v) * 3;
      o[0] = ox; o[1] = oy; o[2] = oz;
    } else {
      if (mine) {
        float* o = verts + ((size_t)(f0 + q) * ns + i) * 3;
        o[0] = ox; o[1] = oy; o[2] = oz;
      }
      // the 16 candidates of a unit sit on the 16 lanes of one DPP row
      const float lx = row16_min(ox), ly = row16_min(oy), lz = row16_min(oz);
      const float hx = row16_max(ox), hy = row16_max(oy), hz = row16_max(oz);
      if ((i & 15) == 0) {
        float* pb = bbox + ((size_t)(f0 + q) * nuc + (i >> 4)) * 6;
        pb[0] = lx; pb[1] = ly; pb[2] = lz;
        pb[3] = hx; pb[4] = hy; pb[5] = hz;
      }
    }
  }
}

struct SkinCachedArgs {
  UuoGridHdr h;
  int F, V, ns;
  uuo_gptr<const int32_t> subset;
  uuo_gptr<const float> C;
  uuo_gptr<const float> ST;
  uuo_gptr<const int> Wi;
  uuo_gptr<const float> Ww;
  uuo_gptr<const float> A;
  uuo_gptr<const float> betas;
  uuo_gptr<const float> trans;
  uuo_gptr<float> verts;
  uuo_gptr<float> bbox;  // null: vertices by vertex id; else the compact form (see skin_cached_body)
};
__global__ __launch_bounds__(256) void k_skin_cached(SkinCachedArgs a) {
  skin_cached_body(a.F, a.V, a.ns, a.subset, a.C, a.ST, a.Wi, a.Ww, a.A, a.betas, a.trans, a.verts, a.bbox);
}
__global__ __launch_bounds__(256) void k_skin_cached_b(const SkinCachedArgs* __restrict__ batch) {
  UUO_BATCH_PICK(SkinCachedArgs, batch)
  skin_cached_body(a.F, a.V, a.ns, a.subset, a.C, a.ST, a.Wi, a.Ww, a.A, a.betas, a.trans, a.verts, a.bbox);
}

int uuo_launch_skin_cached(const uuo_model* m, hipStream_t s, int F, const float* cache, const float* A,
                           const float* betas, const float* trans, const int32_t* subset, int n_subset, float* verts,
                           float* bbox_compact) {
  const int ns = subset ? n_subset : m->V;
  if (F <= 0 || ns <= 0) return 0;
  const int gx = (ns + 255) / 256, gy = (F + SKC_FB - 1) / SKC_FB;
  SkinCachedArgs a{{gx, gy}, F, m->V, ns, subset, cache, m->ST, m->Wi, m->Ww, A, betas, trans, verts, bbox_compact};
  if (uuo_record(UUO_OP_SKIN_CACHED, gx, gy, a)) return 0;
  hipLaunchKernelGGL(k_skin_cached, dim3(gx, gy), dim3(256), 0, s, a);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void k_identity_transforms(int count, float* __restrict__ A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count * 12) {
    const int e = i % 12;
    A[i] = (e == 0 || e == 5 || e == 10) ? 1.f : 0.f;
  }
}

int uuo_launch_identity_transforms(hipStream_t s, int count, float* A) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_identity_transforms, dim3((count * 12 + 255) / 256), dim3(256), 0, s, count, A);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

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
  // Scratch (operand tiles, skinning matrices, posed joints) is kept per stream.  torch hands out stream handles from a
  // small pool, so two host threads may hold the same hipStream_t: the entry's own mutex is held across the reallocation
  // AND the three launches, so the two callers' launch sequences never interleave on the shared buffers (in-stream
  // order then makes the reuse safe), and nobody keeps a copy of pointers another caller may free.
  uuo_model::FwdScratch* scp;
  {
    std::lock_guard<std::mutex> lock(m->fwd_mutex);
    scp = &m->fwd[s];  // std::map nodes never move
  }
  uuo_model::FwdScratch& sc = *scp;
  std::lock_guard<std::mutex> entry_lock(sc.mu);
  if (sc.cap < nFT) {
    // hipFree waits for the device, so kernels of an earlier call that still read the old buffers have finished
    if (sc.pfaT) (void)hipFree(sc.pfaT);
    if (sc.A) (void)hipFree(sc.A);
    if (sc.jp) (void)hipFree(sc.jp);
    sc.pfaT = sc.A = sc.jp = nullptr;
    sc.cap = 0;
    UUO_HIP_CHECK(hipMalloc((void**)&sc.pfaT, (size_t)nFT * UUO_KP * UUO_FT * sizeof(float)));
    UUO_HIP_CHECK(hipMalloc((void**)&sc.A, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 12 * sizeof(float)));
    UUO_HIP_CHECK(hipMalloc((void**)&sc.jp, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 3 * sizeof(float)));
    // on the caller's stream: a null-stream memset is not ordered with torch's non-blocking streams and could land
    // after the first pose_prep has written the tiles
    UUO_HIP_CHECK(hipMemsetAsync(sc.pfaT, 0, (size_t)nFT * UUO_KP * UUO_FT * sizeof(float), s));
    UUO_HIP_CHECK(hipMemsetAsync(sc.A, 0, (size_t)nFT * UUO_FT * UUO_NUM_JOINTS * 12 * sizeof(float), s));
    sc.cap = nFT;
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

// batched launches of this file's kernels (uuo_common.h): 0 = launched, 1 = not one of mine, < 0 = error
int uuo_batched_launch_smpl(int op, hipStream_t s, const void* d_args, int count, int gx, int gy) {
  if (op == UUO_OP_POSE_PREP) {
    hipLaunchKernelGGL(k_pose_prep_b, dim3(gx, gy, count), dim3(64), 0, s, (const PosePrepArgs*)d_args);
  } else if (op == UUO_OP_SKIN_CACHED) {
    hipLaunchKernelGGL(k_skin_cached_b, dim3(gx, gy, count), dim3(256), 0, s, (const SkinCachedArgs*)d_args);
  } else {
    return 1;
  }
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

// replay of one recorded UUO_OP_SKIN call (host copy of its arguments), outside record mode
int uuo_replay_skin_call(hipStream_t s, const void* h_args) {
  const SkinCallArgs* c = (const SkinCallArgs*)h_args;
  if (c->pfa16) return uuo_launch_skin16(c->m, s, c->F, c->pfa16, c->A, c->trans, c->verts, c->bbox);
  return uuo_launch_skin(c->m, s, c->F, c->pfaT, c->A, c->trans, c->verts, c->bbox);
}
