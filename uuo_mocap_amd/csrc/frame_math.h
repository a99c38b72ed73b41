// Per-frame SMPL math shared by the forward (pose_prep) and backward (bwd) kernels.
// Semantics restated from smplx.lbs.batch_rigid_transform and the pytorch3d rotation transforms the
// reference calls (reference optimization.py:66-74,192-200,336-338,662-679).
#pragma once
#include "uuo_common.h"

struct FrameLds {
  float R[UUO_NUM_JOINTS][9];   // per-joint local rotation (after the stage's normalisation)
  float J[UUO_NUM_JOINTS][3];   // rest joints for this frame's betas
  float GR[UUO_NUM_JOINTS][9];  // world rotation  G_j^R
  float Gt[UUO_NUM_JOINTS][3];  // world translation G_j^t
  float beta[10];
  float Mroot[9];               // Rz(z).root before Gram-Schmidt (root modes Z_GS / ZSHARED)
  float Rz[4];                  // c, -s, s, c entries actually used (R00,R01,R10,R11)
};

__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      o[r * 3 + c] = fmaf(a[r * 3 + 2], b[6 + c], fmaf(a[r * 3 + 1], b[3 + c], a[r * 3 + 0] * b[c]));
}

// rotation_6d_to_matrix(matrix_to_rotation_6d(M)): Gram-Schmidt on rows 0,1; row 2 = cross (F.normalize eps 1e-12)
__device__ __forceinline__ void gs6d_forward(const float* a, float* R) {
  float a1x = a[0], a1y = a[1], a1z = a[2], a2x = a[3], a2y = a[4], a2z = a[5];
  float n1 = sqrtf(a1x * a1x + a1y * a1y + a1z * a1z);
  float d1 = fmaxf(n1, 1e-12f);
  float b1x = a1x / d1, b1y = a1y / d1, b1z = a1z / d1;
  float s = b1x * a2x + b1y * a2y + b1z * a2z;
  float ux = a2x - s * b1x, uy = a2y - s * b1y, uz = a2z - s * b1z;
  float n2 = sqrtf(ux * ux + uy * uy + uz * uz);
  float d2 = fmaxf(n2, 1e-12f);
  float b2x = ux / d2, b2y = uy / d2, b2z = uz / d2;
  R[0] = b1x; R[1] = b1y; R[2] = b1z;
  R[3] = b2x; R[4] = b2y; R[5] = b2z;
  R[6] = b1y * b2z - b1z * b2y;
  R[7] = b1z * b2x - b1x * b2z;
  R[8] = b1x * b2y - b1y * b2x;
}

// Backward of gs6d_forward: dR (3 rows) -> gradient on raw rows 0,1 (row 2 receives none).
__device__ __forceinline__ void gs6d_backward(const float* a, const float* dR, float* da /*6*/) {
  float a1[3] = {a[0], a[1], a[2]}, a2[3] = {a[3], a[4], a[5]};
  float n1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
  float d1 = fmaxf(n1, 1e-12f);
  float b1[3] = {a1[0] / d1, a1[1] / d1, a1[2] / d1};
  float s = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  float u2[3] = {a2[0] - s * b1[0], a2[1] - s * b1[1], a2[2] - s * b1[2]};
  float n2 = sqrtf(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  float d2 = fmaxf(n2, 1e-12f);
  float b2[3] = {u2[0] / d2, u2[1] / d2, u2[2] / d2};
  float db1[3] = {dR[0], dR[1], dR[2]}, db2[3] = {dR[3], dR[4], dR[5]}, db3[3] = {dR[6], dR[7], dR[8]};
  // b3 = b1 x b2
  db1[0] += b2[1] * db3[2] - b2[2] * db3[1];
  db1[1] += b2[2] * db3[0] - b2[0] * db3[2];
  db1[2] += b2[0] * db3[1] - b2[1] * db3[0];
  db2[0] += db3[1] * b1[2] - db3[2] * b1[1];
  db2[1] += db3[2] * b1[0] - db3[0] * b1[2];
  db2[2] += db3[0] * b1[1] - db3[1] * b1[0];
  // b2 = u2 / |u2|
  float p2 = b2[0] * db2[0] + b2[1] * db2[1] + b2[2] * db2[2];
  float du2[3] = {(db2[0] - p2 * b2[0]) / d2, (db2[1] - p2 * b2[1]) / d2, (db2[2] - p2 * b2[2]) / d2};
  // u2 = a2 - (b1.a2) b1
  float ds = -(b1[0] * du2[0] + b1[1] * du2[1] + b1[2] * du2[2]);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    da[3 + c] = du2[c] + ds * b1[c];
    db1[c] += -s * du2[c] + ds * a2[c];
  }
  // b1 = a1 / |a1|
  float p1 = b1[0] * db1[0] + b1[1] * db1[1] + b1[2] * db1[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) da[c] = (db1[c] - p1 * b1[c]) / d1;
}

// axis_angle_to_matrix((0,0,z)) through pytorch3d's quaternion route; returns R00,R01,R10,R11 (R22 = 1).
__device__ __forceinline__ void rz_entries(float z, float* o) {
  float angle = fabsf(z);
  float half = angle * 0.5f;
  float sh = (angle < 1e-6f) ? (0.5f - (angle * angle) / 48.f) : (sinf(half) / angle);
  float qr = cosf(half), qk = z * sh;
  float two_s = 2.0f / (qr * qr + qk * qk);
  o[0] = 1.f - two_s * (qk * qk);
  o[1] = two_s * (0.f - qk * qr);
  o[2] = two_s * (0.f + qk * qr);
  o[3] = 1.f - two_s * (qk * qk);
}

// Forward of one frame into LDS: rotations, joints, world transforms.  Every thread of the block must
// call it (it contains barriers); only threads 0..23 do work.
__device__ __forceinline__ void frame_forward(const UuoPoseSrc& src, const UuoTree* __restrict__ tree, int f,
                                              FrameLds& L) {
  const int l = threadIdx.x;
  // the tree's level tables for the kinematic steps below: issued first, so their round trip overlaps the pose loads
  const int max_depth = tree->max_depth;
  const int lk = l / 12, le = l - lk * 12;
  int lvl_j[UUO_MAX_DEPTH], lvl_p[UUO_MAX_DEPTH];
#pragma unroll
  for (int d = 1; d < UUO_MAX_DEPTH; ++d) {
    const bool on = d <= max_depth && l < 12 * UUO_LEVEL_W && lk < tree->level_n[d];
    lvl_j[d] = on ? tree->level_j[d][lk] : -1;
    lvl_p[d] = on ? tree->level_p[d][lk] : 0;
  }
  // every global load of the frame (raw rotations, shape, yaw) is issued before the first barrier: one round trip
  float raw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (l < UUO_NUM_JOINTS) {
    const float* pr = (l == 0) ? src.root + (size_t)f * 9 : src.body + ((size_t)f * 23 + (l - 1)) * 9;
#pragma unroll
    for (int e = 0; e < 9; ++e) raw[e] = pr[e];
  }
  float jt[3] = {0.f, 0.f, 0.f}, js[3][10];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (l < UUO_NUM_JOINTS) jt[c] = tree->Jt[l][c];
#pragma unroll
    for (int k = 0; k < 10; ++k) js[c][k] = (l < UUO_NUM_JOINTS) ? tree->JS[l][c][k] : 0.f;
  }
  if (l < 10) L.beta[l] = src.betas[(size_t)f * src.betas_stride + l];
  if (l == 32 && src.root_mode >= UUO_ROOT_Z_GS) {
    float z = (src.root_mode == UUO_ROOT_Z_GS) ? src.z[f] : src.z[0];
    rz_entries(z, L.Rz);
  }
  __syncthreads();
  if (l < UUO_NUM_JOINTS) {
    float R[9];
    if (l == 0) {
      if (src.root_mode >= UUO_ROOT_Z_GS) {
        float m[9];
        const float c00 = L.Rz[0], c01 = L.Rz[1], c10 = L.Rz[2], c11 = L.Rz[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          m[c] = fmaf(c01, raw[3 + c], c00 * raw[c]);
          m[3 + c] = fmaf(c11, raw[3 + c], c10 * raw[c]);
          m[6 + c] = raw[6 + c];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          L.Mroot[e] = m[e];
          raw[e] = m[e];
        }
      }
      if (src.root_mode == UUO_ROOT_GS || src.root_mode == UUO_ROOT_Z_GS) {
        gs6d_forward(raw, R);
      } else {
#pragma unroll
        for (int e = 0; e < 9; ++e) R[e] = raw[e];
      }
    } else {
      if (src.norm_body) {
        gs6d_forward(raw, R);
      } else {
#pragma unroll
        for (int e = 0; e < 9; ++e) R[e] = raw[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) L.R[l][e] = R[e];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = jt[c];
#pragma unroll
      for (int k = 0; k < 10; ++k) acc = fmaf(js[c][k], L.beta[k], acc);
      L.J[l][c] = acc;
    }
  }
  __syncthreads();
  // Kinematic chain, one depth per step, one lane per (joint of the level, entry): entries 0..8 are G_j^R = G_p^R R_j,
  // 9..11 are G_j^t = G_p^R (J_j - J_p) + G_p^t.  SMPL has <= 5 joints per level, so a step occupies <= 60 lanes of wave 0
  // and costs an LDS round trip and three FMAs (the 24-lane version spent ~40 dependent instructions per joint per
  // step); every entry is the same FMA chain as mat3_mul, so the result is bit-identical to it.
  if (l < 12) {
    if (l < 9) L.GR[0][l] = L.R[0][l];
    else L.Gt[0][l - 9] = L.J[0][l - 9];
  }
  __syncthreads();
#pragma unroll
  for (int d = 1; d < UUO_MAX_DEPTH; ++d) {
    if (d <= max_depth) {  // block-uniform
      const int j = lvl_j[d], p = lvl_p[d];
      if (j >= 0) {
        if (le < 9) {
          const int r = le / 3, c = le - r * 3;
          L.GR[j][le] = fmaf(L.GR[p][r * 3 + 2], L.R[j][6 + c], fmaf(L.GR[p][r * 3 + 1], L.R[j][3 + c], L.GR[p][r * 3] * L.R[j][c]));
        } else {
          const int r = le - 9;
          const float rel0 = L.J[j][0] - L.J[p][0], rel1 = L.J[j][1] - L.J[p][1], rel2 = L.J[j][2] - L.J[p][2];
          L.Gt[j][r] = fmaf(L.GR[p][r * 3 + 2], rel2, fmaf(L.GR[p][r * 3 + 1], rel1, L.GR[p][r * 3] * rel0)) + L.Gt[p][r];
        }
      }
      __syncthreads();
    }
  }
}

// A_j = [G^R | G^t - G^R J_j] as 3x4 row-major (the skinning matrix of smplx.lbs)
__device__ __forceinline__ void frame_skin_matrix(const FrameLds& L, int j, float* A12) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float gj = fmaf(L.GR[j][r * 3 + 2], L.J[j][2], fmaf(L.GR[j][r * 3 + 1], L.J[j][1], L.GR[j][r * 3] * L.J[j][0]));
    A12[r * 4 + 0] = L.GR[j][r * 3 + 0];
    A12[r * 4 + 1] = L.GR[j][r * 3 + 1];
    A12[r * 4 + 2] = L.GR[j][r * 3 + 2];
    A12[r * 4 + 3] = L.Gt[j][r] - gj;
  }
}

// Wave-wide sum on the VALU: four DPP row-rotate adds leave each 16-lane row's sum on all of its lanes, then four
// v_readlane gather the rows (an LDS-crossbar butterfly of ds_bpermute costs ~100 cycles per step on the dependent
// chain; this is ~10 short instructions).  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum_fast(float v) {
  v += dpp_rot<0x128>(v);  // row_ror:8
  v += dpp_rot<0x124>(v);  // row_ror:4
  v += dpp_rot<0x122>(v);  // row_ror:2
  v += dpp_rot<0x121>(v);  // row_ror:1
  const int i = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// fp64 wave sum on the VALU: four row-rotate steps on both halves of the double (DPP), then the four row totals
// are read with v_readlane -- no LDS traffic (the __shfl_xor version is 12 ds_bpermute per call)
template <int CTRL>
__device__ __forceinline__ double dpp_rot_d(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xF, 0xF, false),
                          __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_d_fast(double v) {
  v += dpp_rot_d<0x128>(v);
  v += dpp_rot_d<0x124>(v);
  v += dpp_rot_d<0x122>(v);
  v += dpp_rot_d<0x121>(v);
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
