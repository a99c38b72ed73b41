// The 2D-prior fit of one yaw hypothesis (reference utils/hmr_utils.py:170-425; closure at :281-365) as ONE fused closure.
//
// What the reference evaluates per closure: two SMPL forwards (6 890 vertices x F, the 207 x 20 670 pose blend twice), a
// 45-joint perspective projection and a one-directional chamfer term -- although the body pose, the shape (detached,
// hmr_utils.py:218,292) and the HMR root orientation never change during this solve: its 3F + 4 live parameters are a yaw
// about the camera's vertical axis, a body translation per frame and one camera translation.  Both forwards therefore reduce
// to ONE forward before the solve (joints0 / verts0: the HMR pose with the HMR root orientation and zero translation):
//     joints(f,j)  = joints0(f,j) + inv_t(f),              inv_t = Ry(-yaw) (b_f - c) + c          (hmr_utils.py:300-310)
//     vertex(f,i)  = C (Ry(yaw) W(f,i) + b_f) + j0(f),     W = verts0 - j0,  j0 = joints0(f,0)      (hmr_utils.py:325-333)
// (the root rotation C Ry(yaw) R0 turns the posed body about its pelvis j0; C = HMR -> mocap axes), and the nearest-vertex
// search of the chamfer term becomes a search of M moving query points u = Ry(yaw)^T (C^T (m - j0) - b_f) against the
// CONSTANT cloud W -- no skinning at all inside the closure.  Three launches per evaluation: k_reproj_search (grid
// F x 4 vertex slices: every lane keeps the running minimum of its vertices for 16 markers per pass, the markers in scalar
// registers and two per packed-fp32 instruction; a transposing butterfly merges the wave, packed 64-bit (distance, index)
// keys = pytorch3d's first-index tie order), k_reproj_terms (one wave per frame: the matched pairs' residuals, the 45
// joints through the pinhole camera, both gradients, the frame's partial sums) and k_reproj_sum (one block: double sums).
#include "uuo_common.h"

#define RPJ_T 256    // threads of a search block (4 waves)
#define RPJ_S 4      // vertex slices per frame (grid = F x RPJ_S search blocks: 1 200 at F = 300, all resident at once)
#define RPJ_MC 16    // markers per register pass (their coordinates live in scalar registers, the running minima in 32 VGPRs)
#define RPJ_NJ 64    // joints a frame may have (SMPL + the extra vertex joints: 45)
#define RPJ_PW 8     // doubles of a frame's partial sums: sum of squared key-point residuals (masked), sum of squared
                     // nearest distances, d/d yaw, d/d camera xyz, 2 unused

struct ReprojArgs {
  int F, M, V, J;
  const float* x;
  const float* markers;
  const float* joints0;
  const float* verts0;
  const float* kp_target;
  const float* mask;
  float fx, fy, cx, cy;
  float coef_rep, coef_ch;  // 2 w_reprojection / (F J 2), 2 w_chamfer / (F M)
  float* grad;
  double* part;
  unsigned long long* keys;  // [F][RPJ_S][M] packed (distance bits, vertex) minima of the slices
  float* kp_out;
  int32_t* nn_idx;
};

typedef float rpj2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rpj_uniform(float v) {  // a wave-uniform value, kept in a scalar register
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ double rpj_wave_sum(double v) {  // (sums of many signed fp32 terms: the yaw gradient nearly cancels)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// minima of N keys per lane over the wave's 64 lanes: a transposing butterfly -- at every step a lane keeps the half of its
// keys that its side of the exchange owns and receives the partner's copy of them (N/2 + N/4 + .. exchanges instead of
// 6 N).  Returns the minimum of key (lane >> 2) & (N - 1)... of marker index rpj_owned(lane), valid in every lane.
template <int N>
__device__ __forceinline__ unsigned long long rpj_transpose_min(unsigned long long (&key)[N], int lane) {
  static_assert(N == 16, "16 keys per lane: four halving steps (lane bits 5..2), then two plain steps (bits 1, 0)");
  unsigned long long k8[8], k4[4], k2[2];
  {
    const bool up = lane & 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned long long mine = up ? key[i + 8] : key[i], give = up ? key[i] : key[i + 8];
      const unsigned long long got = __shfl_xor(give, 32, 64);
      k8[i] = got < mine ? got : mine;
    }
  }
  {
    const bool up = lane & 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned long long mine = up ? k8[i + 4] : k8[i], give = up ? k8[i] : k8[i + 4];
      const unsigned long long got = __shfl_xor(give, 16, 64);
      k4[i] = got < mine ? got : mine;
    }
  }
  {
    const bool up = lane & 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned long long mine = up ? k4[i + 2] : k4[i], give = up ? k4[i] : k4[i + 2];
      const unsigned long long got = __shfl_xor(give, 8, 64);
      k2[i] = got < mine ? got : mine;
    }
  }
  unsigned long long k1;
  {
    const bool up = lane & 4;
    const unsigned long long mine = up ? k2[1] : k2[0], give = up ? k2[0] : k2[1];
    const unsigned long long got = __shfl_xor(give, 4, 64);
    k1 = got < mine ? got : mine;
  }
#pragma unroll
  for (int off = 2; off >= 1; off >>= 1) {
    const unsigned long long got = __shfl_xor(k1, off, 64);
    k1 = got < k1 ? got : k1;
  }
  return k1;
}
// the key index a lane holds after rpj_transpose_min<16>: bit 5 chose +8, bit 4 +4, bit 3 +2, bit 2 +1
__device__ __forceinline__ int rpj_owned(int lane) {
  return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

// search: block (f, s) finds, for every marker of frame f, the nearest vertex among slice s of the frame's vertices
__global__ __launch_bounds__(RPJ_T) void k_reproj_search(ReprojArgs a) {
  __shared__ unsigned long long sK[RPJ_T / 64][RPJ_MC];
  __shared__ float sU[RPJ_MC][3];
  const int f = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.M, V = a.V;
  const float yaw = rpj_uniform(a.x[0]);
  const float sn = sinf(yaw), cs = cosf(yaw);
  const float bx = rpj_uniform(a.x[1 + 3 * f]), by = rpj_uniform(a.x[2 + 3 * f]), bz = rpj_uniform(a.x[3 + 3 * f]);
  const float* j0p = a.joints0 + (size_t)f * a.J * 3;
  const float j0x = rpj_uniform(j0p[0]), j0y = rpj_uniform(j0p[1]), j0z = rpj_uniform(j0p[2]);
  const float* vf = a.verts0 + (size_t)f * V * 3;
  const int per = (V + RPJ_S - 1) / RPJ_S, i0 = sl * per, i1 = min(V, i0 + per);
  for (int m0 = 0; m0 < M; m0 += RPJ_MC) {
    // the pass' query points  u = Ry(yaw)^T (C^T (m - j0) - b):  mocap axes -> HMR axes (x, -z, y), then into the body's
    // un-yawed frame; pairs of markers share the packed-fp32 instructions (same IEEE operations, element by element); a pass
    // past the end repeats the last marker (its minima are simply not stored)
    // (computed once per block by 16 lanes and broadcast through LDS: done by every wave it was a quarter of the kernel's
    // vector instructions)
    if (tid < RPJ_MC) {
      const float* mp = a.markers + ((size_t)f * M + min(m0 + tid, M - 1)) * 3;
      const float qx = (mp[0] - j0x) - bx, qy = -(mp[2] - j0z) - by, qz = (mp[1] - j0y) - bz;
      sU[tid][0] = cs * qx - sn * qz;
      sU[tid][1] = qy;
      sU[tid][2] = sn * qx + cs * qz;
    }
    __syncthreads();  // (the previous pass' readers of sU are past its two barriers below)
    rpj2 ux[RPJ_MC / 2], uy[RPJ_MC / 2], uz[RPJ_MC / 2];
#pragma unroll
    for (int k = 0; k < RPJ_MC; ++k) {
      const float vx = rpj_uniform(sU[k][0]), vy = rpj_uniform(sU[k][1]), vz = rpj_uniform(sU[k][2]);
      if (k & 1) { ux[k / 2].y = vx; uy[k / 2].y = vy; uz[k / 2].y = vz; }
      else { ux[k / 2].x = vx; uy[k / 2].x = vy; uz[k / 2].x = vz; }
    }
    unsigned bd[RPJ_MC], bi[RPJ_MC];
#pragma unroll
    for (int k = 0; k < RPJ_MC; ++k) { bd[k] = 0x7F800000u; bi[k] = 0x7FFFFFFFu; }  // +inf, no vertex
    for (int i = i0 + tid; i < i1; i += RPJ_T) {
      const float wx = vf[3 * i] - j0x, wy = vf[3 * i + 1] - j0y, wz = vf[3 * i + 2] - j0z;
      const rpj2 wx2 = rpj2{wx, wx}, wy2 = rpj2{wy, wy}, wz2 = rpj2{wz, wz};
#pragma unroll
      for (int k = 0; k < RPJ_MC / 2; ++k) {
        const rpj2 dx = ux[k] - wx2, dy = uy[k] - wy2, dz = uz[k] - wz2;
        const rpj2 d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // the bits of a non-negative distance order as unsigned exactly as the value does (a NaN orders above +inf and is
          // never taken); a lane visits its vertices in rising order, so strict '<' keeps the first minimum
          const unsigned d = __float_as_uint(h ? d2.y : d2.x);
          const bool better = d < bd[2 * k + h];
          bd[2 * k + h] = better ? d : bd[2 * k + h];
          bi[2 * k + h] = better ? (unsigned)i : bi[2 * k + h];
        }
      }
    }
    unsigned long long key[RPJ_MC];
#pragma unroll
    for (int k = 0; k < RPJ_MC; ++k) key[k] = ((unsigned long long)bd[k] << 32) | bi[k];
    const unsigned long long kmin = rpj_transpose_min<RPJ_MC>(key, lane);
    __syncthreads();  // (the previous pass' sK has been consumed)
    if ((lane & 3) == 0) sK[wave][rpj_owned(lane)] = kmin;
    __syncthreads();
    if (tid < RPJ_MC && m0 + tid < M) {
      unsigned long long k0 = sK[0][tid];
#pragma unroll
      for (int w = 1; w < RPJ_T / 64; ++w) { const unsigned long long o = sK[w][tid]; k0 = o < k0 ? o : k0; }
      a.keys[((size_t)f * RPJ_S + sl) * M + m0 + tid] = k0;
    }
  }
}

// terms: one wave per frame merges the slices' minima and evaluates both terms with their gradients
__global__ __launch_bounds__(64) void k_reproj_terms(ReprojArgs a) {
  const int f = blockIdx.x, lane = threadIdx.x;
  const int F = a.F, M = a.M, V = a.V;
  const float yaw = a.x[0];
  const float sn = sinf(yaw), cs = cosf(yaw);
  const float bx = a.x[1 + 3 * f], by = a.x[2 + 3 * f], bz = a.x[3 + 3 * f];
  const float* j0p = a.joints0 + (size_t)f * a.J * 3;
  const float j0x = j0p[0], j0y = j0p[1], j0z = j0p[2];
  const float* vf = a.verts0 + (size_t)f * V * 3;
  // ---- chamfer term
  double cl = 0.0, cyaw = 0.0, cbx = 0.0, cby = 0.0, cbz = 0.0;  // the terms are fp32 products, their sums double
  for (int m = lane; m < M; m += 64) {
    unsigned long long key = a.keys[((size_t)f * RPJ_S) * M + m];
#pragma unroll
    for (int sl = 1; sl < RPJ_S; ++sl) { const unsigned long long o = a.keys[((size_t)f * RPJ_S + sl) * M + m]; key = o < key ? o : key; }
    const int i = (int)(unsigned)(key & 0xFFFFFFFFull);
    if (i < V) {  // (a NaN marker has no minimum: it contributes nothing, where the reference would propagate the NaN)
      if (a.nn_idx) a.nn_idx[(size_t)f * M + m] = i;
      const float wx = vf[3 * i] - j0x, wy = vf[3 * i + 1] - j0y, wz = vf[3 * i + 2] - j0z;
      const float* mp = a.markers + ((size_t)f * M + m) * 3;
      // d = Ry(yaw) W + b - C^T (m - j0): the residual vertex - marker in HMR axes
      const float rx = cs * wx + sn * wz, rz = -sn * wx + cs * wz;
      const float dx = rx + bx - (mp[0] - j0x), dy = wy + by + (mp[2] - j0z), dz = rz + bz - (mp[1] - j0y);
      cl += (double)(dx * dx + dy * dy + dz * dz);
      const float gx_ = a.coef_ch * dx, gy_ = a.coef_ch * dy, gz_ = a.coef_ch * dz;
      cbx += (double)gx_; cby += (double)gy_; cbz += (double)gz_;
      cyaw += (double)(gx_ * rz) - (double)(gz_ * rx);  // d . (d Ry / d yaw) W,  (d Ry / d yaw) W = (rz, 0, -rx)
    }
  }
  cl = rpj_wave_sum(cl); cyaw = rpj_wave_sum(cyaw); cbx = rpj_wave_sum(cbx); cby = rpj_wave_sum(cby); cbz = rpj_wave_sum(cbz);
  // ---- key-point term: the frame's joints through the pinhole camera
  const float ccx = a.x[1 + 3 * F], ccy = a.x[2 + 3 * F], ccz = a.x[3 + 3 * F];
  const float ex = bx - ccx, ey = by - ccy, ez = bz - ccz;
  // inv_t = Ry(-yaw) (b - c) + c
  const float tx = (cs * ex - sn * ez) + ccx, ty = ey + ccy, tz = (sn * ex + cs * ez) + ccz;
  const float mk = a.mask[f];
  double l = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
  if (lane < a.J) {
    const float* jp = j0p + 3 * lane;
    const float px = (jp[0] + tx) + ccx, py = (jp[1] + ty) + ccy, pz = (jp[2] + tz) + ccz;
    const float kx = ((px / pz) * a.fx + a.cx) + 0.5f, ky = ((py / pz) * a.fy + a.cy) + 0.5f;
    if (a.kp_out) {
      a.kp_out[((size_t)f * a.J + lane) * 2] = kx;
      a.kp_out[((size_t)f * a.J + lane) * 2 + 1] = ky;
    }
    const float* tp = a.kp_target + ((size_t)f * a.J + lane) * 2;
    const float r0 = kx - tp[0], r1 = ky - tp[1];
    l = (double)((r0 * r0 + r1 * r1) * mk);
    const float jx = a.coef_rep * mk * r0 * a.fx / pz, jy = a.coef_rep * mk * r1 * a.fy / pz;
    gx = (double)jx;
    gy = (double)jy;
    gz = (double)(-(jx * px + jy * py) / pz);
  }
  l = rpj_wave_sum(l); gx = rpj_wave_sum(gx); gy = rpj_wave_sum(gy); gz = rpj_wave_sum(gz);
  if (lane == 0) {
    const double rgx = cs * gx + sn * gz, rgz = -sn * gx + cs * gz;  // Ry(yaw) G = (d inv_t / d b)^T G
    a.grad[1 + 3 * f] = (float)(cbx + rgx);
    a.grad[2 + 3 * f] = (float)(cby + gy);
    a.grad[3 + 3 * f] = (float)(cbz + rgz);
    double* pp = a.part + (size_t)f * RPJ_PW;
    pp[0] = l;
    pp[1] = cl;
    // d inv_t / d yaw = (d Ry(-yaw) / d yaw) (b - c) = (-sn ex - cs ez, 0, cs ex - sn ez)
    pp[2] = cyaw + gx * (double)(-sn * ex - cs * ez) + gz * (double)(cs * ex - sn * ez);
    pp[3] = 2.0 * gx - rgx;  // the camera translation enters twice: p = joints0 + Ry(-yaw)(b - c) + c + c
    pp[4] = gy;             // (2 G - Ry(yaw) G)_y
    pp[5] = 2.0 * gz - rgz;
  }
}

struct ReprojSumArgs {
  int F;
  const double* part;
  float scale_rep, scale_ch;  // w_reprojection / (F J 2), w_chamfer / (F M)
  float* grad;
  float* loss;
  int n_tail;  // trailing parameters that receive no gradient (the detached betas)
};

__global__ __launch_bounds__(256) void k_reproj_sum(ReprojSumArgs a) {
  __shared__ double sh[4][6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int f = tid; f < a.F; f += 256) {
    const double* pp = a.part + (size_t)f * RPJ_PW;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] += pp[k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) sh[wave][k] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = ((sh[0][k] + sh[1][k]) + sh[2][k]) + sh[3][k];
    a.loss[0] = (float)(t[0] * (double)a.scale_rep + t[1] * (double)a.scale_ch);
    a.grad[0] = (float)t[2];
    a.grad[1 + 3 * a.F] = (float)t[3];
    a.grad[2 + 3 * a.F] = (float)t[4];
    a.grad[3 + 3 * a.F] = (float)t[5];
  }
  if (tid < a.n_tail) a.grad[4 + 3 * a.F + tid] = 0.f;
}

extern "C" int uuo_reprojection_num_params(const uuo_reprojection_problem_t* p) { return p ? 3 * p->F + 14 : 0; }

extern "C" int uuo_reprojection_create(const uuo_reprojection_problem_t* p, uuo_reprojection_t** out) {
  UUO_REQUIRE(p && out, "uuo_reprojection_create: null argument");
  UUO_REQUIRE(p->F > 0 && p->M > 0 && p->V > 0, "uuo_reprojection_create: F, M and V must be positive");
  UUO_REQUIRE(p->J > 0 && p->J <= RPJ_NJ, "uuo_reprojection_create: 1..64 joints per frame");
  UUO_REQUIRE(p->d_markers && p->d_joints0 && p->d_verts0 && p->d_kp_target && p->d_mask,
              "uuo_reprojection_create: null device pointer");
  UUO_REQUIRE((long long)p->F * p->V * 3 < 0x7FFFFFFFll, "uuo_reprojection_create: F * V too large");
  uuo_reprojection* h = new uuo_reprojection();
  h->p = *p;
  const size_t part_bytes = ((size_t)p->F * RPJ_PW * sizeof(double) + 255) / 256 * 256;
  if (hipMalloc((void**)&h->part, part_bytes + (size_t)p->F * RPJ_S * p->M * sizeof(unsigned long long)) != hipSuccess) {
    delete h;
    uuo_set_error("uuo_reprojection_create: out of device memory");
    return -12;
  }
  h->keys = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(h->part) + part_bytes);
  *out = h;
  return 0;
}

extern "C" int uuo_reprojection_destroy(uuo_reprojection_t* h) {
  if (!h) return 0;
  if (h->part) (void)hipFree(h->part);
  delete h;
  return 0;
}

int uuo_reprojection_eval_impl(uuo_reprojection* h, hipStream_t s, const float* d_x, float* d_loss, float* d_grad,
                               float* d_kp, int32_t* d_nn_idx) {
  const uuo_reprojection_problem_t& p = h->p;
  ReprojArgs a;
  a.F = p.F; a.M = p.M; a.V = p.V; a.J = p.J;
  a.x = d_x;
  a.markers = p.d_markers; a.joints0 = p.d_joints0; a.verts0 = p.d_verts0; a.kp_target = p.d_kp_target; a.mask = p.d_mask;
  a.fx = p.focal[0]; a.fy = p.focal[1]; a.cx = p.center[0]; a.cy = p.center[1];
  const double n_rep = (double)p.F * p.J * 2.0, n_ch = (double)p.F * p.M;
  a.coef_rep = (float)(2.0 * (double)p.w_reprojection / n_rep);
  a.coef_ch = (float)(2.0 * (double)p.w_chamfer / n_ch);
  a.grad = d_grad; a.part = h->part; a.keys = h->keys; a.kp_out = d_kp; a.nn_idx = d_nn_idx;
  hipLaunchKernelGGL(k_reproj_search, dim3(p.F, RPJ_S), dim3(RPJ_T), 0, s, a);
  hipLaunchKernelGGL(k_reproj_terms, dim3(p.F), dim3(64), 0, s, a);
  ReprojSumArgs r;
  r.F = p.F; r.part = h->part;
  r.scale_rep = (float)((double)p.w_reprojection / n_rep);
  r.scale_ch = (float)((double)p.w_chamfer / n_ch);
  r.grad = d_grad; r.loss = d_loss; r.n_tail = 10;
  hipLaunchKernelGGL(k_reproj_sum, dim3(1), dim3(256), 0, s, r);
  UUO_HIP_CHECK(hipGetLastError());
  return 0;
}

extern "C" int uuo_reprojection_eval(uuo_reprojection_t* h, void* stream, const float* d_x, float* d_loss, float* d_grad,
                                     float* d_kp, int32_t* d_nn_idx) {
  UUO_REQUIRE(h && d_x && d_loss && d_grad, "uuo_reprojection_eval: null argument");
  return uuo_reprojection_eval_impl(h, (hipStream_t)stream, d_x, d_loss, d_grad, d_kp, d_nn_idx);
}
