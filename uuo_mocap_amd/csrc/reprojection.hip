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
// CONSTANT cloud W -- no skinning at all inside the closure.  One block per frame does the search (every lane keeps the
// best of its vertex slice for 32 markers in registers, packed 64-bit minima across the block), the projection, both
// gradients and the frame's partial sums; a one-block kernel adds the partials in double.  Two launches per evaluation.
#include "uuo_common.h"

#define RPJ_T 512    // threads of a frame block (8 waves: 64 registers of running minima per lane need the room)
#define RPJ_MC 32    // markers per register pass
#define RPJ_NJ 64    // joints a frame may have (SMPL + the extra vertex joints: 45)
#define RPJ_PW 8     // floats of a frame's partial sums: sum of squared key-point residuals (masked), sum of squared
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
  float* part;
  float* kp_out;
  int32_t* nn_idx;
};

__device__ __forceinline__ unsigned long long rpj_wave_min(unsigned long long k) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(k, off, 64);
    k = o < k ? o : k;
  }
  return k;
}
__device__ __forceinline__ float rpj_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(RPJ_T) void k_reproj_frame(ReprojArgs a) {
  __shared__ float4 sU[RPJ_MC];                             // query points of the pass (w unused)
  __shared__ unsigned long long sK[RPJ_T / 64][RPJ_MC];     // per-wave minima
  __shared__ float sAcc[8];                                 // chamfer part of the frame: loss, d yaw, d b xyz
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int F = a.F, M = a.M, V = a.V;
  const float yaw = a.x[0];
  const float sn = sinf(yaw), cs = cosf(yaw);
  const float bx = a.x[1 + 3 * f], by = a.x[2 + 3 * f], bz = a.x[3 + 3 * f];
  const float* j0p = a.joints0 + (size_t)f * a.J * 3;
  const float j0x = j0p[0], j0y = j0p[1], j0z = j0p[2];
  const float* vf = a.verts0 + (size_t)f * V * 3;
  if (tid < 8) sAcc[tid] = 0.f;

  // ---- chamfer term: nearest vertex of every marker, 32 markers per pass over the frame's vertices
  for (int m0 = 0; m0 < M; m0 += RPJ_MC) {
    const int mc = min(RPJ_MC, M - m0);
    __syncthreads();  // (previous pass has consumed sU / sK)
    if (tid < mc) {
      const float* mp = a.markers + ((size_t)f * M + m0 + tid) * 3;
      // C^T (m - j0) - b : mocap axes -> HMR axes (x, -z, y), then into the body's un-yawed frame
      const float qx = (mp[0] - j0x) - bx, qy = -(mp[2] - j0z) - by, qz = (mp[1] - j0y) - bz;
      sU[tid] = make_float4(cs * qx - sn * qz, qy, sn * qx + cs * qz, 0.f);
    }
    __syncthreads();
    float best[RPJ_MC];
    int bidx[RPJ_MC];
#pragma unroll
    for (int k = 0; k < RPJ_MC; ++k) { best[k] = __builtin_inff(); bidx[k] = 0x7FFFFFFF; }
    for (int i = tid; i < V; i += RPJ_T) {
      const float wx = vf[3 * i] - j0x, wy = vf[3 * i + 1] - j0y, wz = vf[3 * i + 2] - j0z;
#pragma unroll
      for (int k = 0; k < RPJ_MC; ++k) {
        if (k < mc) {  // block-uniform
          const float4 u = sU[k];
          const float dx = u.x - wx, dy = u.y - wy, dz = u.z - wz;
          const float d = dx * dx + dy * dy + dz * dz;
          if (d < best[k]) { best[k] = d; bidx[k] = i; }  // a lane visits its vertices in rising order: first minimum kept
        }
      }
    }
#pragma unroll
    for (int k = 0; k < RPJ_MC; ++k) {
      if (k < mc) {
        const unsigned long long key =
            rpj_wave_min(((unsigned long long)__float_as_uint(best[k]) << 32) | (unsigned)bidx[k]);
        if (lane == 0) sK[wave][k] = key;
      }
    }
    __syncthreads();
    if (wave == 0) {
      float l = 0.f, gy_ = 0.f, gbx = 0.f, gby = 0.f, gbz = 0.f;
      if (lane < mc) {
        unsigned long long key = sK[0][lane];
#pragma unroll
        for (int w = 1; w < RPJ_T / 64; ++w) { const unsigned long long o = sK[w][lane]; key = o < key ? o : key; }
        const int i = (int)(unsigned)(key & 0xFFFFFFFFull);
        if (i < V) {  // (a NaN marker has no minimum: it contributes nothing, where the reference would propagate the NaN)
          if (a.nn_idx) a.nn_idx[(size_t)f * M + m0 + lane] = i;
          const float wx = vf[3 * i] - j0x, wy = vf[3 * i + 1] - j0y, wz = vf[3 * i + 2] - j0z;
          const float* mp = a.markers + ((size_t)f * M + m0 + lane) * 3;
          // d = Ry(yaw) W + b - C^T (m - j0): the residual vertex - marker in HMR axes
          const float rx = cs * wx + sn * wz, rz = -sn * wx + cs * wz;
          const float dx = rx + bx - (mp[0] - j0x), dy = wy + by + (mp[2] - j0z), dz = rz + bz - (mp[1] - j0y);
          l = dx * dx + dy * dy + dz * dz;
          gbx = a.coef_ch * dx; gby = a.coef_ch * dy; gbz = a.coef_ch * dz;
          gy_ = gbx * rz - gbz * rx;  // d . (d Ry / d yaw) W,  (d Ry / d yaw) W = (rz, 0, -rx)
        }
      }
      l = rpj_wave_sum(l); gy_ = rpj_wave_sum(gy_); gbx = rpj_wave_sum(gbx); gby = rpj_wave_sum(gby); gbz = rpj_wave_sum(gbz);
      if (lane == 0) { sAcc[0] += l; sAcc[1] += gy_; sAcc[2] += gbx; sAcc[3] += gby; sAcc[4] += gbz; }
    }
  }
  __syncthreads();

  // ---- key-point term: the frame's joints through the pinhole camera (wave 0)
  if (wave == 0) {
    const float ccx = a.x[1 + 3 * F], ccy = a.x[2 + 3 * F], ccz = a.x[3 + 3 * F];
    const float ex = bx - ccx, ey = by - ccy, ez = bz - ccz;
    // inv_t = Ry(-yaw) (b - c) + c
    const float tx = (cs * ex - sn * ez) + ccx, ty = ey + ccy, tz = (sn * ex + cs * ez) + ccz;
    const float mk = a.mask[f];
    float l = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
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
      l = (r0 * r0 + r1 * r1) * mk;
      gx = a.coef_rep * mk * r0 * a.fx / pz;
      gy = a.coef_rep * mk * r1 * a.fy / pz;
      gz = -(gx * px + gy * py) / pz;
    }
    l = rpj_wave_sum(l); gx = rpj_wave_sum(gx); gy = rpj_wave_sum(gy); gz = rpj_wave_sum(gz);
    if (lane == 0) {
      const float rgx = cs * gx + sn * gz, rgz = -sn * gx + cs * gz;  // Ry(yaw) G = (d inv_t / d b)^T G
      a.grad[1 + 3 * f] = sAcc[2] + rgx;
      a.grad[2 + 3 * f] = sAcc[3] + gy;
      a.grad[3 + 3 * f] = sAcc[4] + rgz;
      float* pp = a.part + (size_t)f * RPJ_PW;
      pp[0] = l;
      pp[1] = sAcc[0];
      // d inv_t / d yaw = (d Ry(-yaw) / d yaw) (b - c) = (-sn ex - cs ez, 0, cs ex - sn ez)
      pp[2] = sAcc[1] + gx * (-sn * ex - cs * ez) + gz * (cs * ex - sn * ez);
      pp[3] = 2.f * gx - rgx;  // the camera translation enters twice: p = joints0 + Ry(-yaw)(b - c) + c + c
      pp[4] = gy;             // (2 G - Ry(yaw) G)_y
      pp[5] = 2.f * gz - rgz;
    }
  }
}

struct ReprojSumArgs {
  int F;
  const float* part;
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
    const float* pp = a.part + (size_t)f * RPJ_PW;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] += (double)pp[k];
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
  if (hipMalloc((void**)&h->part, (size_t)p->F * RPJ_PW * sizeof(float)) != hipSuccess) {
    delete h;
    uuo_set_error("uuo_reprojection_create: out of device memory");
    return -12;
  }
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
  a.grad = d_grad; a.part = h->part; a.kp_out = d_kp; a.nn_idx = d_nn_idx;
  hipLaunchKernelGGL(k_reproj_frame, dim3(p.F), dim3(RPJ_T), 0, s, a);
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
