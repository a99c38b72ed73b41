// Model tables: host-side re-layout for gfx950 and upload.  Replaces smplx.create(...) as used by
// SmplInference.__init__ (reference src/video_mocap/utils/smpl.py:22-27).
#include <cmath>
#include <cstring>
#include <vector>

#include "uuo_common.h"

static thread_local std::string g_last_error;
void uuo_set_error(const std::string& msg) { g_last_error = msg; }
thread_local UuoRecorder* uuo_recorder = nullptr;  // non-null while a lock-step batch records its launches
extern "C" const char* uuo_last_error(void) { return g_last_error.c_str(); }
extern "C" int uuo_abi_version(void) { return 3; }

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
  UUO_HIP_CHECK(hipMalloc((void**)dst, src.size() * sizeof(T)));
  UUO_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int uuo_model_create(const float* vt, const float* S, const float* P, const float* Jreg, const float* W,
                                const int64_t* parents, const int64_t* extra, int V, uuo_model_t** out) {
  UUO_REQUIRE(vt && S && P && Jreg && W && parents && extra && out, "uuo_model_create: null argument");
  UUO_REQUIRE(V > 0 && V < (1 << 24), "uuo_model_create: bad vertex count");
  uuo_model* m = new uuo_model();
  m->V = V;
  m->VP = ((V + 127) / 128) * 128;
  const int VP = m->VP;

  // augmented blend basis in MFMA-operand order.  v_mfma_f32_16x16x4_f32 takes B[k = 4s + (l>>4)][j = l&15] from lane l;
  // a lane's values for 4 consecutive K-steps s are stored contiguously, and the 64 lanes of a (coord, unit, group)
  // back to back, so one global_load_dwordx4 per wave fetches 1 KB = 4 K-steps of B with perfect coalescing.
  //   P3[c][u][g][l][t] = Baug[k = 4*(4g+t) + (l>>4)][v = 16u + (l&15)][c],  Baug rows 0..206 posedirs, 207..216 shapedirs
  const int nunits = VP / 16, ngroups = UUO_KP / 16;
  std::vector<float> P3((size_t)3 * nunits * ngroups * 256, 0.f), vt3((size_t)3 * VP, 0.f);
  auto baug = [&](int k, int v, int c) -> float {
    if (v >= V) v = V - 1;  // padding vertices duplicate the last one: their positions never widen a unit's box
    if (k < UUO_NUM_POSE_FEATS) return P[(size_t)k * V * 3 + v * 3 + c];
    if (k < UUO_NUM_POSE_FEATS + UUO_NUM_BETAS) return S[((size_t)v * 3 + c) * 10 + (k - UUO_NUM_POSE_FEATS)];
    return 0.f;
  };
  for (int c = 0; c < 3; ++c)
    for (int u = 0; u < nunits; ++u)
      for (int g = 0; g < ngroups; ++g)
        for (int l = 0; l < 64; ++l)
          for (int t = 0; t < 4; ++t)
            P3[((((size_t)c * nunits + u) * ngroups + g) * 64 + l) * 4 + t] =
                baug(4 * (4 * g + t) + (l >> 4), 16 * u + (l & 15), c);
  for (int v = 0; v < VP; ++v)
    for (int c = 0; c < 3; ++c) vt3[(size_t)c * VP + v] = vt[(v < V ? v : V - 1) * 3 + c];

  // the same basis for v_mfma_f32_16x16x32_f16 (k_skin3): every entry times a power of two that brings the largest one into
  // [128, 256), split into hi = fp16(x) and lo = fp16(x - hi) (22 significant bits, the scaling is exact); a lane's 8 K-slots of
  // step s are its 4 + 4 values of groups 2 s and 2 s + 1 above, so the A operand is the same regrouping of pfaT (k_pose_prep)
  std::vector<_Float16> P16((size_t)3 * nunits * ngroups * 512);
  {
    float amax = 0.f;
    for (float x : P3) amax = std::fmax(amax, std::fabs(x));
    int e = 0;
    if (amax > 0.f && std::isfinite(amax)) {
      (void)std::frexp(amax, &e);  // amax = f * 2^e, f in [0.5, 1)  ->  amax * 2^(8 - e) in [128, 256)
      e = 8 - e;
    }
    const float bscale = std::ldexp(1.0f, e);
    m->skin16_inv = 1.0f / (UUO_SK16_ASCALE * bscale);
    for (int c = 0; c < 3; ++c)
      for (int u = 0; u < nunits; ++u)
        for (int st = 0; st < ngroups / 2; ++st)
          for (int l = 0; l < 64; ++l)
            for (int t = 0; t < 8; ++t) {
              const float x = bscale * P3[((((size_t)c * nunits + u) * ngroups + 2 * st + (t >> 2)) * 64 + l) * 4 + (t & 3)];
              const _Float16 hi = (_Float16)x;
              const _Float16 lo = (_Float16)(x - (float)hi);
              const size_t base = ((((size_t)c * nunits + u) * (ngroups / 2) + st) * 2) * 512 + (size_t)l * 8 + t;
              P16[base] = hi;
              P16[base + 512] = lo;
            }
  }

  // per-vertex transposed posedirs rows PT[v][c][k] (contiguous 3*208 floats per vertex) for gather-LBS / backward
  std::vector<float> PT((size_t)V * 3 * UUO_KB, 0.f);
  for (int k = 0; k < UUO_NUM_POSE_FEATS; ++k)
    for (int v = 0; v < V; ++v)
      for (int c = 0; c < 3; ++c) PT[((size_t)v * 3 + c) * UUO_KB + k] = P[(size_t)k * V * 3 + v * 3 + c];

  // sparse skin weights, joints ascending (same summation order as the dense j = 0..23 chain)
  std::vector<int> Wi((size_t)VP * 4, 0);
  std::vector<float> Ww((size_t)VP * 4, 0.f);
  int max_nnz = 0;
  for (int v = 0; v < V; ++v) {
    int n = 0;
    for (int j = 0; j < UUO_NUM_JOINTS; ++j) {
      float w = W[(size_t)v * UUO_NUM_JOINTS + j];
      if (w != 0.f) {
        if (n < 4) {
          Wi[(size_t)v * 4 + n] = j;
          Ww[(size_t)v * 4 + n] = w;
        }
        ++n;
      }
    }
    if (n > max_nnz) max_nnz = n;
  }
  for (int v = V; v < VP; ++v)
    for (int n = 0; n < 4; ++n) {
      Wi[(size_t)v * 4 + n] = Wi[(size_t)(V - 1) * 4 + n];
      Ww[(size_t)v * 4 + n] = Ww[(size_t)(V - 1) * 4 + n];
    }
  if (max_nnz > 4) {
    // SMPL's skinning weights have at most four non-zero entries per vertex and every kernel of the fitted path is built on
    // that (sparse joint lists in registers); a dense-weight model is refused here rather than falling off a cliff later
    delete m;
    uuo_set_error("uuo_model_create: a vertex has " + std::to_string(max_nnz) +
                  " non-zero skinning weights; the SMPL family has at most 4 and nothing else is supported");
    return -22;
  }
  m->nnz = max_nnz;

  // kinematic tree + hoisted joint tables (J = Jt + JS.beta, with Jt = Jreg.v_template, JS = Jreg.shapedirs)
  UuoTree& t = m->h_tree;
  std::memset(&t, 0, sizeof(t));
  for (int j = 0; j < UUO_NUM_JOINTS; ++j) t.parent[j] = (j == 0) ? -1 : (int)parents[j];
  t.max_depth = 0;
  for (int j = 0; j < UUO_NUM_JOINTS; ++j) {
    if (j > 0 && (t.parent[j] < 0 || t.parent[j] >= j)) {
      delete m;
      uuo_set_error("uuo_model_create: parents must satisfy 0 <= parents[j] < j");
      return -22;
    }
    t.depth[j] = (j == 0) ? 0 : t.depth[t.parent[j]] + 1;
    if (t.depth[j] > t.max_depth) t.max_depth = t.depth[j];
    if (j > 0) {
      int p = t.parent[j];
      if (t.nchild[p] >= 4 || t.depth[j] >= UUO_MAX_DEPTH) {
        delete m;
        uuo_set_error("uuo_model_create: kinematic tree wider/deeper than supported (4 children, depth 9)");
        return -22;
      }
      t.child[p][t.nchild[p]++] = j;
    }
  }
  for (int j = 0; j < UUO_NUM_JOINTS; ++j) {  // joints by depth, ascending joint id inside a level
    const int d = t.depth[j];
    if (t.level_n[d] >= UUO_LEVEL_W) {
      delete m;
      uuo_set_error("uuo_model_create: more than 5 joints on one level of the kinematic tree (SMPL has at most 5)");
      return -22;
    }
    t.level_p[d][t.level_n[d]] = t.parent[j];
    t.level_j[d][t.level_n[d]++] = j;
  }
  for (int j = 0; j < UUO_NUM_JOINTS; ++j)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int v = 0; v < V; ++v) acc += (double)Jreg[(size_t)j * V + v] * (double)vt[v * 3 + c];
      t.Jt[j][c] = (float)acc;
      for (int l = 0; l < 10; ++l) {
        double a2 = 0.0;
        for (int v = 0; v < V; ++v) a2 += (double)Jreg[(size_t)j * V + v] * (double)S[((size_t)v * 3 + c) * 10 + l];
        t.JS[j][c][l] = (float)a2;
      }
    }
  for (int e = 0; e < UUO_NUM_EXTRA_JOINTS; ++e) {
    if (extra[e] < 0 || extra[e] >= V) {
      delete m;
      uuo_set_error("uuo_model_create: extra joint vertex id out of range");
      return -22;
    }
    t.extra_vids[e] = (int)extra[e];
  }

  // dense backward: the basis as the B operand of the transposed contraction (k_dpf), and the joints' vertex lists (k_dA)
  std::vector<float> PB((size_t)nunits * ngroups * 3 * 256, 0.f);
  for (int u = 0; u < nunits; ++u)
    for (int jt = 0; jt < ngroups; ++jt)
      for (int g = 0; g < 3; ++g)
        for (int l = 0; l < 64; ++l)
          for (int tt = 0; tt < 4; ++tt) {
            const int vc = 16 * g + 4 * tt + (l >> 4), v = 16 * u + vc / 3, c = vc % 3;
            PB[((((size_t)u * ngroups + jt) * 3 + g) * 64 + l) * 4 + tt] = (v < V) ? baug(16 * jt + (l & 15), v, c) : 0.f;
          }
  std::vector<int> JLoff(UUO_NUM_JOINTS + 1, 0), JLv;
  std::vector<float> JLw;
  for (int j = 0; j < UUO_NUM_JOINTS; ++j) {
    JLoff[j] = (int)JLv.size();
    for (int v = 0; v < V; ++v)
      for (int n = 0; n < 4; ++n)
        if (Wi[(size_t)v * 4 + n] == j && Ww[(size_t)v * 4 + n] != 0.f) {
          JLv.push_back(v);
          JLw.push_back(Ww[(size_t)v * 4 + n]);
        }
  }
  JLoff[UUO_NUM_JOINTS] = (int)JLv.size();
  if (JLv.empty()) {  // (a model without any skin weight: keep the uploads non-empty)
    JLv.push_back(0);
    JLw.push_back(0.f);
  }

  int rc = 0;
  rc |= upload(&m->PB, PB);
  rc |= upload(&m->JLoff, JLoff);
  rc |= upload(&m->JLv, JLv);
  rc |= upload(&m->JLw, JLw);
  rc |= upload(&m->P3, P3);
  {
    _Float16* d16 = nullptr;
    rc |= upload(&d16, P16);
    m->P16 = d16;
  }
  rc |= upload(&m->vt3, vt3);
  rc |= upload(&m->PT, PT);
  std::vector<float> ST(S, S + (size_t)V * 30), vtv(vt, vt + (size_t)V * 3);
  rc |= upload(&m->ST, ST);
  rc |= upload(&m->vt, vtv);
  rc |= upload(&m->Wi, Wi);
  rc |= upload(&m->Ww, Ww);
  if (rc == 0) {
    if (hipMalloc((void**)&m->tree, sizeof(UuoTree)) != hipSuccess ||
        hipMemcpy(m->tree, &t, sizeof(UuoTree), hipMemcpyHostToDevice) != hipSuccess) {
      uuo_set_error("uuo_model_create: tree upload failed");
      rc = -5;
    }
  }
  if (rc != 0) {
    uuo_model_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}

extern "C" int uuo_model_destroy(uuo_model_t* m) {
  if (!m) return 0;
  void* ptrs[] = {m->P3, m->P16, m->vt3, m->PT, m->ST, m->vt, m->Wi, m->Ww, m->tree, m->PB, m->JLoff, m->JLv, m->JLw};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& kv : m->bwd) {
    if (kv.second.pfaT) (void)hipFree(kv.second.pfaT);
    if (kv.second.A) (void)hipFree(kv.second.A);
    if (kv.second.frames) (void)hipFree(kv.second.frames);
    if (kv.second.gcopy) (void)hipFree(kv.second.gcopy);
    uuo_dense_ws_destroy(kv.second.ws);
  }
  for (auto& kv : m->fwd) {
    if (kv.second.pfaT) (void)hipFree(kv.second.pfaT);
    if (kv.second.A) (void)hipFree(kv.second.A);
    if (kv.second.jp) (void)hipFree(kv.second.jp);
  }
  delete m;
  return 0;
}

extern "C" int uuo_model_num_verts(const uuo_model_t* m) { return m ? m->V : -22; }
