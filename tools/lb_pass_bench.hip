// Microbenchmark of the two L-BFGS history passes (k_lb_dots, k_lb_direction) at the bench's problem size, against a
// plain streaming read of the same bytes.  Variants are timed back to back over NROT rotating histories (so that neither
// L2 nor the 256-MB Infinity Cache holds the data, as in a fit with twelve solves in flight) and over ONE history (hot).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I uuo_mocap_amd/csrc tools/lb_pass_bench.hip \
//         -o gpurun_out/lb_pass_bench && gpurun_out/lb_pass_bench
#include "../uuo_mocap_amd/csrc/lbfgs_kernels.hip"

#include <cstdio>
#include <vector>

void uuo_set_error(const std::string& msg) { std::printf("error: %s\n", msg.c_str()); }
thread_local UuoRecorder* uuo_recorder = nullptr;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// plain read of the same history bytes with the passes' own layout walk (floor)
__global__ __launch_bounds__(256) void k_floor(const float4* __restrict__ S, const float4* __restrict__ Y, size_t n4,
                                               float* __restrict__ out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + u * stride;
      v[2 * u] = j < n4 ? S[j] : make_float4(0, 0, 0, 0);
      v[2 * u + 1] = j < n4 ? Y[j] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;
}

template <bool ACC32>
__global__ __launch_bounds__(256) void k_dots_v(LbDotsArgs a) {
  lb_dots_body<ACC32>(a.n, a.cap, a.capL, a.head, a.count, a.cand, a.S, a.Y, a.g, a.gp, a.d, a.t, a.ncb, a.gcb, a.part, a.skip_lo,
                      a.skip_hi);
}
template <bool ACC32>
__global__ __launch_bounds__(64 * LB_DQ) void k_dir_v(LbDirArgs a) {
  lb_direction_body<ACC32>(a.n, a.cap, a.capL, a.S, a.Y, a.g, a.st, a.d, a.x, a.t, a.xt, a.map);
}

struct Hist {
  float *S, *Y, *vecs;
  double* part;
  LbDev* st;
};

int main(int argc, char** argv) {
  const int F = 300;
  const int n = argc > 1 ? atoi(argv[1]) : 142 * F + 10;  // compact chamfer packing
  const int hist = 100, cap = hist + 1;
  const int npad = (n + LB_CW - 1) / LB_CW * LB_CW;
  const int ncb = npad / LB_CW;
  const int gcb = (ncb + LB_MAXCHUNK - 1) / LB_MAXCHUNK;
  const int nchunks = (ncb + gcb - 1) / gcb;
  const size_t hist_floats = (size_t)ncb * LB_CBSTRIDE(cap);
  const int NROT = 12;
  std::vector<Hist> H(NROT);
  std::vector<float> hv(hist_floats);
  for (size_t i = 0; i < hist_floats; ++i) hv[i] = 1e-3f * (float)((i * 2654435761u) % 1000) - 0.5f;
  for (int r = 0; r < NROT; ++r) {
    CHECK(hipMalloc(&H[r].S, hist_floats * 4));
    CHECK(hipMalloc(&H[r].Y, hist_floats * 4));
    CHECK(hipMalloc(&H[r].vecs, (size_t)LB_NVEC * npad * 4));
    CHECK(hipMalloc(&H[r].part, (size_t)LB_MAXCHUNK * LB_ROWS * 3 * 8));
    CHECK(hipMalloc(&H[r].st, sizeof(LbDev)));
    CHECK(hipMemcpy(H[r].S, hv.data(), hist_floats * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(H[r].Y, hv.data(), hist_floats * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(H[r].vecs, hv.data(), (size_t)LB_NVEC * npad * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(H[r].st, 0, sizeof(LbDev)));
    const int cnt = hist;
    CHECK(hipMemcpy((char*)H[r].st + offsetof(LbDev, count), &cnt, 4, hipMemcpyHostToDevice));
    std::vector<double> c(LB_MAXH, 1e-3);
    CHECK(hipMemcpy((char*)H[r].st + offsetof(LbDev, cy), c.data(), LB_MAXH * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy((char*)H[r].st + offsetof(LbDev, cs), c.data(), LB_MAXH * 8, hipMemcpyHostToDevice));
  }
  float* out = nullptr;
  CHECK(hipMalloc(&out, 1 << 20));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  UuoIndexMap map;
  std::memset(&map, 0, sizeof(map));
  std::printf("n=%d ncb=%d gcb=%d nchunks=%d history bytes S+Y = %.1f MB\n", n, ncb, gcb, nchunks, 2.0 * hist_floats * 4 / 1e6);
  auto run = [&](const char* name, int variant, int rot) -> int {
    const int iters = 240;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      for (int it = 0; it < iters; ++it) {
        Hist& h = H[rot ? it % NROT : 0];
        float* g = h.vecs + 2 * (size_t)npad;
        float* gp = h.vecs + 3 * (size_t)npad;
        float* d = h.vecs;
        float* x = h.vecs + 4 * (size_t)npad;
        float* xt = h.vecs + 5 * (size_t)npad;
        if (variant == 0) {
          hipLaunchKernelGGL(k_floor, dim3(1024), dim3(256), 0, 0, (const float4*)h.S, (const float4*)h.Y, hist_floats / 4, out);
        } else if (variant == 1 || variant == 3) {
          LbDotsArgs a{{nchunks, LB_DRS}, n, cap, cap, 0, hist - 1, hist - 1, h.S, h.Y, g, gp, d, 0.1f, ncb, gcb, h.part, 0, 0};
          if (variant == 1) hipLaunchKernelGGL(k_dots_v<false>, dim3(nchunks, LB_DRS), dim3(256), 0, 0, a);
          else hipLaunchKernelGGL(k_dots_v<true>, dim3(nchunks, LB_DRS), dim3(256), 0, 0, a);
        } else if (variant == 2 || variant == 4) {
          LbDirArgs a{{2 * ncb, 1}, n, cap, cap, h.S, h.Y, g, h.st, d, x, 0.1f, xt, map};
          if (variant == 2) hipLaunchKernelGGL(k_dir_v<false>, dim3(2 * ncb), dim3(64 * LB_DQ), 0, 0, a);
          else hipLaunchKernelGGL(k_dir_v<true>, dim3(2 * ncb), dim3(64 * LB_DQ), 0, 0, a);
        }
      }
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double us = 1e3 * best / iters;
    std::printf("%-28s %-8s %7.2f us  %5.2f TB/s\n", name, rot ? "rotating" : "hot", us, 2.0 * hist_floats * 4 / us / 1e6);
    return 0;
  };
  for (int rot = 1; rot >= 0; --rot) {
    if (run("floor (plain read)", 0, rot)) return 1;
    if (run("k_lb_dots fp64", 1, rot)) return 1;
    if (run("k_lb_dots fp32-packed", 3, rot)) return 1;
    if (run("k_lb_direction fp64", 2, rot)) return 1;
    if (run("k_lb_direction fp32-packed", 4, rot)) return 1;
  }
  return 0;
}
