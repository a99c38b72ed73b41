// What a plain streaming read of a history-sized buffer achieves on this GPU: the yardstick for k_lb_dots / k_lb_direction
// (two passes over ~52 MB per L-BFGS iteration).  Every variant reads `bytes` once with 16-byte loads and reduces to one
// float per block; between launches a 512-MB buffer is read so that neither L2 nor the 256-MB MALL holds the data.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_bw.hip -o gpurun_out/stream_bw && gpurun_out/stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n4, float* __restrict__ out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n4; i += stride) {
    const float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;  // keeps the loads alive
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t big = 512ull << 20;
  float4 *buf = nullptr, *flush = nullptr;
  float* out = nullptr;
  CHECK(hipMalloc(&buf, big));
  CHECK(hipMalloc(&flush, big));
  CHECK(hipMalloc(&out, 1 << 20));
  CHECK(hipMemset(buf, 0, big));
  CHECK(hipMemset(flush, 0, big));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t sizes[] = {52ull << 20, 105ull << 20, 400ull << 20};
  const int grids[] = {256, 512, 1024, 2048, 4096};
  for (size_t bytes : sizes)
    for (int cold = 1; cold >= 0; --cold)
      for (int g : grids) {
        float best[3] = {1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 6; ++rep)
          for (int v = 0; v < 3; ++v) {
            if (cold) hipLaunchKernelGGL(k_read<4>, dim3(2048), dim3(256), 0, 0, flush, big / 16, out);
            CHECK(hipEventRecord(e0, 0));
            if (v == 0) hipLaunchKernelGGL(k_read<2>, dim3(g), dim3(256), 0, 0, buf, bytes / 16, out);
            if (v == 1) hipLaunchKernelGGL(k_read<4>, dim3(g), dim3(256), 0, 0, buf, bytes / 16, out);
            if (v == 2) hipLaunchKernelGGL(k_read<8>, dim3(g), dim3(256), 0, 0, buf, bytes / 16, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best[v]) best[v] = ms;
          }
        printf("%4zu MB %s grid %5d: loads in flight 2/4/8 per lane -> %6.1f / %6.1f / %6.1f us = %.2f / %.2f / %.2f TB/s\n", bytes >> 20,
               cold ? "cold" : "warm", g, 1e3 * best[0], 1e3 * best[1], 1e3 * best[2], bytes / best[0] / 1e9, bytes / best[1] / 1e9,
               bytes / best[2] / 1e9);
      }
  return 0;
}
