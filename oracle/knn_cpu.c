/* CPU restatement of pytorch3d's K=1 nearest-neighbour loop.  TEST INFRASTRUCTURE (oracle/__init__.py).
 *
 * Restates the published pytorch3d 0.7.x CPU op (pytorch3d/csrc/knn/knn_cpu.cpp, package not
 * vendored in the reference: install.sh:5-7), which the reference reaches through
 * pytorch3d.loss.chamfer_distance at losses/chamfer_distance.py:15-20:
 *
 *     for each query i1:  for each candidate i2:
 *         dist = 0;  for d in 0..D-1:  diff = p1[d] - p2[d];  dist += diff * diff;
 *         keep (dist, i2) if dist < best            -> first index wins on ties
 *
 * Build with -ffp-contract=off so `dist += diff*diff` is a separately rounded multiply and add
 * (pytorch3d wheels are built for baseline x86-64 without FMA).
 */
#include <stdint.h>
#include <math.h>

void knn1_cpu(const float* p1, const float* p2, int64_t N, int64_t P1, int64_t P2, int64_t D,
              float* dists, int64_t* idx) {
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t i1 = 0; i1 < P1; ++i1) {
      const float* a = p1 + (n * P1 + i1) * D;
      float best = INFINITY;
      int64_t besti = 0;
      for (int64_t i2 = 0; i2 < P2; ++i2) {
        const float* b = p2 + (n * P2 + i2) * D;
        float dist = 0.0f;
        for (int64_t d = 0; d < D; ++d) {
          float diff = a[d] - b[d];
          dist += diff * diff;
        }
        if (dist < best) {
          best = dist;
          besti = i2;
        }
      }
      dists[n * P1 + i1] = best;
      idx[n * P1 + i1] = besti;
    }
  }
}

/* numpy semantics of optimization.py:486,598: D_f[m,v] = sqrt((dx*dx+dy*dy)+dz*dz),
 * mean over valid frames = sequential-in-f fp32 accumulate then divide by count, argmin first index. */
void assign_mean_argmin_cpu(const float* verts, const float* markers, const uint8_t* valid, int64_t F,
                            int64_t M, int64_t V, int64_t* out_idx, float* out_mean) {
  int64_t count = 0;
  for (int64_t f = 0; f < F; ++f) count += valid[f] ? 1 : 0;
  for (int64_t m = 0; m < M; ++m) {
    float best = INFINITY;
    int64_t besti = 0;
    for (int64_t v = 0; v < V; ++v) {
      float acc = 0.0f;
      for (int64_t f = 0; f < F; ++f) {
        if (!valid[f]) continue;
        const float* x = markers + (f * M + m) * 3;
        const float* y = verts + (f * V + v) * 3;
        float dx = y[0] - x[0], dy = y[1] - x[1], dz = y[2] - x[2];
        float s = (dx * dx + dy * dy) + dz * dz;
        acc += sqrtf(s);
      }
      float mean = acc / (float)count;
      if (out_mean) out_mean[m * V + v] = mean;
      if (mean < best) { best = mean; besti = v; }
    }
    out_idx[m] = besti;
  }
}
