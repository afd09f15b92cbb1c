// probe: numerics of the N-packed split-f16 product on v_mfma_f32_32x32x16_f16 -- rare outliers?
// D[32 rows][16 samples] = W[32][256] * x[256][16], x = hi + lo (f16 pair), W f16-exact (w_lo = 0) or fp32 (hi + lo)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef unsigned v2uu __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }
// W: [trial][32][256] fp32; X: [trial][16][256] fp32; out: [trial][32][16] totals, out2: hi+lo re-split of relu(total)
__global__ void k(const float *W, const float *X, float *out, float *out2, int split_w) {
  const int trial = blockIdx.x, lane = threadIdx.x, n = lane & 31, h = lane >> 5;
  const float *w = W + (size_t)trial * 32 * 256, *x = X + (size_t)trial * 16 * 256;
  v16f acc = {0};
  for (int t = 0; t < 16; ++t) {
    v8h ah, al, b;
    for (int e = 0; e < 8; ++e) {
      const int kk = 16 * t + 8 * h + e;
      const float wv = w[n * 256 + kk];
      const _Float16 wh = (_Float16)wv;
      ah[e] = wh; al[e] = (_Float16)(wv - (float)wh);
      const float xv = x[(n & 15) * 256 + kk];
      const _Float16 xh = (_Float16)xv;
      b[e] = (n < 16) ? xh : (_Float16)(xv - (float)xh);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b, acc, 0, 0, 0);
    if (split_w) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc, 0, 0, 0);
  }
  for (int j = 0; j < 8; ++j) {
    const float lo = acc[j], hi = acc[j + 8];
    const v2uu r = __builtin_amdgcn_permlane16_swap(f2u(lo), f2u(hi), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    const float s = u2f(r0) + u2f(r1);
    const int jj = (n < 16) ? j : j + 8;
    const int row = (jj & 3) + 8 * (jj >> 2) + 4 * h;
    out[((size_t)trial * 32 + row) * 16 + (n & 15)] = s;
    const float xr = fmaxf(s, 0.0f);
    const _Float16 hh = (_Float16)xr;
    const _Float16 ll = (_Float16)(xr - (float)hh);
    out2[((size_t)trial * 32 + row) * 16 + (n & 15)] = (float)hh + (float)ll;
  }
}
int main(int argc, char **argv) {
  const int T = 4096;
  std::vector<float> W((size_t)T * 32 * 256), X((size_t)T * 16 * 256), O((size_t)T * 32 * 16), O2(O.size());
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX; };
  for (int mode = 0; mode < 2; ++mode) {
    for (auto &v : W) { float f = (rnd() - 0.5f) * 0.25f; v = mode == 0 ? (float)(_Float16)f : f; }
    for (auto &v : X) { float f = rnd(); v = f < 0.4f ? 0.0f : (f - 0.4f) * (f - 0.4f) * 30.0f * rnd(); }
    float *dW, *dX, *dO, *dO2;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dO, O.size() * 4); hipMalloc(&dO2, O.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(T), dim3(64), 0, 0, dW, dX, dO, dO2, mode);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(O2.data(), dO2, O.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst2 = 0; long bad = 0;
    for (int t = 0; t < T; ++t) for (int r = 0; r < 32; ++r) for (int s = 0; s < 16; ++s) {
      double ref = 0, mag = 0;
      for (int kk = 0; kk < 256; ++kk) { double p = (double)W[((size_t)t * 32 + r) * 256 + kk] * X[((size_t)t * 16 + s) * 256 + kk]; ref += p; mag += fabs(p); }
      const double e = fabs(O[((size_t)t * 32 + r) * 16 + s] - ref) / mag;
      const double e2 = fabs(O2[((size_t)t * 32 + r) * 16 + s] - fmax(ref, 0.0)) / mag;
      if (e > worst) worst = e; if (e2 > worst2) worst2 = e2; if (e > 1e-5) ++bad;
    }
    printf("mode %d (%s weights): worst |err| / sum|products| = %.3e (re-split %.3e), outliers > 1e-5: %ld of %ld\n", mode, mode ? "fp32 hi+lo" : "f16-exact", worst, worst2, bad, (long)T * 512);
    hipFree(dW); hipFree(dX); hipFree(dO); hipFree(dO2);
  }
  return 0;
}
