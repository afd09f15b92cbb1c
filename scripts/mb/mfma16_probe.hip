// Lane / register map of v_mfma_f32_16x16x32_f16 on gfx950, checked exhaustively against the hypothesis the 3-product split
// kernel would be built on:  A: lane l -> row l % 16, k = 8 (l / 16) + i;  B: lane l -> col l % 16, k = 8 (l / 16) + i;
// D: lane l, reg r -> row 4 (l / 16) + r, col l % 16.  Also times a dependent chain and an independent stream of them.
//   hipcc --offload-arch=gfx950 -O2 scripts/mb/mfma16_probe.hip -o scripts/mb/bin/mfma16_probe && scripts/mb/bin/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(int *bad, float *out) {
  const int lane = threadIdx.x;
  int nbad = 0;
  for (int row = 0; row < 16; ++row)
    for (int k = 0; k < 32; ++k)
      for (int col = 0; col < 16; col += 5) {
        v8h a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
        if (lane == (k / 8) * 16 + row) a[k % 8] = (_Float16)3.0f;
        if (lane == (k / 8) * 16 + col) b[k % 8] = (_Float16)2.0f;
        v4f c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) {
          const float want = (lane == (row / 4) * 16 + col && r == row % 4) ? 6.0f : 0.0f;
          if (c[r] != want) ++nbad;
        }
      }
  atomicAdd(bad, nbad);
  if (lane == 0) out[0] = 1.0f;
}

__global__ void rate(float *out, int iters, int chains) {
  v8h a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x * 3 + i)); }
  v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    if (chains > 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    if (chains > 2) c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    if (chains > 2) c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0); else if (chains > 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 + chains] = (float)(t1 - t0) / (4.0f * iters);
  out[16 + threadIdx.x % 4] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  int *bad; float *out;
  hipMalloc(&bad, 4); hipMalloc(&out, 256); hipMemset(bad, 0, 4); hipMemset(out, 0, 256);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, bad, out);
  for (int ch = 1; ch <= 4; ch *= 2) hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, out, 4096, ch == 4 ? 4 : ch);
  hipLaunchKernelGGL(rate, dim3(1), dim3(128), 0, 0, out + 32, 4096, 2);     // two waves of one workgroup (different SIMDs)
  int hbad; float h[64];
  hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
  printf("mfma_f32_16x16x32_f16 layout hypothesis: %s (%d mismatches)\n", hbad == 0 ? "CONFIRMED" : "WRONG", hbad);
  printf("cycles per MFMA, one wave: 1 dependent chain %.1f, 2 chains %.1f, 4 chains %.1f\n", h[2], h[3], h[5]);
  return hbad != 0;
}
