// probe: do f16 MFMA inputs / f16 conversions keep subnormals on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
__global__ void k(float a, float b, float *out) {
  v8h A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)(_Float16)a; out[2] = (float)(_Float16)b; }
}
int main() {
  float *d, h[3];
  hipMalloc(&d, sizeof(h));
  const float cases[][2] = {{9.5367431640625e-07f, 1.0f}, {1.0f, 9.5367431640625e-07f}, {3.0e-5f, 1.0f}, {1.0f, 3.0e-5f}, {1.0e-4f, 1.0f}, {3.0e-5f, 3.0e-5f}};
  for (auto &c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a=%g b=%g: mfma sum (16 products) = %g expected %g | cvt a -> %g, cvt b -> %g\n", c[0], c[1], h[0], 16.0 * h[1] * h[2], h[1], h[2]);
  }
  return 0;
}
