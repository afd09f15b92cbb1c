// probe: semantics of __builtin_amdgcn_permlane16_swap on gfx950 (which rows move where)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2uu __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out) {
  unsigned lane = threadIdx.x;
  v2uu r = __builtin_amdgcn_permlane16_swap(lane, 100 + lane, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
}
int main() {
  unsigned *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int j = 0; j < 2; ++j) { printf("r[%d]:", j); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * j + i]); printf("\n"); }
  return 0;
}
