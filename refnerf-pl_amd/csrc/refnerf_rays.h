/*
 * refnerf_rays.h -- on-device ray generation: the step in front of the path
 * (camera_utils.pixels_to_rays, internal/camera_utils.py:502-614, perspective
 * cameras without lens distortion; optional NDC conversion,
 * camera_utils.convert_to_ndc :31-97).  One thread per pixel; 8 B in
 * (pixel coordinates), 56 B out per ray -- HBM-bound, trivially so.
 */
#pragma once
#include <hip/hip_runtime.h>

namespace rn {

struct RayGenArgs {
  const int *pix_x, *pix_y;
  const float *pixtocams;   /* [3,3] shared (stride 0) or per ray (stride 9) */
  const float *camtoworlds; /* [3,4] shared (stride 0) or per ray (stride 12) */
  const float *pixtocam_ndc; /* [3,3] or NULL */
  int p2c_stride, c2w_stride;
  int n;
  float *origins, *directions, *viewdirs, *radii, *imageplane;
};

__device__ __forceinline__ void mat3_vec(const float *M, int ld, const float v[3], float out[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = (M[i * ld] * v[0] + M[i * ld + 1] * v[1]) + M[i * ld + 2] * v[2];
}

/* camera_utils.convert_to_ndc for one ray (near = 1) */
__device__ __forceinline__ void to_ndc(const float o_in[3], const float d[3], const float *p2c, float o_ndc[3], float d_ndc[3]) {
  const float t = -(1.0f + o_in[2]) / d[2];
  float o[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = o_in[i] + t * d[i];
  const float xm = 1.0f / p2c[2], ym = 1.0f / p2c[5];
  o_ndc[0] = xm * o[0] / o[2]; o_ndc[1] = ym * o[1] / o[2]; o_ndc[2] = -1.0f;
  const float inf[3] = {xm * d[0] / d[2], ym * d[1] / d[2], 1.0f};
#pragma unroll
  for (int i = 0; i < 3; ++i) d_ndc[i] = inf[i] - o_ndc[i];
}

__global__ void pixels_to_rays_kernel(const RayGenArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n) return;
  const float *p2c = A.pixtocams + (size_t)i * A.p2c_stride;
  const float *c2w = A.camtoworlds + (size_t)i * A.c2w_stride;
  const float x = (float)A.pix_x[i], y = (float)A.pix_y[i];
  float dir[3][3], cam0[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {                       /* the pixel and its +x / +y neighbours */
    const float pd[3] = {x + (q == 1 ? 1.0f : 0.0f) + 0.5f, y + (q == 2 ? 1.0f : 0.0f) + 0.5f, 1.0f};
    float cam[3];
    mat3_vec(p2c, 3, pd, cam);
    cam[1] = -cam[1]; cam[2] = -cam[2];               /* OpenCV -> OpenGL */
    if (q == 0) { cam0[0] = cam[0]; cam0[1] = cam[1]; cam0[2] = cam[2]; }
    mat3_vec(c2w, 4, cam, dir[q]);
  }
  float o[3] = {c2w[3], c2w[7], c2w[11]};
  const float nrm = sqrtf((dir[0][0] * dir[0][0] + dir[0][1] * dir[0][1]) + dir[0][2] * dir[0][2]);
  float dxn, dyn, d_out[3];
  if (A.pixtocam_ndc == nullptr) {
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = dir[1][c] - dir[0][c], b = dir[2][c] - dir[0][c];
      s1 += a * a; s2 += b * b;
      d_out[c] = dir[0][c];
    }
    dxn = sqrtf(s1); dyn = sqrtf(s2);
  } else {
    float ox[3], oy[3], on[3], tmp[3];
    to_ndc(o, dir[1], A.pixtocam_ndc, ox, tmp);
    to_ndc(o, dir[2], A.pixtocam_ndc, oy, tmp);
    to_ndc(o, dir[0], A.pixtocam_ndc, on, d_out);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = ox[c] - on[c], b = oy[c] - on[c];
      s1 += a * a; s2 += b * b;
      o[c] = on[c];
    }
    dxn = sqrtf(s1); dyn = sqrtf(s2);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    A.origins[(size_t)i * 3 + c] = o[c];
    A.directions[(size_t)i * 3 + c] = d_out[c];
    A.viewdirs[(size_t)i * 3 + c] = dir[0][c] / nrm;   /* from the world-space direction, before NDC */
  }
  A.radii[i] = (0.5f * (dxn + dyn)) * 2.0f / sqrtf(12.0f);
  if (A.imageplane) { A.imageplane[(size_t)i * 2] = cam0[0]; A.imageplane[(size_t)i * 2 + 1] = cam0[1]; }
}

}  // namespace rn
