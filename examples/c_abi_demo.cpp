// Minimal C/C++ host for the C ABI (no PyTorch anywhere): casts a block of pinhole rays on the device,
// packs a weight blob, runs two sampling levels of the Ref-NeRF path in the requested arithmetic mode and
// prints a checksum of the rendered colours.  Used by tests/test_hip_parity.py::test_c_abi_without_torch,
// which feeds it the same seeded weights / camera as the Python host and compares the numbers.
//
//   hipcc -O2 -I include examples/c_abi_demo.cpp -L refnerf-pl_amd/csrc -lrefnerf_hip -o c_abi_demo
//   ./c_abi_demo weights.f32 camtoworld.f32 <width> <height> <focal> <precision 0|1> out_rgb.f32
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "refnerf_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define RN_OK(x) do { int r_ = (x); if (r_ != REFNERF_OK) { fprintf(stderr, "refnerf: %d %s\n", r_, refnerf_last_error()); return 3; } } while (0)

static bool read_file(const char *path, std::vector<float> &v, size_t n) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  v.resize(n);
  size_t got = fread(v.data(), sizeof(float), n, f);
  fclose(f);
  return got == n;
}

template <typename T> static T *dmalloc(size_t n) { void *p = nullptr; return hipMalloc(&p, n * sizeof(T)) == hipSuccess ? (T *)p : nullptr; }

int main(int argc, char **argv) {
  if (argc != 8) { fprintf(stderr, "usage: %s weights.f32 c2w.f32 W H focal precision out.f32\n", argv[0]); return 1; }
  const int W = atoi(argv[3]), H = atoi(argv[4]), prec = atoi(argv[6]);
  const float focal = (float)atof(argv[5]);
  const int R = W * H, N = 64;
  std::vector<float> params, c2w;
  if (!read_file(argv[1], params, REFNERF_NUM_PARAMS) || !read_file(argv[2], c2w, 12)) { fprintf(stderr, "cannot read inputs\n"); return 1; }
  RN_OK(refnerf_device_ok());

  // ---- rays on the device (camera_utils.pixels_to_rays)
  std::vector<int> px(R), py(R);
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { px[y * W + x] = x; py[y * W + x] = y; }
  const float p2c[9] = {1.0f / focal, 0.0f, -0.5f * W / focal, 0.0f, 1.0f / focal, -0.5f * H / focal, 0.0f, 0.0f, 1.0f};
  int *d_px = dmalloc<int>(R), *d_py = dmalloc<int>(R);
  float *d_p2c = dmalloc<float>(9), *d_c2w = dmalloc<float>(12);
  float *d_o = dmalloc<float>(3 * R), *d_d = dmalloc<float>(3 * R), *d_v = dmalloc<float>(3 * R), *d_rad = dmalloc<float>(R);
  float *d_near = dmalloc<float>(R), *d_far = dmalloc<float>(R);
  HIP_OK(hipMemcpy(d_px, px.data(), R * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_py, py.data(), R * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_p2c, p2c, sizeof(p2c), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_c2w, c2w.data(), 12 * sizeof(float), hipMemcpyHostToDevice));
  RN_OK(refnerf_pixels_to_rays(d_px, d_py, d_p2c, 0, d_c2w, 0, nullptr, R, d_o, d_d, d_v, d_rad, nullptr, nullptr));
  std::vector<float> nearv(R, 2.0f), farv(R, 6.0f);
  HIP_OK(hipMemcpy(d_near, nearv.data(), R * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_far, farv.data(), R * sizeof(float), hipMemcpyHostToDevice));

  // ---- weights
  float *d_params = dmalloc<float>(REFNERF_NUM_PARAMS);
  HIP_OK(hipMemcpy(d_params, params.data(), REFNERF_NUM_PARAMS * sizeof(float), hipMemcpyHostToDevice));
  void *d_packed = nullptr;
  HIP_OK(hipMalloc(&d_packed, refnerf_packed_weights_bytes(prec)));
  RN_OK(refnerf_pack_weights(d_params, d_packed, prec, nullptr));

  // ---- two levels (models.py:162-306)
  refnerf_rays rays = {d_o, d_d, d_v, d_rad, d_near, d_far};
  std::vector<float> sd0(2 * R), w0(R, 1.0f);
  for (int r = 0; r < R; ++r) { sd0[2 * r] = 0.0f; sd0[2 * r + 1] = 1.0f; }
  float *d_sd_in = dmalloc<float>(2 * R), *d_w_in = dmalloc<float>(R);
  HIP_OK(hipMemcpy(d_sd_in, sd0.data(), 2 * R * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_w_in, w0.data(), R * sizeof(float), hipMemcpyHostToDevice));
  float *d_sd[2], *d_w[2], *d_rgb[2], *d_dif = dmalloc<float>(3 * R), *d_spc = dmalloc<float>(3 * R), *d_dist = dmalloc<float>(R), *d_acc = dmalloc<float>(R);
  for (int l = 0; l < 2; ++l) { d_sd[l] = dmalloc<float>((size_t)R * (N + 1)); d_w[l] = dmalloc<float>((size_t)R * N); d_rgb[l] = dmalloc<float>(3 * R); }
  for (int l = 0; l < 2; ++l) {
    refnerf_level_cfg cfg;
    refnerf_level_cfg_default(&cfg);
    cfg.n_samples = N;
    cfg.n_in = l == 0 ? 1 : N;
    cfg.compute_extras = 0;
    cfg.precision = prec;
    refnerf_level_out out = {};
    out.d_sdist = d_sd[l]; out.d_weights = d_w[l];
    out.d_r_rgb = d_rgb[l]; out.d_r_diffuse = d_dif; out.d_r_specular = d_spc; out.d_r_distance = d_dist; out.d_r_acc = d_acc;
    RN_OK(refnerf_level_forward(d_packed, &cfg, &rays, R, l == 0 ? d_sd_in : d_sd[0], l == 0 ? d_w_in : d_w[0], &out, nullptr));
  }
  HIP_OK(hipDeviceSynchronize());
  std::vector<float> rgb(3 * R);
  HIP_OK(hipMemcpy(rgb.data(), d_rgb[1], 3 * R * sizeof(float), hipMemcpyDeviceToHost));
  FILE *f = fopen(argv[7], "wb");
  if (!f || fwrite(rgb.data(), sizeof(float), rgb.size(), f) != rgb.size()) { fprintf(stderr, "cannot write output\n"); return 1; }
  fclose(f);
  double s = 0.0;
  for (float x : rgb) s += x;
  printf("c_abi_demo: %d rays, precision %d, mean rgb %.6f\n", R, prec, s / rgb.size());
  return 0;
}
