// Training through the C ABI with no PyTorch in the process: two levels of refnerf_level_forward_train (saved
// layer inputs in caller-owned buffers), the data loss of train_utils.compute_data_loss written out by hand
// (0.1 * mse(level 0) + 1.0 * mse(level 1), lossmult = 1), refnerf_level_backward per level into ONE canonical
// gradient blob.  tests/test_hip_parity.py::test_c_abi_training_without_torch compares loss and gradient with the
// Python host (Model.__call__ autograd nodes) on the same seeded weights / camera / targets.
//
//   hipcc -O2 -I include examples/c_abi_train_demo.cpp -L refnerf-pl_amd/csrc -lrefnerf_hip -o c_abi_train_demo
//   ./c_abi_train_demo weights.f32 camtoworld.f32 <width> <height> <focal> gt_rgb.f32 out_grads.f32
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
  if (argc != 8 && argc != 9) { fprintf(stderr, "usage: %s weights.f32 c2w.f32 W H focal gt.f32 out_grads.f32 [chains 0 = f32 | 3 = split f16]\n", argv[0]); return 1; }
  const int chains = argc == 9 ? atoi(argv[8]) : REFNERF_PREC_F32;   // cfg.precision of both directions: REFNERF_PREC_F32 | REFNERF_PREC_F16X2
  const int W = atoi(argv[3]), H = atoi(argv[4]);
  const float focal = (float)atof(argv[5]);
  const int R = W * H, N = 48;
  std::vector<float> params, c2w, gt;
  if (!read_file(argv[1], params, REFNERF_NUM_PARAMS) || !read_file(argv[2], c2w, 12) || !read_file(argv[6], gt, 3 * (size_t)R)) {
    fprintf(stderr, "cannot read inputs\n");
    return 1;
  }
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
  refnerf_rays rays = {d_o, d_d, d_v, d_rad, d_near, d_far};

  // ---- weights (training runs in the f32 arithmetic mode)
  float *d_params = dmalloc<float>(REFNERF_NUM_PARAMS);
  HIP_OK(hipMemcpy(d_params, params.data(), REFNERF_NUM_PARAMS * sizeof(float), hipMemcpyHostToDevice));
  void *d_packed = nullptr;
  // which weight image a training level of this arithmetic streams: the library's own rule (ABI v11: refnerf_level_image)
  refnerf_level_cfg icfg;
  refnerf_level_cfg_default(&icfg);
  icfg.training = 1;
  icfg.precision = chains;
  const int image = refnerf_level_image(&icfg);
  HIP_OK(hipMalloc(&d_packed, refnerf_packed_weights_bytes(image)));
  RN_OK(refnerf_pack_weights(d_params, d_packed, image, nullptr));

  // ---- training forward of both levels: outputs the loss needs + what the backward reads back
  std::vector<float> sd0(2 * R), w0(R, 1.0f);
  for (int r = 0; r < R; ++r) { sd0[2 * r] = 0.0f; sd0[2 * r + 1] = 1.0f; }
  float *d_sd_in = dmalloc<float>(2 * R), *d_w_in = dmalloc<float>(R);
  HIP_OK(hipMemcpy(d_sd_in, sd0.data(), 2 * R * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_w_in, w0.data(), R * sizeof(float), hipMemcpyHostToDevice));
  const size_t S = (size_t)R * N;
  const size_t act_bytes = refnerf_activation_workspace_bytes(R, N);
  float *d_sd[2], *d_w[2], *d_rgb[2], *d_dens[2], *d_hrgb[2];
  void *d_act[2];
  float *d_dif = dmalloc<float>(3 * R), *d_spc = dmalloc<float>(3 * R), *d_dist = dmalloc<float>(R), *d_acc = dmalloc<float>(R);
  refnerf_level_cfg cfg[2];
  for (int l = 0; l < 2; ++l) {
    d_sd[l] = dmalloc<float>((size_t)R * (N + 1)); d_w[l] = dmalloc<float>(S); d_rgb[l] = dmalloc<float>(3 * R);
    d_dens[l] = dmalloc<float>(S); d_hrgb[l] = dmalloc<float>(3 * S);
    HIP_OK(hipMalloc(&d_act[l], act_bytes));
    refnerf_level_cfg_default(&cfg[l]);
    cfg[l].n_samples = N;
    cfg[l].n_in = l == 0 ? 1 : N;
    cfg[l].training = 1;
    cfg[l].compute_extras = 0;
    cfg[l].precision = chains;          // (the saved-activation format follows from it: refnerf_activations_format below)
    refnerf_level_out out = {};
    out.d_sdist = d_sd[l]; out.d_weights = d_w[l]; out.d_density = d_dens[l]; out.d_rgb = d_hrgb[l];
    out.d_r_rgb = d_rgb[l]; out.d_r_diffuse = d_dif; out.d_r_specular = d_spc; out.d_r_distance = d_dist; out.d_r_acc = d_acc;
    RN_OK(refnerf_level_forward_train(d_packed, &cfg[l], &rays, R, l == 0 ? d_sd_in : d_sd[0], l == 0 ? d_w_in : d_w[0], &out,
                                      d_act[l], act_bytes, nullptr));
  }
  HIP_OK(hipDeviceSynchronize());

  // ---- data loss (train_utils.py:33-88, mse, lossmult = 1) and its gradient w.r.t. the two renderings, on the host
  const float mult[2] = {0.1f, 1.0f};
  double loss = 0.0;
  float *d_g_rgb[2];
  for (int l = 0; l < 2; ++l) {
    std::vector<float> rgb(3 * (size_t)R), g(3 * (size_t)R);
    HIP_OK(hipMemcpy(rgb.data(), d_rgb[l], rgb.size() * sizeof(float), hipMemcpyDeviceToHost));
    double se = 0.0;
    for (size_t i = 0; i < rgb.size(); ++i) {
      const float res = rgb[i] - gt[i];
      se += (double)res * res;
      g[i] = mult[l] * 2.0f * res / (3.0f * R);
    }
    loss += mult[l] * se / (3.0 * R);
    d_g_rgb[l] = dmalloc<float>(3 * (size_t)R);
    HIP_OK(hipMemcpy(d_g_rgb[l], g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice));
  }

  // ---- backward of both levels into one gradient blob (accumulated)
  float *d_grads = dmalloc<float>(REFNERF_NUM_PARAMS);
  HIP_OK(hipMemset(d_grads, 0, REFNERF_NUM_PARAMS * sizeof(float)));
  const size_t ws_bytes = refnerf_backward_workspace_bytes(R, N);
  void *d_ws = nullptr;
  HIP_OK(hipMalloc(&d_ws, ws_bytes));
  for (int l = 1; l >= 0; --l) {
    /* (the format refnerf_level_forward_train wrote for this cfg: fp32 rows, or the split-f16 pair units of REFNERF_PREC_F16X2) */
    refnerf_level_saved saved = {d_sd[l], d_dens[l], d_hrgb[l], d_w[l], d_act[l], refnerf_activations_format(&cfg[l])};
    refnerf_level_grads seeds = {};
    seeds.d_g_r_rgb = d_g_rgb[l];
    RN_OK(refnerf_level_backward(d_packed, &cfg[l], &rays, R, &saved, &seeds, d_grads, d_ws, ws_bytes, nullptr));
  }
  HIP_OK(hipDeviceSynchronize());
  std::vector<float> grads(REFNERF_NUM_PARAMS);
  HIP_OK(hipMemcpy(grads.data(), d_grads, grads.size() * sizeof(float), hipMemcpyDeviceToHost));
  FILE *f = fopen(argv[7], "wb");
  if (!f || fwrite(grads.data(), sizeof(float), grads.size(), f) != grads.size()) { fprintf(stderr, "cannot write output\n"); return 1; }
  fclose(f);
  double n2 = 0.0;
  for (float x : grads) n2 += (double)x * x;
  printf("c_abi_train_demo: %d rays x %d samples x 2 levels, loss %.8f, |grad|^2 %.8e\n", R, N, loss, n2);
  return 0;
}
