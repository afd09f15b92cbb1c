import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import refnerf_pl_amd
from refnerf_pl_amd import configs, models, synthetic, utils, camera_utils, _hip
dev = torch.device("cuda", 0)
configs.clear_config()
configs.parse_config_files_and_bindings(["configs/refnerf_blender.gin"], ["Config.hip_precision = 'bf16'"])
cfg = configs.Config()
model = models.construct_model(None, cfg).to(dev).eval()
model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
c2w, focal = synthetic.blender_camera(seed=1)
def sync(): torch.cuda.synchronize()
with torch.no_grad():
    for rep in range(3):
        sync(); t0 = time.perf_counter()
        img = camera_utils.cast_pinhole_rays(c2w.astype(np.float32), 800, 800, focal, 2.0, 6.0, device=dev)
        sync(); t1 = time.perf_counter()
        rendering = models.render_image(lambda r: model(r, 1.0, True), img, cfg, verbose=False, device=dev)
        sync(); t2 = time.perf_counter()
        # the chunk loop alone
        rays = img.reshape(640000, -1)
        _hip.set_timing(True)
        for idx0 in range(0, 640000, cfg.render_chunk_size):
            model(rays[idx0:idx0 + cfg.render_chunk_size], 1.0, True)
        sync(); t3 = time.perf_counter()
        k_ms, launches = _hip.get_timing(); _hip.set_timing(False)
        print(f"cast {1e3*(t1-t0):.1f} ms  render_image {1e3*(t2-t1):.1f} ms  bare loop {1e3*(t3-t2):.1f} ms  kernels {k_ms:.1f} ms / {launches}")
