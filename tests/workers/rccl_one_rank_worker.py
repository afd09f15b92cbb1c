"""A ONE-rank RCCL process group on a one-GPU box (started fresh by tests/test_hip_shards.py, before anything here touches
the GPU): library load, communicator creation and the device-tensor collectives of the training step --
distributed.allreduce_gradients on the 4.44 MB flat gradient blob (forced past the world-size-1 early return) and
distributed.broadcast_parameters -- executed once on gfx950 before an 8-GPU node ever sees this code.  The reference's DDP
runs on the same library (train.py:84-88, `DDPPlugin`: NCCL = RCCL on ROCm).  Prints one JSON line."""
import datetime
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import configs, distributed, models, synthetic, utils
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev, timeout=datetime.timedelta(seconds=120))
    init_s = time.perf_counter() - t0
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")], ["Config.hip_flat_grads = True"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(dev)
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=4, bias_scale=0.02, sharpen=10.0))
    before = model.nerf_mlp.flat_params().detach().clone()
    distributed.broadcast_parameters(model, src=0, force=True)
    torch.cuda.synchronize()
    assert torch.equal(model.nerf_mlp.flat_params().detach(), before)

    blob = model.nerf_mlp.flat_parameter()
    g = torch.randn(blob.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    blob.grad = g.clone()
    distributed.allreduce_gradients(model, force=True)         # first call: communicator warm-up
    torch.cuda.synchronize()
    assert torch.equal(blob.grad, g), "a one-rank all-reduce (sum, / 1) must return the blob unchanged"
    n = 50
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        distributed.allreduce_gradients(model, force=True)
    ev1.record()
    torch.cuda.synchronize()
    assert torch.equal(blob.grad, g)
    # the per-parameter path (flat = False): 46 gradients flattened, reduced, scattered back
    cfg.hip_flat_grads = False
    model.nerf_mlp.release_flat_parameter()
    for i, p in enumerate(model.parameters()):
        p.grad = torch.full_like(p, float(1 + i % 3))
    distributed.allreduce_gradients(model, force=True)
    torch.cuda.synchronize()
    assert all(torch.equal(p.grad, torch.full_like(p, float(1 + i % 3))) for i, p in enumerate(model.parameters()))
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    assert float(t.sum()) == 4.0
    print(json.dumps({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "init_s": init_s,
                      "allreduce_blob_bytes": int(blob.numel() * 4), "allreduce_ms": ev0.elapsed_time(ev1) / n,
                      "nccl_version": ".".join(map(str, torch.cuda.nccl.version()))}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
