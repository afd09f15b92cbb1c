"""HIP-graph replay of the eval level loop for launch-bound batch sizes.

One `Model.__call__` is two kernel launches plus ~50 small allocations and ctypes calls; at 512 rays per GPU
(the 8-GPU shard of BASELINE config 4) the two kernels take 0.2 ms and the host another 0.25 ms.  Capturing
the call once in a HIP graph (torch.cuda.CUDAGraph: the library launches on the capturing stream) and
replaying it removes the host part: 0.47 -> 0.33 ms per step at 512 rays; at 4096 rays the step is GPU-bound
and nothing changes.  Inference only (no autograd through a replay).
"""
from dataclasses import fields

import torch

from . import utils


class GraphedForward:
    """model(rays, train_frac, compute_extras) for a FIXED ray count, replayed from a HIP graph.

    g = GraphedForward(model, example_rays, train_frac=1.0, compute_extras=True)
    renderings, ray_history = g(rays)      # rays: same shapes as example_rays

    The returned tensors are the graph's static output buffers: consume (or clone) them before the next call.
    Re-create the object after the model's parameters change (the packed weight image is baked into the graph
    by address, and is re-packed in place, so an optimiser step followed by one eager call keeps it valid;
    a resize or a precision switch does not)."""

    def __init__(self, model, example_rays: utils.Rays, train_frac: float = 1.0, compute_extras: bool = True, warmup: int = 3):
        if model.training:
            raise ValueError("GraphedForward replays the inference path: call model.eval() first")
        dev = model.device
        self.model = model
        self.static_rays = utils.Rays(*[torch.as_tensor(getattr(example_rays, f.name)).to(dev).clone() for f in fields(example_rays)])
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                      # pack weights, set kernel attributes, warm the allocator
                model(self.static_rays, train_frac, compute_extras)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.outputs = model(self.static_rays, train_frac, compute_extras)

    def __call__(self, rays: utils.Rays):
        for f in fields(rays):
            dst = getattr(self.static_rays, f.name)
            src = torch.as_tensor(getattr(rays, f.name))
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"GraphedForward was captured for {f.name} of shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.outputs
