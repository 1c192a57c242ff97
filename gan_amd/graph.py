"""Recording a whole training step into one HIP graph.

A step of the reference's models is hundreds of launches (per-layer kernels, normalisation, elementwise glue, the
optimizer); at small per-GPU batches (Pix2Pix: 8 images) the host cannot issue them as fast as the GPU retires them.
GraphedStep runs the step a few times on a side stream, records it once and afterwards replays the recording with one
host call per step.

What the step function may do: anything that only enqueues GPU work on tensors whose storage outlives the recording
(parameters, optimizer state, preallocated inputs that the caller refreshes with copy_()).  What it may not do: read
values back to the host (.item(), prints), rebind a buffer to a new tensor (write into it instead), branch on data, or
draw host-side random numbers that should differ between steps (device-side generators are recorded correctly).

A hazard of this torch / ROCm pair, outside this library: a multi-block torch reduction (sum / mean over some 10^5 elements
or more) recorded in a graph returns wrong values in replays when the same reduction is also launched eagerly between the
replays (tools/probe/sum_graph_probe.py shows it on a bare x.sum()).  The library's own backward formulas do not use torch
reductions (bias gradients come from ops.channel_sum); a step function's loss reductions are affected only if the caller
mixes eager reductions of the same shape between replays -- keep evaluation / logging code in a replay of its own or run
it before the first replay.
"""
import torch

from . import ops


class GraphedStep:
    def __init__(self, step, warmup=2, keep_weight_images=True):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # Packed weight images and bf16 operand copies made during the warm-up must not be found by the recording: their
        # producer kernels would be missing from the graph and replays would keep reading the values of this moment.
        # Exception: a step whose optimizers refresh every cached image right after they move the weights
        # (Adam.repack_after_step; the step says so): the images of the warm-up are current when the recording starts and the
        # recorded refresh launches keep them current in every replay, so the per-layer pack launches stay out of the graph.
        if not (keep_weight_images and getattr(step, "weight_images_refreshed_in_step", False)):
            ops.clear_weight_cache()
        else:
            ops.weight_images_settled()      # (synchronised above: no wait nodes for the warm-up's packs inside the graph)
        ops.clear_cast_cache()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), ops.capturing_on():
            self.out = step()
        ops.clear_cast_cache()

    def __call__(self):
        """Replays the step; returns the tensors the recorded call returned (same storage every time)."""
        self.graph.replay()
        return self.out
