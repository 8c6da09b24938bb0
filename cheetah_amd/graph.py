"""Replaying a tracking step from a device graph (not part of the reference's API: an MI355X-side convenience).

A small-beam step — assign settings, `Segment.track`, read a screen or a beam property, possibly `backward()` — costs a few tens
of microseconds on the GPU and a few hundred in Python. libchx's launches are ordinary launches on the current stream, its
workspaces come from torch's allocator and nothing on the tracking path synchronises, so such a step can be captured ONCE into a
`torch.cuda.CUDAGraph` (hipGraph on ROCm) and replayed with one launch. Everything the kernels read is read through device
pointers at replay time: write new settings IN PLACE (`quad.k1.copy_(value)`, an optimiser's update of a Parameter) and the
replay follows them; assigning a NEW tensor to a setting changes the lattice and needs a new capture.

    step = cheetah_amd.graph.capture(lambda: (segment.track(beam), screen.reading)[1])
    quad.k1.copy_(new_k1)          # in place
    image = step()                 # replays; `image` is the same (static) tensor every time

With gradients: capture BEFORE the trainable Parameters have been through a backward pass on the default stream (their
AccumulateGrad node would carry that stream into the capture, which the HIP runtime does not survive), and read `param.grad`
after each replay (the static gradient tensor of the capture)."""
from __future__ import annotations

import torch


class CapturedStep:
    """Result of `capture`: call it to replay; `outputs` is what the captured function returned (static tensors)."""

    def __init__(self, graph: torch.cuda.CUDAGraph, outputs):
        self.graph = graph
        self.outputs = outputs

    def __call__(self):
        self.graph.replay()
        return self.outputs


def capture(fn, warmup: int = 3) -> CapturedStep:
    """Run `fn` `warmup` times on a side stream (allocations, plans, memoised moments, the space-charge chain's guard settle
    there), then once more under capture. `fn` must be free of host synchronisation and keep the tensors it reads in place."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(int(warmup), 1)):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outputs = fn()
    return CapturedStep(graph, outputs)
