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


def capture(fn, warmup: int = 3, constant_beam: bool = False) -> CapturedStep:
    """Run `fn` `warmup` times on a side stream (allocations, plans, the space-charge chain's guard settle there), then once more
    under capture. `fn` must be free of host synchronisation and keep the tensors it reads in place.

    While recording, the host-side caches that would skip a launch because "nothing changed since the last call" are off
    (`_ops.CAPTURING`): every kernel that derives something from a setting is part of the graph, so the replay follows in-place
    changes of ANY tensor the step reads. `constant_beam=True` keeps one of them: the memoised moments of the beam that enters
    the step (used by the backward pass of a beam property of a linearly tracked beam) — one particle pass less per replay for
    an optimisation over lattice settings with a fixed incoming beam; the beam must then not be edited between replays."""
    from . import _ops

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    _ops.CAPTURING[0] += 1
    keep = _ops.CAPTURE_KEEPS_BEAM_MOMENTS[0]
    _ops.CAPTURE_KEEPS_BEAM_MOMENTS[0] = bool(constant_beam)
    try:
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outputs = fn()
    finally:
        _ops.CAPTURING[0] -= 1
        _ops.CAPTURE_KEEPS_BEAM_MOMENTS[0] = keep
    return CapturedStep(graph, outputs)
