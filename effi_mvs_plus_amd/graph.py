"""HIP-graph replay of the hot path (``Effi_MVS_plus.forward_hot``).

One reference view is ~100 kernel launches; enqueued from Python they cost about as much host time as the GPU needs.
``HotPathGraph`` captures the whole pass once (``torch.cuda.CUDAGraph``: a hipGraph; linear when ``ops.get_branches()`` is off,
the default, otherwise with both streams' dependencies) for a fixed input geometry and replays it with one launch -- a linear
graph replays back to back with no gaps between kernels.  Inputs live in static device
buffers, ``slots`` sets of them (double buffering: the producer of the next view's features fills one slot while the graph
runs on the other; each slot has its own captured graph).  ``graph(features, cnet, proj, depth_values, slot=0)`` copies the
caller's tensors into the slot (device to device) and replays; a producer that writes straight into ``graph.inputs[slot]``
(or ``graph.load(slot, ...)`` ahead of time) calls ``graph.replay(slot)`` and skips the copy.  The returned tensors are the
slot's static outputs -- valid until that slot's next replay (clone what must outlive it).  Replay is bitwise identical to
the eager pass (tools/graph_replay_check.py, tests/test_gpu_model.py).
"""
from __future__ import annotations

import torch


def _flatten(obj, out):
    if isinstance(obj, torch.Tensor):
        out.append(obj)
    elif isinstance(obj, dict):
        for k in sorted(obj):
            _flatten(obj[k], out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _flatten(v, out)
    return out


def _clone_tree(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().clone(memory_format=torch.preserve_format)
    if isinstance(obj, dict):
        return {k: _clone_tree(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_clone_tree(v) for v in obj)
    return obj


class ReplayGraph:
    """hipGraph replay of ``fn(*inputs)`` for a fixed input geometry (see the module docstring); ``fn`` must be free of
    host synchronisation and of data-dependent host control flow (``forward_hot`` and ``forward`` are)."""

    def __init__(self, fn, example, warmup=2, slots=1):
        self.fn = fn
        example = tuple(example)
        self.inputs = [_clone_tree(example) for _ in range(slots)]
        self._flat_in = [_flatten(inp, []) for inp in self.inputs]
        self._sig = [(tuple(t.shape), t.dtype) for t in self._flat_in[0]]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):                     # packs weights, creates the zero page, warms the allocator
                fn(*self.inputs[0])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graphs, self.outputs = [], []
        for inp in self.inputs:
            g = torch.cuda.CUDAGraph()
            # thread_local: CUDA/HIP calls of other threads (e.g. the RCCL watchdog of torch.distributed) do not
            # invalidate the capture
            with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = fn(*inp)
            self.graphs.append(g)
            self.outputs.append(out)

    def load(self, slot, *inputs):
        """Device-to-device copy of one view's inputs into a slot's static buffers."""
        flat = _flatten(tuple(inputs), [])
        if len(flat) != len(self._sig):
            raise ValueError("ReplayGraph: input structure differs from the captured one")
        for src, (shape, dtype) in zip(flat, self._sig):
            if tuple(src.shape) != shape or src.dtype != dtype:
                raise ValueError(f"ReplayGraph: input {tuple(src.shape)}/{src.dtype} differs from the captured {shape}/{dtype}")
        torch._foreach_copy_(self._flat_in[slot], flat)

    def replay(self, slot=0):
        self.graphs[slot].replay()
        return self.outputs[slot]

    def __call__(self, *inputs, slot=0):
        self.load(slot, *inputs)
        return self.replay(slot)


class HotPathGraph(ReplayGraph):
    """``Effi_MVS_plus.forward_hot(features, cnet_depth, proj_matrices, depth_values)`` as a replayable graph."""

    def __init__(self, net, features, cnet_depth, proj_matrices, depth_values, warmup=2, slots=1):
        super().__init__(net.forward_hot, (features, cnet_depth, proj_matrices, depth_values), warmup=warmup, slots=slots)


class ForwardGraph(ReplayGraph):
    """The whole ``Effi_MVS_plus.forward(imgs, proj_matrices, depth_values)`` (feature / context pyramids + hot path) as a
    replayable graph -- what the reference's drivers time per view (test_dtu_dypcd.py:437-442)."""

    def __init__(self, net, imgs, proj_matrices, depth_values, warmup=2, slots=1):
        super().__init__(net.forward, (imgs, proj_matrices, depth_values), warmup=warmup, slots=slots)
