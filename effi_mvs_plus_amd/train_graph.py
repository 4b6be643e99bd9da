"""A whole training step -- forward in train mode -> loss -> backward -> optimizer step, the body of the reference's
``train_sample`` (train.py:229-263) -- captured ONCE into a HIP graph and replayed per sample.

Why: the eager step issues ~3,500 kernel launches (one per operator, forward and backward); their kernels add up to ~32 ms at the DTU
training shape (640x512, 5 views) while the host needs 45-55 ms to issue them, depending on the box.  A replayed graph has no
per-launch host work: the step takes what its kernels take.

What makes the step capturable:
  * every operator of the path launches on torch's current stream through the C ABI and never synchronises (include/effi_mvs_hip.h);
  * the depth range (``depth_values[:, 0]``, ``[:, -1]``) enters two kernels by value: it is fixed at capture time
    (``model.static_depth_range``).  DTU training uses one range for every sample (datasets/dtu_yao.py: 425 mm + 192 x 2.5 mm);
  * the loss is ``mvs_loss_static`` (mean over the valid pixels as sum x mask / count: the reference's boolean indexing has a
    data-dependent shape and synchronises);
  * the optimizer must be constructed with ``capturable=True`` (its step counter lives on the device);
  * BatchNorm's ``num_batches_tracked`` is incremented by the forward kernel.

The weight-gradient arena and the packed-weight cache of ``autograd`` / ``ops`` know about capture (a block / entry made outside the
graph is not reused inside it).
"""
from __future__ import annotations

import copy
from typing import Callable, Dict, Sequence

import torch

from .models.module import mvs_loss_static  # noqa: F401  (re-exported: the loss a graphed step uses)

DLOSS = (1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4)            # train.py:246: stage of each of the 13 depth maps


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, imgs, proj_matrices, depth_values, depth_gt_ms, mask_ms)`` captures the step on
    the given sample (its tensors fix shapes and the depth range); ``loss = step(imgs, proj_matrices, depth_values, depth_gt_ms,
    mask_ms)`` copies a new sample into the captured buffers and replays.  Returns the loss tensor of the captured graph (valid
    until the next call).  ``warmup`` eager steps run first on a side stream (allocator and lazy initialisation; PyTorch's recipe
    for whole-network capture) -- model and optimizer state are restored afterwards, so the first replay is step 1."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, imgs: torch.Tensor, proj_matrices: Dict[str, torch.Tensor],
                 depth_values: torch.Tensor, depth_gt_ms: Dict[str, torch.Tensor], mask_ms: Dict[str, torch.Tensor],
                 dloss: Sequence[int] = DLOSS, loss_fn: Callable = mvs_loss_static, warmup: int = 2):
        for grp in optimizer.param_groups:
            if not grp.get("capturable", False):
                raise ValueError("GraphedTrainStep: construct the optimizer with capturable=True")
        if not model.training:
            raise ValueError("GraphedTrainStep: model.train() first")
        self.model, self.optimizer, self.dloss, self.loss_fn = model, optimizer, tuple(dloss), loss_fn
        clone = lambda t: t.detach().clone()                      # noqa: E731
        self.imgs, self.depth_values = clone(imgs), clone(depth_values)
        self.proj = {k: clone(v) for k, v in proj_matrices.items()}
        self.gt = {k: clone(v) for k, v in depth_gt_ms.items()}
        self.mask = {k: clone(v) for k, v in mask_ms.items()}
        lo, hi = depth_values[:, 0], depth_values[:, -1]
        if not (bool((lo == lo[0]).all()) and bool((hi == hi[0]).all())):
            raise NotImplementedError("GraphedTrainStep: the samples of a batch must share their depth range")
        self.depth_range = (float(lo[0]), float(hi[0]))
        model.static_depth_range = self.depth_range
        # warm-up on a side stream, then restore the state it changed (weights, BatchNorm buffers, optimizer moments)
        model_state = copy.deepcopy(model.state_dict())
        opt_state = copy.deepcopy(optimizer.state_dict())
        side = torch.cuda.Stream(device=imgs.device)
        side.wait_stream(torch.cuda.current_stream(imgs.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._step()
        torch.cuda.current_stream(imgs.device).wait_stream(side)
        torch.cuda.synchronize(imgs.device)
        with torch.no_grad():
            model.load_state_dict(model_state)                    # copies in place: parameter addresses stay
        optimizer.load_state_dict(opt_state)
        if not optimizer.state_dict()["state"]:
            # a fresh optimizer creates its state lazily in step(): that must not happen inside the capture with different
            # addresses per replay -- so take one eager step to create it, then zero it
            self._step()
            torch.cuda.synchronize(imgs.device)
            with torch.no_grad():
                model.load_state_dict(model_state)
                for st in optimizer.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        self._drop_caches()
        with torch.cuda.graph(self.graph):
            self.loss = self._step()
        self._drop_caches()                                       # nothing outside the graph keeps pointing into its memory pool
        self.replays = 0

    @staticmethod
    def _drop_caches():
        from . import autograd, ops
        ops._PACK_CACHE.clear()
        autograd.grad_arena.blocks.clear()

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(self.imgs, self.proj, self.depth_values)
        loss, _ = self.loss_fn(out["depth"], self.gt, self.mask, self.dloss)
        loss.backward()
        self.optimizer.step()
        return loss

    def load_sample(self, imgs, proj_matrices, depth_values, depth_gt_ms, mask_ms):
        """Copy a sample into the captured buffers (asynchronous device copies on the current stream)."""
        self.imgs.copy_(imgs, non_blocking=True)
        self.depth_values.copy_(depth_values, non_blocking=True)
        for k in self.proj:
            self.proj[k].copy_(proj_matrices[k], non_blocking=True)
        for k in self.gt:
            self.gt[k].copy_(depth_gt_ms[k], non_blocking=True)
            self.mask[k].copy_(mask_ms[k], non_blocking=True)

    def __call__(self, imgs=None, proj_matrices=None, depth_values=None, depth_gt_ms=None, mask_ms=None):
        if imgs is not None:
            self.load_sample(imgs, proj_matrices, depth_values, depth_gt_ms, mask_ms)
        self.graph.replay()
        self.replays += 1
        return self.loss
