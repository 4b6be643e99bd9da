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
  * the learning rate must be a TENSOR: a Python-float ``lr`` would be baked into the captured kernels' arguments and every replay
    would train at the capture-time rate, whatever a scheduler sets afterwards (the reference steps ``OneCycleLR`` after every
    sample, train.py:127,510-511).  ``GraphedTrainStep`` converts each group's float ``lr`` to a 0-dim device tensor in place;
    torch's schedulers ``fill_`` a tensor ``lr`` (``_update_param_group_val``), so ``scheduler.step()`` between replays works as
    in the eager loop.  ``betas`` / ``eps`` / ``weight_decay`` ARE captured by value: a scheduler that cycles momentum
    (``OneCycleLR(cycle_momentum=True)``; the reference passes ``False``) is refused;
  * BatchNorm's ``num_batches_tracked`` is incremented by the forward kernel.

Weights are updated by the replayed optimizer on the device, which does not bump ``Tensor._version``: the packed-weight cache of
``ops`` is dropped after every replay, so an eager forward of the same model in between sees the new weights.  (The same holds for
code that writes ``p.data`` directly: call ``ops.drop_pack_cache()`` after such an update.)

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
            if not torch.is_tensor(grp["lr"]):
                # a float would be frozen into the graph (see the module docstring); schedulers fill_ a tensor lr in place
                grp["lr"] = torch.tensor(float(grp["lr"]), dtype=torch.float32, device=imgs.device)
            elif grp["lr"].device != imgs.device:
                raise ValueError("GraphedTrainStep: a tensor lr must live on the model's device")
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
        self._betas = [tuple(grp.get("betas", ())) for grp in optimizer.param_groups]
        # samples on the device whose range differs from the captured one are counted here (no host sync per step): check_ranges()
        self.range_mismatches = torch.zeros((), dtype=torch.int64, device=imgs.device)
        # the range is a constant of the CAPTURED step only: the attribute lives on the model while this constructor runs its eager
        # and captured steps and is removed again, so later eager train-mode forwards of the model read their own sample's range
        model.static_depth_range = self.depth_range
        try:
            self._build(model, optimizer, imgs, warmup)
        finally:
            model.static_depth_range = None

    def _build(self, model, optimizer, imgs, warmup):
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
        """Copy a sample into the captured buffers (asynchronous device copies on the current stream).  The depth range of the
        sample must be the captured one (two kernels hold it by value): a host tensor is checked here and refused; a device tensor
        is checked on the device without a sync -- ``check_ranges()`` reads the count."""
        lo, hi = depth_values[:, 0], depth_values[:, -1]
        if not depth_values.is_cuda:
            if not (bool((lo == self.depth_range[0]).all()) and bool((hi == self.depth_range[1]).all())):
                raise ValueError(f"GraphedTrainStep: sample depth range ({float(lo[0])}, {float(hi[0])}) differs from the captured "
                                 f"{self.depth_range}; capture a step per range (datasets/dtu_yao.py uses one)")
        else:
            self.range_mismatches += ((lo != self.depth_range[0]) | (hi != self.depth_range[1])).any().to(torch.int64)
        self.imgs.copy_(imgs, non_blocking=True)
        self.depth_values.copy_(depth_values, non_blocking=True)
        for k in self.proj:
            self.proj[k].copy_(proj_matrices[k], non_blocking=True)
        for k in self.gt:
            self.gt[k].copy_(depth_gt_ms[k], non_blocking=True)
            self.mask[k].copy_(mask_ms[k], non_blocking=True)

    def check_ranges(self):
        """Synchronises; raises if any device-resident sample loaded so far had another depth range than the captured one."""
        n = int(self.range_mismatches.item())
        if n:
            raise ValueError(f"GraphedTrainStep: {n} replayed sample(s) had a depth range other than the captured {self.depth_range}")

    def __call__(self, imgs=None, proj_matrices=None, depth_values=None, depth_gt_ms=None, mask_ms=None):
        for grp, b in zip(self.optimizer.param_groups, self._betas):
            if tuple(grp.get("betas", ())) != b:
                raise ValueError("GraphedTrainStep: betas changed after capture (a momentum-cycling scheduler?): they are captured by value")
        if imgs is not None:
            self.load_sample(imgs, proj_matrices, depth_values, depth_gt_ms, mask_ms)
        self.graph.replay()
        self.replays += 1
        from . import ops
        ops.drop_pack_cache()              # the replay updated the weights on the device without bumping their version counters
        return self.loss
