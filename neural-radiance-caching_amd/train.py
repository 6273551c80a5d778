"""First piece of the training path: gradients of the proposal levels' density fields and
their data-parallel reduction.

The reference's train step (internal/train_utils.py:3100-3177) is `jax.value_and_grad(loss_fn)` over the whole
model followed by `jax.lax.pmean(grad, "batch")` across devices and the optimizer.  Here:

  * `density_grads(rc, level, points, d_density, d_feature)` -> {tensor name: gradient} for the hash-grid tables and
    the density MLP of one level (rc_density_backward: hand-written backward on the matrix cores + atomic scatter
    into the tables), named like the reference's parameter tree so an optimizer keyed on those names can consume it;
  * `allreduce_grads(flat_buffers)` -> the pmean: ONE all-reduce per level over the flat gradient buffer
    (torch.distributed; backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).  The buffers are the
    bucket: ~45-180 MB per level, large enough to run the xGMI ring at its per-link bound, no per-tensor launches.

The loss, the shader's backward and the optimizer are not part of this row.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import numpy as np


def grads_as_dict(flat, layout) -> Dict[str, object]:
    """Views of the flat gradient buffer by tensor name (layout: RadianceCache.density_grad_layout(level)[0])."""
    return {name: flat[off: off + int(np.prod(shape))].reshape(shape) for name, off, shape in layout}


def density_grads(rc, level: int, points, d_density, d_feature=None, flat=None):
    """-> ({name: gradient view}, flat buffer, density [n])."""
    layout, _ = rc.density_grad_layout(level)
    flat, density = rc.density_backward(level, points, d_density, d_feature, flat)
    return grads_as_dict(flat, layout), flat, density


def allreduce_grads(buffers: Iterable, average: bool = True, group=None) -> List:
    """jax.lax.pmean(grad, axis_name="batch") (internal/train_utils.py:3133-3135) for per-level flat gradient
    buffers: one in-place all-reduce each, divided by the world size."""
    import torch.distributed as dist

    buffers = list(buffers)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return buffers
    world = dist.get_world_size(group)
    handles = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True) for b in buffers]
    for b, hnd in zip(buffers, handles):
        hnd.wait()
        if average:
            b.div_(world)
    return buffers
