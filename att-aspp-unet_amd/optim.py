"""clip_grad_norm_ + AdamW over the engine's flat buffers (pipeline:302, :322-324).

Three launches per step regardless of the number of parameter tensors: squared global
norm, AdamW with the clip coefficient folded in (the step is skipped on non-finite
gradients, which is what GradScaler.step does in the reference), step counter.
The class is a ``torch.optim.Optimizer`` so the reference's LR schedulers
(LinearLR -> CosineAnnealingLR, pipeline:303-306) drive ``param_groups[0]['lr']`` unchanged.
"""
from __future__ import annotations

import torch

from . import _abi, ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4, max_grad_norm=1.0):
        self.model = model
        super().__init__(list(model.parameters()),
                         dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm))

    def _store(self):
        st = self.model.engine.store
        if st is None:
            raise _abi.AauError("FusedAdamW.step() before the first forward/backward on the device")
        if st.m is None:
            st.m = torch.zeros_like(st.flat)
            st.v = torch.zeros_like(st.flat)
        return st

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the engine's flat buffer, which every backward zeroes itself
        if set_to_none:
            for p in self.model.parameters():
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None, inv_scale: float = 1.0):
        st = self._store()
        g = self.param_groups[0]
        ops.grad_sqnorm(st.gflat, st.total, inv_scale, st.norm_ws)
        ops.adamw_step(st.flat, st.m, st.v, st.gflat, st.total, st.norm_ws, st.step_dev, float(g["lr"]),
                       g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], g["max_grad_norm"], inv_scale)
        return None

    def grad_norm(self) -> torch.Tensor:
        """Pre-clip global L2 norm of the last step (device tensor, no sync)."""
        return self.model.engine.store.norm_ws[0].sqrt()
