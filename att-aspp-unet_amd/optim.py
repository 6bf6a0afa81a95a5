"""clip_grad_norm_ + AdamW over the engine's flat buffers (pipeline:302, :322-324).

Three launches per step regardless of the number of parameter tensors: squared global
norm, AdamW with the clip coefficient folded in (the step is skipped on non-finite
gradients, which is what GradScaler.step does in the reference), step counter.
The class is a ``torch.optim.Optimizer`` so the reference's LR schedulers
(LinearLR -> CosineAnnealingLR, pipeline:303-306) drive ``param_groups[i]['lr']`` unchanged.

The learning rate and weight decay of every parameter group live in DEVICE memory
(``hyp`` [G, 2]); the kernel reads them at run time, so a step captured into a hipGraph
follows the schedule: ``refresh_hyper()`` uploads the current ``param_groups`` values
whenever they changed (a 16-byte host-to-device copy on the stream, issued OUTSIDE of
capture).  Several groups (test_ablation.py:576-586: attention parameters at the full rate,
backbone at half) are resolved per 64-element block of the flat buffer by a byte table.
"""
from __future__ import annotations

import torch

from . import _abi, ops
from .engine import ALIGN


class FusedAdamW(torch.optim.Optimizer):
    """``FusedAdamW(model, lr=...)`` or, with groups, ``FusedAdamW(model, groups=[{"params": [...], "lr": ...}, ...])``
    (every parameter of the model must appear in exactly one group; betas / eps / max_grad_norm are global)."""

    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4, max_grad_norm=1.0, groups=None):
        self.model = model
        params = list(model.parameters()) if groups is None else groups
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      max_grad_norm=max_grad_norm))
        if groups is not None:
            have = {id(p) for g in self.param_groups for p in g["params"]}
            missing = [n for n, p in model.named_parameters() if id(p) not in have]
            if missing:
                raise _abi.AauError(f"FusedAdamW: parameters in no group: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        if len(self.param_groups) > 256:
            raise _abi.AauError("FusedAdamW: at most 256 parameter groups")
        self._hyp_dev = self._gob = None
        self._hyp_last = None
        self._store_id = None

    def _store(self):
        st = self.model.engine.store
        if st is None:
            raise _abi.AauError("FusedAdamW.step() before the first forward/backward on the device")
        if st.m is None:
            st.m = torch.zeros_like(st.flat)
            st.v = torch.zeros_like(st.flat)
        if self._store_id != id(st):
            G = len(self.param_groups)
            self._hyp_dev = torch.zeros(G, 2, dtype=torch.float32, device=st.device)
            self._hyp_last = None
            self._gob = None
            if G > 1:
                gid = {id(p): k for k, g in enumerate(self.param_groups) for p in g["params"]}
                tab = torch.zeros((st.total + ALIGN - 1) // ALIGN, dtype=torch.uint8)
                for name, p in zip(st.names, st.params):
                    b = st.offs[name] // ALIGN
                    tab[b:b + (p.numel() + ALIGN - 1) // ALIGN] = gid[id(p)]
                self._gob = tab.to(st.device)
            self._store_id = id(st)
        return st

    def refresh_hyper(self):
        """Upload (lr, weight_decay) of every group if they changed since the last upload.  Called by ``step`` when the
        stream is not capturing; a graphed step calls it before each replay."""
        self._store()
        cur = tuple((float(g["lr"]), float(g["weight_decay"])) for g in self.param_groups)
        if cur != self._hyp_last:
            # a fresh pageable source per upload: the copy is staged at call time, so a later change of the schedule can
            # never overwrite the values of an upload the stream has not executed yet (a reused pinned buffer can)
            self._hyp_dev.copy_(torch.tensor(cur, dtype=torch.float32).view(len(cur), 2))
            self._hyp_last = cur

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the engine's flat buffer, which every backward zeroes itself
        if set_to_none:
            for p in self.model.parameters():
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None, inv_scale: float = 1.0):
        st = self._store()
        if not torch.cuda.is_current_stream_capturing():
            self.refresh_hyper()
        elif self._hyp_last is None:
            raise _abi.AauError("FusedAdamW: refresh_hyper() must run once before the step is captured")
        g = self.param_groups[0]
        ops.grad_sqnorm(st.gflat, st.total, inv_scale, st.norm_ws)
        ops.adamw_step_dev(st.flat, st.m, st.v, st.gflat, st.total, st.norm_ws, st.step_dev, self._hyp_dev, self._gob,
                           len(self.param_groups), g["betas"][0], g["betas"][1], g["eps"], g["max_grad_norm"], inv_scale)
        return None

    def grad_norm(self) -> torch.Tensor:
        """Pre-clip global L2 norm of the last step (device tensor, no sync)."""
        return self.model.engine.store.norm_ws[0].sqrt()
