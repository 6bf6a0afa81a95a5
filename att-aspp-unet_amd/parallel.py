"""Single-node data parallelism: one process per GPU, gradient all-reduce on RCCL over xGMI
overlapped with the engine's backward pass.

The reference has no distributed code; the semantics here are
those of DistributedDataParallel over the reference step: every rank runs the reference
step on its own 8 frames (per-GPU BatchNorm statistics, like eight bs=8 replicas), the
gradients are averaged, and every rank applies the identical clip + AdamW update.

The flat gradient buffer is laid out in registration order (d1..d4, bridge, u4..u1,
out_conv) and the backward pass finishes it from the back, so buckets are contiguous
slices that become final at known points of the recorded backward list ("marks").  Each
mark enqueues one all-reduce (async: RCCL runs it on its own stream behind an event on
the compute stream) while the remaining backward kernels keep the CUs busy.  xGMI is
point-to-point: a few large buckets (here 5: <=46 MB) beat many small ones; the last one, which no
backward kernel can hide, is kept small (d1..d3: 2.6 MB of the 85 MB).  The 1/world
factor is folded into the optimiser's unscale factor, so no extra pass touches the grads.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist

# mark fired by the engine after the backward of a block -> parameter-name prefixes whose grads are then final
BUCKET_PLAN = (
    ("u3", ("u3.", "u2.", "u1.", "out_conv.")),
    ("u4", ("u4.att.", "u4.conv.")),
    # u4.up's weight gradient is computed inside the bridge's grouped launch (engine.py: pre_wg), so it belongs to this bucket
    ("bridge", ("bridge.", "u4.up.")),
    ("d4", ("d4.",)),                      # 8 MB at base_c 48: reduced under the backward of d3..d1
    ("d1", ("d1.", "d2.", "d3.")),         # the only bucket nothing can hide: 2.6 MB
)


def bucket_ranges(names: Sequence[str], offs: dict, numels: dict, total: int, align: int = 64):
    """-> {mark: (begin, end)} contiguous element ranges of the flat buffer, covering it exactly once."""
    out = {}
    for mark, prefixes in BUCKET_PLAN:
        sel = [n for n in names if n.startswith(prefixes)]
        if not sel:
            continue
        b = min(offs[n] for n in sel)
        e = max(offs[n] + (numels[n] + align - 1) // align * align for n in sel)
        out[mark] = (b, min(e, total))
    # sanity: disjoint and covering
    spans = sorted(out.values())
    assert spans[0][0] == 0 and spans[-1][1] == total, spans
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0, spans
    return out


class GradBucketReducer:
    """Launches one async all-reduce per bucket when its mark fires; ``finish()`` makes the current
    stream wait for all of them.  Works on any flat tensor / backend (RCCL on GPU, gloo on CPU)."""

    def __init__(self, flat: torch.Tensor, ranges: dict, group=None):
        self.flat, self.ranges, self.group = flat, ranges, group
        self.works = []
        self.fired = []

    def on_mark(self, mark: str):
        r = self.ranges.get(mark)
        if r is None:
            return
        self.fired.append(mark)
        self.works.append(dist.all_reduce(self.flat[r[0]:r[1]], op=dist.ReduceOp.SUM, group=self.group,
                                          async_op=True))

    def finish(self):
        for w in self.works:
            w.wait()
        missing = set(self.ranges) - set(self.fired)
        self.works, self.fired = [], []
        if missing:
            raise RuntimeError(f"gradient buckets never reduced: {sorted(missing)}")


class DataParallel:
    """Wraps an AttentionASPPUNet for DP training.  Usage:
        dp = DataParallel(model); ... loss.backward() (or TrainStep) ...; dp.finish(); opt.step(inv_scale=dp.inv_scale)
    """

    def __init__(self, model, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model, self.group = model, group
        self.world = dist.get_world_size(group)
        self.inv_scale = 1.0 / self.world
        self.reducer = None
        self.pending = None     # gradient kept over an accumulating backward, added back by finish()
        model.engine.bucket_cb = self._on_mark
        model.engine.dp = self
        # identical starting weights on every rank
        st = model.engine.store
        tensors = [st.flat] if st is not None else [p.data for p in model.parameters()]
        for t in tensors + [b.data for b in model.buffers()]:
            dist.broadcast(t, src=0, group=group)

    def _ensure(self):
        st = self.model.engine.store
        if self.reducer is None or self.reducer.flat is not st.gflat:
            numels = {n: p.numel() for n, p in zip(st.names, st.params)}
            self.reducer = GradBucketReducer(st.gflat, bucket_ranges(st.names, st.offs, numels, st.total), self.group)
        return self.reducer

    def _on_mark(self, mark):
        self._ensure().on_mark(mark)

    def before_backward(self, keep):
        """Called by the autograd node before a backward.  ``keep``: the flat gradient of earlier forward / backward pairs
        (``.grad`` not cleared: torch's accumulation semantics) or None.  The backward rewrites the flat buffer and the
        bucket all-reduces work on it in place and asynchronously, so ``keep`` -- which is ALREADY reduced -- is only
        added once they are done, in ``finish()``; adding it on the compute stream right after the backward could
        run before or during a bucket's all-reduce and be summed world-size times or torn."""
        red = self._ensure()
        if red.works or self.pending is not None:
            raise RuntimeError("DataParallel.finish() must be called after every backward (the previous backward's "
                               "bucket all-reduces are still outstanding)")
        self.pending = keep

    def finish(self):
        red = self._ensure()
        red.finish()
        if self.pending is not None:
            red.flat.add_(self.pending)
            self.pending = None

    def sync_buffers(self, src: int = 0):
        """Broadcast rank ``src``'s module buffers (BatchNorm running mean / var / count): they are updated from each
        rank's own shard and drift apart; the checkpoint and the validation score are rank 0's."""
        for b in self.model.buffers():
            dist.broadcast(b.data, src=src, group=self.group)

    def agree(self, *values: float, src: int = 0):
        """Rank ``src``'s scalars on every rank (control-flow decisions must be collective)."""
        dev = next(self.model.parameters()).device
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        dist.broadcast(t, src=src, group=self.group)
        return tuple(float(v) for v in t.tolist())
