"""Static execution plans for the Attention-ASPP-UNet hot path on one MI355X.

The network graph is fixed (attention_aspp_unet_pipeline_stage.py:123-127), so for every
(batch, height, width, train/eval) the engine records ONCE the full list of kernel
launches of the forward pass and of the hand-written backward pass, with every buffer
pre-allocated and every pointer resolved.  A step is then a replay of those lists:
no allocator traffic, no autograd graph over the internals, no host synchronisation,
and therefore capturable as one hipGraph.

Memory layout (all resident in HBM for the life of the plan):
  * parameters / gradients / Adam moments: three flat fp32 buffers; each nn.Parameter is a
    view into them (4-D weights in channels_last = [O][KH][KW][I] physical order, which is
    the K-contiguous GEMM operand order), so clip + AdamW are launches over one buffer;
  * packed bf16 GEMM operands of all conv weights (forward and data-gradient forms),
    refreshed by one table-driven launch per step;
  * activations: NHWC bf16; decoder concat buffers are written in place by their
    producers (skip half by the encoder / attention gate, upsampled half by the
    ConvTranspose GEMM), never copied.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable

import torch
import torch.nn as nn

from . import _abi, ops
from ._abi import ConvDesc, PackEntry, STAT_REPLICAS

import os as _os

BF16 = torch.bfloat16
F32 = torch.float32
_NO_TRAVERSE = _os.environ.get("AAU_NO_TRAVERSE", "0") == "1"   # experiment switch
ALIGN = 64  # elements; keeps every parameter 256-byte aligned inside the flat buffers


def _ru(x, m):
    return (x + m - 1) // m * m


def _short(conv_name: str) -> str:
    """'u4.conv.0.block.0' -> 'u4.conv.0', 'bridge.blocks.1.0' -> 'bridge.blocks.1' (layer labels of the profiler)."""
    for suf in (".block.0", ".0"):
        if conv_name.endswith(suf):
            return conv_name[:-len(suf)]
    return conv_name


class _Rec:
    """Recorded launch list.  Tensors are turned into raw pointers at record time and kept alive.

    Ops can be tagged for a SIDE stream (weight gradients: they only feed the optimiser, so they run
    concurrently with the data-gradient / BatchNorm chain that the next layer is waiting for).
    ``fork`` makes the side stream wait for everything recorded so far on the main stream; ``join``
    makes the main stream wait for the side stream."""

    def __init__(self):
        self.ops = []
        self.keep = []
        self.kind = _abi.current_precision()    # the library build (bf16 / fp16) these launches were recorded against
        self.side = None        # torch.cuda.Stream, created on first use
        self.uses_side = False
        self.wg_ops = []        # (op index, workspace bytes) of the weight-gradient launches

    label = ""   # caller-side name attached to the ops recorded from now on (layer name; shown by the profiler)

    def add(self, name, *args, side=False):
        f = _abi.fn(name)
        conv = []
        for a in args:
            if isinstance(a, torch.Tensor):
                self.keep.append(a)
                conv.append(a.data_ptr())
            elif isinstance(a, ConvDesc):
                self.keep.append(a)
                conv.append(C.byref(a))
            else:
                conv.append(a)
        si = _abi.STAT_ARG.get(name)
        if si is not None:      # aau_stat pointer: its size in bytes follows it (the library checks it against the channel count)
            t = args[si]
            conv.insert(si + 1, 0 if t is None else t.numel() * t.element_size())
        self.ops.append((f, tuple(conv), name, 1 if side else 0, self.label))
        self.uses_side |= side

    def add_wgrad(self, desc, src, dz, dw, side=False, src_bn=None, dz_bn=None):
        """aau_conv_wgrad with the shared split-K workspace, which is sized and patched in by ``bind_wgrad_ws``.
        ``src_bn`` = (scale, shift): ``src`` is the raw conv output of the producing layer and its BatchNorm + ReLU is applied
        on the operand inside the kernel (aau_conv_wgrad_bnin); ``dz_bn``: the same for the ``dz`` operand (the coarse input
        activation of a ConvTranspose2d, aau_conv_wgrad_bnin_dz)."""
        if dz_bn is not None:
            self.add("aau_conv_wgrad_bnin_dz", desc, src, dz, dz_bn[0], dz_bn[1], dw, None, 0, side=side)
        elif src_bn is None:
            self.add("aau_conv_wgrad", desc, src, dz, dw, None, 0, side=side)
        else:
            self.add("aau_conv_wgrad_bnin", desc, src, src_bn[0], src_bn[1], dz, dw, None, 0, side=side)
        self.wg_ops.append((len(self.ops) - 1, ops.conv_wgrad_ws_bytes(desc)))

    def bind_wgrad_ws(self, device):
        import os
        if not self.wg_ops or os.environ.get("AAU_WGRAD_ATOMIC", "0") == "1":   # experiment: fp32-atomic split-K
            return
        nbytes = max(n for _, n in self.wg_ops)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        self.keep.append(ws)
        for i, _ in self.wg_ops:
            f, a, name, sid, lab = self.ops[i]
            self.ops[i] = (f, a[:-2] + (ws.data_ptr(), nbytes), name, sid, lab)      # (..., ws, ws_bytes) end both forms

    def callback(self, fn: Callable[[], None]):
        self.ops.append((None, fn, "callback", 0, ""))

    def fork(self):
        self.ops.append((None, None, "fork", 0, ""))

    def join(self):
        self.ops.append((None, None, "join", 0, ""))

    def run(self, stream: int):
        # alternate the traversal direction of the streaming kernels (aau_traverse): a consumer that starts where
        # its producer finished finds that end of a > 128 MB tensor still in the 256-MiB Infinity Cache
        trav = _abi.lib(self.kind).aau_traverse
        trav(0 if _NO_TRAVERSE else 1)
        try:
            self._run(stream)
        finally:
            trav(0)

    def segments(self):
        """The launch list cut at its callbacks: [(ops, callback or None), ...].  Used to replay the backward as one
        hipGraph per segment with the gradient-bucket all-reduces issued in between (pipeline.GraphedTrainStep)."""
        out, cur = [], []
        for op in self.ops:
            if op[0] is None and op[2] == "callback":
                out.append((cur, op[1]))
                cur = []
            else:
                cur.append(op)
        out.append((cur, None))
        return out

    def run_ops(self, ops_, stream: int, first: bool, last: bool):
        """One segment of ``segments()``.  The traversal parity (aau_traverse) runs on across the segments."""
        trav = _abi.lib(self.kind).aau_traverse
        if first:
            trav(0 if _NO_TRAVERSE else 1)
        try:
            self._run(stream, ops_)
        finally:
            if last:
                trav(0)

    def _run(self, stream: int, ops_=None):
        side_t = side = None
        if self.uses_side:
            if self.side is None:
                self.side = torch.cuda.Stream()
            side_t = self.side
            side = side_t.cuda_stream
            main_t = torch.cuda.current_stream()
        prof = _abi.PROF_ON
        setlab = _abi.lib(self.kind).aau_prof_label if prof else None
        for f, a, name, sid, lab in (self.ops if ops_ is None else ops_):
            if f is None:
                if name == "fork":
                    if side_t is not None:
                        side_t.wait_stream(main_t)
                elif name == "join":
                    if side_t is not None:
                        main_t.wait_stream(side_t)
                else:
                    a()
                continue
            if prof:
                setlab(f"{lab}:{name[4:]}".encode())
            rc = f(*a, side if sid else stream)
            if rc != 0:
                _abi.check(rc, name)


class _Arena:
    """Bump allocator over device buffers that are cleared together once per pass (``zero_``).  The first buffer is sized
    by the caller's estimate; a ``take`` that does not fit opens another one (the takes themselves size the arena: a new
    fused path that needs a few more words cannot overrun a constant)."""

    def __init__(self, n, device, dtype=F32):
        self.device, self.dtype = device, dtype
        self.grow = _ru(max(int(n), 16), 4)       # whole 16-byte units: the arenas are cleared by aau_zero_multi
        self.bufs = [torch.zeros(self.grow, dtype=dtype, device=device)]
        self.off = 0

    def take(self, n):
        n4 = _ru(int(n), 4)
        if self.off + n4 > self.bufs[-1].numel():
            self.bufs.append(torch.zeros(max(self.grow, n4), dtype=self.dtype, device=self.device))   # n4: multiple of 4
            self.off = 0
        t = self.bufs[-1][self.off:self.off + n]
        self.off += n4
        return t

    def zero_(self):
        for b in self.bufs:
            b.zero_()


class ConvP:
    """One convolution's weights inside the flat store."""
    __slots__ = ("name", "kind", "param", "O", "I", "k", "dil", "w", "dw", "pk_f", "pk_d", "cpad_f", "cpad_d",
                 "bias", "dbias")


class BNP:
    __slots__ = ("name", "C", "gamma", "beta", "rm", "rv", "nbt", "dgamma", "dbeta")


class ParamStore:
    """Flat fp32 parameter / gradient buffers, packed bf16 operands and the pack table."""

    def __init__(self, model: nn.Module, device):
        self.device = device
        named = list(model.named_parameters())
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += _ru(p.numel(), ALIGN)
        self.total = total
        self.flat = torch.zeros(total, dtype=F32, device=device)
        self.gflat = torch.zeros(total, dtype=F32, device=device)
        self.offs = dict(zip(self.names, offs))
        self.pviews, self.gviews = {}, {}
        for (name, p), off in zip(named, offs):
            pv, gv = self._view(self.flat, off, p.shape), self._view(self.gflat, off, p.shape)
            with torch.no_grad():
                pv.copy_(p.data.to(device=device, dtype=F32))
            p.data = pv
            p.grad = None
            self.pviews[name], self.gviews[name] = pv, gv
        self._ptrs = [p.data_ptr() for p in self.params]
        self.m = self.v = None  # Adam moments, allocated by the optimiser
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self.norm_ws = torch.zeros(ops.SQNORM_WS, dtype=F32, device=device)   # [0] = |g|^2, rest: workgroup partials
        self._build_pack(model)

    @staticmethod
    def _view(flat, off, shape):
        n = 1
        for s in shape:
            n *= s
        t = flat[off:off + n]
        if len(shape) == 4:
            d0, d1, d2, d3 = shape
            return t.view(d0, d2, d3, d1).permute(0, 3, 1, 2)  # logical OIHW, physical [O][KH][KW][I]
        return t.view(*shape)

    def intact(self) -> bool:
        return all(p.data_ptr() == q for p, q in zip(self.params, self._ptrs))

    def bind_grads(self):
        for name, p in zip(self.names, self.params):
            g = self.gviews[name]
            if p.grad is not g:
                p.grad = g

    # ---- packed operands ----
    def _build_pack(self, model):
        self.convs: dict[str, ConvP] = {}
        self.bns: dict[str, BNP] = {}
        entries, dst, blk = [], 0, 0

        def add_entry(**kw):
            nonlocal dst, blk
            e = PackEntry(dst_off=dst, blk_begin=blk, **kw)
            entries.append(e)
            n = e.R * e.T * e.Cpad
            off = dst
            dst += _ru(n, 8)
            blk += (n // 8 + 255) // 256
            return off, n

        mods = dict(model.named_modules())
        for mname, m in mods.items():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                cp = ConvP()
                cp.name, cp.param = mname, m.weight
                wname = mname + ".weight"
                cp.w, cp.dw = self.pviews[wname], self.gviews[wname]
                cp.bias = self.pviews.get(mname + ".bias")
                cp.dbias = self.gviews.get(mname + ".bias")
                src = self.offs[wname]
                cp.pk_f = cp.pk_d = None
                cp.cpad_f = cp.cpad_d = 0
                if isinstance(m, nn.ConvTranspose2d):
                    cp.kind, cp.I, cp.O, cp.k, cp.dil = "convT", m.in_channels, m.out_channels, 2, 1
                    I_, O_ = cp.I, cp.O
                    cp.cpad_f, cp.cpad_d = ops.cpad_of(I_), ops.cpad_of(O_)
                    cp.pk_f = add_entry(src_off=src, R=4 * O_, T=1, C=I_, Cpad=cp.cpad_f, s_r=O_, s_t=0, s_c=4 * O_,
                                        t_flip=0, R2=O_, s_r2=1)
                    cp.pk_d = add_entry(src_off=src, R=I_, T=4, C=O_, Cpad=cp.cpad_d, s_r=4 * O_, s_t=O_, s_c=1,
                                        t_flip=0, R2=0, s_r2=0)
                else:
                    cp.O, cp.I, cp.k, cp.dil = m.out_channels, m.in_channels, m.kernel_size[0], m.dilation[0]
                    T = cp.k * cp.k
                    if cp.I % 8 != 0:
                        cp.kind = "first"      # Conv2d(1, C, 3): direct kernel, fp32 weights
                    elif cp.O % 8 != 0:
                        cp.kind = "rowdot"     # Conv2d(C, 1, 1): out_conv / psi, fp32 weights
                    else:
                        cp.kind = "conv"
                        cp.cpad_f, cp.cpad_d = ops.cpad_of(cp.I), ops.cpad_of(cp.O)
                        cp.pk_f = add_entry(src_off=src, R=cp.O, T=T, C=cp.I, Cpad=cp.cpad_f, s_r=T * cp.I,
                                            s_t=cp.I, s_c=1, t_flip=0, R2=0, s_r2=0)
                        cp.pk_d = add_entry(src_off=src, R=cp.I, T=T, C=cp.O, Cpad=cp.cpad_d, s_r=1, s_t=cp.I,
                                            s_c=T * cp.I, t_flip=1, R2=0, s_r2=0)
                self.convs[mname] = cp
            elif isinstance(m, nn.BatchNorm2d):
                b = BNP()
                b.name, b.C = mname, m.num_features
                b.gamma, b.beta = self.pviews[mname + ".weight"], self.pviews[mname + ".bias"]
                b.dgamma, b.dbeta = self.gviews[mname + ".weight"], self.gviews[mname + ".bias"]
                b.rm, b.rv, b.nbt = m.running_mean, m.running_var, m.num_batches_tracked
                self.bns[mname] = b
        # 16-bit operand images written by aau_pack_weights at the head of every forward list: bf16 or (fp16 plans) IEEE
        # half bit patterns -- the torch dtype of this buffer is only a label
        self.packed = torch.zeros(max(dst, 8), dtype=BF16, device=self.device)
        for cp in self.convs.values():
            for key in ("pk_f", "pk_d"):
                v = getattr(cp, key)
                if v is not None:
                    setattr(cp, key, self.packed[v[0]:v[0] + v[1]])
        arr = (PackEntry * len(entries))(*entries)
        self.pack_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self.pack_n, self.pack_blocks = len(entries), blk
        self.bn_channels = sum(b.C for b in self.bns.values())

    def buffers_intact(self, model) -> bool:
        mods = dict(model.named_modules())
        for name, b in self.bns.items():
            m = mods[name]
            if b.rm is not m.running_mean or b.rv is not m.running_var or b.nbt is not m.num_batches_tracked:
                return False
        return True


class Plan:
    """Recorded forward (and backward) launch lists for one input shape / mode."""

    def __init__(self, eng: "Engine", B: int, H: int, W: int, train: bool):
        if B < 1 or H < 16 or W < 16 or H % 16 or W % 16:
            raise _abi.AauError(f"input {B}x1x{H}x{W}: H and W must be positive multiples of 16 (four 2x2 poolings; the "
                                "reference's bilinear fix-up of odd sizes, pipeline:106-107, is not built)")
        if train and B < 2:
            raise _abi.AauError("training-mode BatchNorm of the ASPP image-pool branch needs batch >= 2 (pipeline:75-77: "
                                "'Expected more than 1 value per channel when training')")
        self.eng, self.B, self.H, self.W, self.train = eng, B, H, W, train
        self.dev = eng.store.device
        self.fwd, self.bwd = _Rec(), _Rec()
        self.fwd_gen, self.bwd_gen = 0, -1   # training forwards run / generation whose backward has run (model._NetFn)
        # dropout seed in device memory, advanced on the stream once per training forward (graph-capturable)
        self.drop_seed = torch.tensor([eng.next_seed() >> 1], dtype=torch.int64, device=eng.store.device)
        self.drop_p = eng.dropout_p() if train else 0.0
        st = eng.store
        nbn = st.bn_channels + 8
        # order-independent fixed-point statistics (include/aau.h: aau_stat): int64 arenas, zeroed once per pass
        self.stats_arena = _Arena(STAT_REPLICAS * 4 * nbn + 4 * 64, self.dev, dtype=torch.int64)
        self.bstats_arena = _Arena(STAT_REPLICAS * 4 * 56 * eng.store.convs["d1.0.block.0"].O + 64, self.dev,
                                   dtype=torch.int64) if train else None     # channel sums of dcat (4 levels x 3*Co)
        self.red_arena = _Arena(STAT_REPLICAS * 2 * nbn * 3 + STAT_REPLICAS * nbn * 3, self.dev)
        self.vec_arena = _Arena(nbn * 8 + 64, self.dev)
        # workspace of the fixed-order cross-workgroup sums of the BatchNorm-backward reduce passes (include/aau.h:
        # aau_bn_red_ws_bytes): shared by all layers (reduce -> apply are adjacent on the stream)
        self.red_ws = torch.zeros(ops.bn_red_ws_bytes(max(b.C for b in st.bns.values())) // 4, dtype=F32,
                                  device=self.dev) if train else None
        self.x = torch.zeros(B, 1, H, W, dtype=F32, device=self.dev)
        self.logits = torch.zeros(B, 1, H, W, dtype=F32, device=self.dev)
        self.dlogits = torch.zeros(B, 1, H, W, dtype=F32, device=self.dev) if train else None
        self.psis = {}         # decoder level -> (alpha fp32 [M], h, w) of the residual gates (ablation variant)
        self.bwd_blocks = []   # per forward block: list of (name, args) recorded later in reverse
        self.bucket_hooks = {}  # block index -> callback fired after that block's backward
        self._build()
        self.bwd.bind_wgrad_ws(self.dev)

    # ---- small helpers ----
    def _planar_ok(self, lv, gate_kind, B, h, w, Co, train) -> bool:
        """Two-plane concat at this level: ungated, and every consumer launch serves two-plane operands."""
        import os
        if gate_kind is not None or os.environ.get("AAU_NO_PLANAR", "0") == "1":
            return False
        st = self.eng.store
        cv = st.convs[f"u{lv + 1}.conv.0.block.0"]
        M = B * h * w
        if 2 * M * Co >= 1 << 30:
            return False
        fwd = ops.conv_desc(B, h, w, 2 * Co, Co, h, w, Co, Co, 3, 3, 1, 1, 1, cv.cpad_f, src_split=(Co, M * Co))
        if not ops.conv_split_ok(fwd, 0):
            return False
        if train:
            dg = ops.conv_desc(B, h, w, Co, Co, h, w, 2 * Co, Co, 3, 3, 1, 1, 1, cv.cpad_d, dst_split=(Co, M * Co))
            wg = ops.conv_desc(B, h, w, 2 * Co, Co, h, w, Co, Co, 3, 3, 1, 1, 1, src_split=(Co, M * Co))
            if not (ops.conv_split_ok(dg, 0) and ops.conv_split_ok(wg, 1)):
                return False
        return True

    def _bnin_ok(self, cname, N, H, W, Cprev) -> bool:
        """May conv ``cname`` take the previous ConvBNReLU's RAW output (Cprev channels, dense) with that layer's BatchNorm +
        ReLU applied on the operand -- in the forward conv AND in the weight gradient (both must be served)?"""
        if not self.train or self.eng.no_bnin:
            return False
        if Cprev > self.eng.bnin_max_c:
            return False
        cv = self.eng.store.convs[cname]
        if cv.kind != "conv" or cv.k != 3 or cv.dil != 1 or cv.I != Cprev:
            return False
        fwd = ops.conv_desc(N, H, W, cv.I, cv.I, H, W, cv.O, cv.O, 3, 3, 1, 1, 1, cv.cpad_f)
        wg = ops.conv_desc(N, H, W, cv.I, cv.I, H, W, cv.O, cv.O, 3, 3, 1, 1, 1)
        return ops.conv_bnin_ok(fwd) and ops.conv_wgrad_bnin_ok(wg)

    def new(self, *shape, dtype=None):
        dtype = self.eng.adt if dtype is None else dtype
        return torch.zeros(*shape, dtype=dtype, device=self.dev)

    def bnbuf(self, C_):
        va = self.vec_arena
        return dict(stats=self.stats_arena.take(ops.stat_words(C_)), scale=va.take(C_), shift=va.take(C_),
                    mean=va.take(C_), invstd=va.take(C_), red=self.red_arena.take(STAT_REPLICAS * 2 * C_))

    def _bn_finalize(self, bn: BNP, w, count):
        self.fwd.add("aau_bn_finalize", w["stats"], bn.gamma, bn.beta, bn.rm, bn.rv, bn.nbt, w["scale"], w["shift"],
                     w["mean"], w["invstd"], bn.C, count, 1e-5, 0.1)

    def _bn_fold(self, bn: BNP, w):
        self.fwd.add("aau_bn_fold_eval", bn.gamma, bn.beta, bn.rm, bn.rv, w["scale"], w["shift"], bn.C, 1e-5)

    # ---- ConvBNReLU on MFMA: forward ----
    def cbr_fwd(self, cname, bname, src, sp, N, H, W, ydst, yp, drop=False, bcast_hw=0, pool=None, head=None,
                src_split=(0, 0), group=None, src_bn=None, skip_act=False):
        """conv(cname) -> BN(bname) -> ReLU; returns the per-layer record used by the backward.
        ``head`` (training only): the out_conv ConvP when this is the last ConvBNReLU -- its activation feeds out_conv
        alone, so BN + ReLU + out_conv run as one pass and neither the activation nor its gradient is stored.
        ``skip_act`` / ``src_bn`` (training only): a ConvBNReLU whose ONLY consumer is the next 3x3 conv does not write its
        activation (skip_act); that conv reads the raw output ``src`` = z and applies relu(z * scale + shift), src_bn =
        (scale, shift), on the operand in LDS -- forward and weight gradient (aau_conv_igemm_bnin / aau_conv_wgrad_bnin)."""
        st = self.eng.store
        cv, bn = st.convs[cname], st.bns[bname]
        self.fwd.label = _short(cname)
        M = N * H * W
        pad = cv.dil * (cv.k // 2)
        w = self.bnbuf(bn.C)
        rec = dict(cv=cv, bn=bn, w=w, N=N, H=H, W=W, M=M, src=src, sp=sp, drop=drop, bcast_hw=bcast_hw,
                   src_split=src_split, src_bn=src_bn)
        if self.train:
            z = self.new(M, cv.O)
            d = ops.conv_desc(N, H, W, cv.I, sp, H, W, cv.O, cv.O, cv.k, cv.k, 1, pad, cv.dil, cv.cpad_f,
                              src_split=src_split)
            if group is not None:
                # ``group``: the conv joins one multi-problem launch (aau_conv_igemm_multi) emitted by the caller, who then
                # runs the statistics / activation part recorded here
                lab = self.fwd.label
                group["convs"].append((d, src, cv.pk_f, z, w["stats"], lab))

                def post(bn=bn, w=w, z=z, lab=lab):
                    self.fwd.label = lab
                    self._bn_finalize(bn, w, M)
                    self.fwd.add("aau_bn_act", z, cv.O, ydst, yp, w["scale"], w["shift"], M, cv.O, 1, 0,
                                 self.drop_p if drop else 0.0, self.drop_seed)
                assert head is None and pool is None and not bcast_hw
                group["post"].append(post)
                group.setdefault("bn", []).append((bn, w, z, ydst, yp, drop))
                rec["z"] = z
                return rec
            if src_bn is not None:
                self.fwd.add("aau_conv_igemm_bnin", d, src, src_bn[0], src_bn[1], cv.pk_f, z, None, w["stats"])
            else:
                self.fwd.add("aau_conv_igemm", d, src, cv.pk_f, z, None, None, None, w["stats"])
            self._bn_finalize(bn, w, M)
            Mo = M * bcast_hw if bcast_hw else M
            if skip_act:
                assert head is None and pool is None and not bcast_hw and not drop
            elif head is not None:
                self.fwd.add("aau_bn_act_outconv", z, cv.O, w["scale"], w["shift"], head.w, head.bias, self.logits, M, cv.O)
                rec["head"] = head
            elif pool is not None:
                self.fwd.add("aau_bn_act_pool", z, cv.O, ydst, yp, pool, cv.O, w["scale"], w["shift"], N, H, W, cv.O)
            else:
                self.fwd.add("aau_bn_act", z, cv.O, ydst, yp, w["scale"], w["shift"], Mo, cv.O, 1, bcast_hw,
                             self.drop_p if drop else 0.0, self.drop_seed)
            rec["z"] = z
        else:
            self._bn_fold(bn, w)
            if bcast_hw:
                z = self.new(M, cv.O)
                d = ops.conv_desc(N, H, W, cv.I, sp, H, W, cv.O, cv.O, cv.k, cv.k, 1, pad, cv.dil, cv.cpad_f)
                self.fwd.add("aau_conv_igemm", d, src, cv.pk_f, z, None, None, None, None)
                self.fwd.add("aau_bn_act", z, cv.O, ydst, yp, w["scale"], w["shift"], M * bcast_hw, cv.O, 1,
                             bcast_hw, 0.0, self.drop_seed)
            else:
                d = ops.conv_desc(N, H, W, cv.I, sp, H, W, cv.O, yp, cv.k, cv.k, 1, pad, cv.dil, cv.cpad_f, relu=1,
                                  src_split=src_split)
                self.fwd.add("aau_conv_igemm", d, src, cv.pk_f, ydst, None, w["scale"], w["shift"], None)
                if pool is not None:
                    self.fwd.add("aau_maxpool2", ydst, yp, pool, cv.O, N, H, W, cv.O)
        return rec

    # ---- ConvBNReLU backward: dy (+ pooled gradient) -> dW, dgamma, dbeta, d(input) ----
    def cbr_bwd(self, r, dy, dyp, dpool=None, dpp=0, din=None, dinp=0, accumulate=0, din_stats=None,
                defer_wgrad=None, din_split=(0, 0), defer_dgrad=None, red_next=None, dz_given=None):
        """``red_next``: the record of the ConvBNReLU layer whose output gradient is ``din`` (its only source): where the
        library can (aau_conv_bnred_ok), the data-gradient conv below also accumulates THAT layer's BatchNorm-backward
        sums in its epilogue and the layer's own reduce pass over (z, dy) is not recorded."""
        cv, bn, w = r["cv"], r["bn"], r["w"]
        N, H, W, M = r["N"], r["H"], r["W"], r["M"]
        b = self.bwd
        b.label = _short(cv.name)
        dp_ = self.drop_p if r["drop"] else 0.0
        fuse1 = cv.kind == "first" and dpool is None and dp_ == 0.0 and not self.eng.no_fuse_conv1
        dz = dz_given if dz_given is not None else (None if fuse1 else self.new(M, cv.O))
        head = r.get("head")
        if dz_given is not None:
            pass        # the BatchNorm backward of this layer ran in a multi-layer launch (bridge branches)
        elif head is not None:
            # network head: dy = dlogits x w_out is rank one and never stored (see cbr_fwd)
            b.add("aau_bn_bwd_reduce_outconv", r["z"], cv.O, self.dlogits, head.w, w["scale"], w["shift"], w["mean"],
                  w["invstd"], w["red"], head.dw, head.dbias, self.red_ws, M, cv.O)
            b.add("aau_bn_bwd_apply_rank1", r["z"], cv.O, dz, cv.O, bn.gamma, w["mean"], w["invstd"], w["red"], bn.dgamma,
                  bn.dbeta, M, cv.O, self.dlogits, head.w, w["scale"], w["shift"])
        elif dpool is None:
            # no pooling: the reduce pass only accumulates, the apply pass recomputes the ReLU / dropout mask
            if r.get("red_done"):   # the sums came out of the producing data-gradient conv (red_next above)
                pass
            elif r["z"] is None:    # first layer, z not stored
                b.add("aau_conv1_bn_bwd_reduce", r["src"], cv.w, dy, dyp, w["scale"], w["shift"], w["mean"],
                      w["invstd"], w["red"], N, H, W, cv.O, self.red_ws)
            else:
                b.add("aau_bn_bwd_reduce", r["z"], cv.O, dy, dyp, None, 0, None, cv.O, w["scale"], w["shift"],
                      w["mean"], w["invstd"], w["red"], N, H, W, cv.O, 1, dp_, self.drop_seed, self.red_ws)
            if fuse1:
                # first layer: no input gradient, so the apply pass feeds the weight gradient directly
                b.add("aau_bn_bwd_apply_conv1", r["z"], cv.O, bn.gamma, w["mean"], w["invstd"], w["red"], bn.dgamma,
                      bn.dbeta, N, H, W, cv.O, dy, dyp, w["scale"], w["shift"], r["src"], cv.w, cv.dw, self.red_ws)
                return None
            b.add("aau_bn_bwd_apply", r["z"], cv.O, dz, cv.O, bn.gamma, w["mean"], w["invstd"], w["red"], bn.dgamma,
                  bn.dbeta, M, cv.O, dy, dyp, w["scale"], w["shift"], 1, dp_, self.drop_seed)
        else:
            if self.eng.pool_store_routed:     # default: the reduce pass stores the routed gradient, the apply pass reads it
                b.add("aau_bn_bwd_reduce", r["z"], cv.O, dy, dyp, dpool, dpp, dz, cv.O, w["scale"], w["shift"], w["mean"],
                      w["invstd"], w["red"], N, H, W, cv.O, 1, dp_, self.drop_seed, self.red_ws)
                b.add("aau_bn_bwd_apply", r["z"], cv.O, dz, cv.O, bn.gamma, w["mean"], w["invstd"], w["red"], bn.dgamma,
                      bn.dbeta, M, cv.O, None, 0, None, None, 1, 0.0, self.drop_seed)
            else:
                # opt-in: both passes route the pooled gradient themselves, no intermediate tensor (2x2-window accesses in
                # the apply pass cost more than the 1.25 tensor passes they save)
                b.add("aau_bn_bwd_reduce", r["z"], cv.O, dy, dyp, dpool, dpp, None, cv.O, w["scale"], w["shift"], w["mean"],
                      w["invstd"], w["red"], N, H, W, cv.O, 1, dp_, self.drop_seed, self.red_ws)
                b.add("aau_bn_bwd_apply_pool", r["z"], cv.O, dz, cv.O, bn.gamma, w["mean"], w["invstd"], w["red"],
                      bn.dgamma, bn.dbeta, N, H, W, cv.O, dy, dyp, dpool, dpp, w["scale"], w["shift"], 1)
        pad = cv.dil * (cv.k // 2)
        ov = self.eng.overlap_wgrad
        if ov:
            b.fork()            # dz is final: the weight gradient can run beside the data-gradient chain
        if cv.kind == "first":
            b.add("aau_conv1_wgrad", r["src"], dz, cv.dw, N, H, W, cv.O, side=ov)
            return dz
        dwd = ops.conv_desc(N, H, W, cv.I, r["sp"], H, W, cv.O, cv.O, cv.k, cv.k, 1, pad, cv.dil,
                            src_split=r.get("src_split", (0, 0)))
        if defer_wgrad is not None:
            defer_wgrad.append((dwd, r["src"], dz, cv.dw, b.label))     # emitted later as one grouped launch
        else:
            b.add_wgrad(dwd, r["src"], dz, cv.dw, side=ov, src_bn=r.get("src_bn"))
        if din is not None:
            dd = ops.conv_desc(N, H, W, cv.O, cv.O, H, W, cv.I, dinp, cv.k, cv.k, 1, pad, cv.dil, cv.cpad_d,
                               accumulate=accumulate, dst_split=din_split)
            # din_stats: [R][2][Cin] sums of the produced gradient (channel sums feed the ConvTranspose bias grad)
            nr = red_next
            fuse_red = (nr is not None and defer_dgrad is None and din_stats is None and nr.get("z") is not None
                        and not nr["drop"] and nr.get("head") is None and nr["cv"].O == cv.I and dinp == cv.I
                        and not self.eng.no_fuse_bnred and ops.conv_bnred_ok(dd))
            if defer_dgrad is not None:
                defer_dgrad.append((dd, dz, cv.pk_d, din, b.label))     # emitted later as one grouped launch
            elif fuse_red:
                nw = nr["w"]
                sums = self.bstats_arena.take(ops.stat_words(cv.I))
                b.add("aau_conv_igemm_bnred", dd, dz, cv.pk_d, din, nr["z"], cv.I, nw["scale"], nw["shift"], nw["mean"],
                      nw["invstd"], sums)
                b.add("aau_stats_to_red", sums, cv.I, nw["red"])
                nr["red_done"] = True
            else:
                b.add("aau_conv_igemm", dd, dz, cv.pk_d, din, None, None, None, din_stats)
        return dz

    # ---- graph ----
    def _build(self):
        eng, st, f, B, H, W = self.eng, self.eng.store, self.fwd, self.B, self.H, self.W
        model = eng.model
        c = st.convs["d1.0.block.0"].O
        tr = self.train
        Hs = [H, H // 2, H // 4, H // 8, H // 16]
        Ws = [W, W // 2, W // 4, W // 8, W // 16]
        Ms = [B * h * w for h, w in zip(Hs, Ws)]
        Cs = [c, 2 * c, 4 * c, 8 * c, 16 * c]

        # weights: one table-driven repack per step
        f.label = "pack"
        f.add("aau_pack_weights", st.flat, st.packed, st.pack_table, st.pack_n, st.pack_blocks)

        # ---------------- encoder ----------------
        # an up-block WITHOUT a gate concatenates the encoder output as is: the encoder then writes its output straight
        # into the lower half of that block's concat buffer (u1 always; u2..u4 of the ablation variant without attention)
        gate_kinds = [eng.gate_kind(f"u{lv + 1}") for lv in range(4)]
        # Concat buffers torch.cat([skip, up], 1) (pipeline:108).  Interleaved [M][2C] by default; an ungated level whose
        # consumer kernels take two-plane operands (aau.h, aau_conv_split_ok) keeps TWO DENSE PLANES [2][M][C] instead:
        # at C = 48 the 96-byte half rows at a 192-byte pitch cost their writers 2x and their readers 1.5x per byte
        # (round-2 micro-benchmark) -- the encoder output, the transposed conv and their gradients at level 1.
        planar = [self._planar_ok(lv, gate_kinds[lv], B, Hs[lv], Ws[lv], Cs[lv], tr) for lv in range(4)]
        cats = [self.new(2, Ms[lv], Cs[lv]) if planar[lv] else self.new(Ms[lv], 2 * Cs[lv]) for lv in range(4)]
        cat_p = [Cs[lv] if planar[lv] else 2 * Cs[lv] for lv in range(4)]            # pixel pitch of either half
        cat_split = [(Cs[lv], Ms[lv] * Cs[lv]) if planar[lv] else (0, 0) for lv in range(4)]

        def lo(buf, lv):
            return buf[0] if planar[lv] else buf

        def hi_(buf, lv):
            return buf[1] if planar[lv] else buf[:, Cs[lv]:]
        skips = [lo(cats[lv], lv) if gate_kinds[lv] is None else self.new(Ms[lv], Cs[lv]) for lv in range(4)]
        skip_p = [cat_p[lv] if gate_kinds[lv] is None else Cs[lv] for lv in range(4)]
        pools = [self.new(Ms[i + 1], Cs[i]) for i in range(4)]
        enc = []  # (rec0, rec1) per level
        # d1.0: direct kernel on the fp32 frame
        cv0, bn0 = st.convs["d1.0.block.0"], st.bns["d1.0.block.1"]
        f.label = "d1.0"
        w0 = self.bnbuf(c)
        # training: d1.1 reads the raw first-layer output and applies its BatchNorm + ReLU on the operand (no y10 at all)
        bnin10 = eng.no_recompute_z1 and self._bnin_ok("d1.1.block.0", B, H, W, c)
        y10 = None if bnin10 else self.new(Ms[0], c)
        r10 = dict(cv=cv0, bn=bn0, w=w0, N=B, H=H, W=W, M=Ms[0], src=self.x, sp=1, drop=False, bcast_hw=0)
        if eng.no_recompute_z1:
            z10 = self.new(Ms[0], c)
            f.add("aau_conv1_fwd", self.x, cv0.w, z10, w0["stats"] if tr else None, B, H, W, c)
            if tr:
                self._bn_finalize(bn0, w0, Ms[0])
            else:
                self._bn_fold(bn0, w0)
            if not bnin10:
                f.add("aau_bn_act", z10, c, y10, c, w0["scale"], w0["shift"], Ms[0], c, 1, 0, 0.0, self.drop_seed)
        else:
            # z of the first layer (9 FMAs per value) is never stored: statistics pass, then y straight from the frame;
            # the backward recomputes it the same way (aau_conv1_bn_bwd_reduce, aau_bn_bwd_apply_conv1)
            z10 = None
            if tr:
                f.add("aau_conv1_fwd", self.x, cv0.w, None, w0["stats"], B, H, W, c)
                self._bn_finalize(bn0, w0, Ms[0])
            else:
                self._bn_fold(bn0, w0)
            f.add("aau_conv1_bn_act", self.x, cv0.w, y10, c, w0["scale"], w0["shift"], B, H, W, c)
        r10["z"] = z10
        if bnin10:
            r11 = self.cbr_fwd("d1.1.block.0", "d1.1.block.1", z10, c, B, H, W, skips[0], skip_p[0], pool=pools[0],
                               src_bn=(w0["scale"], w0["shift"]))
        else:
            r11 = self.cbr_fwd("d1.1.block.0", "d1.1.block.1", y10, c, B, H, W, skips[0], skip_p[0], pool=pools[0])
        enc.append((r10, r11))
        for lv in range(1, 4):
            h, w_ = Hs[lv], Ws[lv]
            bnin = self._bnin_ok(f"d{lv + 1}.1.block.0", B, h, w_, Cs[lv])
            ya = None if bnin else self.new(Ms[lv], Cs[lv])
            ra = self.cbr_fwd(f"d{lv + 1}.0.block.0", f"d{lv + 1}.0.block.1", pools[lv - 1], Cs[lv - 1], B, h, w_,
                              ya, Cs[lv], skip_act=bnin)
            rb = self.cbr_fwd(f"d{lv + 1}.1.block.0", f"d{lv + 1}.1.block.1", ra["z"] if bnin else ya, Cs[lv], B, h, w_,
                              skips[lv], skip_p[lv], pool=pools[lv],
                              src_bn=(ra["w"]["scale"], ra["w"]["shift"]) if bnin else None)
            enc.append((ra, rb))

        # ---------------- bridge: ASPP (pipeline:67-83) or ConvBNReLU + Dropout (ablation:194-197) ----------------
        h5, w5, M5 = Hs[4], Ws[4], Ms[4]
        Cb = Cs[4]
        p4 = pools[3]
        bout = self.new(M5, Cb)
        aspp = hasattr(model.bridge, "blocks")
        if aspp:
            nbr = len(model.bridge.blocks)
            ncat = (nbr + 1) * Cb
            cat5 = self.new(M5, ncat)
            # training: the spatial branches' convolutions as ONE multi-problem launch where the library serves it (each
            # of them is one workgroup per CU on its own), longest problem first
            grp = dict(convs=[], post=[]) if tr and 2 <= nbr <= 8 else None
            br = [self.cbr_fwd(f"bridge.blocks.{i}.0", f"bridge.blocks.{i}.1", p4, Cs[3], B, h5, w5,
                               cat5[:, i * Cb:], ncat, group=grp) for i in range(nbr)]
            if grp is not None:
                cvs = grp["convs"]
                order = sorted(range(len(cvs)), key=lambda i: -(cvs[i][0].KH * cvs[i][0].KW) * 100 + cvs[i][0].dil)
                descs = [cvs[i][0] for i in order]
                if not eng.no_igemm_multi and len(descs) <= 4 and ops.conv_igemm_multi_ok(descs):
                    f.label = "bridge(multi)"
                    pack = ops.igemm_multi_args(descs, [cvs[i][1] for i in order], [cvs[i][2] for i in order],
                                                [cvs[i][3] for i in order], [cvs[i][4] for i in order])
                    f.keep.extend([cvs[i][k] for i in order for k in (1, 2, 3, 4)])
                    f.keep.append(pack)
                    f.add("aau_conv_igemm_multi", *pack)
                else:
                    for d_, src_, pk_, z_, st_, lab in cvs:
                        f.label = lab
                        f.add("aau_conv_igemm", d_, src_, pk_, z_, None, None, None, st_)
                bns = grp.get("bn", [])
                if (not eng.no_bn_multi and 2 <= len(bns) <= 8 and len({b_[0].C for b_ in bns}) == 1
                        and len({b_[4] for b_ in bns}) == 1 and not any(b_[5] for b_ in bns)):
                    # the branches' BatchNorm statistics and activations: one launch each for all of them
                    f.label = "bridge(multi)"
                    Cn = bns[0][0].C
                    tabF = ops.ptr_table([[w_["stats"], bn_.gamma, bn_.beta, bn_.rm, bn_.rv, bn_.nbt, w_["scale"], w_["shift"],
                                           w_["mean"], w_["invstd"]] for bn_, w_, _, _, _, _ in bns])
                    tabA = ops.ptr_table([[z_, y_, w_["scale"], w_["shift"]] for _, w_, z_, y_, _, _ in bns])
                    f.keep.extend([tabF, tabA] + [t for bn_, w_, z_, y_, _, _ in bns for t in (bn_.rm, bn_.rv, bn_.nbt, z_, y_)])
                    f.add("aau_bn_finalize_multi", len(bns), tabF, ops.stat_words(Cn) * 8, Cn, M5, 1e-5, 0.1)
                    f.add("aau_bn_act_multi", len(bns), tabA, Cn, bns[0][4], M5, Cn, 1)
                    self.bn_multi_fwd = True
                else:
                    for post in grp["post"]:
                        post()
            pooled = self.new(B, Cs[3])
            gap_ws = self.new(ops.GAP_WS_ROWS, B, max(Cs[3], Cb), dtype=F32)     # one row per pixel slab
            f.label = "bridge.pool"
            f.add("aau_gap_fwd", p4, Cs[3], pooled, gap_ws, B, h5 * w5, Cs[3])
            if tr and B <= 16 and not eng.no_poolbranch:
                # training: conv + BatchNorm statistics over the B samples in one latency-sized launch (poolbranch.hip)
                cvp, bnp = st.convs["bridge.pool.1"], st.bns["bridge.pool.2"]
                wp = self.bnbuf(Cb)
                zp_ = self.new(B, Cb)
                f.add("aau_poolbranch_fwd", pooled, Cs[3], cvp.pk_f, cvp.cpad_f, zp_, bnp.gamma, bnp.beta, bnp.rm, bnp.rv,
                      bnp.nbt, wp["scale"], wp["shift"], wp["mean"], wp["invstd"], B, Cs[3], Cb, 1e-5, 0.1)
                f.add("aau_bn_act", zp_, Cb, cat5[:, nbr * Cb:], ncat, wp["scale"], wp["shift"], M5, Cb, 1, h5 * w5, 0.0,
                      self.drop_seed)
                rpool = dict(cv=cvp, bn=bnp, w=wp, z=zp_, src=pooled, fused=True)
            else:
                rpool = self.cbr_fwd("bridge.pool.1", "bridge.pool.2", pooled, Cs[3], B, 1, 1, cat5[:, nbr * Cb:], ncat,
                                     bcast_hw=h5 * w5)
            rproj = self.cbr_fwd("bridge.project.0", "bridge.project.1", cat5, ncat, B, h5, w5, bout, Cb, drop=True)
        else:
            rplain = self.cbr_fwd("bridge.0.block.0", "bridge.0.block.1", p4, Cs[3], B, h5, w5, bout, Cb, drop=True)

        # ---------------- decoder ----------------
        dec = []
        g_in, g_c = bout, Cb
        g_bn = None      # (scale, shift) when g_in is the RAW output of the previous level's last ConvBNReLU (activation not stored)
        for lv in (3, 2, 1, 0):
            name = f"u{lv + 1}"
            Co = Cs[lv]
            hi, wi = Hs[lv + 1], Ws[lv + 1]     # input grid of the up-conv
            ho, wo, Mo = Hs[lv], Ws[lv], Ms[lv]
            up = st.convs[f"{name}.up"]
            cat = cats[lv]
            kind = gate_kinds[lv]
            dup = ops.conv_desc(B, hi, wi, g_c, g_c, hi, wi, 4 * Co, cat_p[lv], Cpad=up.cpad_f, shuffle2x2=1)
            f.label = f"{name}.up"
            if g_bn is not None:     # the previous level's BatchNorm + ReLU on the transposed conv's operand
                f.add("aau_conv_igemm_bnin", dup, g_in, g_bn[0], g_bn[1], up.pk_f, hi_(cat, lv), up.bias, None)
            else:
                f.add("aau_conv_igemm", dup, g_in, up.pk_f, hi_(cat, lv), up.bias, None, None, None)
            gate = None
            if kind == "res":
                wg, wx, psi = st.convs[f"{name}.att.Wg"], st.convs[f"{name}.att.Wx"], st.convs[f"{name}.att.psi.1"]
                Fi = wg.O
                zg, zx = self.new(Mo, Fi), self.new(Mo, Fi)
                alpha = self.new(Mo, dtype=F32)
                f.label = f"{name}.att"
                dg = ops.conv_desc(B, ho, wo, Co, 2 * Co, ho, wo, Fi, Fi, Cpad=wg.cpad_f)
                dx = ops.conv_desc(B, ho, wo, Co, skip_p[lv], ho, wo, Fi, Fi, Cpad=wx.cpad_f)
                f.add("aau_conv_igemm", dg, cat[:, Co:], wg.pk_f, zg, None, None, None, None)
                f.add("aau_conv_igemm", dx, skips[lv], wx.pk_f, zx, None, None, None, None)
                f.add("aau_gate2_fwd", zg, zx, psi.w, psi.bias, skips[lv], skip_p[lv], alpha, cat, 2 * Co, Mo, Fi, Co)
                gate = dict(kind="res", wg=wg, wx=wx, psi=psi, zg=zg, zx=zx, alpha=alpha, Fi=Fi,
                            rep=self.new(STAT_REPLICAS * (Fi + 8), dtype=F32) if tr else None)
                self.psis[lv] = (alpha, ho, wo)
            elif kind == "bn":
                Fi = Co // 2
                wg, wx = st.convs[f"{name}.att.Wg.0"], st.convs[f"{name}.att.Wx.0"]
                bg, bx, b1 = st.bns[f"{name}.att.Wg.1"], st.bns[f"{name}.att.Wx.1"], st.bns[f"{name}.att.psi.1"]
                psi = st.convs[f"{name}.att.psi.0"]
                wgb, wxb, w1 = self.bnbuf(Fi), self.bnbuf(Fi), self.bnbuf(1)
                zg, zx = self.new(Mo, Fi), self.new(Mo, Fi)
                f.label = f"{name}.att"
                psi_pre = self.new(Mo, dtype=F32)
                alpha = self.new(Mo, dtype=F32)
                dg = ops.conv_desc(B, ho, wo, Co, 2 * Co, ho, wo, Fi, Fi, Cpad=wg.cpad_f)
                dx = ops.conv_desc(B, ho, wo, Co, skip_p[lv], ho, wo, Fi, Fi, Cpad=wx.cpad_f)
                sg_, sx_ = (wgb["stats"], wxb["stats"]) if tr else (None, None)
                if not eng.no_igemm_multi and ops.conv_igemm_multi_ok([dg, dx]):
                    # the gate's two 1x1 GEMMs (W_g on the up-sampled path, W_x on the skip) as one two-problem launch
                    pack = ops.igemm_multi_args([dg, dx], [cat[:, Co:], skips[lv]], [wg.pk_f, wx.pk_f], [zg, zx], [sg_, sx_])
                    f.keep.extend([cat, skips[lv], wg.pk_f, wx.pk_f, zg, zx, pack] + ([sg_, sx_] if tr else []))
                    f.add("aau_conv_igemm_multi", *pack)
                else:
                    f.add("aau_conv_igemm", dg, cat[:, Co:], wg.pk_f, zg, None, None, None, sg_)
                    f.add("aau_conv_igemm", dx, skips[lv], wx.pk_f, zx, None, None, None, sx_)
                if tr and not eng.no_bn_multi and bg.C == bx.C:
                    # the statistics of the gate's two 1x1 convs: one launch
                    tabF = ops.ptr_table([[w_["stats"], bn_.gamma, bn_.beta, bn_.rm, bn_.rv, bn_.nbt, w_["scale"], w_["shift"],
                                           w_["mean"], w_["invstd"]] for bn_, w_ in ((bg, wgb), (bx, wxb))])
                    f.keep.extend([tabF, bg.rm, bg.rv, bg.nbt, bx.rm, bx.rv, bx.nbt])
                    f.add("aau_bn_finalize_multi", 2, tabF, ops.stat_words(bg.C) * 8, bg.C, Mo, 1e-5, 0.1)
                elif tr:
                    self._bn_finalize(bg, wgb, Mo)
                    self._bn_finalize(bx, wxb, Mo)
                else:
                    self._bn_fold(bg, wgb)
                    self._bn_fold(bx, wxb)
                f.add("aau_gate_psi", zg, zx, wgb["scale"], wgb["shift"], wxb["scale"], wxb["shift"], psi.w, psi_pre,
                      w1["stats"] if tr else None, Mo, Fi)
                if tr:
                    self._bn_finalize(b1, w1, Mo)
                else:
                    self._bn_fold(b1, w1)
                f.add("aau_gate_apply", skips[lv], skip_p[lv], psi_pre, w1["scale"], w1["shift"], alpha, cat, 2 * Co,
                      Mo, Co)
                gate = dict(kind="bn", wg=wg, wx=wx, psi=psi, bg=bg, bx=bx, b1=b1, wgb=wgb, wxb=wxb, w1=w1, zg=zg, zx=zx,
                            psi_pre=psi_pre, alpha=alpha, Fi=Fi,
                            wrep=self.red_arena.take(STAT_REPLICAS * Fi) if tr else None)
            bnin = self._bnin_ok(f"{name}.conv.1.block.0", B, ho, wo, Co)
            ya = None if bnin else self.new(Mo, Co)
            ra = self.cbr_fwd(f"{name}.conv.0.block.0", f"{name}.conv.0.block.1", cat, cat_p[lv], B, ho, wo, ya, Co,
                              src_split=cat_split[lv], skip_act=bnin)
            fuse_head = tr and lv == 0 and not eng.no_fuse_head
            # the activation of this level's last ConvBNReLU feeds the next level's transposed conv alone: where that conv's
            # forward and weight-gradient kernels apply BatchNorm + ReLU on their operand, it is not stored either
            up_bnin = False
            if tr and lv > 0 and not eng.no_bnin and not eng.no_bnin_up:
                Cn, upn = Cs[lv - 1], st.convs[f"u{lv}.up"]
                dn = ops.conv_desc(B, ho, wo, Co, Co, ho, wo, 4 * Cn, cat_p[lv - 1], Cpad=upn.cpad_f, shuffle2x2=1)
                un = ops.conv_desc(B, Hs[lv - 1], Ws[lv - 1], Cn, cat_p[lv - 1], ho, wo, Co, Co, 2, 2, 2, 0, 1)
                up_bnin = ops.conv_bnin_ok(dn) and ops.conv_wgrad_bnin_dz_ok(un)
            yb = None if (fuse_head or up_bnin) else self.new(Mo, Co)
            rb = self.cbr_fwd(f"{name}.conv.1.block.0", f"{name}.conv.1.block.1", ra["z"] if bnin else ya, Co, B, ho, wo,
                              yb, Co, head=st.convs["out_conv"] if fuse_head else None,
                              src_bn=(ra["w"]["scale"], ra["w"]["shift"]) if bnin else None, skip_act=up_bnin)
            dec.append(dict(lv=lv, name=name, up=up, cat=cat, gate=gate, ra=ra, rb=rb, g_in=g_in, g_c=g_c, g_bn=g_bn, Co=Co,
                            hi=hi, wi=wi, ho=ho, wo=wo, Mo=Mo, out=yb))
            g_in, g_c = (rb["z"] if up_bnin else yb), Co
            g_bn = (rb["w"]["scale"], rb["w"]["shift"]) if up_bnin else None
        oc = st.convs["out_conv"]
        fused_head = g_in is None
        f.label = "out_conv"
        if not fused_head:
            f.add("aau_outconv_fwd", g_in, c, oc.w, oc.bias, self.logits, Ms[0], c)
        if not tr:
            return

        # =============================== backward ===============================
        b = self.bwd
        mark = self._mark
        rep_ws = self.rep_ws = self.new(STAT_REPLICAS * (max(Cs) + 8), dtype=F32)   # replica scratch of the column reductions
        dy = None
        b.label = "out_conv"
        if not fused_head:
            dy = self.new(Ms[0], c)
            b.add("aau_outconv_bwd", g_in, c, self.dlogits, oc.w, dy, c, oc.dw, oc.dbias, rep_ws, Ms[0], c)
        # gradient of the encoder outputs: its own buffer behind a gate, the lower half of dcat otherwise
        dskip = [None if gate_kinds[lv] is None else self.new(Ms[lv], Cs[lv]) for lv in range(4)]
        dskip_p = [cat_p[lv] if gate_kinds[lv] is None else Cs[lv] for lv in range(4)]
        pre_wg = []                          # weight gradients deferred into the bridge's grouped launch
        for blk in reversed(dec):            # u1, u2, u3, u4
            lv, Co, Mo, ho, wo, hi, wi = blk["lv"], blk["Co"], blk["Mo"], blk["ho"], blk["wo"], blk["hi"], blk["wi"]
            dya = self.new(Mo, Co)
            self.cbr_bwd(blk["rb"], dy, Co, din=dya, dinp=Co, red_next=blk["ra"])
            dcat = self.new(2, Mo, Co) if planar[lv] else self.new(Mo, 2 * Co)
            dcat_hi = hi_(dcat, lv)                # gradient of the transposed conv's output, pixel pitch cat_p[lv]
            # channel sums of dcat (fp32, from the data-gradient epilogue) -> ConvTranspose2d bias gradient below
            sA = None if eng.no_fuse_colsum else self.bstats_arena.take(ops.stat_words(2 * Co))
            self.cbr_bwd(blk["ra"], dya, Co, din=dcat, dinp=cat_p[lv], din_stats=sA, din_split=cat_split[lv])
            if gate_kinds[lv] is None:
                dskip[lv] = lo(dcat, lv)
            cat, gt, up = blk["cat"], blk["gate"], blk["up"]
            if gt is not None and gt["kind"] == "res":
                b.label = f"{blk['name']}.att"
                Fi = gt["Fi"]
                wg, wx, psi = gt["wg"], gt["wx"], gt["psi"]
                ds = self.new(Mo, Fi)
                b.add("aau_gate2_bwd", dcat, 2 * Co, skips[lv], skip_p[lv], gt["alpha"], gt["zg"], gt["zx"], psi.w,
                      dskip[lv], Co, ds, self.red_ws, psi.dw, psi.dbias, Mo, Fi, Co)
                b.add_wgrad(ops.conv_desc(B, ho, wo, Co, 2 * Co, ho, wo, Fi, Fi), cat[:, Co:], ds, wg.dw)
                b.add_wgrad(ops.conv_desc(B, ho, wo, Co, skip_p[lv], ho, wo, Fi, Fi), skips[lv], ds, wx.dw)
                sB = None if sA is None else self.bstats_arena.take(ops.stat_words(Co))
                b.add("aau_conv_igemm", ops.conv_desc(B, ho, wo, Fi, Fi, ho, wo, Co, 2 * Co, Cpad=wg.cpad_d,
                                                      accumulate=1), ds, wg.pk_d, dcat[:, Co:], None, None, None, sB)
                b.add("aau_conv_igemm", ops.conv_desc(B, ho, wo, Fi, Fi, ho, wo, Co, Co, Cpad=wx.cpad_d,
                                                      accumulate=1), ds, wx.pk_d, dskip[lv], None, None, None, None)
            elif gt is not None:
                b.label = f"{blk['name']}.att"
                Fi = gt["Fi"]
                w1, wgb, wxb = gt["w1"], gt["wgb"], gt["wxb"]
                dq = self.new(Mo, dtype=F32)
                ds, dzg, dzx = self.new(Mo, Fi), self.new(Mo, Fi), self.new(Mo, Fi)
                red1 = w1["red"]                         # fp32 [4]: totals of step 1
                tot = self.red_arena.take(4 * Fi)        # fp32 [4][Fi]: totals of step 2
                b.add("aau_gate_bwd1", dcat, 2 * Co, skips[lv], skip_p[lv], gt["alpha"], gt["psi_pre"], w1["mean"],
                      w1["invstd"], dskip[lv], Co, dq, red1, Mo, Co, self.red_ws)
                b.add("aau_gate_bwd2", dq, gt["psi_pre"], red1, gt["b1"].gamma, w1["mean"], w1["invstd"], gt["zg"],
                      gt["zx"], wgb["scale"], wgb["shift"], wxb["scale"], wxb["shift"], wgb["mean"], wgb["invstd"],
                      wxb["mean"], wxb["invstd"], gt["psi"].w, ds, tot, gt["b1"].dgamma, gt["b1"].dbeta, Mo, Fi,
                      self.red_ws)
                b.add("aau_gate_bwd3", ds, gt["zg"], gt["zx"], gt["bg"].gamma, wgb["mean"], wgb["invstd"],
                      gt["bx"].gamma, wxb["mean"], wxb["invstd"], tot, dzg, dzx, gt["bg"].dgamma,
                      gt["bg"].dbeta, gt["bx"].dgamma, gt["bx"].dbeta, gt["psi"].dw, Mo, Fi)
                wg, wx = gt["wg"], gt["wx"]
                ov = eng.overlap_wgrad
                if ov:
                    b.fork()
                b.add_wgrad(ops.conv_desc(B, ho, wo, Co, 2 * Co, ho, wo, Fi, Fi), cat[:, Co:], dzg, wg.dw, side=ov)
                b.add_wgrad(ops.conv_desc(B, ho, wo, Co, skip_p[lv], ho, wo, Fi, Fi), skips[lv], dzx, wx.dw, side=ov)
                sB = None if sA is None else self.bstats_arena.take(ops.stat_words(Co))
                b.add("aau_conv_igemm", ops.conv_desc(B, ho, wo, Fi, Fi, ho, wo, Co, 2 * Co, Cpad=wg.cpad_d,
                                                      accumulate=1), dzg, wg.pk_d, dcat[:, Co:], None, None, None,
                      sB)
                b.add("aau_conv_igemm", ops.conv_desc(B, ho, wo, Fi, Fi, ho, wo, Co, Co, Cpad=wx.cpad_d,
                                                      accumulate=1), dzx, wx.pk_d, dskip[lv], None, None, None, None)
            # ConvTranspose2d backward: bias, weight, input
            gsrc, gc = blk["g_in"], blk["g_c"]
            b.label = f"{blk['name']}.up"
            if sA is None:
                b.add("aau_colsum", dcat_hi, cat_p[lv], up.dbias, rep_ws, Mo, Co)
            else:
                # channel sums of dcat[:, Co:] (+ what the gate's data gradient added to it): one launch for both
                sb_ = sB if gt is not None else None
                b.add("aau_fold_stats_pair", sA, ops.stat_words(2 * Co) * 8, 2 * Co, Co, sb_, ops.stat_words(Co) * 8 if sb_ is not None else 0,
                      Co, 0, Co, up.dbias)
            ov = eng.overlap_wgrad
            if ov:
                b.fork()        # dcat[:, Co:] is final (gate data-gradient accumulated above)
            upd = ops.conv_desc(B, ho, wo, Co, cat_p[lv], hi, wi, gc, gc, 2, 2, 2, 0, 1)
            if lv == 3 and aspp and not eng.no_wgrad_group and not ov and ops.conv_wgrad_group_ok([upd], lone_ok=True):
                # the deepest ConvTranspose2d's weight gradient (32 big tiles over the same 8192 pixels as the bridge's)
                # rides along in the bridge's grouped launch below instead of half-filling the chip on its own
                pre_wg.append((upd, dcat_hi, gsrc, up.dw, b.label))
            else:
                b.add_wgrad(upd, dcat_hi, gsrc, up.dw, side=ov, dz_bn=blk["g_bn"])
            dg_in = self.new(B * hi * wi, gc)
            b.add("aau_conv_igemm", ops.conv_desc(B, ho, wo, Co, cat_p[lv], hi, wi, gc, gc, 2, 2, 2, 0, 1, up.cpad_d),
                  dcat_hi, up.pk_d, dg_in, None, None, None, None)
            dy = dg_in
            mark(blk["name"])
        # bridge
        dp4 = self.new(M5, Cs[3])
        if aspp:
            dcat5 = self.new(M5, ncat)
            # the weight gradients of the projection and of the spatial branches go into ONE grouped launch
            # (aau_conv_wgrad_group) once every branch's dz exists; their data gradients run as before
            wg = list(pre_wg) if not eng.no_wgrad_group else None
            # ... and the branches' data gradients, which all add into dL/dx of the bridge input, into ONE grouped
            # launch as well (aau_conv_igemm_group): the sum stays in registers across the branches
            dg = []
            self.cbr_bwd(rproj, dy, Cb, din=dcat5, dinp=ncat, defer_wgrad=wg)
            dzs = [None] * nbr
            if getattr(self, "bn_multi_fwd", False):
                # the branches' BatchNorm backward as one reduce (+ fold) and one apply launch for all of them
                b.label = "bridge(multi)"
                dzs = [self.new(M5, Cb) for _ in range(nbr)]
                rws = self.new(nbr, 2 * Cb * 1024, dtype=F32)
                dys = [dcat5[:, i * Cb:] for i in range(nbr)]
                tabR = ops.ptr_table([[r["z"], dys[i], r["w"]["scale"], r["w"]["shift"], r["w"]["mean"], r["w"]["invstd"],
                                       r["w"]["red"], rws[i]] for i, r in enumerate(br)])
                tabP = ops.ptr_table([[r["z"], dzs[i], r["bn"].gamma, r["w"]["mean"], r["w"]["invstd"], r["w"]["red"],
                                       r["bn"].dgamma, r["bn"].dbeta, dys[i], r["w"]["scale"], r["w"]["shift"]]
                                      for i, r in enumerate(br)])
                b.keep.extend([tabR, tabP, rws] + dys)
                b.add("aau_bn_bwd_reduce_multi", nbr, tabR, Cb, ncat, B, h5, w5, Cb, 1)
                b.add("aau_bn_bwd_apply_multi", nbr, tabP, Cb, Cb, ncat, M5, Cb, 1)
            for i, r in enumerate(br):
                self.cbr_bwd(r, dcat5[:, i * Cb:], ncat, din=dp4, dinp=Cs[3], accumulate=1 if i > 0 else 0, defer_wgrad=wg,
                             defer_dgrad=dg, dz_given=dzs[i])
            # The grouped weight gradient is latency-bound (one or two workgroups per CU, each waiting ~2 us for its next
            # K-step: scripts/probes/wl_sched.py) and nothing needs its result before the gradient bucket closes.  Opt-in
            # (AAU_BRIDGE_WG_SIDE=1): on the side stream beside the grouped data gradient and the image-pool branch --
            # measured +0.14 ms on the step (same-box A/B, 3 pairs), like every other two-stream form tried on this chip.
            wg_side = False
            if wg:
                descs = [w_[0] for w_ in wg]
                b.label = "bridge(grouped)"
                if len(wg) <= 8 and ops.conv_wgrad_group_ok(descs):
                    pack = ops.wgrad_group_args(descs, [w_[1] for w_ in wg], [w_[2] for w_ in wg], [w_[3] for w_ in wg])
                    b.keep.extend([w_[k] for w_ in wg for k in (1, 2, 3)])
                    b.keep.append(pack)
                    wg_side = eng.bridge_wg_side
                    if wg_side:
                        b.fork()
                    # heads of the launch's work queues: words of the arena that begin_backward clears every step
                    qws = self.red_arena.take(ops.wgrad_group_queue_words())
                    b.add("aau_conv_wgrad_group", *pack, qws, qws.numel() * 4, side=wg_side)
                else:
                    for dwd, src, dz_, dw_, lab in wg:
                        b.label = lab
                        b.add_wgrad(dwd, src, dz_, dw_)
            dgd = [g_[0] for g_ in dg]
            if len(dg) >= 2 and ops.conv_igemm_group_ok(dgd):
                b.label = "bridge(grouped)"
                nws = ops.conv_igemm_group_ws_bytes(dgd) // 4
                gws = self.new(nws, dtype=torch.float32) if nws else None
                pack = ops.igemm_group_args(dgd, [g_[1] for g_ in dg], [g_[2] for g_ in dg], dp4, gws)
                b.keep.extend([g_[k] for g_ in dg for k in (1, 2)])
                b.keep.append(pack)
                b.add("aau_conv_igemm_group", pack[0], pack[1], pack[2], pack[3], dp4, gws)
            else:
                for dd_, dz_, pk_, din_, lab in dg:
                    b.label = lab
                    b.add("aau_conv_igemm", dd_, dz_, pk_, din_, None, None, None, None)
            dpb = self.new(B, Cb)
            b.label = "bridge.pool"
            b.add("aau_spatial_sum", dcat5[:, nbr * Cb:], ncat, dpb, gap_ws, B, h5 * w5, Cb)
            dpooled = self.new(B, Cs[3])
            if rpool.get("fused"):
                cvp, bnp, wp = rpool["cv"], rpool["bn"], rpool["w"]
                dzp = self.new(B, Cb)
                b.add("aau_poolbranch_bwd", dpb, Cb, rpool["z"], rpool["src"], Cs[3], bnp.gamma, wp["scale"], wp["shift"],
                      wp["mean"], wp["invstd"], dzp, bnp.dgamma, bnp.dbeta, cvp.dw, B, Cs[3], Cb)
                b.add("aau_poolbranch_dx", dzp, cvp.pk_d, cvp.cpad_d, dpooled, Cs[3], B, Cs[3], Cb)
            else:
                rpool_b = dict(rpool)
                rpool_b["bcast_hw"] = 0
                self.cbr_bwd(rpool_b, dpb, Cb, din=dpooled, dinp=Cs[3])
            b.label = "bridge.pool"
            b.add("aau_gap_bwd", dpooled, dp4, Cs[3], B, h5 * w5, Cs[3])
            if wg_side:
                b.join()
        else:
            self.cbr_bwd(rplain, dy, Cb, din=dp4, dinp=Cs[3])
        mark("bridge")
        # encoder
        dpool = dp4
        for lv in (3, 2, 1, 0):
            ra, rb = enc[lv]
            dsk, dskp = dskip[lv], dskip_p[lv]
            dya = self.new(Ms[lv], Cs[lv])
            self.cbr_bwd(rb, dsk, dskp, dpool=dpool, dpp=Cs[lv], din=dya, dinp=Cs[lv], red_next=ra)
            if lv > 0:
                dprev = self.new(Ms[lv], Cs[lv - 1])
                self.cbr_bwd(ra, dya, Cs[lv], din=dprev, dinp=Cs[lv - 1])
                dpool = dprev
            else:
                self.cbr_bwd(ra, dya, Cs[lv])
            mark(f"d{lv + 1}")

    def _mark(self, name):
        cb = self.eng.bucket_callback(name)
        if cb is not None:
            self.bwd.join()          # the bucket's weight gradients run on the side stream
            self.bwd.callback(cb)
        elif name == "d1":
            self.bwd.join()          # end of backward: everything is back on the main stream

    def psi_outputs(self):
        """[psi3, psi2] of test_ablation.py:218: the attention maps of u4 and u3 ([B,1,h,w]); a block without a gate
        gives the DummyAttention placeholder zeros(1,1,1,1)."""
        out = []
        for lv in (3, 2):
            if lv in self.psis:
                a, h, w = self.psis[lv]
                out.append(a.view(self.B, 1, h, w).clone())
            else:
                out.append(torch.zeros(1, 1, 1, 1, device=self.dev))
        return out

    # ---- execution ----
    def run_forward(self, x: torch.Tensor):
        stream = torch.cuda.current_stream().cuda_stream
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x.reshape(self.x.shape), non_blocking=True)
        if self.train:
            self.fwd_gen += 1
            # one launch: the statistic accumulators cleared, the dropout seed advanced by an odd increment mod 2^64
            # (a new mask every step)
            ops.zero_multi(self.stats_arena.bufs, self.drop_seed, 0x9E3779B97F4A7C15)
        self.fwd.run(stream)
        return self.logits

    def begin_backward(self, dlogits: torch.Tensor | None = None):
        st = self.eng.store
        if dlogits is not None and dlogits.data_ptr() != self.dlogits.data_ptr():
            self.dlogits.copy_(dlogits.reshape(self.dlogits.shape), non_blocking=True)
        ops.zero_multi(self.red_arena.bufs + self.bstats_arena.bufs + [st.gflat])

    def run_backward(self, dlogits: torch.Tensor | None):
        stream = torch.cuda.current_stream().cuda_stream
        self.begin_backward(dlogits)
        self.bwd.run(stream)
        self.eng.store.bind_grads()


class Engine:
    """Owns the flat parameter store and the plans of one model instance."""

    def __init__(self, model: nn.Module):
        self.model = model
        self.store: ParamStore | None = None
        self.plans: dict = {}
        self.precision = "bf16"  # 16-bit storage type of the launch lists: "bf16", or "fp16" (inference only)
        self._seed = None       # dropout seed chain: drawn from torch's RNG (and the DP rank) on first use
        self.bucket_cb = None   # set by the data-parallel wrapper: name -> callable
        self.dp = None          # the data-parallel wrapper itself (gradient accumulation is finished by it: model._NetFn)
        import os
        self.overlap_wgrad = os.environ.get("AAU_OVERLAP_WGRAD", "0") == "1"   # measured null on MI355X (A/B, same device)
        self.no_fuse_conv1 = os.environ.get("AAU_NO_FUSE_CONV1", "0") == "1"   # experiment switches
        self.no_fuse_colsum = os.environ.get("AAU_NO_FUSE_COLSUM", "0") == "1"
        self.no_fuse_head = os.environ.get("AAU_NO_FUSE_HEAD", "0") == "1"
        self.no_wgrad_group = os.environ.get("AAU_NO_WGRAD_GROUP", "0") == "1"
        self.no_fuse_bnred = os.environ.get("AAU_NO_BNRED", "0") == "1"
        self.no_bnin = os.environ.get("AAU_NO_BNIN", "0") == "1"     # A/B: ConvBNReLU pairs with the activation in memory
        self.no_bn_multi = os.environ.get("AAU_NO_BN_MULTI", "0") == "1"   # A/B: one BatchNorm launch per ASPP branch
        self.no_bnin_up = os.environ.get("AAU_NO_BNIN_UP", "0") == "1"    # A/B: the activation in front of a transposed conv stays in memory
        self.bnin_max_c = int(os.environ.get("AAU_BNIN_MAX_C", "96"))     # A/B: widest producing layer whose BN + ReLU moves onto the operand
        # opt-in (measured +0.04 ms on the step): the pooled layers' apply pass redoes the max-pool routing instead of
        # reading the routed gradient the reduce pass stored
        self.pool_store_routed = os.environ.get("AAU_POOL_APPLY_ROUTES", "0") != "1"
        self.bridge_wg_side = os.environ.get("AAU_BRIDGE_WG_SIDE", "0") == "1"   # opt-in, measured negative
        self.no_igemm_multi = os.environ.get("AAU_NO_IGEMM_MULTI", "0") == "1"   # A/B: one launch per ASPP branch
        self.no_poolbranch = os.environ.get("AAU_NO_POOLBRANCH", "0") == "1"   # A/B: the generic launches for bridge.pool
        # z of the first layer recomputed from the frame instead of stored (-201 MB of HBM at bs 8 / 512^2): measured
        # 0.08 ms SLOWER per step (the three recomputing kernels are VALU / latency bound, not byte bound), so opt-in
        self.no_recompute_z1 = os.environ.get("AAU_RECOMPUTE_Z1", "0") != "1" or self.no_fuse_conv1

    def dropout_p(self) -> float:
        """p of the bridge's Dropout: ASPP.project[3] (pipeline:78) or Sequential(ConvBNReLU, Dropout)[1] (ablation:196)."""
        for m in self.model.bridge.modules():
            if isinstance(m, nn.Dropout):
                return float(m.p)
        return 0.0

    def gate_kind(self, name: str):
        """'bn': pipeline:85-92 (BatchNorm gate, x*a); 'res': ablation:128-143 (no BN, x*a + x); None: DummyAttention."""
        att = getattr(self.model, name).att
        if hasattr(att, "Wg"):
            return "bn" if isinstance(att.Wg, nn.Sequential) else "res"
        return None

    def next_seed(self) -> int:
        """Seeds of the Dropout masks follow ``torch.manual_seed`` (pipeline:33-52 ``set_seed``): the chain starts from
        torch's seed mixed with the data-parallel rank, so two seeds -- or two ranks -- give different
        mask streams and the same seed reproduces them (``initial_seed`` is read, the generator state is not consumed)."""
        if self._seed is None:
            import torch.distributed as dist
            base = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            self._seed = (base ^ (0x9E3779B97F4A7C15 * (rank + 1))) & 0xFFFFFFFFFFFFFFFF
        self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self._seed

    def bucket_callback(self, name):
        if self.bucket_cb is None:
            return None
        return lambda n=name: self.bucket_cb(n)

    def set_precision(self, kind: str):
        """"bf16" (default; training and inference) or "fp16" (IEEE half, the reference's fp16 inference configuration
        ``pipeline:320,437``: forward only -- there is no loss scaling here, training stays bf16)."""
        if kind not in _abi.LIB_PATHS:
            raise _abi.AauError(f"unknown precision {kind!r} (bf16 | fp16)")
        if kind != self.precision:
            _abi.lib(kind)          # fail loudly if that build of the library is missing
            self.precision = kind   # plans are keyed by it: launch lists are recorded against one build of the library

    @property
    def adt(self):
        """torch dtype of the 16-bit activation tensors of this engine's plans."""
        return torch.float16 if self.precision == "fp16" else BF16

    def ensure(self, device):
        _abi.lib(self.precision)  # fail loudly if the HIP library is missing
        if device.type != "cuda":
            raise _abi.AauError("the MI355X path needs CUDA/HIP tensors; there is no CPU fallback "
                                "(the CPU restatement lives in oracle/ and is test infrastructure only)")
        if self.store is None or self.store.device != device or not self.store.intact() \
                or not self.store.buffers_intact(self.model):
            first = next(self.model.parameters())
            if first.device != device:
                raise _abi.AauError(f"model parameters are on {first.device}, input on {device}")
            old = self.store
            self.store = ParamStore(self.model, device)
            if old is not None and old.m is not None and old.total == self.store.total:
                self.store.m, self.store.v, self.store.step_dev = old.m, old.v, old.step_dev
            self.plans.clear()
        return self.store

    def plan(self, B, H, W, train) -> Plan:
        key = (B, H, W, bool(train), self.dropout_p() if train else 0.0,
               self.bucket_cb is not None, self.overlap_wgrad, self.precision)
        p = self.plans.get(key)
        if p is None:
            if train and self.precision != "bf16":
                raise _abi.AauError("fp16 is an inference precision here (no loss scaling): call model.eval() or "
                                    "set_precision('bf16') to train")
            with _abi.precision(self.precision):
                p = Plan(self, B, H, W, bool(train))
            self.plans[key] = p
        return p
