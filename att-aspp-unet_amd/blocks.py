"""Standalone (single-block) forward execution of the reference's building blocks on the HIP kernels:
``ConvBNReLU`` (pipeline:59-65), ``ASPP`` (:67-83), ``AttentionGate`` (:85-92), ``UpBlock`` (:98-109).

The training path never goes through here -- ``AttentionASPPUNet`` replays a whole-network plan
(engine.py).  This module exists so that the block classes keep their reference call signature
(``block(x)``, ``gate(g, x)``, ``up(g, x)``: NCHW fp32 in, NCHW fp32 out) for inspection, unit tests and
feature extraction.  Forward only: BatchNorm uses batch statistics (and updates the running statistics)
in ``train()`` mode and running statistics in ``eval()`` mode; no autograd graph is built.
"""
from __future__ import annotations

import torch

from . import _abi, ops

BF16, F32 = torch.bfloat16, torch.float32


def _check(x, c):
    if x.device.type != "cuda":
        raise _abi.AauError("block execution needs HIP tensors (no CPU fallback)")
    if x.dim() != 4 or x.shape[1] != c:
        raise _abi.AauError(f"expected [B,{c},H,W], got {tuple(x.shape)}")
    if c % 8 != 0:
        raise _abi.AauError("standalone blocks need channel counts that are multiples of 8")


def to_nhwc(x):
    B, C_, H, W = x.shape
    out = torch.empty(B, H, W, C_, dtype=BF16, device=x.device)
    ops.nchw_to_nhwc(x.float().contiguous(), out, C_, B, C_, H, W)
    return out


def to_nchw(t):
    B, H, W, C_ = t.shape
    out = torch.empty(B, C_, H, W, dtype=F32, device=t.device)
    ops.nhwc_to_nchw(t, ops.pitch_of(t), out, B, C_, H, W)
    return out


def _pack_conv(w):
    """OIHW fp32 -> bf16 [O][T][Cpad] (layout plumbing of a handful of weights; the engine uses aau_pack_weights)."""
    O, I, kh, kw = w.shape
    cp = ops.cpad_of(I)
    out = torch.zeros(O, kh * kw, cp, dtype=BF16, device=w.device)
    out[:, :, :I] = w.detach().permute(0, 2, 3, 1).reshape(O, kh * kw, I).to(BF16)
    return out, cp


def _bn_scale_shift(bn, stats, count, training):
    C_ = bn.num_features
    dev = bn.weight.device
    scale, shift = torch.empty(C_, device=dev), torch.empty(C_, device=dev)
    if training:
        mean, invstd = torch.empty(C_, device=dev), torch.empty(C_, device=dev)
        ops.bn_finalize(stats, bn.weight.detach().contiguous(), bn.bias.detach().contiguous(), bn.running_mean,
                        bn.running_var, bn.num_batches_tracked, scale, shift, mean, invstd, C_, count)
    else:
        ops.bn_fold_eval(bn.weight.detach().contiguous(), bn.bias.detach().contiguous(), bn.running_mean,
                         bn.running_var, scale, shift, C_)
    return scale, shift


def conv_bn(conv, bn, x, out=None, out_pitch=None, relu=True, training=True, bcast_hw=0):
    """x: NHWC bf16 (view with pitch) -> y NHWC bf16 = [relu](bn(conv(x)))."""
    B, H, W, Cin = x.shape
    O = conv.out_channels
    k, dil = conv.kernel_size[0], conv.dilation[0]
    wpk, cp = _pack_conv(conv.weight)
    M = B * H * W
    z = torch.empty(B, H, W, O, dtype=BF16, device=x.device)
    stats = ops.stats_buffer(O, x.device) if training else None
    d = ops.conv_desc(B, H, W, Cin, ops.pitch_of(x), H, W, O, O, k, k, 1, dil * (k // 2), dil, cp)
    ops.conv_igemm(d, x, wpk, z, stats=stats)
    scale, shift = _bn_scale_shift(bn, stats, M, training)
    if out is None:
        Mo = M * bcast_hw if bcast_hw else M
        out = torch.empty(Mo, O, dtype=BF16, device=x.device)
        out_pitch = O
    ops.bn_act(z, O, out, out_pitch, scale, shift, M * bcast_hw if bcast_hw else M, O, 1 if relu else 0, bcast_hw)
    return out, z, scale, shift


def convbnrelu_forward(mod, x):
    conv, bn = mod.block[0], mod.block[1]
    _check(x, conv.in_channels)
    xh = to_nhwc(x)
    y, _, _, _ = conv_bn(conv, bn, xh, training=mod.training)
    B, _, H, W = x.shape
    return to_nchw(y.view(B, H, W, conv.out_channels))


def aspp_forward(mod, x):
    in_c = mod.blocks[0][0].in_channels
    _check(x, in_c)
    B, _, H, W = x.shape
    xh = to_nhwc(x)
    Cb = mod.blocks[0][0].out_channels
    nbr = len(mod.blocks)
    ncat = (nbr + 1) * Cb
    cat = torch.empty(B * H * W, ncat, dtype=BF16, device=x.device)
    for i, blk in enumerate(mod.blocks):
        conv_bn(blk[0], blk[1], xh, out=cat[:, i * Cb:], out_pitch=ncat, training=mod.training)
    pooled = torch.empty(B, 1, 1, in_c, dtype=BF16, device=x.device)
    ops.gap_fwd(xh, in_c, pooled, torch.empty(ops.GAP_WS_ROWS, B, in_c, device=x.device), B, H * W, in_c)
    conv_bn(mod.pool[1], mod.pool[2], pooled, out=cat[:, nbr * Cb:], out_pitch=ncat, training=mod.training,
            bcast_hw=H * W)
    y, _, _, _ = conv_bn(mod.project[0], mod.project[1], cat.view(B, H, W, ncat), training=mod.training)
    if mod.training and mod.project[3].p > 0:
        raise _abi.AauError("standalone ASPP in train() mode needs project[3].p == 0 (dropout belongs to the network plan)")
    return to_nchw(y.view(B, H, W, Cb))


def _gate(mod, gh, xh, out, out_pitch, training):
    """gh, xh: NHWC bf16 views; writes x*alpha into out (pitch out_pitch)."""
    B, H, W, C_ = xh.shape
    M = B * H * W
    Fi = mod.Wg[0].out_channels
    dev = xh.device

    def branch(seq, src):
        wpk, cp = _pack_conv(seq[0].weight)
        z = torch.empty(M, Fi, dtype=BF16, device=dev)
        st = ops.stats_buffer(Fi, dev) if training else None
        d = ops.conv_desc(B, H, W, src.shape[-1], ops.pitch_of(src), H, W, Fi, Fi, Cpad=cp)
        ops.conv_igemm(d, src, wpk, z, stats=st)
        sc, sh = _bn_scale_shift(seq[1], st, M, training)
        return z, sc, sh

    zg, sg, hg = branch(mod.Wg, gh)
    zx, sx, hx = branch(mod.Wx, xh)
    psi_pre = torch.empty(M, device=dev)
    st1 = ops.stats_buffer(1, dev) if training else None
    ops.gate_psi(zg, zx, sg, hg, sx, hx, mod.psi[0].weight.detach().reshape(-1).contiguous(), psi_pre, st1, M, Fi)
    sc1, sh1 = _bn_scale_shift(mod.psi[1], st1, M, training)
    ops.gate_apply(xh, ops.pitch_of(xh), psi_pre, sc1, sh1, None, out, out_pitch, M, C_)


def gate_forward(mod, g, x):
    _check(g, mod.Wg[0].in_channels)
    _check(x, mod.Wx[0].in_channels)
    B, C_, H, W = x.shape
    out = torch.empty(B, H, W, C_, dtype=BF16, device=x.device)
    _gate(mod, to_nhwc(g), to_nhwc(x), out, C_, mod.training)
    return to_nchw(out)


def upblock_forward(mod, g, x):
    in_c, Co = mod.up.in_channels, mod.up.out_channels
    _check(g, in_c)
    _check(x, Co)
    B, _, hi, wi = g.shape
    H, W = x.shape[2:]
    if (H, W) != (2 * hi, 2 * wi):
        raise _abi.AauError("UpBlock: the skip tensor must be exactly 2x the resolution of g "
                            "(the reference's bilinear fix-up branch, pipeline:106-107, is not on the HIP path)")
    gh, xh = to_nhwc(g), to_nhwc(x)
    cat = torch.empty(B, H, W, 2 * Co, dtype=BF16, device=x.device)
    w = mod.up.weight.detach()                                       # IOHW
    cp = ops.cpad_of(in_c)
    wp = torch.zeros(4 * Co, 1, cp, dtype=BF16, device=x.device)
    wp[:, 0, :in_c] = w.permute(2, 3, 1, 0).reshape(4 * Co, in_c).to(BF16)
    d = ops.conv_desc(B, hi, wi, in_c, in_c, hi, wi, 4 * Co, 2 * Co, Cpad=cp, shuffle2x2=1)
    ops.conv_igemm(d, gh, wp, cat[..., Co:], bias=mod.up.bias.detach().contiguous())
    if hasattr(mod.att, "Wg"):
        _gate(mod.att, cat[..., Co:], xh, cat, 2 * Co, mod.training)
    else:
        cat[..., :Co] = xh                                           # DummyAttention: identity on the skip
    ya, _, _, _ = conv_bn(mod.conv[0].block[0], mod.conv[0].block[1], cat, training=mod.training)
    yb, _, _, _ = conv_bn(mod.conv[1].block[0], mod.conv[1].block[1], ya.view(B, H, W, Co), training=mod.training)
    return to_nchw(yb.view(B, H, W, Co))
