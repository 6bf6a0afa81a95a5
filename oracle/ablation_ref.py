"""TEST INFRASTRUCTURE -- CPU (ATen fp32) restatement of the ablation variant of the network,
test_ablation.py:73-218 (ConvBNReLU :73-84, ASPP :86-126, AttentionGate :128-143, DummyAttention :145-147,
UpBlock :149-166, AttentionASPPUNet :168-218).  Pinned by tests/golden/g6_ablation.npz, which
oracle/make_golden_ablation.py produced by importing the reference file itself in the build container.
Only tests/ may import this module; the product (att-aspp-unet_amd/ablation.py) never does."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_cpu import ASPP, ConvBNReLU


class AttentionGate(nn.Module):
    def __init__(self, Fg, Fl, Fint=None):
        super().__init__()
        if Fint is None:
            Fint = max(8, min(Fg, Fl) // 4)
        self.Wg = nn.Conv2d(Fg, Fint, 1, bias=False)
        self.Wx = nn.Conv2d(Fl, Fint, 1, bias=False)
        self.psi = nn.Sequential(nn.ReLU(True), nn.Conv2d(Fint, 1, 1, bias=True), nn.Sigmoid())

    def forward(self, g, x):
        a = self.psi(self.Wg(g) + self.Wx(x))
        return x * a + x, a


class DummyAttention(nn.Module):
    def forward(self, g, x):
        return x, torch.zeros(1, 1, 1, 1, device=x.device)


class UpBlock(nn.Module):
    def __init__(self, in_c, out_c, use_att=True):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_c, out_c, 2, 2)
        self.att = AttentionGate(out_c, out_c) if use_att else DummyAttention()
        self.conv = nn.Sequential(ConvBNReLU(in_c, out_c), ConvBNReLU(out_c, out_c))

    def forward(self, g, x):
        g = self.up(g)
        if g.shape[-2:] != x.shape[-2:]:
            g = F.interpolate(g, size=x.shape[-2:], mode="bilinear", align_corners=False)
        x_att, psi = self.att(g, x)
        return self.conv(torch.cat([x_att, g], 1)), psi


class AttentionASPPUNet(nn.Module):
    def __init__(self, in_channels=1, num_classes=1, base_c=32, use_att=True, use_aspp=True, att_depth=4):
        super().__init__()
        c = base_c
        self.d1 = nn.Sequential(ConvBNReLU(in_channels, c), ConvBNReLU(c, c)); self.p1 = nn.MaxPool2d(2)
        self.d2 = nn.Sequential(ConvBNReLU(c, 2 * c), ConvBNReLU(2 * c, 2 * c)); self.p2 = nn.MaxPool2d(2)
        self.d3 = nn.Sequential(ConvBNReLU(2 * c, 4 * c), ConvBNReLU(4 * c, 4 * c)); self.p3 = nn.MaxPool2d(2)
        self.d4 = nn.Sequential(ConvBNReLU(4 * c, 8 * c), ConvBNReLU(8 * c, 8 * c)); self.p4 = nn.MaxPool2d(2)
        self.bridge = ASPP(8 * c, 16 * c) if use_aspp else nn.Sequential(ConvBNReLU(8 * c, 16 * c, 3), nn.Dropout(0.1))
        self.u4 = UpBlock(16 * c, 8 * c, use_att and att_depth >= 4)
        self.u3 = UpBlock(8 * c, 4 * c, use_att and att_depth >= 3)
        self.u2 = UpBlock(4 * c, 2 * c, False)
        self.u1 = UpBlock(2 * c, c, False)
        self.out_conv = nn.Conv2d(c, num_classes, 1)

    def forward(self, x):
        x1 = self.d1(x)
        x2 = self.d2(self.p1(x1))
        x3 = self.d3(self.p2(x2))
        x4 = self.d4(self.p3(x3))
        b = self.bridge(self.p4(x4))
        d4, psi3 = self.u4(b, x4)
        d3, psi2 = self.u3(d4, x3)
        d2, _ = self.u2(d3, x2)
        d1, _ = self.u1(d2, x1)
        return self.out_conv(d1), [psi3, psi2]


def dropout_module(net):
    for m in net.bridge.modules():
        if isinstance(m, nn.Dropout):
            return m
    return None


VARIANTS = {"full": {}, "noatt": dict(use_att=False), "noaspp": dict(use_aspp=False),
            "plain": dict(use_att=False, use_aspp=False), "depth3": dict(att_depth=3)}


def emulate_bf16_storage(net):
    """ref_cpu.emulate_bf16_storage (round where the MI355X path stores bf16) plus the gated skip of the residual gate."""
    from . import ref_cpu
    handles = ref_cpu.emulate_bf16_storage(net)
    for m in net.modules():
        if isinstance(m, AttentionGate):
            handles.append(m.register_forward_hook(lambda mod, inp, out: (ref_cpu._RoundSTE.apply(out[0]), out[1])))
    return handles
