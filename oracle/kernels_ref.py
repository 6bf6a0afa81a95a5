"""CPU fp32 references for single kernels (TEST INFRASTRUCTURE; torch CPU ops only).

Layout helpers convert between the library's NHWC bf16 tensors and the NCHW fp32
tensors that the reference call sites (attention_aspp_unet_pipeline_stage.py:63,71-78,
88-90,101,115-122) hand to ATen.
"""
import torch
import torch.nn.functional as F


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def nhwc_to_nchw(t: torch.Tensor) -> torch.Tensor:
    return t.float().permute(0, 3, 1, 2).contiguous()


def nchw_to_nhwc(t: torch.Tensor) -> torch.Tensor:
    return t.permute(0, 2, 3, 1).contiguous()


def conv_fwd(x_nhwc, w_oihw, dil=1, bias=None):
    """Conv2d with "same" padding for odd kernels (pipeline:63,71-74)."""
    k = w_oihw.shape[-1]
    y = F.conv2d(nhwc_to_nchw(x_nhwc), w_oihw.float(), bias, padding=dil * (k // 2), dilation=dil)
    return nchw_to_nhwc(y)


def conv_dgrad(dy_nhwc, w_oihw, in_hw, dil=1):
    k = w_oihw.shape[-1]
    N = dy_nhwc.shape[0]
    dx = torch.nn.grad.conv2d_input((N, w_oihw.shape[1], *in_hw), w_oihw.float(), nhwc_to_nchw(dy_nhwc),
                                    padding=dil * (k // 2), dilation=dil)
    return nchw_to_nhwc(dx)


def conv_wgrad(x_nhwc, dy_nhwc, w_shape, dil=1):
    k = w_shape[-1]
    return torch.nn.grad.conv2d_weight(nhwc_to_nchw(x_nhwc), w_shape, nhwc_to_nchw(dy_nhwc),
                                       padding=dil * (k // 2), dilation=dil)


def convT_fwd(g_nhwc, w_iohw, bias):
    """ConvTranspose2d(in, out, 2, 2) (pipeline:101)."""
    return nchw_to_nhwc(F.conv_transpose2d(nhwc_to_nchw(g_nhwc), w_iohw.float(), bias, stride=2))


def convT_dgrad(dy_nhwc, w_iohw):
    return nchw_to_nhwc(F.conv2d(nhwc_to_nchw(dy_nhwc), w_iohw.float(), None, stride=2))


def convT_wgrad(g_nhwc, dy_nhwc, w_iohw):
    g = nhwc_to_nchw(g_nhwc).requires_grad_(False)
    w = w_iohw.float().clone().requires_grad_(True)
    y = F.conv_transpose2d(g, w, None, stride=2)
    y.backward(nhwc_to_nchw(dy_nhwc))
    return w.grad


def bn_train(z_nhwc, gamma, beta, eps=1e-5):
    """Training-mode BatchNorm2d statistics and output (pipeline:64)."""
    z = z_nhwc.float()
    C = z.shape[-1]
    flat = z.reshape(-1, C)
    mean = flat.mean(0)
    var = flat.var(0, unbiased=False)
    y = (z - mean) / torch.sqrt(var + eps) * gamma + beta
    return y, mean, var
