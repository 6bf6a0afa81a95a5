"""Thin tensor-level wrappers over the C ABI (one Python function per entry point).

Tensors are device memory handles only: every function extracts ``data_ptr()`` and the
current HIP stream and calls into ``libaau.so``.  NHWC bf16 activations are passed as
torch tensors of dtype bfloat16 whose last dimension is the channel dimension; a
channel slice of a wider buffer is passed as (tensor view, pitch).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from ._abi import ConvDesc, PackEntry, STAT_REPLICAS, check, fn


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t) -> int | None:
    return None if t is None else t.data_ptr()


def _seed_ptr(seed, drop_p):
    """Dropout seeds live in device memory (a captured step must see a new one per replay).  Accepts a device int64
    tensor (the engine's) or, for one-off calls, a Python int."""
    if drop_p <= 0.0 or seed is None:
        return None, None
    if isinstance(seed, torch.Tensor):
        return seed.data_ptr(), seed
    t = torch.tensor([int(seed) - (1 << 64) if int(seed) >= (1 << 63) else int(seed)], dtype=torch.int64, device="cuda")
    return t.data_ptr(), t


def stat_words(C: int) -> int:
    """int64 words of an order-independent statistics buffer for C channels (aau.h: AAU_STAT_WORDS)."""
    return STAT_REPLICAS * 2 * C * 2 + 2


def stats_buffer(C: int, device="cuda") -> torch.Tensor:
    """Zeroed aau_stat buffer (int64) for C channels."""
    return torch.zeros(stat_words(C), dtype=torch.int64, device=device)


def _nb(t) -> int:
    """Size in bytes of a tensor handed to the C ABI (0 for None)."""
    return 0 if t is None else t.numel() * t.element_size()


def _check_stats(stats, C: int, what: str):
    """The kernels add into [R][2][C] two-limb int64 slots + a poison word: a buffer sized for anything else (e.g. the
    fp32 [R][2][C] of round 1) would be written out of bounds, so the size is checked on the host."""
    if stats is None:
        return
    if stats.dtype != torch.int64 or stats.numel() < stat_words(C) or not stats.is_contiguous():
        raise _abi.AauError(f"{what}: stats must be a contiguous int64 tensor of >= {stat_words(C)} words for {C} channels "
                            f"(ops.stats_buffer), got {stats.dtype} x {stats.numel()}")


def stats_totals(stats: torch.Tensor, C: int) -> torch.Tensor:
    """fp64 [2, C]: (sum, sum of squares) per channel of an aau_stat buffer."""
    out = torch.empty(2, C, dtype=torch.float64, device=stats.device)
    check(fn("aau_stats_to_f64")(_p(stats), _nb(stats), C, _p(out), _stream()), "aau_stats_to_f64")
    return out


def fold_stats(stats, C, which, c_begin, n, out):
    _check_stats(stats, C, "fold_stats")
    check(fn("aau_fold_stats")(_p(stats), _nb(stats), C, which, c_begin, n, _p(out), _stream()), "aau_fold_stats")


def pitch_of(t: torch.Tensor) -> int:
    """Pixel pitch (elements) of an NHWC tensor/view whose last dim is contiguous."""
    assert t.stride(-1) == 1, "channel dimension must be contiguous"
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def cpad_of(cin: int) -> int:
    """Per-tap channel count of packed weights (the kernel steps K by 64 when this is a
    multiple of 64, else by 32; channels past ``cin`` are zero)."""
    return round_up(cin, 32)


def conv_desc(N, H, W, Cin, src_pitch, Ho, Wo, Cout, dst_pitch, KH=1, KW=1, stride=1, pad=0, dil=1,
              Cpad=None, shuffle2x2=0, accumulate=0, relu=0, src_split=(0, 0), dst_split=(0, 0)) -> ConvDesc:
    """``src_split`` / ``dst_split`` = (split_c, split_off): channels >= split_c live in a second dense plane, element
    offset split_off from the base (aau.h: two-plane operands)."""
    if Cpad is None:
        Cpad = cpad_of(Cin)
    return ConvDesc(N, H, W, Cin, src_pitch, Ho, Wo, Cout, dst_pitch, KH, KW, stride, pad, dil, Cpad,
                    shuffle2x2, accumulate, relu, src_split[0], src_split[1], dst_split[0], dst_split[1])


def conv_split_ok(desc: ConvDesc, mode: int) -> bool:
    return bool(fn("aau_conv_split_ok")(C.byref(desc), mode))


def conv_igemm(desc: ConvDesc, src, wpk, dst, bias=None, scale=None, shift=None, stats=None):
    _check_stats(stats, desc.Cout, "conv_igemm")
    check(fn("aau_conv_igemm")(C.byref(desc), _p(src), _p(wpk), _p(dst), _p(bias), _p(scale), _p(shift),
                               _p(stats), _nb(stats), _stream()), "aau_conv_igemm")


def conv_is_halo3x3(desc: ConvDesc) -> bool:
    return bool(fn("aau_conv_is_halo3x3")(C.byref(desc)))


def conv_wgrad_ws_bytes(desc: ConvDesc) -> int:
    n = C.c_int64(0)
    check(fn("aau_conv_wgrad_ws_bytes")(C.byref(desc), C.byref(n)), "aau_conv_wgrad_ws_bytes")
    return int(n.value)


def conv_wgrad(desc: ConvDesc, src, dz, dw, ws=None):
    """ws: fp32 scratch of >= conv_wgrad_ws_bytes(desc) bytes (deterministic split-K) or None (fp32 atomics)."""
    nb = 0 if ws is None else ws.numel() * ws.element_size()
    check(fn("aau_conv_wgrad")(C.byref(desc), _p(src), _p(dz), _p(dw), _p(ws), nb, _stream()), "aau_conv_wgrad")


def wgrad_group_args(descs, srcs, dzs, dws):
    """ctypes argument pack of aau_conv_wgrad_group (keep the returned tuple alive while the call may run)."""
    n = len(descs)
    da = (ConvDesc * n)(*descs)
    sa = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    za = (C.c_void_p * n)(*[t.data_ptr() for t in dzs])
    wa = (C.c_void_p * n)(*[t.data_ptr() for t in dws])
    return da, sa, za, wa, n


def conv_wgrad_group_ok(descs, lone_ok: bool = False) -> bool:
    """``lone_ok``: may this ONE problem join a grouped launch (the whole group is checked again where it is emitted)."""
    n = len(descs)
    if lone_ok:
        return n == 1 and bool(fn("aau_conv_wgrad_group_member_ok")(C.byref(descs[0])))
    return bool(fn("aau_conv_wgrad_group_ok")((ConvDesc * n)(*descs), n))


def ptr_table(rows):
    """Host array of device pointers for the *_multi entry points (include/aau.h): rows of tensors (None -> NULL), row
    major.  Keep the returned object -- and the tensors -- alive while launches that use it may run."""
    flat = [0 if t is None else t.data_ptr() for row in rows for t in row]
    return (C.c_void_p * len(flat))(*flat)


def wgrad_group_queue_words() -> int:
    """int32 words of the work-queue heads of aau_conv_wgrad_group (the caller zeroes them before every call)."""
    return int(fn("aau_conv_wgrad_group_queue_bytes")()) // 4


def conv_wgrad_group(descs, srcs, dzs, dws, queue=None):
    """dw_i += weight gradient of problem i, all problems in one launch (see include/aau.h).  ``queue``: zeroed int32
    tensor of ``wgrad_group_queue_words()`` words (a fresh one is made when omitted)."""
    da, sa, za, wa, n = wgrad_group_args(descs, srcs, dzs, dws)
    if queue is None:
        queue = torch.zeros(wgrad_group_queue_words(), dtype=torch.int32, device=dws[0].device)
    check(fn("aau_conv_wgrad_group")(da, sa, za, wa, n, _p(queue), queue.numel() * queue.element_size(), _stream()),
          "aau_conv_wgrad_group")


def igemm_group_args(descs, srcs, wpks, dst, ws):
    """ctypes argument pack of aau_conv_igemm_group (keep the returned tuple alive while the call may run)."""
    n = len(descs)
    da = (ConvDesc * n)(*descs)
    sa = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    wa = (C.c_void_p * n)(*[t.data_ptr() for t in wpks])
    return da, sa, wa, n, dst, ws


def igemm_multi_args(descs, srcs, wpks, dsts, stats):
    """ctypes argument pack of aau_conv_igemm_multi (keep the returned tuple alive while the call may run)."""
    n = len(descs)
    da = (ConvDesc * n)(*descs)
    sa = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    wa = (C.c_void_p * n)(*[t.data_ptr() for t in wpks])
    oa = (C.c_void_p * n)(*[t.data_ptr() for t in dsts])
    ta = (C.c_void_p * n)(*[None if t is None else t.data_ptr() for t in stats])
    ba = (C.c_int64 * n)(*[0 if t is None else t.numel() * t.element_size() for t in stats])
    return da, sa, wa, oa, ta, ba, n


def conv_igemm_multi_ok(descs) -> bool:
    n = len(descs)
    return bool(fn("aau_conv_igemm_multi_ok")((ConvDesc * n)(*descs), n))


def conv_igemm_multi(descs, srcs, wpks, dsts, stats):
    """Problem i = conv_igemm(descs[i], srcs[i], wpks[i], dsts[i], stats=stats[i]), all in one launch (see include/aau.h)."""
    for d, t in zip(descs, stats):
        _check_stats(t, d.Cout, "conv_igemm_multi")
    pack = igemm_multi_args(descs, srcs, wpks, dsts, stats)
    check(fn("aau_conv_igemm_multi")(*pack, _stream()), "aau_conv_igemm_multi")


def conv_igemm_group_ok(descs) -> bool:
    n = len(descs)
    return bool(fn("aau_conv_igemm_group_ok")((ConvDesc * n)(*descs), n))


def conv_igemm_group_ws_bytes(descs) -> int:
    n = len(descs)
    return int(fn("aau_conv_igemm_group_ws_bytes")((ConvDesc * n)(*descs), n))


def conv_igemm_group(descs, srcs, wpks, dst, ws=None):
    """dst = (descs[0].accumulate ? dst : 0) + sum_i conv_i(srcs[i], wpks[i]) in one launch (see include/aau.h)."""
    da, sa, wa, n, _, _ = igemm_group_args(descs, srcs, wpks, dst, ws)
    need = conv_igemm_group_ws_bytes(descs)
    if need and (ws is None or ws.numel() * ws.element_size() < need):
        raise AauError(f"conv_igemm_group: workspace of {need} bytes required")
    check(fn("aau_conv_igemm_group")(da, sa, wa, n, _p(dst), _p(ws), _stream()), "aau_conv_igemm_group")


def conv_bnred_ok(desc: ConvDesc) -> bool:
    return bool(fn("aau_conv_bnred_ok")(C.byref(desc)))


def conv_igemm_bnred(desc: ConvDesc, src, wpk, dst, z, zp, scale, shift, smean, sinvstd, sums):
    """Data-gradient conv + the BatchNorm-backward sums of the layer that consumes ``dst`` (include/aau.h)."""
    _check_stats(sums, desc.Cout, "conv_igemm_bnred")
    check(fn("aau_conv_igemm_bnred")(C.byref(desc), _p(src), _p(wpk), _p(dst), _p(z), zp, _p(scale), _p(shift), _p(smean),
                                     _p(sinvstd), _p(sums), _nb(sums), _stream()), "aau_conv_igemm_bnred")


def conv_bnin_ok(desc: ConvDesc) -> bool:
    return bool(fn("aau_conv_bnin_ok")(C.byref(desc)))


def conv_igemm_bnin(desc: ConvDesc, src, in_scale, in_shift, wpk, dst, stats=None, bias=None):
    """dst = conv(relu(src * in_scale + in_shift)): the producing layer's BatchNorm + ReLU applied on the operand in LDS
    (include/aau.h: bit for bit aau_bn_act followed by aau_conv_igemm)."""
    if stats is not None:
        _check_stats(stats, desc.Cout, "conv_igemm_bnin")
    check(fn("aau_conv_igemm_bnin")(C.byref(desc), _p(src), _p(in_scale), _p(in_shift), _p(wpk), _p(dst), _p(bias), _p(stats),
                                    _nb(stats) if stats is not None else 0, _stream()), "aau_conv_igemm_bnin")


def conv_wgrad_bnin_ok(desc: ConvDesc) -> bool:
    return bool(fn("aau_conv_wgrad_bnin_ok")(C.byref(desc)))


def conv_wgrad_bnin(desc: ConvDesc, src, in_scale, in_shift, dz, dw, ws=None):
    """dw += weight gradient with relu(src * in_scale + in_shift) as the input (include/aau.h)."""
    nb = 0 if ws is None else ws.numel() * ws.element_size()
    check(fn("aau_conv_wgrad_bnin")(C.byref(desc), _p(src), _p(in_scale), _p(in_shift), _p(dz), _p(dw), _p(ws), nb, _stream()),
          "aau_conv_wgrad_bnin")


def conv_wgrad_bnin_dz_ok(desc: ConvDesc) -> bool:
    return bool(fn("aau_conv_wgrad_bnin_dz_ok")(C.byref(desc)))


def conv_wgrad_bnin_dz(desc: ConvDesc, src, dz, dz_scale, dz_shift, dw, ws=None):
    """dw += weight gradient with relu(dz * dz_scale + dz_shift) as the `dz` operand (include/aau.h: ConvTranspose2d)."""
    nb = 0 if ws is None else ws.numel() * ws.element_size()
    check(fn("aau_conv_wgrad_bnin_dz")(C.byref(desc), _p(src), _p(dz), _p(dz_scale), _p(dz_shift), _p(dw), _p(ws), nb, _stream()),
          "aau_conv_wgrad_bnin_dz")


def stats_to_red(stats, Cc, red):
    _check_stats(stats, Cc, "stats_to_red")
    check(fn("aau_stats_to_red")(_p(stats), _nb(stats), Cc, _p(red), _stream()), "aau_stats_to_red")


def conv1_fwd(x, w, z, stats, N, H, W, Cc):
    _check_stats(stats, Cc, "conv1_fwd")
    check(fn("aau_conv1_fwd")(_p(x), _p(w), _p(z), _p(stats), _nb(stats), N, H, W, Cc, _stream()), "aau_conv1_fwd")


def conv1_wgrad(x, dz, dw, N, H, W, Cc):
    check(fn("aau_conv1_wgrad")(_p(x), _p(dz), _p(dw), N, H, W, Cc, _stream()), "aau_conv1_wgrad")


def pack_weights(flat, packed, table_dev, n_entries, total_blocks):
    check(fn("aau_pack_weights")(_p(flat), _p(packed), _p(table_dev), n_entries, total_blocks, _stream()),
          "aau_pack_weights")


def zero_multi(tensors, counter=None, counter_inc=0):
    """Clears the given device tensors in one launch (groups of 8) and bumps an int64 counter tensor (aau.h: aau_zero_multi)."""
    import ctypes as C
    tensors = list(tensors)
    first = True
    while tensors or (first and counter is not None):
        chunk, tensors = tensors[:8], tensors[8:]
        ptrs = (C.c_void_p * 8)(*[t.data_ptr() for t in chunk])
        sizes = (C.c_int64 * 8)(*[t.numel() * t.element_size() for t in chunk])
        cp = counter.data_ptr() if (first and counter is not None) else None
        check(fn("aau_zero_multi")(ptrs, sizes, len(chunk), cp, int(counter_inc) & 0xFFFFFFFFFFFFFFFF, _stream()),
              "aau_zero_multi")
        first = False


def bn_finalize(stats, gamma, beta, rmean, rvar, nbt, scale, shift, smean, sinvstd, Cc, count,
                eps=1e-5, momentum=0.1):
    _check_stats(stats, Cc, "bn_finalize")
    check(fn("aau_bn_finalize")(_p(stats), _nb(stats), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(nbt), _p(scale),
                                _p(shift), _p(smean), _p(sinvstd), Cc, count, eps, momentum, _stream()),
          "aau_bn_finalize")


def bn_fold_eval(gamma, beta, rmean, rvar, scale, shift, Cc, eps=1e-5):
    check(fn("aau_bn_fold_eval")(_p(gamma), _p(beta), _p(rmean), _p(rvar), _p(scale), _p(shift), Cc, eps,
                                 _stream()), "aau_bn_fold_eval")


def bn_act(z, zp, y, yp, scale, shift, M, Cc, relu=1, bcast_hw=0, drop_p=0.0, drop_seed=0):
    sp, _keep = _seed_ptr(drop_seed, drop_p)
    check(fn("aau_bn_act")(_p(z), zp, _p(y), yp, _p(scale), _p(shift), M, Cc, relu, bcast_hw, drop_p,
                           sp, _stream()), "aau_bn_act")


def bn_act_pool(z, zp, y, yp, p, pp, scale, shift, N, H, W, Cc):
    check(fn("aau_bn_act_pool")(_p(z), zp, _p(y), yp, _p(p), pp, _p(scale), _p(shift), N, H, W, Cc, _stream()),
          "aau_bn_act_pool")


def maxpool2(y, yp, p, pp, N, H, W, Cc):
    check(fn("aau_maxpool2")(_p(y), yp, _p(p), pp, N, H, W, Cc, _stream()), "aau_maxpool2")


def bn_red_ws_bytes(Cc: int) -> int:
    """Bytes of the reusable workspace of the BatchNorm-backward reduce passes for <= Cc channels."""
    return int(_abi.lib().aau_bn_red_ws_bytes(int(Cc)))


_red_ws = {}


def bn_red_ws(Cc: int, device):
    """A process-wide workspace for eager calls (one per device, grown on demand)."""
    need = bn_red_ws_bytes(Cc) // 4
    ws = _red_ws.get(str(device))
    if ws is None or ws.numel() < need:
        ws = torch.zeros(need, dtype=torch.float32, device=device)
        _red_ws[str(device)] = ws
    return ws


def bn_bwd_reduce(z, zp, dy, dyp, dpool, dpp, dz, dzp, scale, shift, smean, sinvstd, red, N, H, W, Cc,
                  relu=1, drop_p=0.0, drop_seed=0, ws=None):
    """red: fp32 [2][C], overwritten with the totals (sum g, sum g*zhat)."""
    sp, _keep = _seed_ptr(drop_seed, drop_p)
    ws = bn_red_ws(Cc, red.device) if ws is None else ws
    check(fn("aau_bn_bwd_reduce")(_p(z), zp, _p(dy), dyp, _p(dpool), dpp, _p(dz), dzp, _p(scale), _p(shift),
                                  _p(smean), _p(sinvstd), _p(red), N, H, W, Cc, relu, drop_p, sp, _p(ws),
                                  _stream()), "aau_bn_bwd_reduce")


def bn_bwd_apply(z, zp, dz, dzp, gamma, smean, sinvstd, red, dgamma, dbeta, M, Cc, dy=None, dyp=0, scale=None,
                 shift=None, relu=1, drop_p=0.0, drop_seed=0):
    sp, _keep = _seed_ptr(drop_seed, drop_p)
    check(fn("aau_bn_bwd_apply")(_p(z), zp, _p(dz), dzp, _p(gamma), _p(smean), _p(sinvstd), _p(red),
                                 _p(dgamma), _p(dbeta), M, Cc, _p(dy), dyp, _p(scale), _p(shift), relu, drop_p,
                                 sp, _stream()), "aau_bn_bwd_apply")


def bn_bwd_apply_pool(z, zp, dz, dzp, gamma, smean, sinvstd, red, dgamma, dbeta, N, H, W, Cc, dy, dyp, dpool, dpp, scale,
                      shift, relu=1):
    """Apply pass of a pooled layer: redoes the max-pool routing (dy may be None), so bn_bwd_reduce needs no dz."""
    check(fn("aau_bn_bwd_apply_pool")(_p(z), zp, _p(dz), dzp, _p(gamma), _p(smean), _p(sinvstd), _p(red), _p(dgamma),
                                      _p(dbeta), N, H, W, Cc, _p(dy), dyp, _p(dpool), dpp, _p(scale), _p(shift), relu,
                                      _stream()), "aau_bn_bwd_apply_pool")


def bn_bwd_apply_conv1(z, zp, gamma, smean, sinvstd, red, dgamma, dbeta, N, H, W, Cc, dy, dyp, scale, shift, x, dw, ws,
                       w=None):
    """z None: recomputed from x and the conv weights ``w`` [C][9].  ws: None or >= bn_red_ws_bytes(Cc)."""
    if ws is None or ws.numel() * ws.element_size() < bn_red_ws_bytes(Cc):
        ws = bn_red_ws(Cc, dw.device)
    check(fn("aau_bn_bwd_apply_conv1")(_p(z), zp, _p(gamma), _p(smean), _p(sinvstd), _p(red), _p(dgamma), _p(dbeta),
                                       N, H, W, Cc, _p(dy), dyp, _p(scale), _p(shift), _p(x), _p(w), _p(dw), _p(ws),
                                       _stream()), "aau_bn_bwd_apply_conv1")


def conv1_bn_act(x, w, y, yp, scale, shift, N, H, W, Cc):
    check(fn("aau_conv1_bn_act")(_p(x), _p(w), _p(y), yp, _p(scale), _p(shift), N, H, W, Cc, _stream()), "aau_conv1_bn_act")


def conv1_bn_bwd_reduce(x, w, dy, dyp, scale, shift, smean, sinvstd, red, N, H, W, Cc, ws=None):
    ws = bn_red_ws(Cc, red.device) if ws is None else ws
    check(fn("aau_conv1_bn_bwd_reduce")(_p(x), _p(w), _p(dy), dyp, _p(scale), _p(shift), _p(smean), _p(sinvstd), _p(red),
                                        N, H, W, Cc, _p(ws), _stream()), "aau_conv1_bn_bwd_reduce")


def _check_gap_ws(ws, N, Cc, what):
    if ws.dtype != torch.float32 or ws.numel() < GAP_WS_ROWS * N * Cc:
        raise _abi.AauError(f"{what}: workspace must be fp32 [{GAP_WS_ROWS}][N][C] = {GAP_WS_ROWS * N * Cc} floats, got {ws.numel()}")


def gap_fwd(x, xp, pooled, ws, N, HW, Cc):
    _check_gap_ws(ws, N, Cc, "gap_fwd")
    check(fn("aau_gap_fwd")(_p(x), xp, _p(pooled), _p(ws), N, HW, Cc, _stream()), "aau_gap_fwd")


def gap_bwd(dpooled, dx, dxp, N, HW, Cc):
    check(fn("aau_gap_bwd")(_p(dpooled), _p(dx), dxp, N, HW, Cc, _stream()), "aau_gap_bwd")


def poolbranch_fwd(x, xp, wpk, cpad, z, gamma, beta, rm, rv, nbt, scale, shift, mean, invstd, B, Cin, Cout, eps=1e-5,
                   momentum=0.1):
    """Image-pool branch, conv + BatchNorm statistics over the batch (aau.h: aau_poolbranch_fwd)."""
    check(fn("aau_poolbranch_fwd")(_p(x), xp, _p(wpk), cpad, _p(z), _p(gamma), _p(beta), _p(rm), _p(rv), _p(nbt), _p(scale),
                                   _p(shift), _p(mean), _p(invstd), B, Cin, Cout, eps, momentum, _stream()), "aau_poolbranch_fwd")


def poolbranch_bwd(dy, dyp, z, x, xp, gamma, scale, shift, mean, invstd, dz, dgamma, dbeta, dw, B, Cin, Cout):
    check(fn("aau_poolbranch_bwd")(_p(dy), dyp, _p(z), _p(x), xp, _p(gamma), _p(scale), _p(shift), _p(mean), _p(invstd), _p(dz),
                                   _p(dgamma), _p(dbeta), _p(dw), B, Cin, Cout, _stream()), "aau_poolbranch_bwd")


def poolbranch_dx(dz, wpd, cpad_d, dx, dxp, B, Cin, Cout):
    check(fn("aau_poolbranch_dx")(_p(dz), _p(wpd), cpad_d, _p(dx), dxp, B, Cin, Cout, _stream()), "aau_poolbranch_dx")


def spatial_sum(src, sp, out, ws, N, HW, Cc):
    _check_gap_ws(ws, N, Cc, "spatial_sum")
    check(fn("aau_spatial_sum")(_p(src), sp, _p(out), _p(ws), N, HW, Cc, _stream()), "aau_spatial_sum")


def gate_psi(zg, zx, sg, hg, sx, hx, wpsi, psi_pre, stats, M, Fi):
    _check_stats(stats, 1, "gate_psi")
    check(fn("aau_gate_psi")(_p(zg), _p(zx), _p(sg), _p(hg), _p(sx), _p(hx), _p(wpsi), _p(psi_pre),
                             _p(stats), _nb(stats), M, Fi, _stream()), "aau_gate_psi")


def gate_apply(x, xp, psi_pre, scale1, shift1, alpha, out, op, M, Cc):
    check(fn("aau_gate_apply")(_p(x), xp, _p(psi_pre), _p(scale1), _p(shift1), _p(alpha), _p(out), op, M, Cc,
                               _stream()), "aau_gate_apply")


def gate_bwd1(dout, dop, x, xp, alpha, psi_pre, mean1, invstd1, dx, dxp, dq, red1, M, Cc, ws=None):
    """red1: fp32 [4], overwritten with (sum dq, sum dq*psihat, 0, 0)."""
    ws = bn_red_ws(Cc, red1.device) if ws is None else ws
    check(fn("aau_gate_bwd1")(_p(dout), dop, _p(x), xp, _p(alpha), _p(psi_pre), _p(mean1), _p(invstd1),
                              _p(dx), dxp, _p(dq), _p(red1), M, Cc, _p(ws), _stream()), "aau_gate_bwd1")


def gate_bwd2(dq, psi_pre, red1, gamma1, mean1, invstd1, zg, zx, sg, hg, sx, hx, mean_g, invstd_g, mean_x,
              invstd_x, wpsi, ds, tot, dgamma1, dbeta1, M, Fi, ws=None):
    """tot: fp32 [4][Fi], overwritten with (psi weight gradient, sum ds, sum ds*zhat_g, sum ds*zhat_x)."""
    ws = bn_red_ws(2 * Fi, tot.device) if ws is None else ws
    args = [dq, psi_pre, red1, gamma1, mean1, invstd1, zg, zx, sg, hg, sx, hx, mean_g, invstd_g, mean_x,
            invstd_x, wpsi, ds, tot, dgamma1, dbeta1]
    check(fn("aau_gate_bwd2")(*[_p(a) for a in args], M, Fi, _p(ws), _stream()), "aau_gate_bwd2")


def gate_bwd3(ds, zg, zx, gamma_g, mean_g, invstd_g, gamma_x, mean_x, invstd_x, tot, dzg, dzx,
              dgamma_g, dbeta_g, dgamma_x, dbeta_x, dwpsi, M, Fi):
    args = [ds, zg, zx, gamma_g, mean_g, invstd_g, gamma_x, mean_x, invstd_x, tot, dzg, dzx,
            dgamma_g, dbeta_g, dgamma_x, dbeta_x, dwpsi]
    check(fn("aau_gate_bwd3")(*[_p(a) for a in args], M, Fi, _stream()), "aau_gate_bwd3")


def outconv_fwd(y, yp, w, b, logits, M, Cc):
    check(fn("aau_outconv_fwd")(_p(y), yp, _p(w), _p(b), _p(logits), M, Cc, _stream()), "aau_outconv_fwd")


def outconv_bwd(y, yp, dlogits, w, dy, dyp, dw, db, ws, M, Cc):
    check(fn("aau_outconv_bwd")(_p(y), yp, _p(dlogits), _p(w), _p(dy), dyp, _p(dw), _p(db), _p(ws), M, Cc,
                                _stream()), "aau_outconv_bwd")


def bn_act_outconv(z, zp, scale, shift, w, b, logits, M, Cc):
    check(fn("aau_bn_act_outconv")(_p(z), zp, _p(scale), _p(shift), _p(w), _p(b), _p(logits), M, Cc, _stream()),
          "aau_bn_act_outconv")


def bn_bwd_reduce_outconv(z, zp, dlogits, w, scale, shift, smean, sinvstd, red, dw, db, ws, M, Cc):
    ws = bn_red_ws(Cc, red.device) if ws is None else ws
    if ws.numel() * ws.element_size() < bn_red_ws_bytes(Cc):
        raise _abi.AauError(f"bn_bwd_reduce_outconv: workspace of {ws.numel() * ws.element_size()} B, need {bn_red_ws_bytes(Cc)} B")
    check(fn("aau_bn_bwd_reduce_outconv")(_p(z), zp, _p(dlogits), _p(w), _p(scale), _p(shift), _p(smean), _p(sinvstd),
                                          _p(red), _p(dw), _p(db), _p(ws), M, Cc, _stream()), "aau_bn_bwd_reduce_outconv")


def bn_bwd_apply_rank1(z, zp, dz, dzp, gamma, smean, sinvstd, red, dgamma, dbeta, M, Cc, dlogits, w_out, scale, shift):
    check(fn("aau_bn_bwd_apply_rank1")(_p(z), zp, _p(dz), dzp, _p(gamma), _p(smean), _p(sinvstd), _p(red), _p(dgamma),
                                       _p(dbeta), M, Cc, _p(dlogits), _p(w_out), _p(scale), _p(shift), _stream()),
          "aau_bn_bwd_apply_rank1")


def fold_replicas(ws, stride, out, n):
    check(fn("aau_fold_replicas")(_p(ws), stride, _p(out), n, _stream()), "aau_fold_replicas")


def colsum(src, sp, out, ws, M, Cc):
    check(fn("aau_colsum")(_p(src), sp, _p(out), _p(ws), M, Cc, _stream()), "aau_colsum")


def criterion(logits, targets, sums, loss_out, dlogits, B, H, W, finetune=False, neg_bce_w=0.05, edge_w=0.05,
              loss_scale=1.0):
    check(fn("aau_criterion")(_p(logits), _p(targets), _p(sums), _p(loss_out), _p(dlogits), B, H, W,
                              1 if finetune else 0, neg_bce_w, edge_w, loss_scale, _stream()), "aau_criterion")


def loss_terms(logits, targets, sums, loss_out, dlogits, B, H, W, coef9):
    """coef9 = (w_ratio, nu, s_n, d_tp, d_p, d_t, s_d, w_bce, w_edge); see include/aau.h."""
    arr = (C.c_float * 9)(*[float(v) for v in coef9])
    check(fn("aau_loss_terms")(_p(logits), _p(targets), _p(sums), _p(loss_out), _p(dlogits), B, H, W, arr, _stream()),
          "aau_loss_terms")


def seg_metrics(logits, targets, sums, out, B, H, W, thr=0.5):
    check(fn("aau_seg_metrics")(_p(logits), _p(targets), _p(sums), _p(out), B, H, W, thr, _stream()),
          "aau_seg_metrics")


def seg_counts(a, b):
    """-> (|a > 0|, |b > 0|, |both|) as Python ints; a, b device tensors of equal shape (evalseg:41-49)."""
    def prep(t):
        if t.dtype == torch.bool:
            t = t.view(torch.uint8)
        elif t.dtype not in (torch.uint8, torch.float32):
            t = (t > 0).to(torch.uint8)
        return t.contiguous()
    a, b = prep(a), prep(b)
    out = torch.empty(3, dtype=torch.int64, device=a.device)
    check(fn("aau_seg_counts")(_p(a), int(a.dtype == torch.float32), _p(b), int(b.dtype == torch.float32), a.numel(),
                               _p(out), _stream()), "aau_seg_counts")
    na, nb, ni = out.tolist()
    return int(na), int(nb), int(ni)


SQNORM_WS = 1028   # aau.h: AAU_SQNORM_WS
GAP_WS_ROWS = 64   # aau.h: AAU_GAP_WS_ROWS


def grad_sqnorm(grad, n, inv_scale, ws):
    if ws.dtype != torch.float32 or ws.numel() < SQNORM_WS:
        raise _abi.AauError(f"grad_sqnorm: workspace must be fp32 [{SQNORM_WS}] (ws[0] = result), got {ws.dtype} x {ws.numel()}")
    check(fn("aau_grad_sqnorm")(_p(grad), n, inv_scale, _p(ws), _stream()), "aau_grad_sqnorm")


def adamw_step(p, m, v, g, n, norm_ws, step_dev, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-4,
               max_norm=1.0, inv_scale=1.0):
    check(fn("aau_adamw_step")(_p(p), _p(m), _p(v), _p(g), n, _p(norm_ws), _p(step_dev), lr, beta1, beta2, eps,
                               weight_decay, max_norm, inv_scale, _stream()), "aau_adamw_step")


def adamw_step_dev(p, m, v, g, n, norm_ws, step_dev, hyp, group_of_block, n_groups, beta1=0.9, beta2=0.999, eps=1e-8,
                   max_norm=1.0, inv_scale=1.0):
    """AdamW with (lr, weight_decay) per parameter group read from the device tensor ``hyp`` [n_groups, 2]."""
    check(fn("aau_adamw_step_dev")(_p(p), _p(m), _p(v), _p(g), n, _p(norm_ws), _p(step_dev), _p(hyp),
                                   _p(group_of_block), n_groups, beta1, beta2, eps, max_norm, inv_scale, _stream()),
          "aau_adamw_step_dev")


def f32_to_bf16(src, dst, n):
    check(fn("aau_f32_to_bf16")(_p(src), _p(dst), n, _stream()), "aau_f32_to_bf16")


def bf16_to_f32(src, dst, n):
    check(fn("aau_bf16_to_f32")(_p(src), _p(dst), n, _stream()), "aau_bf16_to_f32")


def nchw_to_nhwc(src, dst, dst_pitch, N, Cc, H, W):
    check(fn("aau_nchw_to_nhwc")(_p(src), _p(dst), dst_pitch, N, Cc, H, W, _stream()), "aau_nchw_to_nhwc")


def nhwc_to_nchw(src, src_pitch, dst, N, Cc, H, W):
    check(fn("aau_nhwc_to_nchw")(_p(src), src_pitch, _p(dst), N, Cc, H, W, _stream()), "aau_nhwc_to_nchw")


def hflip_f32(src, dst, N, H, W):
    check(fn("aau_hflip_f32")(_p(src), _p(dst), N, H, W, _stream()), "aau_hflip_f32")


def tta_merge(l, lf, prob, N, H, W):
    check(fn("aau_tta_merge")(_p(l), _p(lf), _p(prob), N, H, W, _stream()), "aau_tta_merge")


def window_blend(win_logits, out, H, W, win, stride, ny, nx, sigma):
    check(fn("aau_window_blend")(_p(win_logits), _p(out), H, W, win, stride, ny, nx, sigma, _stream()),
          "aau_window_blend")
