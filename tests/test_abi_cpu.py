"""CPU-side checks of the C ABI: the library loads without a GPU and exports every
symbol that include/aau.h declares (no compute calls here)."""
import ctypes
import os

import pytest


def test_library_exports_every_declared_symbol():
    from att_aspp_unet_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = _abi.lib()
    names = _abi.declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/aau.h but not exported"
    assert lib.aau_version() >= 1
    # every bound signature belongs to a declared symbol
    assert set(_abi._SIGS) <= set(names)


def test_struct_layouts_match_header():
    from att_aspp_unet_amd import _abi
    assert ctypes.sizeof(_abi.ConvDesc) == 22 * 4
    import re
    hdr = open(_abi.HEADER).read()
    body = hdr[hdr.index("typedef struct aau_conv_desc {"):hdr.index("} aau_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n.strip() for decl in re.findall(r"int32_t ([^;]+);", body) for n in decl.split(",")]
    assert names == [f[0] for f in _abi.ConvDesc._fields_]
    assert ctypes.sizeof(_abi.PackEntry) == 8 + 8 + 10 * 4 + 8


def test_invalid_arguments_fail_loudly_without_gpu():
    from att_aspp_unet_amd import _abi
    lib = _abi.lib()
    d = _abi.ConvDesc()
    d.Cin = 7  # not a multiple of 8
    rc = lib.aau_conv_igemm(ctypes.byref(d), 16, 16, 16, None, None, None, None, None)
    assert rc == -1
    assert b"Cin" in lib.aau_last_error()
    with pytest.raises(_abi.AauError):
        _abi.check(rc, "aau_conv_igemm")


def test_statistics_buffers_are_size_checked_on_the_host():
    """A statistics buffer of the wrong type / size (e.g. the fp32 [R][2][C] of round 1) is refused before any kernel
    could add past its end (ops._check_stats): the wrappers raise without touching the GPU."""
    import torch
    from att_aspp_unet_amd import ops, _abi
    d = ops.conv_desc(1, 16, 16, 32, 32, 16, 16, 64, 64, 3, 3, 1, 1, 1, 32)
    dummy = torch.zeros(8)
    for bad in (torch.zeros(ops.STAT_REPLICAS, 2, 64), torch.zeros(ops.stat_words(64) - 1, dtype=torch.int64)):
        with pytest.raises(_abi.AauError, match="stats must be"):
            ops.conv_igemm(d, dummy, dummy, dummy, stats=bad)
        with pytest.raises(_abi.AauError, match="stats must be"):
            ops.bn_finalize(bad, *([dummy] * 9), 64, 256.0)
    assert ops.stats_buffer(64, device="cpu").numel() == ops.stat_words(64)
