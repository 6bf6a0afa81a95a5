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
    rc = lib.aau_conv_igemm(ctypes.byref(d), 16, 16, 16, None, None, None, None, 0, None)
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


def test_statistics_buffers_are_size_checked_inside_the_library():
    """VERDICT round 2, finding 11: a C caller gets the same protection as the Python wrappers -- every entry point that
    takes an aau_stat buffer takes its size in bytes and refuses one smaller than AAU_STAT_WORDS(C) * 8 before any kernel
    is launched (host-side check: runs without a GPU)."""
    from att_aspp_unet_amd import _abi, ops
    lib = _abi.lib()
    need = ops.stat_words(64) * 8
    buf = (ctypes.c_int64 * ops.stat_words(64))()
    p = ctypes.addressof(buf)
    d = ops.conv_desc(1, 16, 16, 32, 32, 16, 16, 64, 64, 3, 3, 1, 1, 1, 32)
    for nbytes in (0, need - 8, 32 * 2 * 64 * 4):            # the last one: round 1's fp32 [32][2][C] buffer
        assert lib.aau_conv_igemm(ctypes.byref(d), p, p, p, None, None, None, p, nbytes, None) == -1
        assert b"statistics buffer" in lib.aau_last_error()
        assert lib.aau_bn_finalize(p, nbytes, p, p, p, p, p, p, p, p, p, 64, 256, 1e-5, 0.1, None) == -1
        assert b"statistics buffer" in lib.aau_last_error()
        assert lib.aau_fold_stats(p, nbytes, 64, 0, 0, 64, p, None) == -1 and b"statistics buffer" in lib.aau_last_error()
        assert lib.aau_stats_to_f64(p, nbytes, 64, p, None) == -1 and b"statistics buffer" in lib.aau_last_error()
        assert lib.aau_conv1_fwd(p, p, p, p, nbytes, 1, 16, 16, 64, None) == -1 and b"statistics buffer" in lib.aau_last_error()
    assert lib.aau_gate_psi(p, p, p, p, p, p, p, p, p, 8, 256, 64, None) == -1 and b"statistics buffer" in lib.aau_last_error()
    # the pointer argument of every such entry point is followed by its size in the ctypes table
    for name, i in _abi.STAT_ARG.items():
        sig = _abi._SIGS[name]
        assert sig[i] is ctypes.c_void_p and sig[i + 1] is ctypes.c_int64, name


def test_bench_starts_its_own_ranks_when_asked_for_several_gpus():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must spawn one fresh process per rank itself:
    checked without a GPU through the launcher self-test (gloo rendezvous, rank 0 prints the one JSON line)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_selftest": 2, "sum_of_rank_plus_one": 3.0}
    # a rank that fails takes the whole launch down with its exit code instead of leaving peers in a rendezvous
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=240)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs an MI355X" in r.stderr
