"""Builds the gfx950 shared library ``lib/libaau.so`` (C ABI: include/aau.h) in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so
travels to the GPU box with the snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libaau.so")
SOURCES = ["runtime.hip", "igemm.hip", "igemm_group.hip", "conv3x3.hip", "conv3x3s.hip", "wgrad.hip", "wgrad3x3.hip", "wgradL.hip", "wgrad3x3r.hip", "bn.hip", "pointwise.hip", "poolbranch.hip", "gate.hip", "loss.hip", "optim.hip", "imgproc.hip", "augment.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-Wno-unused-result", "-Wno-unused-value"]


# per-file flags: the image kernels restate published algorithms operation by operation (separately rounded multiplies
# and adds), so hipcc's default fused-multiply-add contraction is off for that file
EXTRA_FLAGS = {"imgproc.hip": ["-ffp-contract=off"], "augment.hip": ["-ffp-contract=off"]}


def source_hash() -> str:
    """16 hex digits over every source the library is built from: profiles/*_pmc_traffic.json carry it, and bench.py only
    quotes a profile's traffic figure when it was measured on THESE kernels."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted([os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "c3args.h"), os.path.join(CSRC, "igemm_body.inc"), os.path.join(ROOT, "include", "aau.h")]):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _newer(src: str, dst: str) -> bool:
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = True, defines=(), tag: str = "") -> str:
    """``defines`` / ``tag`` build an ablation variant ``lib/libaau_<tag>.so`` (see scripts/)."""
    global LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(PKG, "build" + ("_" + tag if tag else ""))
    lib_out = os.path.join(LIBDIR, f"libaau_{tag}.so") if tag else LIB
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [os.path.join(ROOT, "include", "aau.h"), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "c3args.h"), os.path.join(CSRC, "igemm_body.inc")]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), *[f"-D{d}" for d in defines], "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for src, rc, out in ex.map(compile_one, jobs):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(lib_out):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_out, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}{r.stderr}")
    return lib_out


def build_all(force: bool = False, verbose: bool = True) -> list:
    """Both storage types: libaau.so (bfloat16) and libaau_f16.so (IEEE half, inference)."""
    return [build(force, verbose), build(force, verbose, defines=("AAU_F16",), tag="f16")]


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    tags = [a[6:] for a in sys.argv[1:] if a.startswith("--tag=")]
    if "--source-hash" in sys.argv:
        print(source_hash())
    elif not defs and not tags:
        print(build_all(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv, defines=defs, tag=tags[0] if tags else ""))
