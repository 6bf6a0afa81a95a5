"""MetaImage (.mha / .mhd) reader and writer -- the container the reference reads and writes through SimpleITK
(``attention_aspp_unet_pipeline_stage.py:160-162,485-536``, ``model_attention_aspp.py:14-18``), restated from the
published MetaIO header format so that ``predict`` and the directory reader work where SimpleITK is not installed.

Supported: ``ObjectType = Image``, 2-D / 3-D, the scalar element types below, little- and big-endian data, raw or
zlib-compressed (``CompressedData = True``), data in the same file (``ElementDataFile = LOCAL``) or in a separate raw file
(.mhd).  Arrays use SimpleITK's index order: ``[z, y, x]`` for ``DimSize = x y z``.
"""
from __future__ import annotations

import zlib
from pathlib import Path
from typing import Dict, Tuple

import numpy as np

_TYPES = {"MET_UCHAR": "u1", "MET_CHAR": "i1", "MET_USHORT": "u2", "MET_SHORT": "i2", "MET_UINT": "u4", "MET_INT": "i4",
          "MET_ULONG": "u4", "MET_LONG": "i4", "MET_ULONG_LONG": "u8", "MET_LONG_LONG": "i8", "MET_FLOAT": "f4",
          "MET_DOUBLE": "f8"}          # MetaIO's MET_ValueTypeSize: LONG is 4 bytes, LONG_LONG is the 8-byte type
_NAMES = {"u1": "MET_UCHAR", "i1": "MET_CHAR", "u2": "MET_USHORT", "i2": "MET_SHORT", "u4": "MET_UINT", "i4": "MET_INT",
          "u8": "MET_ULONG_LONG", "i8": "MET_LONG_LONG", "f4": "MET_FLOAT", "f8": "MET_DOUBLE"}


def _truth(v: str) -> bool:
    return v.strip().lower() in ("true", "1", "yes")


def read_header(path) -> Tuple[Dict[str, str], int]:
    """-> (ordered header fields, byte offset of the data that follows ``ElementDataFile``)."""
    hdr: Dict[str, str] = {}
    with open(path, "rb") as f:
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: no ElementDataFile line")
            text = line.decode("latin-1").strip()
            if not text or "=" not in text:
                continue
            k, v = text.split("=", 1)
            hdr[k.strip()] = v.strip()
            if k.strip() == "ElementDataFile":
                return hdr, f.tell()


def read(path) -> Tuple[np.ndarray, Dict[str, str]]:
    """-> (array [z, y, x] or [y, x], header).  ``spacing(header)`` gives (sx, sy, sz)."""
    path = Path(path)
    hdr, off = read_header(path)
    if hdr.get("ObjectType", "Image") != "Image":
        raise ValueError(f"{path}: ObjectType {hdr.get('ObjectType')} is not an image")
    nd = int(hdr["NDims"])
    dims = [int(t) for t in hdr["DimSize"].split()]
    if len(dims) != nd or nd not in (2, 3):
        raise ValueError(f"{path}: NDims {nd} / DimSize {dims}")
    if int(hdr.get("ElementNumberOfChannels", "1")) != 1:
        raise ValueError(f"{path}: multi-channel elements are not supported")
    et = hdr["ElementType"]
    if et not in _TYPES:
        raise ValueError(f"{path}: ElementType {et}")
    msb = _truth(hdr.get("BinaryDataByteOrderMSB", hdr.get("ElementByteOrderMSB", "False")))
    dt = np.dtype((">" if msb else "<") + _TYPES[et])
    src = hdr["ElementDataFile"]
    if src == "LOCAL":
        with open(path, "rb") as f:
            f.seek(off)
            raw = f.read()
    else:
        raw = (path.parent / src).read_bytes()
    if _truth(hdr.get("CompressedData", "False")):
        n = hdr.get("CompressedDataSize")
        raw = zlib.decompress(raw[:int(n)] if n else raw)
    count = int(np.prod(dims))
    if len(raw) < count * dt.itemsize:
        raise ValueError(f"{path}: {len(raw)} data bytes, {count * dt.itemsize} expected")
    arr = np.frombuffer(raw, dtype=dt, count=count).reshape(dims[::-1])
    return arr.astype(dt.newbyteorder("=")), hdr


def spacing(hdr: Dict[str, str]) -> Tuple[float, ...]:
    """(sx, sy[, sz]) in mm: ``ElementSpacing`` (or the older ``ElementSize``), 1.0 when absent."""
    v = hdr.get("ElementSpacing", hdr.get("ElementSize"))
    nd = int(hdr.get("NDims", "3"))
    return tuple(float(t) for t in v.split()) if v else (1.0,) * nd


def write(path, arr: np.ndarray, like: Dict[str, str] = None, compress: bool = True) -> None:
    """Writes ``arr`` ([z, y, x] or [y, x]) as a single-file MetaImage.  ``like``: a header whose geometry (spacing, offset,
    direction, anatomical orientation) is copied -- SimpleITK's ``CopyInformation``."""
    arr = np.ascontiguousarray(arr)
    key = arr.dtype.str[1:]
    if key not in _NAMES or arr.ndim not in (2, 3):
        raise ValueError(f"cannot store dtype {arr.dtype} / {arr.ndim} dimensions as MetaImage")
    nd = arr.ndim
    hdr = [("ObjectType", "Image"), ("NDims", str(nd)), ("BinaryData", "True"), ("BinaryDataByteOrderMSB", "False"),
           ("CompressedData", "True" if compress else "False")]
    like = like or {}
    defaults = {"TransformMatrix": " ".join("1" if i == j else "0" for i in range(nd) for j in range(nd)),
                "Offset": " ".join(["0"] * nd), "CenterOfRotation": " ".join(["0"] * nd),
                "AnatomicalOrientation": "RAI"[:nd] if nd == 3 else "RA", "ElementSpacing": " ".join(["1"] * nd)}
    for k in ("TransformMatrix", "Offset", "CenterOfRotation", "AnatomicalOrientation", "ElementSpacing"):
        v = like.get(k, defaults[k])
        if k != "AnatomicalOrientation" and len(v.split()) != len(defaults[k].split()):
            v = defaults[k]                     # geometry of another dimensionality: fall back
        hdr.append((k, v))
    data = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    if compress:
        data = zlib.compress(data)
        hdr.append(("CompressedDataSize", str(len(data))))
    hdr += [("DimSize", " ".join(str(s) for s in arr.shape[::-1])), ("ElementType", _NAMES[key]), ("ElementDataFile", "LOCAL")]
    with open(path, "wb") as f:
        f.write("".join(f"{k} = {v}\n" for k, v in hdr).encode("latin-1"))
        f.write(data)
