"""MetaImage reader / writer (mhaio.py): the container of the reference's sweeps and output masks
(pipeline:160-162,485-536).  SimpleITK is not installed, so the checks are round trips, hand-written headers in the
variants the format allows, and the index order SimpleITK users expect ([z, y, x] for DimSize = x y z)."""
import importlib
import zlib

import numpy as np
import pytest

mhaio = importlib.import_module("att-aspp-unet_amd.mhaio")


@pytest.mark.parametrize("dtype", ["uint8", "int16", "uint16", "float32", "float64", "int32", "int64", "uint64"])
@pytest.mark.parametrize("compress", [False, True])
def test_round_trip(tmp_path, dtype, compress):
    rng = np.random.default_rng(3)
    a = (rng.random((5, 7, 9)) * 200).astype(dtype)
    like = {"ElementSpacing": "0.28 0.31 1.5", "Offset": "1 2 3", "TransformMatrix": "1 0 0 0 1 0 0 0 1", "AnatomicalOrientation": "RAI"}
    mhaio.write(tmp_path / "v.mha", a, like=like, compress=compress)
    b, h = mhaio.read(tmp_path / "v.mha")
    assert b.dtype == a.dtype and b.shape == (5, 7, 9) and np.array_equal(a, b)
    assert h["DimSize"] == "9 7 5" and mhaio.spacing(h) == (0.28, 0.31, 1.5) and h["Offset"] == "1 2 3"
    a2 = a[2]
    mhaio.write(tmp_path / "s.mha", a2, like=like, compress=compress)      # 3-D geometry does not fit 2-D: defaults
    b2, h2 = mhaio.read(tmp_path / "s.mha")
    assert np.array_equal(a2, b2) and h2["NDims"] == "2" and mhaio.spacing(h2) == (1.0, 1.0)


def test_hand_written_headers(tmp_path):
    vol = np.arange(2 * 3 * 4, dtype=np.int16).reshape(2, 3, 4)
    # big-endian raw data, the older ElementSize key, comment-free minimal header
    (tmp_path / "be.mha").write_bytes(b"ObjectType = Image\nNDims = 3\nDimSize = 4 3 2\nElementType = MET_SHORT\n"
                                      b"ElementSize = 0.5 0.25 2\nBinaryDataByteOrderMSB = True\nElementDataFile = LOCAL\n"
                                      + vol.astype(">i2").tobytes())
    a, h = mhaio.read(tmp_path / "be.mha")
    assert np.array_equal(a, vol) and a[1, 2, 3] == 23 and mhaio.spacing(h) == (0.5, 0.25, 2.0)
    # .mhd with the data in a separate zlib-compressed file
    z = zlib.compress(vol.astype("<i2").tobytes())
    (tmp_path / "x.zraw").write_bytes(z)
    (tmp_path / "x.mhd").write_text(f"ObjectType = Image\nNDims = 3\nDimSize = 4 3 2\nElementType = MET_SHORT\nCompressedData = True\n"
                                    f"CompressedDataSize = {len(z)}\nElementDataFile = x.zraw\n")
    a, _ = mhaio.read(tmp_path / "x.mhd")
    assert np.array_equal(a, vol)
    # errors: truncated data, unknown element type, vector pixels
    (tmp_path / "short.mha").write_bytes(b"NDims = 2\nDimSize = 4 4\nElementType = MET_UCHAR\nElementDataFile = LOCAL\n" + b"\0" * 5)
    with pytest.raises(ValueError):
        mhaio.read(tmp_path / "short.mha")
    (tmp_path / "t.mha").write_bytes(b"NDims = 2\nDimSize = 1 1\nElementType = MET_WHAT\nElementDataFile = LOCAL\n\0")
    with pytest.raises(ValueError):
        mhaio.read(tmp_path / "t.mha")
    (tmp_path / "c.mha").write_bytes(b"NDims = 2\nDimSize = 1 1\nElementNumberOfChannels = 3\nElementType = MET_UCHAR\nElementDataFile = LOCAL\n\0\0\0")
    with pytest.raises(ValueError):
        mhaio.read(tmp_path / "c.mha")
    with pytest.raises(ValueError):
        mhaio.write(tmp_path / "b.mha", np.zeros((2, 2), dtype=bool))


def test_long_types_follow_metaio_sizes(tmp_path):
    """MET_LONG / MET_ULONG are 4-byte types in MetaIO; the 8-byte ones are MET_LONG_LONG / MET_ULONG_LONG (what ITK
    writes for int64)."""
    a = np.arange(24, dtype=np.int64).reshape(2, 3, 4) - 5
    mhaio.write(tmp_path / "l.mha", a, compress=False)
    assert mhaio.read_header(tmp_path / "l.mha")[0]["ElementType"] == "MET_LONG_LONG"
    b, _ = mhaio.read(tmp_path / "l.mha")
    assert b.dtype == np.int64 and np.array_equal(a, b)
    v = np.arange(6, dtype="<i4").reshape(2, 3)
    (tmp_path / "m.mha").write_bytes(b"ObjectType = Image\nNDims = 2\nDimSize = 3 2\nElementType = MET_LONG\nElementDataFile = LOCAL\n" + v.tobytes())
    c, _ = mhaio.read(tmp_path / "m.mha")
    assert c.dtype == np.int32 and np.array_equal(c, v)
