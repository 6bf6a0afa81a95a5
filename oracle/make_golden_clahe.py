"""Build-container only: store the six debugging frames the reference holds as a golden fixture.

/root/reference/output/images/frame{000,064,127}_{orig,enh}.png were written by inference.py:181-183 with real cv2:
``enh = medianBlur(createCLAHE(clipLimit, (8, 8)).apply(orig), 3)``.  They are DATA (inputs and expected outputs of
the cv2 calls), decoded here with PIL and committed as arrays -> tests/golden/g8_clahe_frames.npz.
The clip limit is not recorded with the frames: inference.py:168 reads ``clipLimit=1.0`` with the note "0.8 is fine
too"; the frames are reproduced bit for bit by 0.8 and by no other value of the sweep below (recorded in the npz).

    python oracle/make_golden_clahe.py
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import imgproc_ref as R  # noqa: E402

SRC = "/root/reference/output/images"


def main():
    out = {}
    for i in (0, 64, 127):
        for kind in ("orig", "enh"):
            a = np.asarray(Image.open(f"{SRC}/frame{i:03d}_{kind}.png"))
            assert a.ndim == 2 and a.dtype == np.uint8, (a.shape, a.dtype)
            out[f"frame{i:03d}_{kind}"] = a
    sweep = []
    for clip in (0.5, 0.6, 0.7, 0.75, 0.8, 0.85, 0.9, 1.0, 1.2, 2.0):
        bad = [int((R.median3(R.clahe(out[f"frame{i:03d}_orig"], clip, 8)) != out[f"frame{i:03d}_enh"]).sum()) for i in (0, 64, 127)]
        sweep.append((clip, *bad))
        print(f"clipLimit {clip}: mismatching pixels {bad}")
    out["clip_sweep"] = np.asarray(sweep, np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g8_clahe_frames.npz"), **out)


if __name__ == "__main__":
    main()
