"""Grand-Challenge inference wrapper of the reference (model_attention_aspp.py:14-97) on the HIP path.

``FetalAbdomenSegmentation.predict_array`` takes a raw uint8 sweep [F, H, W] and runs, GPU resident:
CLAHE + median preprocessing (:14-18) -> 128 evenly spaced frames (:43) -> ROI-224 crop around the bright centroid
(:20-31) -> forward in batches of 8 (:52) -> sigmoid, paste back into full frames (:55-59) and returns the probability
stack.  ``postprocess`` (:66-85) thresholds at 0.05, picks the frame with the largest area, dilates it 3x3 and keeps its
largest 8-connected component.  ``select_fetal_abdomen_mask_and_frame`` is :87-93.

The reference file cannot be imported at all (it imports a module ``attention_aspp_unet`` that the repository does not
contain and constructs the model with keyword names the real class does not have); the alias
module ``attention_aspp_unet`` at the repository root supplies that name with that signature.  ``.mha`` sweeps are read
by ``mhaio.py`` (no SimpleITK); ``.tiff`` sweeps are refused with a clear error; ``predict_array`` takes the array directly.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import imgproc
from .model import AttentionASPPUNet

__all__ = ["FetalAbdomenSegmentation", "select_fetal_abdomen_mask_and_frame", "load_image_file_as_array", "run",
           "write_array_as_image_file", "convert_2d_mask_to_3d"]


def load_image_file_as_array(*, location: Path):
    """model_attention_aspp.py:14-18 (MetaImage read, then the GPU preprocessing) -> fp32 [1, F, H, W] on the device."""
    from . import mhaio            # MetaImage container read without SimpleITK
    arr, _ = mhaio.read(location)
    return preprocess_sweep(torch.from_numpy(np.ascontiguousarray(arr.astype(np.uint8))).cuda())[None]


def preprocess_sweep(u8: torch.Tensor) -> torch.Tensor:
    """:16-18: per slice medianBlur(CLAHE(normalize(sl)), 3) / 255 -> fp32 [F, H, W]."""
    return imgproc.to_float(imgproc.median3(imgproc.clahe(imgproc.normalize_minmax(u8))))


class FetalAbdomenSegmentation:
    def __init__(self, checkpoint_path=None, base=16, device="cuda", precision="fp16"):
        self.device = torch.device(device)
        self.net = AttentionASPPUNet(in_channels=1, num_classes=1, base_c=base).to(self.device)
        if checkpoint_path is not None:
            sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            miss, unexp = self.net.load_state_dict(sd, strict=False)
            print(f"[DEBUG] load_state — missing:{len(miss)} unexpected:{len(unexp)}")
        self.net.eval().set_precision(precision)     # the reference wrapper runs fp32; half is the closest 16-bit type
        self.case_id = None

    @torch.no_grad()
    def predict_array(self, sweep_u8, nframes: int = 128, roi: int = 224, batch: int = 8):
        """uint8 [F, H, W] -> (probabilities fp32 [nframes, H, W] on the device, sampled frame indices)."""
        if isinstance(sweep_u8, np.ndarray):
            sweep_u8 = torch.from_numpy(np.ascontiguousarray(sweep_u8))
        vol = preprocess_sweep(sweep_u8.to(self.device))                       # [F, H, W] in [0, 1]
        F_, H, W = vol.shape
        if H < roi or W < roi:
            raise ValueError(f"frames of {H}x{W} are smaller than the {roi}-pixel ROI")
        idxs = np.linspace(0, F_ - 1, nframes).astype(int)                      # :43
        vol = vol[torch.from_numpy(idxs).to(self.device)].contiguous()
        org = imgproc.roi_origin(vol, roi)
        patches = imgproc.roi_crop(vol, org, roi)[:, None]                      # [N, 1, roi, roi]
        N = patches.shape[0]
        logits = torch.empty(N, roi, roi, device=self.device)
        for i in range(0, N, batch):
            logits[i:i + batch] = self.net(patches[i:i + batch].contiguous())[:, 0]
        return imgproc.roi_paste_sigmoid(logits, org, (H, W)), idxs

    def predict(self, input_img_path, save_probabilities=False):
        """:40-64 on an .mha path (read by mhaio.py) -> numpy [128, H, W]."""
        from . import mhaio
        self.case_id = Path(input_img_path[0]).stem
        arr, _ = mhaio.read(input_img_path[0])
        prob, _ = self.predict_array(arr.astype(np.uint8))
        return prob.cpu().numpy()

    def postprocess(self, probability_map):
        """:66-85 -> uint8 [N, H, W] (device tensor in -> device tensor out; numpy in -> numpy out)."""
        as_np = isinstance(probability_map, np.ndarray)
        p = torch.from_numpy(np.ascontiguousarray(probability_map, dtype=np.float32)).to(self.device) if as_np else probability_map
        areas = imgproc.frame_areas(p, 0.05)
        frame_idx = int(torch.argmax(areas).item())                             # first maximum, as np.argmax
        mask = torch.zeros(p.shape, dtype=torch.uint8, device=p.device)
        if int(areas[frame_idx].item()) > 0:
            frame = imgproc.dilate3(imgproc.threshold(p[frame_idx], 0.05))
            mask[frame_idx] = imgproc.keep_largest_component(frame, min_area=1, conn8=True)
        return mask.cpu().numpy() if as_np else mask


def select_fetal_abdomen_mask_and_frame(mask_3d):
    """model_attention_aspp.py:87-93."""
    if isinstance(mask_3d, torch.Tensor):
        mask_3d = mask_3d.cpu().numpy()
    if mask_3d.ndim == 2:
        return (mask_3d > 0).astype(np.uint8), 0
    areas = mask_3d.sum((1, 2))
    idx = int(areas.argmax())
    if areas[idx] == 0:
        return np.zeros(mask_3d.shape[1:], np.uint8), -1
    return (mask_3d[idx] > 0).astype(np.uint8), idx


def convert_2d_mask_to_3d(*, mask_2d, frame_number, number_of_frames):
    """inference.py:253-269: the 2-D mask (1 -> 2) in frame ``frame_number`` of an empty volume; -1 = no abdomen found."""
    m = np.where(np.asarray(mask_2d) == 1, 2, 0).astype(np.uint8)
    vol = np.zeros((number_of_frames,) + m.shape, np.uint8)
    if frame_number == -1:
        return vol
    if frame_number is not None and 0 <= frame_number < number_of_frames:
        vol[frame_number] = m
        return vol
    raise ValueError(f"frame_number must be between -1 and {number_of_frames - 1}, got {frame_number}.")


def write_array_as_image_file(*, location, array, frame_number=None, number_of_frames=128, filename="output.mha"):
    """inference.py:206-251: a compressed uint8 {0, 1} MetaImage volume with 0.28 mm spacing."""
    from . import mhaio
    location = Path(location)
    location.mkdir(parents=True, exist_ok=True)
    array = np.squeeze(np.asarray(array))
    if array.ndim != 2:
        raise ValueError(f"Expected a 2D array, got {array.ndim}D.")
    vol = convert_2d_mask_to_3d(mask_2d=array.astype(np.float32), frame_number=frame_number, number_of_frames=number_of_frames)
    vol = np.where(vol > 0.5, 1, 0).astype(np.uint8)
    mhaio.write(location / filename, vol, like={"ElementSpacing": "0.28 0.28 0.28"}, compress=True)


def run(input_path="./test/input", output_path="./test/output", case_id="output", checkpoint_path=None, base=16):
    """inference.py:50-133, the Grand-Challenge entry point: the sweep under ``<input>/images/stacked-fetal-ultrasound``
    -> probability maps of 128 sampled frames -> post-processing -> the chosen frame's mask written as
    ``<output>/images/fetal-abdomen-segmentation/<case_id>.mha`` (a volume with as many frames as the sweep) and
    ``<output>/fetal-abdomen-frame-number.json``."""
    import json
    from glob import glob
    from . import mhaio
    input_path, output_path = Path(input_path), Path(output_path)
    loc = input_path / "images" / "stacked-fetal-ultrasound"
    files = sorted(glob(str(loc / "*.mha")))
    if not files:
        if glob(str(loc / "*.tiff")):
            raise NotImplementedError(f"{loc}: .tiff sweeps are not supported (MetaImage .mha only)")
        raise FileNotFoundError(f"no .mha sweep under {loc}")
    algorithm = FetalAbdomenSegmentation(checkpoint_path=checkpoint_path, base=base)
    prob = algorithm.predict(files, save_probabilities=True)
    post = algorithm.postprocess(prob)
    seg, frame = select_fetal_abdomen_mask_and_frame(post)
    hdr, _ = mhaio.read_header(files[0])
    dims = [int(t) for t in hdr["DimSize"].split()]
    ref_w, ref_h, n_frames = dims[0], dims[1], dims[2]
    if seg.shape != (ref_h, ref_w):            # nearest-neighbour resize to the sweep's frame size (inference.py:96-101)
        yi = (np.arange(ref_h) * (seg.shape[0] / ref_h)).astype(int).clip(0, seg.shape[0] - 1)
        xi = (np.arange(ref_w) * (seg.shape[1] / ref_w)).astype(int).clip(0, seg.shape[1] - 1)
        seg = seg[yi][:, xi]
    seg = (seg > 0).astype(np.uint8)
    write_array_as_image_file(location=output_path / "images" / "fetal-abdomen-segmentation", array=seg, frame_number=frame,
                              number_of_frames=n_frames, filename=f"{case_id}.mha")
    with open(output_path / "fetal-abdomen-frame-number.json", "w") as f:
        f.write(json.dumps(int(frame), indent=4))
    return 0
