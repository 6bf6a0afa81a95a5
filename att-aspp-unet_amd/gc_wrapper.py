"""Grand-Challenge inference wrapper of the reference (model_attention_aspp.py:14-97) on the HIP path.

``FetalAbdomenSegmentation.predict_array`` takes a raw uint8 sweep [F, H, W] and runs, GPU resident:
CLAHE + median preprocessing (:14-18) -> 128 evenly spaced frames (:43) -> ROI-224 crop around the bright centroid
(:20-31) -> forward in batches of 8 (:52) -> sigmoid, paste back into full frames (:55-59) and returns the probability
stack.  ``postprocess`` (:66-85) thresholds at 0.05, picks the frame with the largest area, dilates it 3x3 and keeps its
largest 8-connected component.  ``select_fetal_abdomen_mask_and_frame`` is :87-93.

The reference file cannot be imported at all (it imports a module ``attention_aspp_unet`` that the repository does not
contain and constructs the model with keyword names the real class does not have); the alias
module ``attention_aspp_unet`` at the repository root supplies that name with that signature.  Reading .mha files needs
SimpleITK, which is not installed here: ``predict`` raises if it is missing, ``predict_array`` takes the array directly.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import imgproc
from .model import AttentionASPPUNet

__all__ = ["FetalAbdomenSegmentation", "select_fetal_abdomen_mask_and_frame", "load_image_file_as_array"]


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
