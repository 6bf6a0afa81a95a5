"""MI355X-native Attention-ASPP-UNet hot path.

The public names mirror attention_aspp_unet_pipeline_stage.py of the reference so that
``from att_aspp_unet_amd import AttentionASPPUNet, build_criterion, evaluate, train`` is a
drop-in for that path.  Everything computes through ``lib/libaau.so`` (hand-written
gfx950 kernels, C ABI in include/aau.h); nothing falls back to CPU or ATen.
"""
from .model import ASPP, AttentionASPPUNet, AttentionGate, ConvBNReLU, DummyAttention, UpBlock  # noqa: F401
from .losses import (ComboLoss, DiceLoss, EdgeLoss, TverskyLoss, build_criterion, iou_score,  # noqa: F401
                     seg_metrics)
from .optim import FusedAdamW  # noqa: F401
from .pipeline import (EARLY_STOP_PATIENCE, GRAD_CLIP, IMG_SIZE, SEED, WEIGHT_DECAY, GraphedForward, GraphedTrainStep, SyntheticLoader,  # noqa: F401
                       TrainStep, evaluate, get_args, load_state_dict_compat, lr_at_epoch, predict_prob_tta, predict_sliding_window, set_seed,
                       train)
from . import ablation, dataset, evalseg, gc_wrapper, imgproc, measure, mhaio  # noqa: F401
from .measure import measure_ac_mm, select_best  # noqa: F401
from .imgproc import refine_mask  # noqa: F401
from .pipeline import calibrate, predict, predict_masks, predict_prob_tta_batch  # noqa: F401
from .parallel import DataParallel, GradBucketReducer, bucket_ranges  # noqa: F401
from ._abi import AauError, precision  # noqa: F401
