"""The module name that model_attention_aspp.py:6 imports (``from attention_aspp_unet import AttentionASPPUNet``) and the
reference repository never shipped.  The wrapper calls ``AttentionASPPUNet(in_ch=1, num_classes=1, base=16)``
(model_attention_aspp.py:36); those keyword names are mapped onto the real constructor
(attention_aspp_unet_pipeline_stage.py:112: ``in_channels, num_classes, base_c``)."""
from att_aspp_unet_amd import AttentionASPPUNet as _Net
from att_aspp_unet_amd.gc_wrapper import FetalAbdomenSegmentation, select_fetal_abdomen_mask_and_frame  # noqa: F401


def AttentionASPPUNet(in_ch=1, num_classes=1, base=16, **kw):
    return _Net(in_channels=kw.pop("in_channels", in_ch), num_classes=num_classes, base_c=kw.pop("base_c", base), **kw)
