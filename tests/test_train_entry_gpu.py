"""The train() entry point (pipeline:244-333 restated) and the data-parallel wiring on one GPU."""
import os
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import att_aspp_unet_amd as a
    return a


def test_train_entry_point_learns_and_checkpoint_loads_into_reference_model(A, tmp_path):
    args = Namespace(stage="main", seed=3, output_dir=str(tmp_path), pretrained=None, epochs=4, batch_size=4, lr=2e-3,
                     base_c=8, edge_w=0.05, neg_bce_w=0.05, synthetic_batches=12, img_size=64)
    model, hist = A.train(args)
    assert len(hist) == 4 and hist[-1][0] < hist[0][0]          # training loss decreases
    ckpts = list((tmp_path / "ckpt_main").glob("best_*.pt"))
    assert len(ckpts) == 1
    sd = torch.load(ckpts[0], map_location="cpu", weights_only=True)
    ref = O.AttentionASPPUNet(base_c=8)
    ref.load_state_dict(sd, strict=True)                          # checkpoint schema == reference schema
    # the reference-shaped model reproduces the HIP model's validation metrics from that checkpoint
    from att_aspp_unet_amd import synth
    x, y = synth.make_frames(4, 64, seed=77, neg_frac=0.0)
    m2 = A.AttentionASPPUNet(base_c=8)
    m2.load_state_dict(sd, strict=True)
    m2 = m2.cuda()
    d_hip, i_hip = A.evaluate(m2, [(x.cuda(), y.cuda())], torch.device("cuda"))
    d_ref, i_ref = O.evaluate(ref, [(x, y)], torch.device("cpu"))
    assert abs(d_hip - d_ref) < 5e-3 and abs(i_hip - i_ref) < 2e-2
    # finetune stage: loads the checkpoint through the legacy-key shim and runs
    args2 = Namespace(**{**vars(args), "stage": "finetune", "pretrained": str(ckpts[0]), "epochs": 1})
    _, hist2 = A.train(args2)
    assert len(hist2) == 1


def test_data_parallel_wiring_single_rank_matches_plain_step(A):
    """world size 1 over RCCL: the bucketed all-reduce path must leave gradients / weights identical to the plain step."""
    import torch.distributed as dist
    from att_aspp_unet_amd import synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
        x, y = synth.make_frames(2, 64, seed=5)
        x, y = x.cuda(), y.cuda()
        outs = []
        for use_dp in (False, True):
            torch.manual_seed(1)
            m = A.AttentionASPPUNet(base_c=8).cuda().train()
            m.bridge.project[3].p = 0.0
            dp = A.DataParallel(m) if use_dp else None
            step = A.TrainStep(m, A.FusedAdamW(m, lr=1e-3), args, dp)
            loss = float(step(x, y).item())
            outs.append((loss, m.engine.store.gflat.clone()))
            if use_dp:
                assert dp.reducer is not None and not dp.reducer.works       # every bucket fired and was waited for
        # Two runs of the SAME plain step already differ (fp32 atomics in the BN statistics flip a few bf16
        # roundings / ReLU decisions of this tiny random-init network), so the comparison is statistical.
        assert outs[0][0] == pytest.approx(outs[1][0], rel=1e-4)
        g0, g1 = outs[0][1], outs[1][1]
        cos = float(torch.dot(g0, g1) / g0.norm() / g1.norm())
        assert cos > 0.999, cos
        assert abs(float(g0.norm()) - float(g1.norm())) < 0.02 * float(g0.norm())
    finally:
        dist.destroy_process_group()
