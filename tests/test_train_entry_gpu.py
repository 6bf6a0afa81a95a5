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


def test_data_parallel_wiring_single_rank_matches_plain_step(A, golden):
    """world size 1 over RCCL: the bucketed all-reduce path must leave the loss / gradients of the plain step
    unchanged.  The step is not bitwise reproducible (fp32 atomics in the BN statistics flip a few bf16 roundings),
    so the comparison is against the measured run-to-run spread of the PLAIN step on the trained fixture."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g = golden("g4_trained_c8_128.npz")
        sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
        args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
        x, y = torch.from_numpy(g["x"][:4]).cuda(), torch.from_numpy(g["y"][:4]).cuda()
        outs = []
        for use_dp in (False, False, True):
            m = A.AttentionASPPUNet(base_c=8)
            m.load_state_dict(sd, strict=True)
            m = m.cuda().train()
            m.bridge.project[3].p = 0.0
            dp = A.DataParallel(m) if use_dp else None
            step = A.TrainStep(m, A.FusedAdamW(m, lr=1e-3), args, dp)
            loss = float(step(x, y).item())
            outs.append((loss, m.engine.store.gflat.clone()))
            if use_dp:
                assert dp.reducer is not None and not dp.reducer.works       # every bucket fired and was waited for

        def cos(a, b):
            return float(torch.dot(a, b) / a.norm() / b.norm())

        (l0, g0), (l1, g1), (l2, g2) = outs
        noise_cos = cos(g0, g1)
        noise_norm = abs(float(g0.norm()) - float(g1.norm())) / float(g0.norm())
        assert noise_cos > 0.98, noise_cos                                   # the fixture itself is well conditioned
        assert l2 == pytest.approx(l0, rel=1e-3)             # measured run-to-run spread of the plain step: 1.3e-4 (scripts/stress_step.py)
        assert cos(g0, g2) > 1 - 4 * (1 - noise_cos) - 1e-3, (cos(g0, g2), noise_cos)
        assert abs(float(g0.norm()) - float(g2.norm())) < (4 * noise_norm + 0.01) * float(g0.norm())
    finally:
        dist.destroy_process_group()


def test_graphed_train_step_segments_match_eager_with_and_without_data_parallel(A, golden):
    """GraphedTrainStep: one hipGraph (no DP) / one graph per gradient-bucket segment with the RCCL all-reduces issued
    in between (world size 1 here) must train like the eager TrainStep."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29518")
    g = golden("g4_trained_c8_128.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    x, y = torch.from_numpy(g["x"][:4]).cuda(), torch.from_numpy(g["y"][:4]).cuda()

    def run(use_dp, graphed, steps=6):
        m = A.AttentionASPPUNet(base_c=8)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        m.bridge.project[3].p = 0.0
        dp = A.DataParallel(m) if use_dp else None
        step = A.TrainStep(m, A.FusedAdamW(m, lr=1e-3), args, dp)
        fn = A.GraphedTrainStep(step, x, y, warmup=0) if graphed else step
        skip = 1 if graphed else 0                   # the graphed form ran one eager step to build its state
        losses = [float(fn(x, y).item()) for _ in range(steps - skip)]
        if use_dp:
            assert not dp.reducer.works
        return losses, torch.cat([p.detach().flatten() for p in m.parameters()]).clone()

    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        le, we = run(False, False)
        for use_dp in (False, True):
            lg, wg = run(use_dp, True)
            assert len(lg) == len(le) - 1
            for a_, b_ in zip(lg, le[1:]):
                assert a_ == pytest.approx(b_, rel=2e-3), (lg, le)
            assert le[-1] < le[0]
            cos = float(torch.dot(wg - we, wg - we) ** 0.5 / we.norm())
            assert cos < 5e-3, cos                  # same weights after 6 Adam steps, up to the atomics' run-to-run spread
    finally:
        dist.destroy_process_group()
