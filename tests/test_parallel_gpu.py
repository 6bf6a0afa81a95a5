"""Data-parallel semantics of the ENGINE with more than one rank, with the means at hand: one
MI355X, so the two ranks are two child processes on the same card and the process group is ``gloo`` (RCCL refuses two
ranks on one device).  Everything else is the product path: ``DataParallel`` + ``TrainStep`` -> bucketed all-reduces
fired from the marks inside the recorded backward -> clip + AdamW with 1/world folded into the unscale factor.

Oracle (the reference has no DP; section 8e defines it): the MEAN over ranks of the per-rank reference gradients, each
rank's loss normalised by its own counts, then ONE clip + AdamW step on that mean -- computed here with the CPU oracle
at the reference-trained fixture weights (base_c 8, 128x128, four frames per rank).
"""
import os
import socket
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O

ARGS = dict(stage="main", edge_w=0.05, neg_bce_w=0.05)
LR = 3e-4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sd, x, y, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import att_aspp_unet_amd as A
        torch.cuda.set_device(0)
        torch.manual_seed(2025)
        m = A.AttentionASPPUNet(base_c=8)
        if rank == 0:                       # only rank 0 holds the real weights: the wrapper must broadcast them
            m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        m = m.cuda().train()
        m.bridge.project[3].p = 0.0
        m._engine.ensure(torch.device("cuda", 0))
        dp = A.DataParallel(m)
        opt = A.FusedAdamW(m, lr=LR, weight_decay=A.WEIGHT_DECAY, max_grad_norm=A.GRAD_CLIP)
        step = A.TrainStep(m, opt, Namespace(**ARGS), dp)
        xs, ys = torch.from_numpy(x[rank]).cuda(), torch.from_numpy(y[rank]).cuda()
        loss = float(step(xs, ys).item())
        torch.cuda.synchronize()
        st = m.engine.store
        plan = m._plan_for(xs)
        out = dict(rank=rank, loss=loss, gsum=st.gflat.detach().cpu().numpy(), flat=st.flat.detach().cpu().numpy(),
                   names=list(st.names), offs=dict(st.offs), drop_seed=int(plan.drop_seed.item()),
                   sd={k: v.detach().cpu().numpy() for k, v in m.state_dict().items()})
        # a second step must not hang or diverge (bucket marks re-arm, every rank applies the same update)
        step(xs, ys)
        torch.cuda.synchronize()
        out["flat2"] = st.flat.detach().cpu().numpy()
        # gradient ACCUMULATION under data parallelism (autograd path): two backward calls without clearing the gradients
        # must leave reduce(g_a) + reduce(g_b) on both ranks -- the kept gradient is added behind the bucket all-reduces
        crit = A.build_criterion(Namespace(**ARGS), A.ComboLoss(), A.EdgeLoss())
        xa, ya, xb, yb = xs[:2], ys[:2], xs[2:], ys[2:]

        def grads(x_, y_, clear=True):
            if clear:
                opt.zero_grad()
            crit(m(x_), y_).backward()
            dp.finish()
            torch.cuda.synchronize()
            return st.gflat.detach().cpu().numpy().copy()
        ga, gb = grads(xa, ya), grads(xb, yb)
        grads(xa, ya)
        out["acc"], out["ga"], out["gb"] = grads(xb, yb, clear=False), ga, gb
        try:                                # a backward while the previous one's all-reduces are outstanding is refused
            opt.zero_grad()
            crit(m(xa), ya).backward()
            crit(m(xb), yb).backward()
            out["refused"] = False
        except RuntimeError:
            out["refused"] = True
        dp.reducer.works, dp.reducer.fired, dp.pending = [], [], None
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(dict(rank=rank, error=f"{e!r}\n{traceback.format_exc()}"))


def test_engine_step_under_data_parallel_world2_matches_mean_of_per_rank_reference_gradients(golden):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    g = golden("g4_trained_c8_128.npz")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd/")}
    x = [g["x"][:4], g["x"][4:]]
    y = [g["y"][:4], g["y"][4:]]
    world, port = 2, _free_port()
    ctx = torch.multiprocessing.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sd, x, y, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=420) for _ in range(world)], key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=120)
    assert all("error" not in r for r in res), [r.get("error") for r in res]

    # ---- the oracle: per-rank reference step pieces on the CPU ----
    crit = O.build_criterion(Namespace(**ARGS), O.ComboLoss(), O.EdgeLoss())
    refs, losses, grads = [], [], []
    for r in range(world):
        ref = O.AttentionASPPUNet(base_c=8)
        ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        ref.train()
        ref.bridge.project[3].p = 0.0
        loss = crit(ref(torch.from_numpy(x[r])), torch.from_numpy(y[r]))
        loss.backward()
        refs.append(ref); losses.append(float(loss)); grads.append({k: p.grad.clone() for k, p in ref.named_parameters()})
    mean = {k: sum(gr[k] for gr in grads) / world for k in grads[0]}
    upd = refs[0]
    for k, p in upd.named_parameters():
        p.grad = mean[k].clone()
    opt = O.make_optimizer(upd, LR)
    torch.nn.utils.clip_grad_norm_(upd.parameters(), O.GRAD_CLIP)
    opt.step()

    names, offs = res[0]["names"], res[0]["offs"]

    def view(flat, k, shape):
        n = int(np.prod(shape))
        t = torch.from_numpy(flat[offs[k]:offs[k] + n])
        if len(shape) == 4:                       # physical [O][KH][KW][I] -> logical OIHW
            d0, d1, d2, d3 = shape
            return t.view(d0, d2, d3, d1).permute(0, 3, 1, 2)
        return t.view(*shape)

    # 1. each rank's loss is its own shard's reference loss
    for r in range(world):
        assert abs(res[r]["loss"] - losses[r]) < 2e-3 * abs(losses[r]), (r, res[r]["loss"], losses[r])
    # 2. the reduced gradient buffer is identical on both ranks, and (sum / world) is the mean of the reference gradients
    assert np.array_equal(res[0]["gsum"], res[1]["gsum"])
    ge = torch.cat([(view(res[0]["gsum"], k, p.shape) / world).reshape(-1) for k, p in upd.named_parameters()]).double()
    gr = torch.cat([mean[k].reshape(-1) for k, _ in upd.named_parameters()]).double()
    cos = float(torch.dot(ge, gr) / ge.norm() / gr.norm())
    assert cos > 0.999, cos
    assert abs(float(ge.norm()) - float(gr.norm())) < 0.01 * float(gr.norm())
    # ... and it is NOT either rank's own gradient (the shards differ)
    g0 = torch.cat([grads[0][k].reshape(-1) for k, _ in upd.named_parameters()]).double()
    assert float((ge - g0).norm() / g0.norm()) > 0.05
    # 3. identical weights on both ranks after one and after two steps, equal to the oracle's update of the mean gradient
    assert np.array_equal(res[0]["flat"], res[1]["flat"]) and np.array_equal(res[0]["flat2"], res[1]["flat2"])
    assert not np.array_equal(res[0]["flat"], res[0]["flat2"])
    worst = 0.0
    for k, p in upd.named_parameters():
        w = view(res[0]["flat"], k, p.shape)
        worst = max(worst, float((w - p.detach()).abs().max()))
    # AdamW's first step moves every weight by ~lr * sign(g): a gradient that differs in sign (near-zero entries) costs
    # at most 2 lr; everything else agrees to a fraction of lr
    assert worst <= 2.05 * LR, worst
    moved = torch.cat([(view(res[0]["flat"], k, p.shape) - torch.from_numpy(sd[k])).reshape(-1) for k, p in upd.named_parameters()])
    movedr = torch.cat([(p.detach() - torch.from_numpy(sd[k])).reshape(-1) for k, p in upd.named_parameters()])
    assert float(torch.dot(moved, movedr) / moved.norm() / movedr.norm()) > 0.98
    # 4. BatchNorm running statistics stay PER RANK (no SyncBN in the reference): they differ between the ranks and each
    #    equals its own shard's reference statistics
    k = "d1.1.block.1.running_mean"
    assert not np.array_equal(res[0]["sd"][k], res[1]["sd"][k])
    for r in range(world):
        ref_rm = refs[r].state_dict()[k].numpy()
        assert np.abs(res[r]["sd"][k] - ref_rm).max() < 4e-2 * np.abs(ref_rm).max() + 1e-4
    # 5. the dropout seed chain is mixed with the rank
    assert res[0]["drop_seed"] != res[1]["drop_seed"]
    # 6. accumulation over two forward / backward pairs: reduce(g_a) + reduce(g_b), bit for bit, on both ranks
    assert np.array_equal(res[0]["acc"], res[1]["acc"])
    assert np.array_equal(res[0]["ga"], res[1]["ga"]) and not np.array_equal(res[0]["ga"], res[0]["gb"])
    assert np.array_equal(res[0]["acc"], res[0]["ga"] + res[0]["gb"])
    assert res[0]["refused"] and res[1]["refused"]


def _train_worker(rank, world, port, out_dir, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import att_aspp_unet_amd as A
        torch.cuda.set_device(0)
        args = A.pipeline.get_args(["train", "--epochs", "3", "--batch_size", "4", "--base_c", "8", "--img_size", "64",
                                    "--synthetic_batches", "6", "--output_dir", out_dir, "--seed", "11"])
        model, hist = A.train(args)
        torch.cuda.synchronize()
        q.put(dict(rank=rank, hist=hist, w=model.engine.store.flat.detach().cpu().numpy(),
                   rm=model.state_dict()["d1.1.block.1.running_mean"].cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(dict(rank=rank, error=f"{e!r}\n{traceback.format_exc()}"))


def test_train_loop_under_data_parallel_makes_collective_decisions(tmp_path):
    """pipeline:316-333 with two ranks: every rank trains on its own synthetic shard, the epoch's validation
    score and BatchNorm buffers are rank 0's on every rank (broadcast), so best-checkpoint / early-stop decisions agree and
    nobody is left waiting in an all-reduce; weights stay identical; only rank 0 writes the checkpoint."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, port = 2, _free_port()
    ctx = torch.multiprocessing.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, str(tmp_path / "ck"), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=420) for _ in range(world)], key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=120)
    assert all("error" not in r for r in res), [r.get("error") for r in res]
    h0, h1 = res[0]["hist"], res[1]["hist"]
    assert len(h0) == len(h1) == 3
    for a, b in zip(h0, h1):
        assert a[1] == b[1] and a[2] == b[2]                 # validation Dice / IoU: rank 0's values on both ranks
        assert a[0] != b[0]                                  # the training loss is each rank's own shard's
    assert np.array_equal(res[0]["w"], res[1]["w"])          # identical weights after every step
    assert np.array_equal(res[0]["rm"], res[1]["rm"])        # buffers were synchronised for validation
    assert len(list((tmp_path / "ck" / "ckpt_main").glob("best_*.pt"))) >= 1


def test_bench_runs_two_ranks_end_to_end():
    """`python bench.py --gpus 2` (ranks started by bench.py itself): the whole control flow of the N > 1 benchmark --
    DataParallel train steps, barrier + max-over-ranks timing, the per-launch profiling pass (EVERY rank runs its steps: they
    contain collectives), rank 0's one JSON line -- rehearsed on one card with the gloo backend (AAU_BENCH_BACKEND)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["AAU_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--base_c", "8",
                        "--size", "128", "--batch", "4"], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["roofline"] is not None and d["cpu_baseline"] is None
    # multi-GPU evidence fields: what the collective library connected, per-GPU rate, per-bucket overlap
    c = d["config"]
    assert c["rccl_ranks_seen"] == 2 and c["collective_backend"] == "gloo"
    assert abs(c["images_per_sec_per_gpu"] * 2 - d["value"]) < 1e-6 * d["value"]
    ov = c["overlap"]
    assert set(ov["buckets"]) == {"u3", "u4", "bridge", "d4", "d1"}
    for b in ov["buckets"].values():
        assert b["mbytes"] > 0 and b["allreduce_ms_alone"] > 0 and b["backward_ms_behind_its_mark"] >= 0
    assert ov["eager_step_ms_without_allreduce"] > 0 and ov["eager_step_ms_with_allreduce"] > 0
