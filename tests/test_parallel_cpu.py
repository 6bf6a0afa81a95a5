"""Data-parallel gradient reduction logic on CPU: world size 2, gloo backend.

The bucket plan, the mark-driven async all-reduces and the averaging convention are
backend independent; the GPU path only swaps gloo for RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import att_aspp_unet_amd as A
        from att_aspp_unet_amd.engine import ParamStore
        torch.manual_seed(0)
        model = A.AttentionASPPUNet(base_c=8)
        st = ParamStore(model, torch.device("cpu"))
        numels = {n: p.numel() for n, p in zip(st.names, st.params)}
        ranges = A.bucket_ranges(st.names, st.offs, numels, st.total)
        # every element belongs to exactly one bucket
        cover = torch.zeros(st.total, dtype=torch.int32)
        for b, e in ranges.values():
            cover[b:e] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1
        # each rank "computes" different gradients; marks fire in the engine's backward order
        g = torch.Generator().manual_seed(100 + rank)
        st.gflat.copy_(torch.randn(st.total, generator=g))
        mine = st.gflat.clone()
        red = A.GradBucketReducer(st.gflat, ranges)
        for mark in ("u1", "u2", "u3", "u4", "bridge", "d4", "d3", "d2", "d1"):
            red.on_mark(mark)
        red.finish()
        others = [torch.randn(st.total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        expect = sum(others)
        assert torch.allclose(st.gflat, expect, atol=1e-6)
        assert torch.equal(others[rank], mine)
        # parameter views see the reduced gradients, and averaging = 1/world in the optimiser
        st.bind_grads()
        name = "bridge.project.0.weight"
        p = dict(model.named_parameters())[name]
        assert torch.allclose(p.grad, st.gviews[name]) and p.grad.shape == p.shape
        # a bucket that never fires is an error, not a silent stale gradient
        red.on_mark("u3")
        try:
            red.finish()
            ok = False
        except RuntimeError:
            ok = True
        assert ok
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_bucket_plan_is_contiguous_and_ordered():
    import att_aspp_unet_amd as A
    from att_aspp_unet_amd.engine import ParamStore
    model = A.AttentionASPPUNet(base_c=8)
    st = ParamStore(model, torch.device("cpu"))
    numels = {n: p.numel() for n, p in zip(st.names, st.params)}
    r = A.bucket_ranges(st.names, st.offs, numels, st.total)
    assert set(r) == {"u3", "u4", "bridge", "d4", "d1"}
    # flat order is registration order: encoder first, decoder + out_conv last
    assert r["d1"][0] == 0 and r["d1"][1] == r["d4"][0] and r["d4"][1] == r["bridge"][0]
    assert r["bridge"][1] == r["u4"][0] and r["u4"][1] == r["u3"][0]
    # the bucket that fires last (nothing left to overlap it with) is the small one
    assert r["d1"][1] - r["d1"][0] < 0.4 * (r["d4"][1] - r["d4"][0])
    assert r["u3"][1] == st.total
    # channels_last parameter views: logical OIHW shape, K-contiguous physical order
    w = dict(model.named_parameters())["d2.0.block.0.weight"]
    assert tuple(w.shape) == (16, 8, 3, 3) and w.stride() == (72, 1, 24, 8)


def _accum_worker(rank, world, port, q):
    """DataParallel's part of gradient accumulation (model._NetFn hands it the kept gradient): the kept, already reduced
    gradient is added only behind the bucket all-reduces of the new one."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import att_aspp_unet_amd as A
        from att_aspp_unet_amd.engine import ParamStore
        torch.manual_seed(0)
        model = A.AttentionASPPUNet(base_c=8)
        model.engine.store = ParamStore(model, torch.device("cpu"))
        st = model.engine.store
        dp = A.DataParallel(model)
        assert model.engine.dp is dp and model.engine.bucket_cb is not None
        marks = ("u1", "u2", "u3", "u4", "bridge", "d4", "d3", "d2", "d1")

        def local(step, r):
            return torch.randn(st.total, generator=torch.Generator().manual_seed(1000 * step + r))

        def backward(step, keep):
            dp.before_backward(keep)                 # what model._NetFn.backward does
            st.gflat.copy_(local(step, rank))        # the engine rewrites the flat buffer ...
            for mk in marks:                         # ... and fires the bucket marks on the way
                model.engine.bucket_cb(mk)
        backward(1, None)
        dp.finish()
        g1 = sum(local(1, r) for r in range(world))
        assert torch.allclose(st.gflat, g1, atol=1e-6)
        backward(2, st.gflat.clone())                # gradients not cleared: accumulate
        dp.finish()
        g2 = sum(local(2, r) for r in range(world))
        assert torch.allclose(st.gflat, g1 + g2, atol=1e-5), float((st.gflat - g1 - g2).abs().max())
        assert dp.pending is None
        # a backward before finish() of the previous one is refused
        backward(3, None)
        try:
            dp.before_backward(None)
            ok = False
        except RuntimeError:
            ok = True
        assert ok
        dp.finish()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gradient_accumulation_under_data_parallel_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
