"""Headline benchmark: images/sec of the full train step (forward + criterion + backward +
[gradient all-reduce] + clip + AdamW) of Attention-ASPP-UNet, base_c 48, 1x512x512 frames,
batch 8 per GPU, bf16 activations / fp32 master weights, synthetic ultrasound phantoms.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  Extra objects:
  roofline     the DOMINANT KERNEL of the step (the kernel variant with the largest summed time among
               the MFMA convolution launches; today conv3x3p<96>): algorithmic FLOPs of its launches /
               their summed duration, measured live with HIP events around every launch on the launch
               stream in a second (eager) pass of the same steps.  Beside it: `conv_family` (all conv
               launches against the 680.05 GFLOP/image of BASELINE.md section 2), `by_kernel` (every conv
               kernel variant) and `per_layer` (every conv launch of one step against
               min(MFMA peak, arithmetic intensity x HBM bandwidth)).
  cpu_baseline the CPU oracle (oracle/ref_cpu.py, an ATen fp32 restatement of the reference step)
               timed on this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_IMAGE = {48: 226.76, 32: 100.84, 16: 25.25}
PEAK_BF16_TFLOPS = 2500.0                                     # MI355X dense bf16 MFMA (guide, spec)
PEAK_HBM_TBS = 8.0                                            # HBM3E (guide, spec; 6.3 TB/s measured copy rate)


def train_gflop_per_image(base_c, size):
    f = FWD_GFLOP_PER_IMAGE.get(base_c, 226.76 * (base_c / 48.0) ** 2) * (size / 512.0) ** 2
    first_dgrad = 2 * 9 * base_c * size * size / 1e9           # no input gradient for the first layer
    return 3 * f - first_dgrad


def host_cores():
    """CPU share actually available to this process (cgroup quota / affinity): all of it unless AAU_CPU_THREADS says less."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    if os.environ.get("AAU_CPU_THREADS"):
        n = min(n, int(os.environ["AAU_CPU_THREADS"]))
    return max(1, n)


def host_cpu_string():
    """'<model name> x <sockets> sockets, <logical cpus> logical CPUs' from /proc/cpuinfo (the lscpu facts)."""
    try:
        model, phys, n = "?", set(), 0
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); n += 1
            elif line.startswith("physical id"):
                phys.add(line.split(":", 1)[1].strip())
        return f"{model}, {max(len(phys), 1)} socket(s), {n} logical CPUs on the host"
    except Exception:
        return "unknown"


def cpu_baseline(base_c, size, batch=8, seconds_budget=90.0):
    """Time the CPU oracle's train step on a bounded sample of the same workload: the metric's own batch (8), the
    protocol of BASELINE.md section 3 (2 warm-up + 5 timed steps) when that fits ~90 s, fewer timed steps otherwise."""
    from argparse import Namespace
    from oracle import ref_cpu as O
    from att_aspp_unet_amd import synth
    threads = host_cores()
    torch.set_num_threads(threads)
    torch.manual_seed(2025)
    net = O.AttentionASPPUNet(base_c=base_c)
    net.train()
    opt = O.make_optimizer(net, 3e-4)
    crit = O.build_criterion(Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05), O.ComboLoss(), O.EdgeLoss())
    x, y = synth.make_frames(batch, size, seed=2025)
    t0 = time.perf_counter()
    O.train_step(net, opt, crit, x, y)                          # warm-up 1 (also sizes the sample)
    t1 = time.perf_counter()
    O.train_step(net, opt, crit, x, y)                          # warm-up 2
    warm = time.perf_counter() - t1
    spent = time.perf_counter() - t0
    n = max(1, min(5, int((seconds_budget - spent) / max(warm, 1e-3))))
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        O.train_step(net, opt, crit, x, y)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": batch / med, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} timed + 2 warm-up fp32 train steps (median) of the CPU oracle at batch {batch}, "
                      f"{size}x{size}, base_c {base_c}: the metric's own step (BASELINE.md section 3: 2 + 5, the timed "
                      f"count cut to a ~{int(seconds_budget)} s budget when a step is slower than that allows)",
            "host_cpu": host_cpu_string(), "sec_per_step": med, "sec_per_step_min": ts[0]}


def library_source_hash():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_aau_build", os.path.join(ROOT, "att-aspp-unet_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def measured_traffic(dom_tag, base_c, size, batch):
    """HBM bytes per launch of the dominant kernel from the newest profiles/r*_pmc_traffic.json whose ``source_hash``
    equals the hash of the kernel sources this library was built from (the PMC passes are separate rocprofv3 runs of this
    very command, scripts/gpu_profile.sh).  A profile of other kernels is stale: -> (None, reason)."""
    import glob
    if not (base_c == 48 and size == 512 and batch == 8):
        return None, "not the profiled configuration"
    want = library_source_hash()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True)
    for f in files:
        try:
            pm = json.load(open(f))
        except Exception:
            continue
        if pm.get("source_hash") != want:
            continue
        k = pm["by_kernel"].get(dom_tag.split(" ")[0])
        return (k["bytes_per_launch"], os.path.basename(f)) if k else (None, f"{os.path.basename(f)} has no row for {dom_tag}")
    return None, f"no profiles/r*_pmc_traffic.json stamped with source hash {want}"


def roofline_from_records(recs, nprof, alg_flops, step_s, base_c, size, batch):
    """Roofline object of the JSON line from the per-launch event records of `nprof` eager steps."""
    per = len(recs) // nprof
    conv = ("igemm", "wgrad")
    fam = {}
    for r in recs:
        f = fam.setdefault(r["family"], {"ms": 0.0, "flops": 0.0, "launches": 0})
        f["ms"] += r["ms"]; f["flops"] += r["flops"]; f["launches"] += 1
    by = {}
    for r in recs:
        if r["family"] in conv:
            k = by.setdefault(r["tag"] or "?", {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            k["ms"] += r["ms"]; k["flops"] += r["flops"]; k["bytes"] += r["bytes"]; k["launches"] += 1
    by_kernel = {}
    for tag, k in sorted(by.items(), key=lambda kv: -kv[1]["ms"]):
        tf = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
        by_kernel[tag] = {"launches_per_step": k["launches"] / nprof, "gflop_per_step": k["flops"] / nprof / 1e9,
                          "ms_per_step": k["ms"] / nprof, "tflops": tf, "frac_of_mfma_peak": tf / PEAK_BF16_TFLOPS,
                          "alg_gbytes_per_step": k["bytes"] / nprof / 1e9}
    dom_tag = next(iter(by_kernel))
    dom = by_kernel[dom_tag]
    # every conv launch of a step (mean over the profiled steps) against its own roofline
    per_layer = []
    if per * nprof == len(recs):
        for i in range(per):
            r0 = recs[i]
            if r0["family"] not in conv or r0["flops"] <= 0:
                continue
            ms = sum(recs[s * per + i]["ms"] for s in range(nprof)) / nprof
            tf = r0["flops"] / (ms * 1e-3) / 1e12
            ai = r0["flops"] / r0["bytes"] if r0["bytes"] > 0 else float("inf")
            roof = min(PEAK_BF16_TFLOPS, ai * PEAK_HBM_TBS)
            per_layer.append([r0["label"], r0["tag"], round(r0["flops"] / 1e9, 2), round(r0["bytes"] / 1e6, 1),
                              round(ms * 1e3, 1), round(tf, 1), round(roof, 1), round(tf / roof, 3)])
    conv_ms = sum(fam.get(k, {"ms": 0.0})["ms"] for k in conv) / nprof
    conv_launches = sum(fam.get(k, {"launches": 0})["launches"] for k in conv) // nprof
    traffic, traffic_src = measured_traffic(dom_tag, base_c, size, batch)
    return {"bound": "mfma", "achieved": dom["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": dom["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": dom_tag, "dominant_kernel": dict(name=dom_tag, **dom),
            "avg_launch_ms": dom["ms_per_step"] / max(dom["launches_per_step"], 1e-9),
            "alg_bytes_per_launch": dom["alg_gbytes_per_step"] * 1e9 / max(dom["launches_per_step"], 1e-9),
            "conv_family": {"launches_per_step": conv_launches, "ms_per_step": conv_ms,
                            "algorithmic_flops_per_step": alg_flops,
                            "tflops": alg_flops / (conv_ms * 1e-3) / 1e12,
                            "frac": alg_flops / (conv_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS},
            "ms_per_step_by_family": {k: v["ms"] / nprof for k, v in fam.items()},
            "whole_step_frac": alg_flops / step_s / 1e12 / PEAK_BF16_TFLOPS,
            "by_kernel": by_kernel,
            "per_layer_columns": ["layer:call", "kernel", "GFLOP", "alg MB", "us", "TFLOP/s",
                                  "roof = min(2500, AI x 8 TB/s)", "frac of roof"],
            "per_layer": per_layer}


def inference_numbers(dev):
    """BASELINE configs 2 and 5 (not the metric): eval forward bs 4 and the 1024^2 sliding window, hipGraph replays."""
    import att_aspp_unet_amd as A
    from att_aspp_unet_amd import synth

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    torch.manual_seed(2025)
    m = A.AttentionASPPUNet(base_c=48).to(dev).eval()
    x, _ = synth.make_frames(4, 512, seed=1)
    x = x.to(dev)
    gf = A.GraphedForward(m, (4, 1, 512, 512), device=dev)
    t2 = timeit(lambda: gf(x))
    m5 = A.AttentionASPPUNet(base_c=48, rates=(6, 12, 18, 24)).to(dev).eval()
    big = torch.rand(1, 1, 1024, 1024, device=dev)
    gf9 = A.GraphedForward(m5, (9, 1, 512, 512), device=dev)
    t5 = timeit(lambda: A.predict_sliding_window(m5, big, 512, 256, forward=gf9), n=10)
    # the same window batch in IEEE half (libaau_f16.so), the dtype BASELINE.json states for config 5
    m5.set_precision("fp16")
    gf9h = A.GraphedForward(m5, (9, 1, 512, 512), device=dev)
    t5h = timeit(lambda: A.predict_sliding_window(m5, big, 512, 256, forward=gf9h), n=10)
    return {"config2_fwd_bs4_512_bf16_hipgraph": {"ms": t2 * 1e3, "images_per_sec": 4 / t2,
                                                  "conv_tflops": 4 * 226.76e9 / t2 / 1e12},
            "config5_1024_sliding_window_9x512_rates_6_12_18_24_fp16_hipgraph": {"ms_per_frame": t5h * 1e3,
                                                                                 "frames_per_sec": 1 / t5h},
            "config5_1024_sliding_window_9x512_rates_6_12_18_24_bf16_hipgraph": {"ms_per_frame": t5 * 1e3,
                                                                                 "frames_per_sec": 1 / t5}}


def spawn_ranks(n, argv):
    """One child process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, as torch.distributed.run
    sets them); rank 0's stdout is ours.  If a rank dies the others are terminated instead of waiting in a rendezvous
    nobody will join.  -> exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in live:
                    q.terminate()
    return rc


def launch_selftest(world, rank):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_selftest": world, "sum_of_rank_plus_one": float(t.item())}))
    dist.destroy_process_group()


def ranks_seen(world, dev):
    """Every rank adds 1 over the process group (RCCL at N > 1): the sum is the number of ranks the collective library
    really connected, which `n_gpus` (WORLD_SIZE echoed back) does not prove.  1 without a process group."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    return int(round(float(t.item())))


def dp_overlap_numbers(dp, step, x, y, reps=5):
    """Data-parallel overlap evidence (every rank runs this; collectives inside): per gradient bucket the bytes, the
    all-reduce time ALONE (blocking, nothing else on the device, median of `reps`) and the time of the backward kernels
    that run BEHIND its mark in a real step (HIP events on the compute stream at the mark and at the end of the
    backward) -- a bucket is hidden when the second exceeds the first; plus the step without any all-reduce."""
    import torch.distributed as dist
    red = dp._ensure()
    ev = {}
    orig = red.on_mark

    def on_mark(mark):
        if mark in red.ranges:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev[mark] = e
        orig(mark)
    red.on_mark = on_mark
    fin = dp.finish
    end = torch.cuda.Event(enable_timing=True)

    def finish():
        end.record()                      # the backward list has been enqueued; the waits for the buckets follow
        fin()
    dp.finish = finish
    step(x, y)
    torch.cuda.synchronize()
    red.on_mark, dp.finish = orig, fin
    behind = {m: e.elapsed_time(end) for m, e in ev.items()}
    out = {}
    for mark, (b, e) in red.ranges.items():
        buf = torch.zeros(e - b, device=x.device)
        ts = []
        for _ in range(reps):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.all_reduce(buf)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out[mark] = {"mbytes": (e - b) * 4 / 1e6, "allreduce_ms_alone": ts[len(ts) // 2],
                     "backward_ms_behind_its_mark": behind.get(mark)}
    # the same steps with the all-reduces left out (every rank trains on its own gradient: timing only)
    red.on_mark = lambda mark: red.fired.append(mark) if mark in red.ranges else None
    for _ in range(2):
        step(x, y)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step(x, y)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / reps * 1e3
    red.on_mark = orig
    dist.barrier()
    return {"buckets": out, "eager_step_ms_without_allreduce": alone}


def loader_numbers(dev, n=96):
    """Feed rate of the input pipeline (not the metric): n synthetic 562x744 PNG frames + masks on
    local disk -> DirectoryLoader (thread-pool decode, GPU Resize + the random augmentations of pipeline:149-153) -> batches
    of 8 x 1x512x512, one warm-up epoch and one timed epoch."""
    import tempfile
    import numpy as np
    from pathlib import Path
    from PIL import Image
    from att_aspp_unet_amd import dataset, synth
    with tempfile.TemporaryDirectory() as td:
        root = Path(td)
        (root / "images").mkdir(); (root / "masks").mkdir()
        x, y = synth.make_frames(8, 512, seed=3)
        for k in range(n):
            fr = (np.pad(x[k % 8, 0].numpy(), ((25, 25), (116, 116))) * 255).astype(np.uint8)     # 562 x 744, the native size
            Image.fromarray(np.roll(fr, 3 * k, axis=1)).save(root / "images" / f"c{k:03d}.png")
            if k % 5:
                Image.fromarray((np.pad(y[k % 8, 0].numpy(), ((25, 25), (116, 116))) * 255).astype(np.uint8)).save(root / "masks" / f"c{k:03d}.png")
        imgs, msks = dataset.collect_pair(root / "images", root / "masks")
        out = {}
        for workers in (1, host_cores()):
            ld = dataset.DirectoryLoader(imgs, msks, 8, 512, train=True, seed=2025, device=dev, workers=workers)
            list(ld)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cnt = sum(b[0].shape[0] for b in ld)
            torch.cuda.synchronize()
            out[f"frames_per_sec_{workers}_decode_threads"] = cnt / (time.perf_counter() - t0)
        out["what"] = f"{n} PNG frames 562x744 (+ masks) from local disk -> augmented 8 x 1x512x512 device batches"
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--base_c", type=int, default=48)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8, help="per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch table (every kernel of one step) to this file")
    ap.add_argument("--no-infer", action="store_true", help="skip the inference side numbers (configs 2 and 5)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("AAU_BENCH_GRAPH", "1")),
                    help="replay the step as one hipGraph when possible")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="launcher check without a GPU: the ranks rendezvous over gloo, rank 0 prints one JSON line")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start one fresh process per GPU ourselves.  Nothing in this process has touched
        # the GPU yet (importing torch does not), and the children are new interpreters, not an exec of this one.
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:]))
    if a.selftest_launch:
        return launch_selftest(world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # AAU_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (the ranks then share
    # the cards; RCCL refuses two ranks on one device).  Never the measured configuration.
    backend = os.environ.get("AAU_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import att_aspp_unet_amd as A
    from att_aspp_unet_amd import _abi, synth
    from argparse import Namespace
    args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    torch.manual_seed(2025)
    model = A.AttentionASPPUNet(base_c=a.base_c).to(dev).train()
    force_dp = os.environ.get("AAU_FORCE_DP") == "1"     # exercise the DP path with one rank (sanity runs)
    if force_dp and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dp = A.DataParallel(model) if (world > 1 or force_dp) else None
    opt = A.FusedAdamW(model, lr=3e-4, weight_decay=A.WEIGHT_DECAY, max_grad_norm=A.GRAD_CLIP)
    step = A.TrainStep(model, opt, args, dp)
    x, y = synth.make_frames(a.batch, a.size, seed=2025 + rank)     # weak scaling: 8 new frames per rank
    x, y = x.to(dev), y.to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    seen = ranks_seen(world, dev)            # before the timed loop: what the collective library connected

    run = step
    graphed = False
    for i in range(a.warmup):
        loss = run(x, y)
    barrier()
    if a.graph and (dp is None or os.environ.get("AAU_BENCH_DP_GRAPH") == "1"):
        # the whole step as ONE hipGraph at N = 1.  With data parallelism GraphedTrainStep can replay one graph per
        # gradient-bucket segment of the backward with the RCCL all-reduces in between, but that measured 1 % SLOWER
        # than the eager launch list (which already runs within 1 % of the single graph), so N > 1 stays eager.
        try:
            gstep = A.GraphedTrainStep(step, x, y, warmup=0)
            gstep(x, y)
            torch.cuda.synchronize()
            run = gstep
            graphed = True
        except Exception as e:  # capture is an optimisation; the eager replay list is the same work
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); timing the eager launch list",
                  file=sys.stderr)
            torch.cuda.synchronize()
            run = step
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = run(x, y)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_val = float(loss.item())

    roof = None
    if not a.no_roofline:
        # second pass of the same steps (eager launch list) with every launch bracketed by HIP events on its stream.
        # EVERY rank runs these steps (a data-parallel step contains collectives: rank 0 alone would wait for peers that
        # have moved on); only rank 0 records.
        nprof = min(a.steps, 5)
        step(x, y)
        torch.cuda.synchronize()
        if rank == 0:
            _abi.prof_enable(True)
        for i in range(nprof):
            step(x, y)
        torch.cuda.synchronize()
        if rank == 0:
            _abi.prof_enable(False)
    if not a.no_roofline and rank == 0:
        recs = _abi.prof_collect_launches()
        if a.dump_launches and len(recs) % nprof == 0:
            per = len(recs) // nprof
            with open(a.dump_launches, "w") as f:
                tot = 0.0
                for i in range(per):
                    us = sum(recs[s_ * per + i]["ms"] for s_ in range(nprof)) / nprof * 1e3
                    r0 = recs[i]
                    tot += us
                    gbs = r0["bytes"] / us / 1e3 if r0["bytes"] > 0 else 0.0
                    f.write(f"{i:4d} {r0['family']:12s} {r0['label'][:44]:44s} {r0['tag'][:28]:28s} {us:8.1f} us  "
                            f"{r0['bytes'] / 1e6:8.1f} MB {gbs:7.0f} GB/s  {r0['flops'] / 1e9:8.2f} GF\n")
                f.write(f"total {tot:.1f} us over {per} launches\n")
        alg = train_gflop_per_image(a.base_c, a.size) * a.batch * 1e9
        roof = roofline_from_records(recs, nprof, alg, dt / a.steps, a.base_c, a.size, a.batch)
    # The reference loop hands HOST batches to the device every step (pipeline:319: 2 x 8 MB at bs 8).  `value` is
    # measured with the inputs resident in HBM; this extra pass times the same steps with a pinned-host -> HBM copy of
    # the frames and the masks issued in front of each step (same stream: the worst case, no prefetch overlap).
    h2d = None
    if rank == 0 and world == 1 and not a.no_roofline:
        xh, yh = x.cpu().pin_memory(), y.cpu().pin_memory()
        nh = max(5, a.steps // 2)
        for _ in range(2):
            x.copy_(xh, non_blocking=True); y.copy_(yh, non_blocking=True); run(x, y)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nh):
            x.copy_(xh, non_blocking=True); y.copy_(yh, non_blocking=True); run(x, y)
        torch.cuda.synchronize()
        h2d = a.batch * nh / (time.perf_counter() - t1)
    overlap = None
    if dp is not None and not a.no_roofline:
        overlap = dp_overlap_numbers(dp, step, x, y)
        t1 = time.perf_counter()
        for _ in range(3):
            step(x, y)
        torch.cuda.synchronize()
        overlap["eager_step_ms_with_allreduce"] = (time.perf_counter() - t1) / 3 * 1e3
        overlap["exposed_allreduce_ms"] = overlap["eager_step_ms_with_allreduce"] - overlap["eager_step_ms_without_allreduce"]
    if world > 1:
        dist.barrier()

    infer = None
    if rank == 0 and world == 1 and not a.no_roofline and a.base_c == 48 and a.size == 512 and not a.no_infer:
        try:
            infer = inference_numbers(dev)
        except Exception as e:   # reported, never fatal for the metric line
            infer = {"error": f"{type(e).__name__}: {e}"}
    loader = None
    if rank == 0 and world == 1 and not a.no_roofline and not a.no_infer:
        try:
            loader = loader_numbers(dev)
        except Exception as e:   # reported, never fatal for the metric line
            loader = {"error": f"{type(e).__name__}: {e}"}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.base_c, a.size, a.batch)

    if rank == 0:
        imgs = a.batch * world * a.steps
        out = {
            "metric": "images/sec (train step) 1x512x512 bs=8/GPU",
            "value": imgs / dt, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Attention-ASPP-UNet full train step (fwd+Dice/BCE/edge loss+bwd+clip+AdamW), "
                                   f"base_c {a.base_c}, 1x{a.size}x{a.size}, batch {a.batch}/GPU, bf16 activations, "
                                   f"fp32 master weights" + (", RCCL grad all-reduce overlapped with backward" if world > 1 else ""),
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "hipgraph": graphed,
                       "rccl_ranks_seen": seen, "collective_backend": (backend if dist.is_initialized() else None),
                       "images_per_sec_per_gpu": imgs / dt / world, "overlap": overlap,
                       "final_loss": loss_val, "images_per_sec_with_h2d_of_inputs": h2d,
                       "inference_not_the_metric": infer, "input_pipeline_not_the_metric": loader},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
