"""HBM traffic of one train step from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM / rocprofv3 section):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcF -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW -o runc -- python3 $R/bench.py --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline
    python scripts/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW profiles/<name>.json

Units / corrections as the guide prescribes: both counters tick in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced streaming reads (128-B requests tallied at 64 B), so it is doubled; WRITE_SIZE is exact for
16-B-per-lane stores and float atomics.  The last full step (between two pack_kernel launches) is summed; "conv"
launches are the MFMA implicit-GEMM kernels (igemm / conv3x3* / conv1x1* / wgrad* and the split-K reduce)."""
import csv, glob, json, os, sys

CONV = ("igemm_kernel", "conv3x3", "conv1x1", "wgrad", "wg_reduce")


def last_step(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
    return rows[idx[-2]:idx[-1]]


def main():
    dF, dW, out = sys.argv[1:4]
    rd, wr = last_step(dF, "FETCH_SIZE"), last_step(dW, "WRITE_SIZE")
    is_conv = lambda r: any(k in r["Kernel_Name"] for k in CONV)
    kib = lambda rows: sum(float(r["Counter_Value"]) for r in rows) * 1024.0
    n_conv = sum(1 for r in rd if is_conv(r) and "wg_reduce" not in r["Kernel_Name"])
    conv_rd, conv_wr = 2.0 * kib([r for r in rd if is_conv(r)]), kib([r for r in wr if is_conv(r)])
    by, nl = {}, {}
    for rows, mul in ((rd, 2.0), (wr, 1.0)):
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aau::", "")
            by[k] = by.get(k, 0.0) + mul * float(r["Counter_Value"]) * 1024.0
            if mul == 2.0:
                nl[k] = nl.get(k, 0) + 1

    def canon(k):   # "conv3x3g_kernel<96>" -> "conv3x3g<96>" (the tag bench.py's live profiler reports)
        k = k.replace("_kernel", "").replace(" ", "").replace("false", "0").replace("true", "1")
        return "wgrad3x3<3,8>" if k.startswith("wgrad3x3<3,8") else k
    import importlib.util
    spec = importlib.util.spec_from_file_location("_aau_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "att-aspp-unet_amd", "build.py"))
    bld = importlib.util.module_from_spec(spec); spec.loader.exec_module(bld)
    res = {
        "source_hash": bld.source_hash(),     # bench.py quotes these figures only for the kernels they were measured on
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --graph 0; last step",
        "correction": "FETCH_SIZE x 1024 B x 2 (gfx950 counts 128-B read requests as 64 B on wide coalesced streams), WRITE_SIZE x 1024 B",
        "conv_kernels": {"launches_per_step": n_conv, "read_bytes_per_step": conv_rd, "write_bytes_per_step": conv_wr,
                         "bytes_per_launch": (conv_rd + conv_wr) / max(n_conv, 1)},
        "whole_step": {"read_bytes": 2.0 * kib(rd), "write_bytes": kib(wr)},
        "bytes_by_kernel": dict(sorted(by.items(), key=lambda kv: -kv[1])),
        "by_kernel": {canon(k): {"bytes_per_step": v, "launches_per_step": nl.get(k, 0),
                                 "bytes_per_launch": v / max(nl.get(k, 1), 1)}
                      for k, v in sorted(by.items(), key=lambda kv: -kv[1])},
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["conv_kernels"]), json.dumps(res["whole_step"]))


if __name__ == "__main__":
    main()
