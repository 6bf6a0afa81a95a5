"""L2 (TCC) hit rate per kernel of the real train step, from one rocprofv3 PMC pass:

    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmcL -o runc \
              -- python3 $R/bench.py --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-infer
    python scripts/pmc_l2.py gpurun_out/pmcL out.txt

Sums over the dispatches of the last full step (between two pack_kernel launches); requests are 128-byte lines."""
import csv, glob, os, sys


def main():
    d, out = sys.argv[1:3]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    did = "Dispatch_Id" if "Dispatch_Id" in rows[0] else None
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    disp, seen = [], set()
    for r in rows:
        key = r[did] if did else (r["Start_Timestamp"], r["Kernel_Name"])
        if key not in seen:
            seen.add(key)
            disp.append((key, r["Kernel_Name"]))
    packs = [i for i, (_, k) in enumerate(disp) if "pack_kernel" in k]
    keep = {k for k, _ in disp[packs[-2]:packs[-1]]} if len(packs) >= 2 else {k for k, _ in disp}
    agg = {}
    for r in rows:
        key = r[did] if did else (r["Start_Timestamp"], r["Kernel_Name"])
        if key not in keep:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aau::", "")
        a = agg.setdefault(name, {"n": set()})
        a["n"].add(key)
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    lines = [f"{'kernel':48s} {'disp':>4s} {'L2 hits':>14s} {'L2 misses':>14s} {'hit rate':>8s} {'miss MB/disp':>12s}"]
    for name, a in sorted(agg.items(), key=lambda kv: -(kv[1].get("TCC_HIT_sum", 0) + kv[1].get("TCC_MISS_sum", 0))):
        h, m = a.get("TCC_HIT_sum", 0.0), a.get("TCC_MISS_sum", 0.0)
        if h + m <= 0:
            continue
        lines.append(f"{name[:48]:48s} {len(a['n']):4d} {h:14.0f} {m:14.0f} {h / (h + m):8.3f} {m * 128 / 1e6 / len(a['n']):12.1f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


main()
