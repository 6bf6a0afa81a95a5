"""MFMA utilisation / LDS bank conflicts per kernel of the REAL train step, from one rocprofv3 PMC pass:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES \
              SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmcM -o runc \
              -- python3 $R/bench.py --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline
    python scripts/pmc_mfma.py gpurun_out/pmcM profiles/r02_pmc_mfma.txt

Per kernel name, over the dispatches of the last full step (between two pack_kernel launches):
  kernel cycles   = SQ_BUSY_CYCLES / 32          (the counter is summed over the 32 shader engines)
  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)
  LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
Profiled passes run at a lower clock than un-profiled ones (guide, DVFS item 2): utilisation is a ratio of cycle
counts and is not affected; durations are not quoted from this pass."""
import csv, glob, os, sys


def main():
    d, out = sys.argv[1:3]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    did = "Dispatch_Id" if "Dispatch_Id" in rows[0] else None
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # dispatch order -> step boundaries at pack_kernel
    disp = []
    seen = set()
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
    lines = ["rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES "
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
             "-- python3 bench.py --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline   (last full train step, bs 8, 512^2, c 48)",
             "sums over the dispatches of one step; kernel cycles = SQ_BUSY_CYCLES / 32 shader engines;",
             "MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); LDS conflict = BANK_CONFLICT / IDX_ACTIVE;",
             "wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (quad-cycles both)", "",
             f"{'kernel':44s} {'disp':>4s} {'kernel cyc':>12s} {'MFMA busy':>14s} {'MFMA util':>9s} {'LDS confl':>9s} {'wait':>6s} {'issue stall':>11s}"]
    order = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0))
    for name, a in order:
        busy = a.get("SQ_BUSY_CYCLES", 0.0) / 32.0
        mf = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0:
            continue
        util = mf / (1024.0 * busy)
        lds = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(a.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
        wc = max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        lines.append(f"{name[:44]:44s} {len(a['n']):4d} {busy:12.0f} {mf:14.0f} {util:9.3f} {lds:9.3f} "
                     f"{a.get('SQ_WAIT_ANY', 0.0) / wc:6.3f} {a.get('SQ_WAIT_INST_ANY', 0.0) / wc:11.3f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
