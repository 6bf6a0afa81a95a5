"""Summarise a rocprofv3 kernel_trace.csv: dispatches of the last full train step in launch order."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
tot = 0
agg = {}
for r in step:
    name = r["Kernel_Name"]
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("aau::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    agg[short] = agg.get(short, 0) + d
    if len(sys.argv) > 2:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {short[:44]:44s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} x{r['Grid_Size_Y']:>3s} lds {r['LDS_Block_Size']:>6s} vgpr {r['VGPR_Count']:>3s}+{r['Accum_VGPR_Count']:>3s} {d:8.1f} us")
print(f"step wall {(int(step[-1]['End_Timestamp'])-t0)/1e3:.1f} us, sum of kernels {tot:.1f} us, {len(step)} dispatches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"  {v:9.1f} us  {k}")
