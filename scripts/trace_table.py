"""Per (kernel, grid) time table from a rocprofv3 kernel trace CSV: python scripts/trace_table.py <trace.csv> <steps>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:44]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    agg[key][0] += 1; agg[key][1] += d
tot = sum(v[1] for v in agg.values())
print(f"total kernel time per step: {tot/steps/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{k[0]:44s} grid={int(k[1])//int(k[3]):6d}x{k[2]:>4s} calls/step={v[0]/steps:6.1f} avg={v[1]/v[0]:8.1f}us  per-step={v[1]/steps/1e3:7.3f} ms")
