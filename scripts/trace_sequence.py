"""Kernel sequence of the last step in a rocprofv3 kernel trace CSV (start order): python scripts/trace_sequence.py <trace.csv> <marker kernel>
The marker (default mask_generate_kernel, the first kernel of a step) separates steps; prints name, grid, duration and the gap
to the previous kernel's end for the last whole step."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "mask_generate_kernel"
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = ends[-2], ends[-1]
prev_end = int(rows[lo - 1]["End_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:60]
    print(f"{name:60s} grid={int(r['Grid_Size_X'])//max(int(r['Workgroup_Size_X']),1):6d} q={r.get('Queue_Id','?'):>2s} dur={(e-s)/1e3:8.1f} gap={(s-prev_end)/1e3:7.1f}")
    prev_end = max(prev_end, e)
