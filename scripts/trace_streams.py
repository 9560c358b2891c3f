"""Per-queue busy time and overlap from a rocprofv3 kernel trace CSV: python scripts/trace_streams.py <trace.csv> <steps>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
busy = collections.defaultdict(float)
iv = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[r.get(qkey, "?")] += (e - s) * 1e-6
    iv.append((s, e))
iv.sort()
# union of intervals = time with at least one kernel running
tot, cs, ce = 0, None, None
for s, e in iv:
    if cs is None: cs, ce = s, e
    elif s <= ce: ce = max(ce, e)
    else: tot += ce - cs; cs, ce = s, e
tot += ce - cs
print("columns:", list(rows[0].keys())[:14])
for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"queue {q}: {b/steps:8.2f} ms/step of kernel time")
print(f"union (>=1 kernel running): {tot*1e-6/steps:8.2f} ms/step; sum of kernel times {sum(busy.values())/steps:8.2f} ms/step")
