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
# gaps on the busiest queue: time between the end of a kernel and the start of the next one on that queue, split by
# length and by whether any other queue was running a kernel during the gap
main_q = max(busy, key=busy.get)
mine = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r.get(qkey, "?") == main_q)
others = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r.get(qkey, "?") != main_q)
import bisect
ostarts = [s for s, _ in others]
def covered(a, b):
    """nanoseconds of [a, b) during which a kernel of another queue runs"""
    i = max(0, bisect.bisect_left(ostarts, a) - 64)
    tot, last = 0, a
    for s, e in others[i:]:
        if s >= b: break
        lo, hi = max(s, last), min(e, b)
        if hi > lo: tot += hi - lo; last = hi
    return tot
buckets = collections.OrderedDict((k, [0, 0.0, 0.0]) for k in ("<3us", "3-10us", "10-100us", ">=100us"))
for (s0, e0), (s1, e1) in zip(mine, mine[1:]):
    g = s1 - e0
    if g <= 0: continue
    k = "<3us" if g < 3000 else "3-10us" if g < 10000 else "10-100us" if g < 100000 else ">=100us"
    buckets[k][0] += 1; buckets[k][1] += g * 1e-6; buckets[k][2] += covered(e0, s1) * 1e-6
print(f"gaps on queue {main_q} (count/step, ms/step, of which another queue busy):")
for k, (n, t, c) in buckets.items():
    print(f"  {k:9s} {n/steps:8.1f} {t/steps:8.3f} {c/steps:8.3f}")
