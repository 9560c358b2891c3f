"""Idle gaps of the busiest queue inside the last `window_ms` of a rocprofv3 kernel trace (the timed steps of bench.py):
python scripts/trace_gaps.py <kernel_trace.csv> [window_ms]"""
import csv, sys, collections, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 100e6
t_end = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - win]
busy = collections.defaultdict(float)
for r in rows:
    busy[r["Stream_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
main_q = max(busy, key=busy.get)
print({q: round(b / win, 3) for q, b in busy.items()}, "fraction of the window with a kernel of that queue running; kernels:", len(rows))
mine = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows if r["Stream_Id"] == main_q)
others = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Stream_Id"] != main_q)
ostarts = [s for s, _ in others]
def covered(a, b):
    i = max(0, bisect.bisect_left(ostarts, a) - 64)
    tot, last = 0, a
    for s, e in others[i:]:
        if s >= b: break
        lo, hi = max(s, last), min(e, b)
        if hi > lo: tot += hi - lo; last = hi
    return tot
edges = [0, 1000, 3000, 10000, 30000, 100000, 10**12]
names = ["<1us", "1-3us", "3-10us", "10-30us", "30-100us", ">=100us"]
stat = [[0, 0.0, 0.0] for _ in names]
overlap_pairs = 0
after = collections.Counter()
prev_end = mine[0][1]
for (s0, e0, n0), (s1, e1, n1) in zip(mine, mine[1:]):
    g = s1 - prev_end
    if g <= 0:
        overlap_pairs += 1
    else:
        k = bisect.bisect_right(edges, g) - 1
        stat[k][0] += 1; stat[k][1] += g; stat[k][2] += covered(prev_end, s1)
        if g >= 10000: after[(n0, n1)] += g
    prev_end = max(prev_end, e1)
print(f"stream {main_q}: {len(mine)} kernels, {overlap_pairs} start before the previous one ends")
print("gap length   count   % of window   % of window with another queue busy")
for nme, (n, t, c) in zip(names, stat):
    print(f"  {nme:9s} {n:6d} {100*t/win:8.2f} {100*c/win:8.2f}")
print("largest contributors to gaps >= 10 us (previous kernel -> next kernel, % of window):")
for (a, b), t in after.most_common(12):
    print(f"  {100*t/win:6.2f}  {a} -> {b}")
