"""HBM traffic per launch from two rocprofv3 counter passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE):

  python scripts/pmc_table.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]

Per (kernel, grid): launches, average FETCH_SIZE and WRITE_SIZE (KiB, as rocprofv3 reports them) and
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced
reads at 64 bytes, so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import json
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "1")) or 1)
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
            key = f"{name} grid={grid // max(wg, 1)}"
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key in sorted(set(fetch) | set(write)):
        nf, sf = fetch.get(key, (0, 0.0))
        nw, sw = write.get(key, (0, 0.0))
        f_kib = sf / nf if nf else 0.0
        w_kib = sw / nw if nw else 0.0
        out[key] = {"launches": max(nf, nw), "fetch_size_kib": round(f_kib, 1), "write_size_kib": round(w_kib, 1),
                    "hbm_bytes": int((2 * f_kib + w_kib) * 1024)}
    for key, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:50]:
        print(f"{key[:70]:70s} n={v['launches']:5d} fetch={v['fetch_size_kib']:11.1f} KiB write={v['write_size_kib']:11.1f} KiB "
              f"hbm={v['hbm_bytes'] / 1e6:9.1f} MB")
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
