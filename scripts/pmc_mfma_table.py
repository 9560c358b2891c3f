"""MFMA utilisation per kernel from one rocprofv3 counter pass:

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py ...
  python scripts/pmc_mfma_table.py <counter_collection.csv> [out.json]

Per (kernel, grid), averaged over its launches:
  mfma_busy   SQ_VALU_MFMA_BUSY_CYCLES: cycles with the matrix pipe busy, summed over the chip's 1,024 SIMDs
  gui_active  GRBM_GUI_ACTIVE: cycles the graphics engine was busy with the dispatch, summed over the 8 XCDs
  util        = mfma_busy / (gui_active / 8 * 1024): the share of SIMD-cycles of the dispatch in which an MFMA executed
  mops        SQ_INSTS_VALU_MFMA_MOPS_BF16: executed bf16 MFMA work in units of 512 FLOP; flops = 512 * mops includes the
              padded rows / keys a kernel multiplies (compare with the algorithmic FLOPs of the launch)
  tflops_at_clock = flops / (gui_active / 8 / clock): rate over the dispatch's own busy cycles at the given clock (2.4 GHz
              peak; the chip runs lower under load, so this bounds the HIP-event rate from above)"""
import collections
import csv
import json
import sys

N_SIMD = 1024
CLOCK_HZ = 2.4e9


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "1")) or 1)
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
            key = f"{name} grid={grid // max(wg, 1)}"
            c = agg[key][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    out = {}
    for key, cs in agg.items():
        avg = {k: v[1] / v[0] for k, v in cs.items()}
        n = max(v[0] for v in cs.values())
        busy, gui = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), avg.get("GRBM_GUI_ACTIVE", 0.0)
        mops = avg.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        cycles = gui / 8.0
        out[key] = {"launches": n, "mfma_busy_cycles": busy, "gui_active": gui, "sq_busy_cycles": avg.get("SQ_BUSY_CYCLES", 0.0),
                    "mfma_util": round(busy / (cycles * N_SIMD), 4) if cycles > 0 else None,
                    "mops_bf16": mops, "flops": 512.0 * mops,
                    "tflops_at_clock": round(512.0 * mops / (cycles / CLOCK_HZ) / 1e12, 1) if cycles > 0 and mops > 0 else None}
    rows = sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"] * kv[1]["launches"])
    for key, v in rows[:40]:
        if v["mfma_busy_cycles"] <= 0:
            continue
        print(f"{key[:64]:64s} n={v['launches']:5d} util={v['mfma_util']!s:>7} gflop/launch={v['flops'] / 1e9:9.2f} "
              f"TF/s@2.4GHz={v['tflops_at_clock']!s:>7} busy={v['mfma_busy_cycles']:.3e} gui={v['gui_active']:.3e}")
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
