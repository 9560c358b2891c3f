"""LDS bank-conflict share per kernel from a rocprofv3 counter pass with
--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES: python scripts/pmc_lds.py <counter_collection.csv>"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:44] + " g" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:16]:
    c, a = v.get("SQ_LDS_BANK_CONFLICT", 0.0), v.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"{k:58s} conflict cycles / active cycles = {c / max(a, 1.0):.3f}   LDS instructions {v.get('SQ_INSTS_LDS', 0):.3g}")
