"""Reduce a rocprofv3 --pmc pass of SQ counters (counter_collection.csv) to per-kernel sums.

    python profiles/pmc_sq.py <counter_collection.csv> out.json

Per kernel name (largest launch only): the raw counters, the instruction mix and
valu_time_share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of the waves' resident time in
which they execute vector-ALU instructions; both counters tick in the same unit)."""
import csv, json, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
by_dispatch = defaultdict(dict)
meta = {}
for r in rows:
    d = r["Dispatch_Id"]
    by_dispatch[d][r["Counter_Name"]] = by_dispatch[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[d] = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
best = {}
for d, c in by_dispatch.items():
    name, grid, vgpr, lds = meta[d]
    if not name.startswith("void adn::") and not name.startswith("adn::"):
        continue
    if name not in best or grid > best[name][0]:
        best[name] = (grid, vgpr, lds, c)
out = {}
for name, (grid, vgpr, lds, c) in sorted(best.items(), key=lambda kv: -kv[1][3].get("SQ_WAVE_CYCLES", 0)):
    waves = grid / 64
    e = {"grid_threads": grid, "vgprs": vgpr, "lds_bytes_per_workgroup": lds, "counters": c}
    if c.get("SQ_WAVE_CYCLES"):
        e["valu_time_share"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"], 4)
        e["lds_time_share"] = round(c.get("SQ_ACTIVE_INST_LDS", 0) / c["SQ_WAVE_CYCLES"], 4)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
        if k in c:
            e[k.lower() + "_per_wave"] = round(c[k] / waves, 1)
    out[name] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in list(out.items())[:8]:
    print(k[:70].ljust(72), {a: b for a, b in v.items() if a != "counters"})
