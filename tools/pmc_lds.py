#!/usr/bin/env python3
"""LDS counters per kernel from one rocprofv3 pass:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS ...
    python3 tools/pmc_lds.py <dir> [kernel-substring]
Prints the averaged raw counters for the largest grid of every kernel whose name contains the substring."""
import csv, glob, os, sys
from collections import defaultdict
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ntt")
rows = defaultdict(dict)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        e = rows[r["Dispatch_Id"]]
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["name"] = r["Kernel_Name"].split("(")[0].replace("p2k::", "").replace("void ", "")
        e["grid"] = int(r["Grid_Size"])
        e["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
by = defaultdict(list)
for e in rows.values():
    if sub in e["name"]:
        by[(e["name"], e["grid"])].append(e)
for (name, grid), es in sorted(by.items(), key=lambda kv: -kv[0][1])[:6]:
    keys = sorted(k for k in es[0] if k not in ("name", "grid"))
    print(name, "grid", grid, "launches", len(es))
    for k in keys:
        print("   %-24s %.4g" % (k, sum(e[k] for e in es) / len(es)))
