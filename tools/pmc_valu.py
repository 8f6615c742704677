#!/usr/bin/env python3
"""Per-kernel VALU issue summary from one rocprofv3 counter pass:

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
        -d gpurun_out/pmc_valu_<tag> --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python3 tools/pmc_valu.py gpurun_out/pmc_valu_<tag> [--src-id gpurun_out/<tag>_srcid.json] [--git-head SHA] > profiles/<tag>_valu.json

For every kernel (launches of one name and grid size are averaged; the largest grid of a name is reported):
    duration_ms     dispatch End - Start of the SAME pass (counter collection serialises kernels and slows them a little)
    clock_GHz       GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, 'DVFS give-back')
    valu_insts      SQ_INSTS_VALU: wave-instructions issued
    lane_ops_per_s  valu_insts * 64 / duration
    peak            256 CU x 4 SIMD x 32 lanes x clock: one wave64 instruction per SIMD per 2 cycles, the datasheet rate
    frac            lane_ops_per_s / peak
    cycles_per_inst SIMD-cycles per VALU wave-instruction = 1024 SIMDs * clock * duration / valu_insts  (2 = datasheet peak;
                    tools/microbench/valu_rates.hip measures 2.3 for plain two-source ops and 4.1 for v_mad_u64_u32, carry
                    ops, v_cndmask, three-source ops: a stream of those CANNOT go below ~4.1)
    insts_per_wave  valu_insts / SQ_WAVES
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

NUM_SIMD = 256 * 4


def main():
    d = sys.argv[1]
    rows = defaultdict(dict)   # dispatch id -> {counter: value, name, grid, dur}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            e = rows[r["Dispatch_Id"]]
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            e["name"] = r["Kernel_Name"].split("(")[0].replace("p2k::", "").replace("void ", "")
            e["grid"] = int(r["Grid_Size"])
            e["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            e["vgpr"] = int(r["VGPR_Count"])
    by = defaultdict(list)
    for e in rows.values():
        by[(e["name"], e["grid"])].append(e)
    best = {}
    for (name, grid), es in by.items():
        if name not in best or grid > best[name][0]:
            best[name] = (grid, es)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from src_id import load_src_id
    out = {"method": __doc__.strip().split("\n\n")[1], "source": load_src_id(sys.argv), "kernels": {}}
    total = sum(e["dur"] for e in rows.values())
    for name, (grid, es) in sorted(best.items(), key=lambda kv: -sum(e["dur"] for e in kv[1][1])):
        n = len(es)
        dur = sum(e["dur"] for e in es) / n
        g = lambda k: sum(e.get(k, 0.0) for e in es) / n
        insts, waves = g("SQ_INSTS_VALU"), g("SQ_WAVES")
        if dur <= 0 or insts <= 0:
            continue
        clock = g("GRBM_GUI_ACTIVE") / 8.0 / dur
        peak = NUM_SIMD * 32 * clock
        ent = {"launches": n, "grid": grid, "vgprs": es[0]["vgpr"], "duration_ms": round(dur * 1e3, 4), "clock_GHz": round(clock / 1e9, 3),
               "valu_insts": int(insts), "waves": int(waves), "insts_per_wave": round(insts / max(waves, 1), 1),
               "lane_ops_per_s": round(insts * 64 / dur / 1e12, 3), "peak_lane_ops_per_s": round(peak / 1e12, 3),
               "frac": round(insts * 64 / dur / peak, 4), "cycles_per_inst": round(NUM_SIMD * clock * dur / insts, 3),
               "sq_active_inst_valu": int(g("SQ_ACTIVE_INST_VALU")), "sq_busy_cycles": int(g("SQ_BUSY_CYCLES")),
               "share_of_pass_time": round(sum(e["dur"] for e in es) / total, 4), "unit": "T lane-ops/s"}
        out["kernels"][name] = ent
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
