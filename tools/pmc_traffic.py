#!/usr/bin/env python3
"""Turn two rocprofv3 counter-collection CSVs (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE) into the
per-launch HBM traffic JSON that bench.py reads (profiles/r01_traffic.json).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <proofs per chunk> [--src-id gpurun_out/<tag>_srcid.json] [--git-head SHA] > profiles/<tag>_traffic.json

Counters are KiB.  gfx950 correction (tools/microbench/traffic_calib.hip, profiles/r01_pmc_calibration_*.csv): a kernel
streaming 2 GiB with 8 B/lane loads reports FETCH_SIZE = 1.0 GiB, so FETCH_SIZE is doubled; WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    rows = defaultdict(list)  # kernel short name -> [(grid, KiB)]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("p2k::", "").replace("void ", "")
            rows[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return rows


def main():
    fd, wd, chunk = sys.argv[1], sys.argv[2], int(sys.argv[3])
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from src_id import load_src_id
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    out = {"method": __doc__.split("\n\n")[1].strip() + "  Counters are KiB; FETCH_SIZE doubled (8 B/lane loads), WRITE_SIZE exact.",
           "chunk": chunk, "source": load_src_id(sys.argv), "per_launch_avg_bytes": {}}
    for name in sorted(fe, key=lambda k: -sum(v for _, v in fe[k])):
        f, w = fe[name], wr.get(name, [])
        if len(f) != len(w):
            continue
        ent = {"launches": len(f), "read": int(2 * 1024 * sum(v for _, v in f) / len(f)), "written": int(1024 * sum(v for _, v in w) / len(w))}
        ent["total"] = ent["read"] + ent["written"]
        if name == "k_hash_leaves":
            # the three trees of a chunk (wires, zs, quotient) share one grid (a thread per leaf per proof); the
            # circuit-setup tree is one smaller launch and is left out of the per-chunk average bench.py reports
            gmax = max(g for g, _ in f)
            ff, ww = [v for g, v in f if g == gmax], [v for g, v in w if g == gmax]
            ent["chunk_launches"] = len(ff)
            ent["chunk_launches_avg"] = int(2 * 1024 * sum(ff) / len(ff) + 1024 * sum(ww) / len(ww))
        out["per_launch_avg_bytes"][name] = ent
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
