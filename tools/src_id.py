#!/usr/bin/env python3
"""Identity of the kernel sources a measurement was taken on: SHA-256 over plonky2-aes_amd/csrc/* (names and contents, sorted).
Printed as JSON; the PMC post-processors (tools/pmc_traffic.py, tools/pmc_valu.py) embed it in profiles/<tag>_*.json and bench.py
compares it with the sources it is running on, so that a counter file taken on OTHER kernels shows as stale in the bench line
instead of silently pricing the wrong binary.

    python tools/src_id.py > gpurun_out/<tag>_srcid.json      # on the GPU box, next to the rocprofv3 passes"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_id():
    d = os.path.join(ROOT, "plonky2-aes_amd", "csrc")
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(d) if not f.startswith("."))
    for f in names:
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(d, f), "rb").read())
    return {"csrc_sha16": h.hexdigest()[:16], "csrc_files": len(names)}


def load_src_id(argv):
    """--src-id FILE (written on the GPU box by this script) and --git-head SHA from a post-processor's command line."""
    out = {}
    if "--src-id" in argv:
        out.update(json.load(open(argv[argv.index("--src-id") + 1])))
    if "--git-head" in argv:
        out["git_head"] = argv[argv.index("--git-head") + 1]
    return out


if __name__ == "__main__":
    json.dump(csrc_id(), sys.stdout)
    print()
