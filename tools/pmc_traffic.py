#!/usr/bin/env python3
"""HBM traffic per launch of the block kernels from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, each collected in a run of its own).

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [commit] > profiles/traffic.json

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests
of wide streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "oroch" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("<")[0].split("(")[0].split("::")[-1]
        agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main(fetch_dir, write_dir, commit=None):
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * 1024.0 * fetch.get(k, 0.0)
        wr = 1024.0 * write.get(k, 0.0)
        out[k] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
    import oroch_amd.sources as sources

    json.dump({"unit": "bytes per launch", "correction": "FETCH_SIZE x 2 (gfx950), KiB -> bytes",
               "commit": commit, "kernel_sources": sources.digest(), "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:4])
