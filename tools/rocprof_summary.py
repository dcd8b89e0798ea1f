#!/usr/bin/env python3
"""Per-kernel summary (the --stats table) of a rocprofv3 results database.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs",
                "VGPRs", "AccumVGPRs", "SGPRs", "LDSBytes", "ScratchBytes", "GridX", "WorkgroupX"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / total), r[4], r[5]] + list(r[6:]))


if __name__ == "__main__":
    main(sys.argv[1])
