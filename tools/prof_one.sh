#!/bin/bash
# tools/prof_one.sh <tag> <program and arguments>: rocprofv3 kernel statistics of one run, summary printed
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/$TAG
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/raw -o p -- "$@" > $D/run.log 2>&1
f=$(find $D/raw -name "*kernel_stats.csv" | head -n 1)
python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "oroch" in r["Name"]:
        print("%-70s %4s %9.1f us" % (r["Name"].split("(")[0][-68:], r["Calls"], float(r["AverageNs"]) / 1e3))
P
grep -v amdgpu.ids $D/run.log | tail -n 4
rm -rf $D/raw
