#!/bin/bash
# instruction counters of the encode kernels on the headline shape: [LIB=tools/lib_x.so] [PMC="..."] tools/pmc_octo.sh <out name>
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export LOG2N=${LOG2N:-28} SKIP_CHECK=1
[ -n "$LIB" ] && export OROCH_HIP_LIBRARY=$R/$LIB
mkdir -p $R/gpurun_out
timeout -s KILL 200 rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY} -d $R/gpurun_out/$1 -o pmc --output-format csv -- python3 $R/tools/check_octo.py > $R/gpurun_out/$1.log 2>&1
python3 - $R/gpurun_out/$1 <<'PY'
import csv, collections, sys, glob, os
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
nblocks = (1 << int(os.environ.get("LOG2N", "28"))) // 256
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'oroch' not in k: continue
    name = k.split('<')[0].split('::')[-1][:24]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in agg.items():
    if 'tile' in name or 'scan' in name or 'finish' in name: continue
    print(name, {c: round(sum(v)/len(v)/nblocks,1) for c, v in sorted(cs.items())})
PY
