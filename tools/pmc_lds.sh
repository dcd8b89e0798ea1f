#!/bin/bash
# LDS counters of the encode kernels: tools/pmc_lds.sh <library> <encode path> <out name>
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export OROCH_HIP_LIBRARY=$R/$1 ENCODE_PATH=$2 LOG2N=${LOG2N:-26}
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/gpurun_out/$3 -o pmc --output-format csv -- python3 $R/tools/run_fused_once.py > $R/gpurun_out/$3.log 2>&1
python3 - $R/gpurun_out/$3 <<'PY'
import csv, collections, sys, glob, os
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
nblocks = (1 << int(os.environ.get("LOG2N", "26"))) // 256
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'oroch' not in k: continue
    name = k.split('<')[0].split('::')[-1][:24]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in agg.items():
    print(name, {c: round(sum(v)/len(v)/nblocks,1) for c, v in sorted(cs.items())})
PY
