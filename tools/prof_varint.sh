#!/bin/bash
# kernel times of the whole-sequence varint codec: tools/prof_varint.sh <name under gpurun_out/>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/$1
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/raw -o p -- python3 $R/tools/time_varint.py > $D/run.log 2>&1
tail -n 1 $D/run.log
python3 - $D <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/raw/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'varint' in r['Name']:
        print('%-16s %3s calls  %.4f ms' % (r['Name'].split('<')[0].split('::')[-1], r['Calls'], float(r['AverageNs']) / 1e6))
PY
