#!/bin/bash
# Collects what profiles/ holds for a round, on the GPU box:
#   tools/collect_profiles.sh <tag, e.g. r2_a> <commit the kernels were built from>
# 1. rocprofv3 kernel statistics and the bench line of the headline (int32, 2^30 values)
# 2. HBM traffic of the same kernels from two counter passes (FETCH_SIZE, WRITE_SIZE), each a run of its own
# 3. kernel statistics of the int64 block mix (2^30 values), of int16 and of uint64 inputs
# Summaries go to gpurun_out/<tag>/; copy them into profiles/ afterwards.
set -e
TAG=$1
COMMIT=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
DEST=$R/gpurun_out/$TAG
mkdir -p $DEST
cd /tmp && export TMPDIR=/tmp
stats() { # <name> <program and arguments ...>
	local name=$1; shift
	timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $DEST/raw_$name -o p -- "$@" > $DEST/${name}.log 2>&1
	local f=$(find $DEST/raw_$name -name "*kernel_stats.csv" | head -n 1)
	(head -n 1 $f; grep oroch_hip $f) > $DEST/${TAG}_${name}_kernel_stats.csv
}
stats headline python3 $R/bench.py --steps 20 --no-cpu-baseline --no-end-to-end --no-other-configs
grep '^{' $DEST/headline.log | tail -n 1 > $DEST/${TAG}_bench.json
for c in FETCH_SIZE WRITE_SIZE; do
	timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $DEST/pmc_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-other-configs > $DEST/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py $DEST/pmc_FETCH_SIZE $DEST/pmc_WRITE_SIZE $COMMIT > $DEST/traffic_int32_2p30.json
stats int64mix python3 $R/bench.py --config int64mix --steps 10 --no-cpu-baseline --no-end-to-end
grep '^{' $DEST/int64mix.log | tail -n 1 > $DEST/${TAG}_int64mix_bench.json
DT=int16 KIND=outlier BODY=9 OUT=15 LOG2N=29 stats int16 python3 $R/tools/time_any.py
DT=uint64 KIND=outlier BODY=20 OUT=50 LOG2N=28 stats uint64 python3 $R/tools/time_any.py
DT=int32 KIND=uniform BODY=13 LOG2N=28 stats int32_uniform13 python3 $R/tools/time_any.py
DT=int16 KIND=outlier BODY=9 OUT=16 PERMILLE=5 SHIFT=-256 SEED=5 LOG2N=28 stats int16_config5 python3 $R/tools/time_any.py
stats sequences python3 $R/tools/time_sequences.py
grep -h "frac of 8TB/s\|of 8 TB/s" $DEST/int16.log $DEST/uint64.log $DEST/int32_uniform13.log $DEST/int16_config5.log $DEST/sequences.log > $DEST/${TAG}_other_inputs.txt
# the bench as the driver runs it (no profiler): the line with every part, the traffic
# just measured included (the box's copy of profiles/ gets the new record)
mkdir -p $R/profiles && cp $DEST/traffic_int32_2p30.json $R/profiles/
cd $R && timeout -k 10 600 python3 bench.py > $DEST/${TAG}_bench_plain.json 2> $DEST/bench_plain.err
ls -la $DEST
