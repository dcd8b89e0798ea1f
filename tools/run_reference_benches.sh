#!/bin/bash
# The reference's bench programs (tests/bench/*.cc, built by `make -C oracle benches`) on the
# GPU box: against this repository's headers (one PCIe round trip per call) and against the
# reference's own headers on the host.  tools/run_reference_benches.sh <out file>
# (array-bench makes ten million find() calls and prints nothing meanwhile: a line
# goes to the output every half minute so that the run is seen to be alive.)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1
: > $OUT
for b in ${BENCHES:-varint group array}; do
	for v in "" -cpu; do
		echo "== $b-bench$v" >> $OUT
		start=$(date +%s)
		timeout -k 10 ${LIMIT:-900} $R/oracle/_ref/$b-bench$v >> $OUT 2>&1 &
		pid=$!
		while kill -0 $pid 2>/dev/null; do
			sleep 5
			now=$(date +%s)
			if [ $(( (now - start) % 30 )) -lt 5 ] && [ $now -gt $((start + 20)) ]; then echo "   ... $((now - start)) s" >> $OUT; fi
		done
		wait $pid
		echo "exit $? after $(( $(date +%s) - start )) s" >> $OUT
	done
done
cat $OUT
