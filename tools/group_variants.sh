#!/bin/bash
# builds variants of the library that differ in group.hip alone: tools/group_variants.sh "name:-DFLAG ..." ...
set -e
cd "$(dirname "$0")/.."
B=oroch_amd/csrc/build
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -Wno-pass-failed -c oroch_amd/csrc/group.hip -o /tmp/group_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib_$name.so /tmp/group_$name.o $B/pack.o $B/varint.o $B/array.o $B/sequence.o $B/pipeline.o $B/api.o ) &
done
wait
ls -la tools/lib_*.so
