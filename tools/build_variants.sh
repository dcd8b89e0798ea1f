#!/bin/bash
# builds experiment variants of the library into tools/ (lib_<name>.so); usage: tools/build_variants.sh "name:-DFLAG ..." ...
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -o tools/lib_$name.so ${ONLY:--DOROCH_ONLY_I32} -Wno-pass-failed oroch_amd/csrc/group.hip oroch_amd/csrc/pack.hip oroch_amd/csrc/api.hip oroch_amd/csrc/varint.hip oroch_amd/csrc/array.hip oroch_amd/csrc/sequence.hip oroch_amd/csrc/pipeline.hip &
done
wait
ls -la tools/*.so
