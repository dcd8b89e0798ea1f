#!/bin/bash
# builds experiment variants of the library into tools/ (lib_<name>.so); usage: tools/build_variants.sh "name:-DFLAG ..." ...
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -o tools/lib_$name.so oroch_amd/csrc/group.hip oroch_amd/csrc/pack.hip oroch_amd/csrc/api.hip &
done
wait
ls -la tools/*.so
