#!/bin/bash
# timing of variants of the library (tools/lib_<name>.so) on one box: VARIANTS="base fake" bash tools/octo_ab.sh
for v in ${VARIANTS:-base}; do
  echo "== $v"
  OROCH_HIP_LIBRARY=$PWD/tools/lib_$v.so SKIP_CHECK=${SKIP_CHECK-1} LOG2N=${LOG2N:-28} timeout -k 10 200 python3 -u tools/check_octo.py 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-1}
done
