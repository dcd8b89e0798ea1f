# A/B of builds of the library on one box (timings differ by several percent from box to box,
# so variants are only comparable within one call):
#   cp oroch_amd/liboroch_hip.so oroch_amd/variants/<name>.so   for each build, then
#   VARIANTS="head new" bash tools/ab.sh
# Prints the kernel timings of tools/time_any.py on the headline shape and on wider ones, twice.
set -e
for rep in 1 2; do
for v in ${VARIANTS:-head new}; do
  export OROCH_HIP_LIBRARY=$PWD/oroch_amd/variants/$v.so
  echo "== $v (rep $rep)"
  for spec in "outlier 10 28" "outlier 16 30" "outlier 20 30" "outlier 24 31" "uniform 18 0" "mix 0 0"; do
    set -- $spec
    DT=int32 LOG2N=28 KIND=$1 BODY=$2 OUT=$3 timeout -k 10 120 python3 tools/time_any.py 2>/dev/null | sed 's/frac.*//'
  done
done
done
