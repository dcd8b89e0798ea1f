# A/B of builds of the library on the inputs of tools/time_inputs.py, on one box:
#   VARIANTS="head new" INPUTS="headline int16cfg5" bash tools/ab_inputs.sh
set -e
for rep in 1 2; do
for v in ${VARIANTS:-head new}; do
  export OROCH_HIP_LIBRARY=$PWD/oroch_amd/variants/$v.so
  echo "== $v (rep $rep)"
  timeout -k 10 200 python3 tools/time_inputs.py ${INPUTS:-headline int16cfg5} 2>/dev/null | sed 's/frac of.*//'
done
done
