#!/bin/bash
# Same-box A/B of differently COMPILED engine libraries (measurement builds with -D switches): scripts/ab_libs.sh OUTDIR lib1.so lib2.so ...
# Each library (and the default one) runs the headline bench; optional extra environment in $ABENV.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
mkdir -p $OUT
ARGS=${ARGS:-"--steps 4"}
for rep in 1 2; do
for lib in "" "$@"; do
  name=${lib:-default}
  if [ -n "$lib" ]; then export EIGEN_HIP_LIB=$R/evolutionary_illusion_generator_amd/$lib; else unset EIGEN_HIP_LIB; fi
  env $ABENV python $R/bench.py $ARGS --warmup 2 --no-cpu-baseline --no-parity --no-supplementary 2>/dev/null | tail -1 > $OUT/${name%.so}_$rep.json
  python - <<PY
import json
d=json.loads(open("$OUT/${name%.so}_$rep.json").read())
r=d["roofline"]
print("%-28s rep $rep %9.2f evals/s  lstm %.4f  all-conv %.4f  conv ms %.2f" % ("$name $ABENV", d["value"], r["frac"], r["all_conv_kernels"]["frac"], r["all_conv_kernels"]["total_ms"]))
PY
done
done
