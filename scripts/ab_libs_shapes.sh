#!/bin/bash
# scripts/ab_libs_shapes.sh OUTDIR lib1.so ... : ab_libs.sh over the small shapes (one repetition each)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
mkdir -p $OUT
for args in "--shape ref160 --steps 10" "--shape c2 --steps 20" "--pop 32 --steps 8"; do
for lib in "" "$@"; do
  name=${lib:-default}
  if [ -n "$lib" ]; then export EIGEN_HIP_LIB=$R/evolutionary_illusion_generator_amd/$lib; else unset EIGEN_HIP_LIB; fi
  env $ABENV python $R/bench.py $args --warmup 2 --no-cpu-baseline --no-parity --no-supplementary 2>/dev/null | tail -1 > $OUT/last.json
  python - <<PY
import json
d=json.loads(open("$OUT/last.json").read())
r=d["roofline"]
print("%-28s %-24s %9.2f evals/s  lstm %.4f  all-conv %.4f  conv ms %.2f" % ("$args", "$name $ABENV", d["value"], r["frac"], r["all_conv_kernels"]["frac"], r["all_conv_kernels"]["total_ms"]))
PY
done
done
