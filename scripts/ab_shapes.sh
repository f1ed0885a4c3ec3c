#!/bin/bash
# Same-box A/B of engine switches over several shapes: scripts/ab_shapes.sh OUTDIR "ENV=.. [ENV2=..]" ... ; shapes from $SHAPES
# (default: headline, pop 32, ref160, c2).  Every variant and the default, interleaved, one line each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
mkdir -p $OUT
SHAPES=${SHAPES:-"--steps@4 --pop@32@--steps@8 --shape@ref160@--steps@10 --shape@c2@--steps@20"}
for sh in $SHAPES; do
  args=${sh//@/ }
  for v in "" "$@"; do
    name=${v:-default}
    env $v python $R/bench.py $args --warmup 2 --no-cpu-baseline --no-parity --no-supplementary 2>/dev/null | tail -1 > $OUT/last.json
    python - <<PY
import json
d=json.loads(open("$OUT/last.json").read())
r=d["roofline"]
print("%-26s %-22s %9.2f evals/s  lstm %.4f  all-conv %.4f  conv ms %.2f" % ("$args"[:26], "$name", d["value"], r["frac"], r["all_conv_kernels"]["frac"], r["all_conv_kernels"]["total_ms"]))
PY
  done
done
