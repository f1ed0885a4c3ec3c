#!/bin/bash
# Same-box A/B of engine switches: scripts/ab_bench.sh OUTDIR "ENV=1" ["ENV2=1" ...] -- every variant and the default, interleaved twice
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for v in "" "$@"; do
    name=${v:-default}
    env $v python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${name%%=*}_$rep.json
    python - <<PY
import json
d=json.loads(open("$OUT/${name%%=*}_$rep.json").read())
r=d["roofline"]
print("%-18s rep $rep  %.2f evals/s  lstm %.4f  all-conv %.4f  conv ms %.1f" % ("$name", d["value"], r["frac"], r["all_conv_kernels"]["frac"], r["all_conv_kernels"]["total_ms"]))
PY
  done
done
