#!/bin/bash
# Same-box A/B of ENVIRONMENT settings of one library: scripts/ab_env.sh OUTDIR "NAME1=VAL1 ..." "NAME2=..." ...   (an empty string = the default)
# Optional bench arguments in $ARGS (default "--steps 4"), repetitions in $REPS (default 2).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
mkdir -p $OUT
ARGS=${ARGS:-"--steps 4"}
for rep in $(seq 1 ${REPS:-2}); do
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python $R/bench.py $ARGS --warmup 2 --no-cpu-baseline --no-parity --no-supplementary 2>$OUT/err_${i}_$rep.txt | tail -1 > $OUT/ab_${i}_$rep.json
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/ab_${i}_$rep.json").read())
    r=d["roofline"]
    ops={(o["layer"],o["op"]):o["ms"] for o in r["per_op"]}
    print("%-34s rep $rep %9.2f evals/s  lstm %.4f  all-conv %.4f  conv ms %.2f | lstm1-3 %s convA2-3 %s convP1-3 %s" % ("[$envs]", d["value"], r["frac"], r["all_conv_kernels"]["frac"], r["all_conv_kernels"]["total_ms"],
        [round(ops.get((l,"lstm"),0),1) for l in (1,2,3)], [round(ops.get((l,"convA"),0),1) for l in (2,3)], [round(ops.get((l,"convP"),0),1) for l in (1,2,3)]))
except Exception as e:
    print("[$envs] rep $rep FAILED", e, open("$OUT/err_${i}_$rep.txt").read()[-800:])
PY
done
done
