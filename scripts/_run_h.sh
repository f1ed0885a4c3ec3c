cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp EIGEN_WINO_MODE=4
O=gpurun_out/r04_h; mkdir -p $O
for sh in "ref160 10" "c2 20" "ref640 4" "c5 2"; do
  set -- $sh
  for w in 0 2 6 14; do
    EIGEN_WINOGRAD=$w python bench.py --shape $1 --steps $2 --warmup 2 $( [ $1 = c5 ] && echo "--pop 64" ) 2>$O/bench.err | tail -1 > $O/bench_$1_w$w.json
    python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1_w$w.json").read())
    r=d["roofline"]
    print("$1 EIGEN_WINOGRAD=$w: %.2f evals/s  conv ms %.2f" % (d["value"], r["all_conv_kernels"]["total_ms"]), [(o["layer"], round(o["ms"],2)) for o in r["per_op"] if o["op"]=="lstm" and o["layer"]>0])
except Exception as e: print("$1 w=$w failed", e); print(open("$O/bench.err").read()[-800:])
PY
  done
done
