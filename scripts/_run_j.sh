cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_j; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -s -k "winograd" > $O/pytest_wino.log 2>&1; grep "WINO\|passed\|failed\|Error" $O/pytest_wino.log | tail -40
for w in 0x0000000E 0x00FFFFFE; do
  EIGEN_WINOGRAD=$w python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>$O/bench.err | tail -1 > $O/bench_$w.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$w.json").read())
    r=d["roofline"]
    print("EIGEN_WINOGRAD=$w: %.2f evals/s  nonzero %d  conv ms %.1f" % (d["value"], d["nonzero_fitness"], r["all_conv_kernels"]["total_ms"]))
    print("   ", [(o["layer"], o["op"][:5], round(o["ms"],1)) for o in r["per_op"] if o["ms"] > 3])
except Exception as e: print("$w failed", e); print(open("$O/bench.err").read()[-1500:])
PY
done
for sh in "ref160 10" "ref640 4"; do
  set -- $sh
  python bench.py --shape $1 --steps $2 --warmup 2 2>$O/bench.err | tail -1 > $O/bench_$1.json
  python -c "
import json
d=json.loads(open('$O/bench_$1.json').read()); r=d['roofline']
print('$1 default: %.2f evals/s conv ms %.2f' % (d['value'], r['all_conv_kernels']['total_ms']), [(o['layer'], o['op'][:5], round(o['ms'],2)) for o in r['per_op'] if o['ms']>1])"
done
