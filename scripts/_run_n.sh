cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_n; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -s -k "winograd" > $O/pytest_wino.log 2>&1; grep "MISMATCH\|WINO_\|passed\|failed\|Error\|error" $O/pytest_wino.log | tail -12
for f in 0 1 0 1; do
  EIGEN_WINO_FUSEUP=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>$O/bench.err | tail -1 > $O/bench_f$f.json
  python -c "
import json
d=json.loads(open('$O/bench_f$f.json').read()); r=d['roofline']
print('EIGEN_WINO_FUSEUP=$f: %.2f evals/s conv ms %.1f  nonzero %d' % (d['value'], r['all_conv_kernels']['total_ms'], d['nonzero_fitness']), [(o['layer'], o['op'][:5], round(o['ms'],1)) for o in r['per_op'] if o['ms']>10 and o['layer']>0])" || tail -5 $O/bench.err
done
for sh in "ref160 10" "ref640 4"; do
  set -- $sh
  for f in 0 1; do
  EIGEN_WINO_FUSEUP=$f python bench.py --shape $1 --steps $2 --warmup 2 2>$O/bench.err | tail -1 > $O/bench_$1.json
  python -c "
import json
d=json.loads(open('$O/bench_$1.json').read()); r=d['roofline']
print('$1 fuseup $f: %.2f evals/s conv ms %.2f' % (d['value'], r['all_conv_kernels']['total_ms']))"
  done
done
